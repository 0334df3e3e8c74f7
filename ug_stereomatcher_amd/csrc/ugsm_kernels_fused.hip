// ugsm_kernels_fused.hip -- kernel_path 0: the production gfx950 kernels.
//
// K-cost  (k_cost_fused):   one matcher iteration's warp + 5-shift squared-NCC cost (3 channels) +
//                           parabola + confidence blend + disparity update, one launch, LDS tiled.
// K-smooth (k_smooth_fused): up to 5 confidence-weighted Jacobi passes + the 3x3 box, one launch.
// K-pyr / K-sq:             blur+decimate evaluated only at the sampled sites; G_clamp*(L^2).
//
// The reference does this with ~120 single-op launches and ~100 device-to-device plane copies per
// iteration.  Arithmetic is bit-identical to the one-stage-per-kernel path
// (ugsm_kernels_ref.hip) and to the CPU oracle: same IEEE operations in the same order, no
// contraction (see ugsm_device.hpp).  No MFMA: this is a stencil with data-dependent gathers.
//
// Citations: /root/reference/src/gpu_matcher/<file>:<line>.
#include "ugsm_device.hpp"
#include "ugsm_launch.hpp"

namespace ugsm {

// =========================================================================================
// K-cost
// =========================================================================================
//
// Tile TX x TY = 32 x 28 output pixels per 256-thread workgroup (4 waves).  A thread owns a
// "quad" (4 consecutive x) so that every LDS access is a 16-byte ds_read/write_b128; thread
// (qx = tid&7, row = tid>>3) -> quad column qx, tile row `row`.
//
// LDS images (float, tile-relative column c stored at [c + OX]):
//   sIdx  [34][38]  source offset of the warped fetch for every pixel of tile+halo3 (all channels)
//   sR    [34][48]  R' = warped right plane, tile+halo3, edge-replicated (texture clamp)   OX=8
//   sL    [32][40]  left plane, tile+halo2, ZERO outside the image (smem-conv zero padding) OX=4
//   sRow  [5][32][32] row-pass of the five product images, rows tile+halo2
//   sBrow [34][40]  row-pass of R'^2, rows tile+halo3, cols tile+halo4                       OX=4
//   sB    [30][40]  B = G_clamp*(R'^2), tile+halo1 (only in-image entries are ever read)     OX=4
// = 47.5 KB -> 3 workgroups (12 waves) per CU.
//
// Per channel: P1 fill sL,sR | barrier | P2 row passes | barrier | P2.5 B column pass | barrier |
// P3 column pass of the 5 products + correlation, accumulated over channels in registers.
constexpr int TX = 32, TY = 28, QX = TX / 4;
constexpr int SR_W = TX + 16, SR_H = TY + 6, SR_OX = 8;
constexpr int SL_W = TX + 8, SL_H = TY + 4, SL_OX = 4;
constexpr int ROW_W = TX, ROW_H = TY + 4;
constexpr int SB_W = TX + 8, SBROW_H = TY + 6, SB_H = TY + 2, SB_OX = 4;
constexpr int IDX_W = TX + 6, IDX_H = TY + 6;

struct f4 {
    float v[4];
};
__device__ __forceinline__ void ld4(const float *p, float *o)
{
    const float4 t = *reinterpret_cast<const float4 *>(p);
    o[0] = t.x; o[1] = t.y; o[2] = t.z; o[3] = t.w;
}
__device__ __forceinline__ void st4(float *p, const float *o)
{
    *reinterpret_cast<float4 *>(p) = make_float4(o[0], o[1], o[2], o[3]);
}

// products are >= +0, so "0 + x" is x exactly and the first add of tap5 can be dropped
__device__ __forceinline__ float tap5p(float a, float b, float c, float d, float e)
{
    float sum = a * UGSM_G0;
    sum += b * UGSM_G1;
    sum += c * UGSM_G2;
    sum += d * UGSM_G1;
    sum += e * UGSM_G0;
    return sum;
}

__global__ __launch_bounds__(256) void k_cost_fused(Img3 L, Img3 R, const float *__restrict__ A3, const float *__restrict__ d3,
                                                    float *__restrict__ nd3, int W, int H, float thr, int blend)
{
    __shared__ __attribute__((aligned(16))) float sR[SR_H * SR_W];
    __shared__ __attribute__((aligned(16))) float sL[SL_H * SL_W];
    __shared__ __attribute__((aligned(16))) float sRow[5 * ROW_H * ROW_W];
    __shared__ __attribute__((aligned(16))) float sBrow[SBROW_H * SB_W];
    __shared__ __attribute__((aligned(16))) float sB[SB_H * SB_W];
    __shared__ int sIdx[IDX_H * IDX_W];

    const int tid = threadIdx.x;
    const int x0 = blockIdx.x * TX, y0 = blockIdx.y * TY;
    const size_t n = (size_t)W * H;
    const int qx = tid & (QX - 1), trow = tid >> 3;  // 8 quad columns x 32 rows

    // ---- P0: warped source offsets for tile+halo3 (warpAbyB, MatchLib.cu:510-515) -------------
    for (int it = tid; it < IDX_H * IDX_W; it += 256) {
        const int r = it / IDX_W, c = it - r * IDX_W;
        const int gx = clampi(x0 + c - 3, 0, W - 1), gy = clampi(y0 + r - 3, 0, H - 1);
        const size_t at = (size_t)gy * W + gx;
        const int sx = tex_index(((float)gx + 0.5f) + d3[at], W);
        const int sy = tex_index(((float)gy + 0.5f) + d3[n + at], H);
        sIdx[it] = sy * R.pitch + sx;
    }

    float Q[5][4];
#pragma unroll
    for (int s = 0; s < 5; s++)
#pragma unroll
        for (int i = 0; i < 4; i++) Q[s][i] = 0.0f;

    for (int k = 0; k < 3; k++) {
        const float *Lk = L.p + (size_t)k * L.plane;
        const float *Rk = R.p + (size_t)k * R.plane;
        __syncthreads();  // sIdx ready (k==0); previous channel's P3 finished with sRow/sB
        // ---- P1: stage L (zero outside) and R' (gather, edge replicated) ------------------
        for (int it = tid; it < IDX_H * IDX_W; it += 256) {
            const int r = it / IDX_W, c = it - r * IDX_W;
            sR[r * SR_W + (c - 3 + SR_OX)] = Rk[sIdx[it]];
        }
        for (int it = tid; it < SL_H * (TX + 4); it += 256) {
            const int r = it / (TX + 4), c = it - r * (TX + 4);  // c: tile column + 2
            const int gx = x0 + c - 2, gy = y0 + r - 2;
            float v = 0.0f;
            if (gx >= 0 && gx < W && gy >= 0 && gy < H) v = Lk[(size_t)gy * L.pitch + gx];
            sL[r * SL_W + (c - 2 + SL_OX)] = v;
        }
        __syncthreads();
        // ---- P2a: row pass of the five products (CompareMove + convolutionRowsKernel) -----
        {
            const int r = trow;  // 0..31 <-> tile row r-2
            const int cx = qx * 4;
            float l[12], rc[12], ru[12], rd[12];
            const float *pl = &sL[r * SL_W + cx - 4 + SL_OX];
            ld4(pl, l); ld4(pl + 4, l + 4); ld4(pl + 8, l + 8);
            const float *pr = &sR[(r + 1) * SR_W + cx - 4 + SR_OX];  // sR row index = tile row + 3 = (r-2)+3
            ld4(pr, rc); ld4(pr + 4, rc + 4); ld4(pr + 8, rc + 8);
            ld4(pr - SR_W, ru); ld4(pr - SR_W + 4, ru + 4); ld4(pr - SR_W + 8, ru + 8);
            ld4(pr + SR_W, rd); ld4(pr + SR_W + 4, rd + 4); ld4(pr + SR_W + 8, rd + 8);
            // arrays hold tile columns cx-4 .. cx+7; pixel cx+i sits at [i+4]
            float p[5][8];  // products at columns cx-2 .. cx+5
#pragma unroll
            for (int j = 0; j < 8; j++) {
                const float lv = l[j + 2];
                p[0][j] = lv * rc[j + 1];  // shift (-1, 0)
                p[1][j] = lv * rc[j + 3];  // shift (+1, 0)
                p[2][j] = lv * ru[j + 2];  // shift (0, -1)
                p[3][j] = lv * rd[j + 2];  // shift (0, +1)
                p[4][j] = lv * rc[j + 2];  // shift (0, 0)
            }
#pragma unroll
            for (int s = 0; s < 5; s++) {
                float o[4];
#pragma unroll
                for (int i = 0; i < 4; i++) o[i] = tap5p(p[s][i], p[s][i + 1], p[s][i + 2], p[s][i + 3], p[s][i + 4]);
                st4(&sRow[(s * ROW_H + r) * ROW_W + cx], o);
            }
        }
        // ---- P2b: row pass of R'^2 (Square + convolutionRowsKernelT), cols -4..TX+3 -------
        for (int it = tid; it < SBROW_H * (SB_W / 4); it += 256) {
            const int r = it / (SB_W / 4), q = it - r * (SB_W / 4);
            const int cx = q * 4 - 4;
            float v[12];
            const float *pr = &sR[r * SR_W + cx - 4 + SR_OX];
            ld4(pr, v); ld4(pr + 4, v + 4); ld4(pr + 8, v + 8);
            float sq[8];
#pragma unroll
            for (int j = 0; j < 8; j++) sq[j] = v[j + 2] * v[j + 2];
            float o[4];
#pragma unroll
            for (int i = 0; i < 4; i++) o[i] = tap5p(sq[i], sq[i + 1], sq[i + 2], sq[i + 3], sq[i + 4]);
            st4(&sBrow[r * SB_W + cx + SB_OX], o);
        }
        __syncthreads();
        // ---- P2.5: column pass of R'^2 -> B on tile+halo1 (convolutionColumnsKernelT) -----
        for (int it = tid; it < SB_H * (SB_W / 4); it += 256) {
            const int r = it / (SB_W / 4), q = it - r * (SB_W / 4);  // r: tile row + 1
            float a[4], b[4], c[4], d[4], e[4], o[4];
            const float *pb = &sBrow[r * SB_W + q * 4];  // sBrow row index = tile row + 3; rows (r-1)-2+3 = r .. r+4
            ld4(pb, a); ld4(pb + SB_W, b); ld4(pb + 2 * SB_W, c); ld4(pb + 3 * SB_W, d); ld4(pb + 4 * SB_W, e);
#pragma unroll
            for (int i = 0; i < 4; i++) o[i] = tap5p(a[i], b[i], c[i], d[i], e[i]);
            st4(&sB[r * SB_W + q * 4], o);
        }
        __syncthreads();
        // ---- P3: column pass of the products, correlation, channel accumulate ---------------
        if (trow < TY) {
            const int cx = qx * 4;
            const int gy = y0 + trow, gx0 = x0 + cx;
            if (gy < H && gx0 < W) {
                // B at the five clamped positions (MoveCorrelation's texdispy fetch, MatchLib.cu:683)
                float bc[12], bu[4], bd[4];
                const float *pb = &sB[(trow + 1) * SB_W + cx - 4 + SB_OX];
                ld4(pb, bc); ld4(pb + 4, bc + 4); ld4(pb + 8, bc + 8);  // columns cx-4 .. cx+7, pixel i at [i+4]
                ld4(pb - SB_W + 4, bu);
                ld4(pb + SB_W + 4, bd);
                const bool top = (gy == 0), bot = (gy == H - 1);
                float a4[4];
#pragma unroll
                for (int i = 0; i < 4; i++) a4[i] = (gx0 + i < W) ? A3[k * n + (size_t)gy * W + gx0 + i] : 1.0f;
#pragma unroll
                for (int s = 0; s < 5; s++) {
                    float r0[4], r1[4], r2[4], r3[4], r4[4];
                    const float *ps = &sRow[(s * ROW_H + trow) * ROW_W + cx];  // rows trow .. trow+4 <-> tile rows trow-2..trow+2
                    ld4(ps, r0); ld4(ps + ROW_W, r1); ld4(ps + 2 * ROW_W, r2); ld4(ps + 3 * ROW_W, r3); ld4(ps + 4 * ROW_W, r4);
#pragma unroll
                    for (int i = 0; i < 4; i++) {
                        const float N = tap5p(r0[i], r1[i], r2[i], r3[i], r4[i]);
                        const int gx = gx0 + i;
                        float b;
                        if (s == 0) b = (gx == 0) ? bc[i + 4] : bc[i + 3];
                        else if (s == 1) b = (gx >= W - 1) ? bc[i + 4] : bc[i + 5];
                        else if (s == 2) b = top ? bc[i + 4] : bu[i];
                        else if (s == 3) b = bot ? bc[i + 4] : bd[i];
                        else b = bc[i + 4];
                        const float q = ncc2(N, a4[i], b);
                        if (k == 0) Q[s][i] = q;
                        else if (k == 1) Q[s][i] = q + Q[s][i];
                        else Q[s][i] = (Q[s][i] + q) / 3.0f;
                    }
                }
            }
        }
    }

    // ---- epilogue: parabola x/y, correlation product, update, confidence blend --------------
    if (trow < TY) {
        const int gy = y0 + trow, gx0 = x0 + qx * 4;
        if (gy < H) {
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const int gx = gx0 + i;
                if (gx < W) {
                    const size_t at = (size_t)gy * W + gx;
                    float ddx, ddy, cx_, cy_;
                    poly(Q[4][i], Q[0][i], Q[1][i], thr, ddx, cx_);
                    poly(Q[4][i], Q[2][i], Q[3][i], thr, ddy, cy_);
                    float kap = cy_ * cx_;
                    const float ndx = d3[at] + ddx;
                    const float ndy = d3[n + at] + ddy;
                    if (blend) kap = blend_conf(d3[2 * n + at], kap);
                    nd3[at] = ndx;
                    nd3[n + at] = ndy;
                    nd3[2 * n + at] = kap;
                }
            }
        }
    }
}

// =========================================================================================
// K-smooth: P (<=5) Jacobi passes of smoothKernel (MatchLib.cu:1092-1145) and, optionally, the
// box filter (convolutionRows/ColumnsKernelTa, :1593-1697) in one launch.
// =========================================================================================
//
// Tile STX x STY outputs, halo h = P (+2 when the box follows: its outer taps have weight 0 but
// are still multiplied, exactly as in the reference).  The three fields live in LDS for the whole
// tile+halo; each pass computes into registers, barrier, writes back, barrier.  A pass leaves
// pixels of global row 0 / column 0 untouched (ix>0 && iy>0 guard) and clamps x+1 / y+1 at the
// image edge.  Values outside the image are never needed: every tap is either inside the image or
// clamped onto it.
template <int STX, int STY, int NT>
__global__ __launch_bounds__(NT) void k_smooth_fused(const float *__restrict__ s3, float *__restrict__ o3, int W, int H, int P, int do_box)
{
    constexpr int HMAX = 7;
    constexpr int LW = STX + 2 * HMAX + 2;  // +2 keeps the row stride off a multiple of 32 banks
    constexpr int LH = STY + 2 * HMAX;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float *f0 = smem, *f1 = smem + LH * LW, *f2 = smem + 2 * LH * LW;

    const int tid = threadIdx.x;
    const int h = P + (do_box ? 2 : 0);
    const int x0 = blockIdx.x * STX - h, y0 = blockIdx.y * STY - h;  // global coords of LDS (0,0)
    const int rw = STX + 2 * h, rh = STY + 2 * h;                  // region held in LDS
    const size_t n = (size_t)W * H;

    // load region; positions outside the image hold the clamped pixel (never used un-clamped)
    for (int it = tid; it < rw * rh; it += NT) {
        const int r = it / rw, c = it - r * rw;
        const int gx = clampi(x0 + c, 0, W - 1), gy = clampi(y0 + r, 0, H - 1);
        const size_t at = (size_t)gy * W + gx;
        f0[r * LW + c] = s3[at];
        f1[r * LW + c] = s3[n + at];
        f2[r * LW + c] = s3[2 * n + at];
    }
    __syncthreads();

    constexpr int MAXPX = ((STX + 2 * HMAX) * (STY + 2 * HMAX) + NT - 1) / NT;
    for (int p = 1; p <= P; p++) {
        // after pass p the valid region is the loaded region shrunk by p on every side
        const int lo = p, wv = rw - 2 * p, hv = rh - 2 * p;
        float n0[MAXPX], n1[MAXPX], n2[MAXPX];
#pragma unroll
        for (int u = 0; u < MAXPX; u++) {
            const int it = tid + u * NT;
            if (it < wv * hv) {
                const int r = lo + it / wv, c = lo + it % wv;
                const int gx = x0 + c, gy = y0 + r;
                const int at = r * LW + c;
                float v0 = f0[at], v1 = f1[at], v2 = f2[at];
                if (gx > 0 && gy > 0 && gx < W && gy < H) {
                    const int ae = (gx + 1 <= W - 1) ? at + 1 : at;
                    const int as = (gy + 1 <= H - 1) ? at + LW : at;
                    const int aw = at - 1, an = at - LW;
                    const float wc = f2[at], ww = f2[aw], we = f2[ae], wn = f2[an], ws = f2[as];
                    float sumCorr = 0.0f;
                    sumCorr = sumCorr + wc;
                    sumCorr = sumCorr + ww;
                    sumCorr = sumCorr + we;
                    sumCorr = sumCorr + wn;
                    sumCorr = sumCorr + ws;
                    float a0 = 0.0f, a1 = 0.0f, a2 = 0.0f;
                    a0 = f0[at] * wc + a0; a1 = f1[at] * wc + a1; a2 = wc * wc + a2;
                    a0 = f0[aw] * ww + a0; a1 = f1[aw] * ww + a1; a2 = ww * ww + a2;
                    a0 = f0[ae] * we + a0; a1 = f1[ae] * we + a1; a2 = we * we + a2;
                    a0 = f0[an] * wn + a0; a1 = f1[an] * wn + a1; a2 = wn * wn + a2;
                    a0 = f0[as] * ws + a0; a1 = f1[as] * ws + a1; a2 = ws * ws + a2;
                    v0 = a0 / sumCorr;
                    v1 = a1 / sumCorr;
                    v2 = a2 / sumCorr;
                }
                n0[u] = v0; n1[u] = v1; n2[u] = v2;
            }
        }
        __syncthreads();
#pragma unroll
        for (int u = 0; u < MAXPX; u++) {
            const int it = tid + u * NT;
            if (it < wv * hv) {
                const int at = (lo + it / wv) * LW + lo + it % wv;
                f0[at] = n0[u]; f1[at] = n1[u]; f2[at] = n2[u];
            }
        }
        __syncthreads();
    }

    if (do_box) {
        // rows (Ta) on the tile + 2 rows above/below, rounded to f32, then columns (Ta)
        const int lo = P;  // valid region after the passes starts at P; box row pass needs +-2 columns
        float *t0 = f0, *t1 = f1, *t2 = f2;
        constexpr int BMAX = (STX * (STY + 4) + NT - 1) / NT;
        float b0[BMAX], b1[BMAX], b2[BMAX];
#pragma unroll
        for (int u = 0; u < BMAX; u++) {
            const int it = tid + u * NT;
            if (it < STX * (STY + 4)) {
                const int r = lo + it / STX, c = lo + 2 + it % STX;  // rows tile-2 .. tile+STY+1, tile columns
                const int gx = x0 + c, gy = clampi(y0 + r, 0, H - 1);
                const int rr = gy - y0;  // clamp rows onto the image (texture clamp)
                if (gx < W) {
                    int cm2 = clampi(gx - 2, 0, W - 1) - x0, cm1 = clampi(gx - 1, 0, W - 1) - x0;
                    int cp1 = clampi(gx + 1, 0, W - 1) - x0, cp2 = clampi(gx + 2, 0, W - 1) - x0;
                    const float *q0 = t0 + rr * LW, *q1 = t1 + rr * LW, *q2 = t2 + rr * LW;
                    b0[u] = box5(q0[cm2], q0[cm1], q0[c], q0[cp1], q0[cp2]);
                    b1[u] = box5(q1[cm2], q1[cm1], q1[c], q1[cp1], q1[cp2]);
                    b2[u] = box5(q2[cm2], q2[cm1], q2[c], q2[cp1], q2[cp2]);
                }
            }
        }
        __syncthreads();
#pragma unroll
        for (int u = 0; u < BMAX; u++) {
            const int it = tid + u * NT;
            if (it < STX * (STY + 4)) {
                const int r = lo + it / STX, c = lo + 2 + it % STX;
                if (x0 + c < W) {
                    t0[r * LW + c] = b0[u]; t1[r * LW + c] = b1[u]; t2[r * LW + c] = b2[u];
                }
            }
        }
        __syncthreads();
        for (int it = tid; it < STX * STY; it += NT) {
            const int r = h + it / STX, c = h + it % STX;
            const int gx = x0 + c, gy = y0 + r;
            if (gx < W && gy < H) {
                // row-pass image rows gy-2..gy+2 clamped: a clamped row's row-pass equals the edge row's
                const int rm2 = clampi(gy - 2, 0, H - 1) - y0, rm1 = clampi(gy - 1, 0, H - 1) - y0;
                const int rp1 = clampi(gy + 1, 0, H - 1) - y0, rp2 = clampi(gy + 2, 0, H - 1) - y0;
                const size_t at = (size_t)gy * W + gx;
                o3[at] = box5(t0[rm2 * LW + c], t0[rm1 * LW + c], t0[r * LW + c], t0[rp1 * LW + c], t0[rp2 * LW + c]);
                o3[n + at] = box5(t1[rm2 * LW + c], t1[rm1 * LW + c], t1[r * LW + c], t1[rp1 * LW + c], t1[rp2 * LW + c]);
                o3[2 * n + at] = box5(t2[rm2 * LW + c], t2[rm1 * LW + c], t2[r * LW + c], t2[rp1 * LW + c], t2[rp2 * LW + c]);
            }
        }
    } else {
        for (int it = tid; it < STX * STY; it += NT) {
            const int r = h + it / STX, c = h + it % STX;
            const int gx = x0 + c, gy = y0 + r;
            if (gx < W && gy < H) {
                const size_t at = (size_t)gy * W + gx;
                o3[at] = f0[r * LW + c];
                o3[n + at] = f1[r * LW + c];
                o3[2 * n + at] = f2[r * LW + c];
            }
        }
    }
}

// ---- launchers ------------------------------------------------------------------------------

void launch_cost_fused(hipStream_t st, Img3 L, Img3 R, const float *A3, const float *d3, float *nd3, int W, int H, float thr, int blend)
{
    dim3 grid((W + TX - 1) / TX, (H + TY - 1) / TY);
    hipLaunchKernelGGL(k_cost_fused, grid, dim3(256), 0, st, L, R, A3, d3, nd3, W, H, thr, blend);
}

template <int STX, int STY, int NT>
static void launch_smooth_t(hipStream_t st, const float *s3, float *o3, int W, int H, int passes, int do_box)
{
    constexpr int LW = STX + 16, LH = STY + 14;
    constexpr size_t bytes = 3 * (size_t)LH * LW * sizeof(float);
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&k_smooth_fused<STX, STY, NT>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
        attr_set = true;
    }
    dim3 grid((W + STX - 1) / STX, (H + STY - 1) / STY);
    hipLaunchKernelGGL((k_smooth_fused<STX, STY, NT>), grid, dim3(NT), bytes, st, s3, o3, W, H, passes, do_box);
}

void launch_smooth_fused(hipStream_t st, const float *s3, float *o3, int W, int H, int passes, int do_box)
{
    // big levels: 128x64 tiles (halo redundancy ~1.2x); small levels: 64x32 so the chip still fills
    if ((size_t)W * H >= (size_t)1 << 20) launch_smooth_t<128, 64, 1024>(st, s3, o3, W, H, passes, do_box);
    else launch_smooth_t<64, 32, 256>(st, s3, o3, W, H, passes, do_box);
}

void launch_blur_decimate(hipStream_t st, const float *src3, int W, int H, float *dst3, int W2, int H2, float sf)
{
    launch_blur_decimate_ref(st, src3, W, H, dst3, W2, H2, sf);
}
void launch_sqblur_clamp(hipStream_t st, Img3 src, int W, int H, float *dst3) { launch_sqblur_clamp_ref(st, src, W, H, dst3); }

}  // namespace ugsm
