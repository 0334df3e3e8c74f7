// ugsm_kernels_fused.hip -- kernel_path 0: the production gfx950 kernels.
//
// K-cost  (k_cost_split; k_cost_fused = the one-thread-per-quad variant kept for A/B):
//                           one matcher iteration's warp + 5-shift squared-NCC cost (3 channels) +
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
#include "ugsm_exact.hpp"
#include "ugsm_launch.hpp"
#include <atomic>
#include <type_traits>

namespace ugsm {

// =========================================================================================
// K-cost
// =========================================================================================
//
// Tile TX x TY = 32 x 28 output pixels per workgroup: 256 threads in k_cost_fused (a thread owns a "quad", 4
// consecutive x, so that every LDS access is a 16-byte ds_read/write_b128; thread (row = tid&31, qx = tid>>5)
// -> tile row `row`, quad column qx), 512 in the production k_cost_split (two threads per quad, below).
//
// LDS images (float, tile-relative column c stored at [c + OX]):
//   sR    [34][52]  R' = warped right plane, tile+halo3, edge-replicated (texture clamp)   OX=8
//   sL    [32][44]  left plane, tile+halo2, ZERO outside the image (smem-conv zero padding) OX=4
//   sRow  [5][32][36] row-pass of the five product images, rows tile+halo2
//   sBrow [34][44]  row-pass of R'^2, rows tile+halo3, cols tile+halo4                       OX=4
//   sB    [30][44]  B = G_clamp*(R'^2), tile+halo1 (only in-image entries are ever read)     OX=4
//   sA    [28][36]  A = G_clamp*(L^2) of the tile, current channel
// = 51 KB; k_cost_split: 96 VGPRs -> 2 workgroups (16 waves) per CU.
//
// Per channel: P1 fill sL,sR | barrier | P2 row passes | barrier | P2.5 B column pass | barrier |
// P3 column pass of the 5 products + correlation, accumulated over channels in registers.
// Row strides are an ODD number of quads (52, 44, 36 floats) and a wave's lanes walk DOWN the rows
// (row = tid & 31, quad column = tid >> 5): the 16-lane groups of ds_read_b128 then hit 16 distinct
// 4-bank slots (odd multiplier mod 16 is a bijection) -- conflict-free; with lanes walking along a
// row the same reads cost 2-3x (rows of 40/48 floats alias in the 64 banks).  Measured:
// SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE still reads 47 %; tools/ldsbench.hip times these layouts within 5 % of
// the conflict-free floor.
constexpr int TX = 32, TY = 28;
constexpr int SR_W = TX + 20, SR_H = TY + 6, SR_OX = 8;
constexpr int SL_W = TX + 12, SL_H = TY + 4, SL_OX = 4;
constexpr int ROW_W = TX + 4, ROW_H = TY + 4;
constexpr int SB_W = TX + 12, SB_Q = (TX + 8) / 4, SBROW_H = TY + 6, SB_H = TY + 2, SB_OX = 4;
constexpr int IDX_W = TX + 6, IDX_H = TY + 6;  // tile + halo 3: the pixels whose warped fetch the tile needs


// ABL: development-only ablation mask (tools/kbench.hip times variants with phases removed to see
// where the time goes); the product instantiates ABL = 0 only.
template <int ABL>
__global__ __launch_bounds__(256) void k_cost_fused(Img3 L, Img3 R, const float *__restrict__ A3, const float *__restrict__ d3,
                                                    float *__restrict__ nd3, int W, int H, float thr, int blend, int tiles_x, int n_tiles)
{
    __shared__ __attribute__((aligned(16))) float sR[SR_H * SR_W];
    __shared__ __attribute__((aligned(16))) float sL[SL_H * SL_W];
    __shared__ __attribute__((aligned(16))) float sRow[5 * ROW_H * ROW_W];
    __shared__ __attribute__((aligned(16))) float sBrow[SBROW_H * SB_W];
    __shared__ __attribute__((aligned(16))) float sB[SB_H * SB_W];
    __shared__ __attribute__((aligned(16))) float sA[TY * ROW_W];  // A = G_clamp*(L^2) of the tile, current channel

    const int tid = threadIdx.x;
    int tile_x, tile_y;
    xcd_tile(n_tiles, tiles_x, tile_x, tile_y);
    const int x0 = tile_x * TX, y0 = tile_y * TY;
    const size_t n = (size_t)W * H;
    const int trow = tid & 31, qx = tid >> 5;  // 32 rows x 8 quad columns, lanes walk down the rows

    // ---- P0: every global read of the tile is issued up front and parked in registers, so the
    // three channel rounds below touch LDS only (one exposed memory latency per workgroup instead
    // of one per channel).  A thread stages the same halo items for all three channels.
    constexpr int NR = (IDX_H * IDX_W + 255) / 256;        // R' items per thread (tile+halo3)
    constexpr int NL = (SL_H * (TX + 4) + 255) / 256;      // L items per thread (tile+halo2)
    float rv[3][NR], lv[3][NL];
    int ridx[NR];
#pragma unroll
    for (int u = 0; u < NR; u++) {  // warped source offsets (warpAbyB, MatchLib.cu:510-515)
        const int it = tid + u * 256;
        ridx[u] = -1;
        if (it < IDX_H * IDX_W) {
            const int r = it / IDX_W, c = it - r * IDX_W;
            const int gx = clampi(x0 + c - 3, 0, W - 1), gy = clampi(y0 + r - 3, 0, H - 1);
            const size_t at = (size_t)gy * W + gx;
            if constexpr (ABL & (1 | 64)) {
                ridx[u] = gy * R.pitch + gx;
            } else {
                const int sx = tex_index(((float)gx + 0.5f) + d3[at], W);
                const int sy = tex_index(((float)gy + 0.5f) + d3[n + at], H);
                ridx[u] = sy * R.pitch + sx;
            }
        }
    }
    // Issue order is CHANNEL-major: vmcnt retires in order, so the wait in front of channel 0's P1
    // covers only channel 0's loads and the other two channels land while channel 0 is computed.
    constexpr int NA = (TX * TY + 255) / 256;
    float aq[3][NA], od[3][NA];
#pragma unroll
    for (int k = 0; k < 3; k++) {
#pragma unroll
        for (int u = 0; u < NR; u++) {
            if constexpr (ABL & 1) rv[k][u] = (float)(ridx[u] & 255) + k;
            else rv[k][u] = (ridx[u] >= 0) ? R.p[(size_t)k * R.plane + ridx[u]] : 0.0f;
        }
#pragma unroll
        for (int u = 0; u < NL; u++) {
            const int it = tid + u * 256;
            const int r = it / (TX + 4), c = it - r * (TX + 4);  // c: tile column + 2
            const int gx = x0 + c - 2, gy = y0 + r - 2;
            const bool in = it < SL_H * (TX + 4) && gx >= 0 && gx < W && gy >= 0 && gy < H;
            if constexpr (ABL & (1 | 128)) lv[k][u] = in ? (float)(gx + k) : 0.0f;
            else lv[k][u] = in ? L.p[(size_t)k * L.plane + (size_t)gy * L.pitch + gx] : 0.0f;
        }
        // A of the tile: read with lanes along the rows (coalesced 128-B segments) and handed to the
        // row-walking compute threads through LDS.  Reading it -- or (dx,dy,conf), or writing the result --
        // directly in the compute mapping costs one cache line per LANE: 32 lines per wave instruction,
        // which made the texture-address unit the bottleneck (ablation: -106 us of 636 at 16 MP).
#pragma unroll
        for (int u = 0; u < NA; u++) {
            const int it = tid + u * 256;
            const int r = it / TX, c = it - r * TX;
            const bool in = it < TX * TY && x0 + c < W && y0 + r < H;
            const size_t at = in ? (size_t)(y0 + r) * W + x0 + c : 0;
            if constexpr (ABL & (1 | 128)) aq[k][u] = 1.0e4f + c;
            else aq[k][u] = in ? A3[k * n + at] : 1.0f;
        }
    }
    // the tile's own (dx, dy, conf) for the update/blend at the very end: issued now, used last
#pragma unroll
    for (int u = 0; u < NA; u++) {
        const int it = tid + u * 256;
        const int r = it / TX, c = it - r * TX;
        const bool in = it < TX * TY && x0 + c < W && y0 + r < H;
        const size_t at = in ? (size_t)(y0 + r) * W + x0 + c : 0;
#pragma unroll
        for (int k = 0; k < 3; k++) {
            if constexpr (ABL & (1 | 128)) od[k][u] = 0.5f;
            else od[k][u] = in ? d3[k * n + at] : 0.0f;
        }
    }

    float Q[5][4];
#pragma unroll
    for (int s = 0; s < 5; s++)
#pragma unroll
        for (int i = 0; i < 4; i++) Q[s][i] = 0.0f;

#pragma unroll
    for (int k = 0; k < 3; k++) {
        // No barrier needed here: P1 rewrites sL/sR, whose last readers (P2 of the previous channel) are
        // two barriers back; sRow/sB (still being read by slower threads' P3) are rewritten only after
        // the barrier that ends P1.
        // ---- P1: stage L (zero outside) and R' (gather, edge replicated) into LDS ------------
#pragma unroll
        for (int u = 0; u < NR; u++) {
            const int it = tid + u * 256;
            if (it < IDX_H * IDX_W) {
                const int r = it / IDX_W, c = it - r * IDX_W;
                sR[r * SR_W + (c - 3 + SR_OX)] = rv[k][u];
            }
        }
#pragma unroll
        for (int u = 0; u < NL; u++) {
            const int it = tid + u * 256;
            if (it < SL_H * (TX + 4)) {
                const int r = it / (TX + 4), c = it - r * (TX + 4);
                sL[r * SL_W + (c - 2 + SL_OX)] = lv[k][u];
            }
        }
        __syncthreads();
        // sA is rewritten only now: its readers (P3 of the previous channel) are all behind the barrier above
#pragma unroll
        for (int u = 0; u < NA; u++) {
            const int it = tid + u * 256;
            if (it < TX * TY) sA[(it / TX) * ROW_W + (it % TX)] = aq[k][u];
        }
        // ---- P2a: row pass of the five products (CompareMove + convolutionRowsKernel) -----
        if constexpr (!(ABL & 2)) {
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
                const float lvj = l[j + 2];
                p[0][j] = lvj * rc[j + 1];  // shift (-1, 0)
                p[1][j] = lvj * rc[j + 3];  // shift (+1, 0)
                p[2][j] = lvj * ru[j + 2];  // shift (0, -1)
                p[3][j] = lvj * rd[j + 2];  // shift (0, +1)
                p[4][j] = lvj * rc[j + 2];  // shift (0, 0)
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
        for (int it = tid; it < SBROW_H * SB_Q; it += 256) {
            const int q = it / SBROW_H, r = it - q * SBROW_H;  // lanes walk down the rows
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
        }
        __syncthreads();
        // ---- P2.5: column pass of R'^2 -> B on tile+halo1 (convolutionColumnsKernelT) -----
        if constexpr (!(ABL & 4)) {
        for (int it = tid; it < SB_H * SB_Q; it += 256) {
            const int q = it / SB_H, r = it - q * SB_H;  // r: tile row + 1; lanes walk down the rows
            float a[4], b[4], c[4], d[4], e[4], o[4];
            const float *pb = &sBrow[r * SB_W + q * 4];  // sBrow row index = tile row + 3; rows r .. r+4
            ld4(pb, a); ld4(pb + SB_W, b); ld4(pb + 2 * SB_W, c); ld4(pb + 3 * SB_W, d); ld4(pb + 4 * SB_W, e);
#pragma unroll
            for (int i = 0; i < 4; i++) o[i] = tap5p(a[i], b[i], c[i], d[i], e[i]);
            st4(&sB[r * SB_W + q * 4], o);
        }
        }
        __syncthreads();
        // ---- P3: column pass of the products, correlation, channel accumulate ---------------
        if constexpr ((ABL & 8)) {
        Q[k][0] += sRow[tid] + sB[tid];
        } else {
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
                ld4(&sA[trow * ROW_W + cx], a4);
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
                        float q;
                        if constexpr (ABL & 16) q = N * a4[i] * b;
                        else q = ncc2_nn(N, a4[i], b);
                        if (k == 0) Q[s][i] = q;
                        else if (k == 1) Q[s][i] = q + Q[s][i];
                        else Q[s][i] = div3_nonneg(Q[s][i] + q);
                    }
                }
            }
        }
        }
    }

    // ---- epilogue: parabola x/y and correlation product in the compute mapping, then the update,
    // confidence blend and all global I/O with lanes along the rows (coalesced), through LDS ------
    __syncthreads();  // every P3 is done with sRow: planes 0..2 become the hand-over buffers
    if (trow < TY) {
        float ex[4], ey[4], ek[4];
#pragma unroll
        for (int i = 0; i < 4; i++) {
            float cx_, cy_;
            if constexpr (ABL & 32) {
                ex[i] = Q[0][i] + Q[1][i] + Q[4][i]; ey[i] = Q[2][i] + Q[3][i]; cx_ = thr; cy_ = ex[i];
            } else {
                poly_fast(Q[4][i], Q[0][i], Q[1][i], thr, ex[i], cx_);
                poly_fast(Q[4][i], Q[2][i], Q[3][i], thr, ey[i], cy_);
            }
            ek[i] = cy_ * cx_;
        }
        st4(&sRow[(0 * ROW_H + trow) * ROW_W + qx * 4], ex);
        st4(&sRow[(1 * ROW_H + trow) * ROW_W + qx * 4], ey);
        st4(&sRow[(2 * ROW_H + trow) * ROW_W + qx * 4], ek);
    }
    __syncthreads();
#pragma unroll
    for (int u = 0; u < NA; u++) {
        const int it = tid + u * 256;
        const int r = it / TX, c = it - r * TX;
        const int gx = x0 + c, gy = y0 + r;
        if (it < TX * TY && gx < W && gy < H) {
            const size_t at = (size_t)gy * W + gx;
            const float ddx = sRow[(0 * ROW_H + r) * ROW_W + c], ddy = sRow[(1 * ROW_H + r) * ROW_W + c];
            float kap = sRow[(2 * ROW_H + r) * ROW_W + c];
            if constexpr (!(ABL & 32)) {
                if (blend) kap = blend_conf(od[2][u], kap);
            }
            nd3[at] = od[0][u] + ddx;
            nd3[n + at] = od[1][u] + ddy;
            nd3[2 * n + at] = kap;
        }
    }
}

// -----------------------------------------------------------------------------------------
// k_cost_split: the same tile and LDS images as k_cost_fused, but TWO threads per quad (512-thread
// workgroup): the LDS footprint per tile -- not registers -- caps the workgroups per CU, so the waves per LDS
// byte are doubled by splitting each quad's work by correlation shift:
//   role 0 (threads 0..255)   = shifts (-1,0), (+1,0), pixels 0,1 of shift (0,0), parabola x
//   role 1 (threads 256..511) = shifts (0,-1), (0,+1), pixels 2,3 of shift (0,0), the R'^2 row pass, parabola y
// (the B column pass is shared by both).  The roles are wave-uniform (no divergence).  They meet once per tile:
// each publishes its two pixels of Q(0,0), then the x / y parabola results (LDS exchanges in the dead sRow planes).
// Where the time goes (SQ counters, profiles/): 1 880 VALU instructions per wave, 0.23 per SIMD cycle at 4 waves/SIMD,
// about three quarters of the issue capacity at the measured ~3.3 cycles per instruction.
// development only (tools/kbench.hip, ABL & 256): s_memtime stamps at the phase boundaries of wave 0 (role 0) and
// wave 4 (role 1) of the first workgroups, 16 stamps each
__device__ long long *g_cost_stamps = nullptr;
#define UGSM_STAMP(i)                                                                                              \
    do {                                                                                                           \
        if constexpr (ABL & 256) {                                                                                 \
            if ((tid & 255) == 0 && blockIdx.x < 8192 && g_cost_stamps)                                            \
                g_cost_stamps[((size_t)blockIdx.x * 2 + (tid >> 8)) * 24 + (i)] = (long long)__builtin_readcyclecounter(); \
        }                                                                                                          \
    } while (0)

#ifndef UGSM_SPLIT_WAVES
#define UGSM_SPLIT_WAVES 4  // waves per SIMD the register allocation aims at (2 workgroups per CU)
#endif
// The body of k_cost_split.  INTERIOR: the tile and its halo of 3 lie inside the image, so the address clamps of P0, the
// zero-padding and validity selects, the clamped B fetches of P3 and the store bounds are compiled out (a workgroup-uniform
// choice made by the kernel below; about 95 % of the tiles of a 16 MP level).
template <int ABL, bool INTERIOR>
__device__ __forceinline__ void cost_split_body(const Img3 &L, const Img3 &R, const float *__restrict__ A3, const float *__restrict__ d3,
                                                float *__restrict__ nd3, const int W, const int H, const float thr, const int blend, const int x0,
                                                const int y0, float *__restrict__ sR, float *__restrict__ sL, float *__restrict__ sRow,
                                                float *__restrict__ sBrow, float *__restrict__ sB, float *__restrict__ sA)
{
    const int tid = threadIdx.x;
    const int role = tid >> 8, t = tid & 255;
    const size_t n = (size_t)W * H;
    const int trow = t & 31, qx = t >> 5;  // 32 rows x 8 quad columns, lanes walk down the rows
    const int cx = qx * 4;
    const int gy = y0 + trow, gx0 = x0 + cx;
    const bool live = INTERIOR ? (trow < TY) : (trow < TY && gy < H && gx0 < W);
    UGSM_STAMP(0);

    // ---- P0: all global reads of the tile up front (see k_cost_fused).  Every load is unconditional on a
    // clamped address and the out-of-range value is selected afterwards: loads inside `if (in)` blocks were
    // compiled to one branch per load, and the six (dx,dy) loads that the warp addresses depend on to six
    // load -> s_waitcnt vmcnt(0) round trips in a row (8.4k of the 48k cycles a tile takes, tools/kbench stamps).
    // Addresses are a 32-bit byte offset against a uniform plane base (a plane is < 4 GiB).
    constexpr int NR = (IDX_H * IDX_W + 511) / 512;
    constexpr int NL = (SL_H * (TX + 4) + 511) / 512;
    gchar_c *const Lb[3] = {uniform_base(L.p), uniform_base(L.p + L.plane), uniform_base(L.p + 2 * L.plane)};
    gchar_c *const Rb[3] = {uniform_base(R.p), uniform_base(R.p + R.plane), uniform_base(R.p + 2 * R.plane)};
    gchar_c *const Ab[3] = {uniform_base(A3), uniform_base(A3 + n), uniform_base(A3 + 2 * n)};
    gchar_c *const Db[3] = {uniform_base(d3), uniform_base(d3 + n), uniform_base(d3 + 2 * n)};
    float rv[3][NR], lv[3][NL];
    int ridx[NR];
    {
        float ddx[NR], ddy[NR];
        int gxh[NR], gyh[NR];
#pragma unroll
        for (int u = 0; u < NR; u++) {  // (dx,dy) at the pixels whose warped fetch the tile needs
            const int it = min(tid + u * 512, IDX_H * IDX_W - 1);
            const int r = it / IDX_W, c = it - r * IDX_W;
            gxh[u] = INTERIOR ? x0 + c - 3 : clampi(x0 + c - 3, 0, W - 1);
            gyh[u] = INTERIOR ? y0 + r - 3 : clampi(y0 + r - 3, 0, H - 1);
            const unsigned off = ((unsigned)gyh[u] * (unsigned)W + (unsigned)gxh[u]) * 4u;
            ddx[u] = ld_at(Db[0], off);
            ddy[u] = ld_at(Db[1], off);
        }
        UGSM_STAMP(16);
#pragma unroll
        for (int u = 0; u < NL; u++) {
            const int it = min(tid + u * 512, SL_H * (TX + 4) - 1);
            const int r = it / (TX + 4), c = it - r * (TX + 4);  // c: tile column + 2
            const int gxl = x0 + c - 2, gyl = y0 + r - 2;
            const bool in = INTERIOR || (gxl >= 0 && gxl < W && gyl >= 0 && gyl < H);
            const unsigned off = INTERIOR ? ((unsigned)gyl * (unsigned)L.pitch + (unsigned)gxl) * 4u
                                          : ((unsigned)clampi(gyl, 0, H - 1) * (unsigned)L.pitch + (unsigned)clampi(gxl, 0, W - 1)) * 4u;
#pragma unroll
            for (int k = 0; k < 3; k++) {
                const float t = ld_at(Lb[k], off);
                lv[k][u] = in ? t : 0.0f;
            }
        }
        UGSM_STAMP(17);
#pragma unroll
        for (int u = 0; u < NR; u++) {  // warped source offsets (warpAbyB, MatchLib.cu:510-515)
            const int sx = tex_index(((float)gxh[u] + 0.5f) + ddx[u], W);
            const int sy = tex_index(((float)gyh[u] + 0.5f) + ddy[u], H);
            ridx[u] = (sy * R.pitch + sx) * 4;
        }
    }
#pragma unroll
    for (int u = 0; u < NR; u++)
#pragma unroll
        for (int k = 0; k < 3; k++) rv[k][u] = ld_at(Rb[k], (unsigned)ridx[u]);
    UGSM_STAMP(18);
    constexpr int NA = (TX * TY + 511) / 512;
    float aq[3][NA], od[3][NA];  // A and the tile's own (dx,dy,conf), lanes along the rows (coalesced)
#pragma unroll
    for (int u = 0; u < NA; u++) {
        const int it = tid + u * 512;
        const int r = it / TX, c = it - r * TX;
        const bool in = it < TX * TY && (INTERIOR || (x0 + c < W && y0 + r < H));
        const unsigned off = in ? ((unsigned)(y0 + r) * (unsigned)W + (unsigned)(x0 + c)) * 4u : 0u;
#pragma unroll
        for (int kk = 0; kk < 3; kk++) {
            const float ta = ld_at(Ab[kk], off), td = ld_at(Db[kk], off);
            aq[kk][u] = in ? ta : 1.0f;
            od[kk][u] = in ? td : 0.0f;
        }
    }

    float Q[3][4];  // role 0: shifts 0,1 and pixels 0,1 of shift 4; role 1: shifts 2,3 and pixels 2,3 of shift 4
#pragma unroll
    for (int s = 0; s < 3; s++)
#pragma unroll
        for (int i = 0; i < 4; i++) Q[s][i] = 0.0f;
    UGSM_STAMP(1);

#pragma unroll
    for (int k = 0; k < 3; k++) {
        // ---- P1: registers -> LDS (sR edge-replicated, sL zero outside the image) -------------
#pragma unroll
        for (int u = 0; u < NR; u++) {
            const int it = tid + u * 512;
            if (it < IDX_H * IDX_W) {
                const int r = it / IDX_W, c = it - r * IDX_W;
                sR[r * SR_W + (c - 3 + SR_OX)] = rv[k][u];
            }
        }
#pragma unroll
        for (int u = 0; u < NL; u++) {
            const int it = tid + u * 512;
            if (it < SL_H * (TX + 4)) {
                const int r = it / (TX + 4), c = it - r * (TX + 4);
                sL[r * SL_W + (c - 2 + SL_OX)] = lv[k][u];
            }
        }
        __syncthreads();
        UGSM_STAMP(2 + 4 * k);
#pragma unroll
        for (int u = 0; u < NA; u++) {  // sA's readers (P3 of the previous channel) are behind the barrier above
            const int it = tid + u * 512;
            if (it < TX * TY) sA[(it / TX) * ROW_W + (it % TX)] = aq[k][u];
        }
        // ---- P2: row passes -----------------------------------------------------------------------
        {
            const int r = trow;  // 0..31 <-> tile row r-2
            float l[12];
            const float *pl = &sL[r * SL_W + cx - 4 + SL_OX];
            ld4(pl, l); ld4(pl + 4, l + 4); ld4(pl + 8, l + 8);
            const float *pr = &sR[(r + 1) * SR_W + cx - 4 + SR_OX];  // sR row index = tile row + 3
            if (role == 0) {
                float rc[12];
                ld4(pr, rc); ld4(pr + 4, rc + 4); ld4(pr + 8, rc + 8);
                // arrays hold tile columns cx-4 .. cx+7; products at columns cx-2 .. cx+5
                float p0[8], p1[8], p4[8];
#pragma unroll
                for (int j = 0; j < 8; j++) {
                    const float lvj = l[j + 2];
                    p0[j] = lvj * rc[j + 1];  // shift (-1, 0)
                    p1[j] = lvj * rc[j + 3];  // shift (+1, 0)
                    p4[j] = lvj * rc[j + 2];  // shift (0, 0)
                }
                float o[4];
#pragma unroll
                for (int i = 0; i < 4; i++) o[i] = tap5p(p0[i], p0[i + 1], p0[i + 2], p0[i + 3], p0[i + 4]);
                st4(&sRow[(0 * ROW_H + r) * ROW_W + cx], o);
#pragma unroll
                for (int i = 0; i < 4; i++) o[i] = tap5p(p1[i], p1[i + 1], p1[i + 2], p1[i + 3], p1[i + 4]);
                st4(&sRow[(1 * ROW_H + r) * ROW_W + cx], o);
#pragma unroll
                for (int i = 0; i < 4; i++) o[i] = tap5p(p4[i], p4[i + 1], p4[i + 2], p4[i + 3], p4[i + 4]);
                st4(&sRow[(4 * ROW_H + r) * ROW_W + cx], o);
            } else {
                float ru[12], rd[12];
                ld4(pr - SR_W, ru); ld4(pr - SR_W + 4, ru + 4); ld4(pr - SR_W + 8, ru + 8);
                ld4(pr + SR_W, rd); ld4(pr + SR_W + 4, rd + 4); ld4(pr + SR_W + 8, rd + 8);
                float p2[8], p3[8];
#pragma unroll
                for (int j = 0; j < 8; j++) {
                    const float lvj = l[j + 2];
                    p2[j] = lvj * ru[j + 2];  // shift (0, -1)
                    p3[j] = lvj * rd[j + 2];  // shift (0, +1)
                }
                float o[4];
#pragma unroll
                for (int i = 0; i < 4; i++) o[i] = tap5p(p2[i], p2[i + 1], p2[i + 2], p2[i + 3], p2[i + 4]);
                st4(&sRow[(2 * ROW_H + r) * ROW_W + cx], o);
#pragma unroll
                for (int i = 0; i < 4; i++) o[i] = tap5p(p3[i], p3[i + 1], p3[i + 2], p3[i + 3], p3[i + 4]);
                st4(&sRow[(3 * ROW_H + r) * ROW_W + cx], o);
                // row pass of R'^2 (Square + convolutionRowsKernelT), cols -4..TX+3
                for (int it = t; it < SBROW_H * SB_Q; it += 256) {
                    const int q = it / SBROW_H, rr = it - q * SBROW_H;
                    const int cb = q * 4 - 4;
                    float v[12];
                    const float *pq = &sR[rr * SR_W + cb - 4 + SR_OX];
                    ld4(pq, v); ld4(pq + 4, v + 4); ld4(pq + 8, v + 8);
                    float sq[8];
#pragma unroll
                    for (int j = 0; j < 8; j++) sq[j] = v[j + 2] * v[j + 2];
#pragma unroll
                    for (int i = 0; i < 4; i++) o[i] = tap5p(sq[i], sq[i + 1], sq[i + 2], sq[i + 3], sq[i + 4]);
                    st4(&sBrow[rr * SB_W + cb + SB_OX], o);
                }
            }
        }
        __syncthreads();
        UGSM_STAMP(3 + 4 * k);
        // ---- P2.5: column pass of R'^2 -> B on tile+halo1: 300 quads, one per thread over both roles ----
        if (tid < SB_H * SB_Q) {
            const int q = tid / SB_H, r = tid - q * SB_H;  // r: tile row + 1
            float a[4], b[4], c[4], d[4], e[4], o[4];
            const float *pb = &sBrow[r * SB_W + q * 4];
            ld4(pb, a); ld4(pb + SB_W, b); ld4(pb + 2 * SB_W, c); ld4(pb + 3 * SB_W, d); ld4(pb + 4 * SB_W, e);
#pragma unroll
            for (int i = 0; i < 4; i++) o[i] = tap5p(a[i], b[i], c[i], d[i], e[i]);
            st4(&sB[r * SB_W + q * 4], o);
        }
        __syncthreads();
        UGSM_STAMP(4 + 4 * k);
        // ---- P3: column pass of the products, correlation, channel accumulate ---------------
        if (live) {
            const float *pb = &sB[(trow + 1) * SB_W + cx - 4 + SB_OX];
            float a4[4];
            ld4(&sA[trow * ROW_W + cx], a4);
            auto colpass = [&](int s, float *N) {
                float r0[4], r1[4], r2[4], r3[4], r4[4];
                const float *ps = &sRow[(s * ROW_H + trow) * ROW_W + cx];  // rows trow .. trow+4 <-> tile rows trow-2..trow+2
                ld4(ps, r0); ld4(ps + ROW_W, r1); ld4(ps + 2 * ROW_W, r2); ld4(ps + 3 * ROW_W, r3); ld4(ps + 4 * ROW_W, r4);
#pragma unroll
                for (int i = 0; i < 4; i++) N[i] = tap5p(r0[i], r1[i], r2[i], r3[i], r4[i]);
            };
            auto accum = [&](int slot, const float *N, const float *b) {
#pragma unroll
                for (int i = 0; i < 4; i++) {
                    const float q = ncc2_nn(N[i], a4[i], b[i]);
                    if (k == 0) Q[slot][i] = q;
                    else if (k == 1) Q[slot][i] = q + Q[slot][i];
                    else Q[slot][i] = div3_nonneg(Q[slot][i] + q);
                }
            };
            // shift (0,0): each role takes two pixels of the quad (role 0: 0,1; role 1: 2,3), which evens out the
            // three-shifts / two-shifts split of the phase; Q[2][2h], Q[2][2h+1] hold them
            auto half4 = [&](const int hsel, const float *bq) {
                float r0[2], r1[2], r2[2], r3[2], r4[2];
                const float *ps = &sRow[(4 * ROW_H + trow) * ROW_W + cx + 2 * hsel];
                ld2(ps, r0); ld2(ps + ROW_W, r1); ld2(ps + 2 * ROW_W, r2); ld2(ps + 3 * ROW_W, r3); ld2(ps + 4 * ROW_W, r4);
#pragma unroll
                for (int j = 0; j < 2; j++) {
                    const int i = 2 * hsel + j;
                    const float Nn = tap5p(r0[j], r1[j], r2[j], r3[j], r4[j]);
                    const float q = ncc2_nn(Nn, a4[i], bq[i]);
                    if (k == 0) Q[2][i] = q;
                    else if (k == 1) Q[2][i] = q + Q[2][i];
                    else Q[2][i] = div3_nonneg(Q[2][i] + q);
                }
            };
            float N[4], b[4];
            if (role == 0) {
                float bc[12];
                ld4(pb, bc); ld4(pb + 4, bc + 4); ld4(pb + 8, bc + 8);  // columns cx-4 .. cx+7, pixel i at [i+4]
                colpass(0, N);
#pragma unroll
                for (int i = 0; i < 4; i++) b[i] = (!INTERIOR && gx0 + i == 0) ? bc[i + 4] : bc[i + 3];
                accum(0, N, b);
                colpass(1, N);
#pragma unroll
                for (int i = 0; i < 4; i++) b[i] = (!INTERIOR && gx0 + i >= W - 1) ? bc[i + 4] : bc[i + 5];
                accum(1, N, b);
                half4(0, bc + 4);
            } else {
                float bm[4], bu[4], bd[4];
                ld4(pb + 4, bm); ld4(pb - SB_W + 4, bu); ld4(pb + SB_W + 4, bd);
                const bool top = !INTERIOR && (gy == 0), bot = !INTERIOR && (gy == H - 1);
                colpass(2, N);
#pragma unroll
                for (int i = 0; i < 4; i++) b[i] = top ? bm[i] : bu[i];
                accum(0, N, b);
                colpass(3, N);
#pragma unroll
                for (int i = 0; i < 4; i++) b[i] = bot ? bm[i] : bd[i];
                accum(1, N, b);
                half4(1, bm);
            }
        }
        UGSM_STAMP(5 + 4 * k);
    }

    // ---- epilogue: parabola x (role 0) / y (role 1) in the compute mapping, hand-over through LDS, then
    // update + blend + coalesced stores with lanes along the rows ---------------------------------------
    __syncthreads();  // every P3 is done with sRow: planes 0..3 become hand-over buffers
    UGSM_STAMP(14);
    float *xq = &sRow[(0 * ROW_H + trow) * ROW_W + cx];
    if (live) {  // Q(0,0): each role publishes its two pixels, both parabolas need the four
        if (role == 0) { xq[0] = Q[2][0]; xq[1] = Q[2][1]; }
        else { xq[2] = Q[2][2]; xq[3] = Q[2][3]; }
    }
    __syncthreads();
    if (live) {
        float c4[4], dd[4], rho[4];
        ld4(xq, c4);
#pragma unroll
        for (int i = 0; i < 4; i++) poly_fast(c4[i], Q[0][i], Q[1][i], thr, dd[i], rho[i]);  // x: shifts (-1,0),(+1,0); y: (0,-1),(0,+1)
        st4(&sRow[((1 + role) * ROW_H + trow) * ROW_W + cx], dd);   // plane 1: delta x, plane 2: delta y
        st4(&sRow[((3 + role) * ROW_H + trow) * ROW_W + cx], rho);  // plane 3: rho x,  plane 4: rho y
    }
    __syncthreads();
    gchar_c *const Nb[3] = {uniform_base(nd3), uniform_base(nd3 + n), uniform_base(nd3 + 2 * n)};
#pragma unroll
    for (int u = 0; u < NA; u++) {
        const int it = tid + u * 512;
        const int r = it / TX, c = it - r * TX;
        const int gxo = x0 + c, gyo = y0 + r;
        if (it < TX * TY && (INTERIOR || (gxo < W && gyo < H))) {
            const unsigned off = ((unsigned)gyo * (unsigned)W + (unsigned)gxo) * 4u;
            const float ddx = sRow[(1 * ROW_H + r) * ROW_W + c], ddy = sRow[(2 * ROW_H + r) * ROW_W + c];
            float kap = sRow[(4 * ROW_H + r) * ROW_W + c] * sRow[(3 * ROW_H + r) * ROW_W + c];  // rho_y * rho_x
            if (blend) kap = blend_conf(od[2][u], kap);
            st_at(Nb[0], off, od[0][u] + ddx);
            st_at(Nb[1], off, od[1][u] + ddy);
            st_at(Nb[2], off, kap);
        }
    }
    UGSM_STAMP(15);
}

template <int ABL, int WAVES = UGSM_SPLIT_WAVES>
__global__ __launch_bounds__(512, WAVES) void k_cost_split(Img3 L, Img3 R, const float *__restrict__ A3, const float *__restrict__ d3,
                                                    float *__restrict__ nd3, int W, int H, float thr, int blend, int tiles_x, int n_tiles)
{
    __shared__ __attribute__((aligned(16))) float sR[SR_H * SR_W];
    __shared__ __attribute__((aligned(16))) float sL[SL_H * SL_W];
    __shared__ __attribute__((aligned(16))) float sRow[5 * ROW_H * ROW_W];
    __shared__ __attribute__((aligned(16))) float sBrow[SBROW_H * SB_W];
    __shared__ __attribute__((aligned(16))) float sB[SB_H * SB_W];
    __shared__ __attribute__((aligned(16))) float sA[TY * ROW_W];
    int tile_x, tile_y;
    xcd_tile(n_tiles, tiles_x, tile_x, tile_y);
    const int x0 = tile_x * TX, y0 = tile_y * TY;
#ifndef UGSM_COST_INTERIOR
#define UGSM_COST_INTERIOR 1
#endif
    const bool interior = UGSM_COST_INTERIOR && !(ABL & 512) && x0 >= 3 && y0 >= 3 && x0 + TX + 3 <= W && y0 + TY + 3 <= H;
    if (interior) cost_split_body<ABL, true>(L, R, A3, d3, nd3, W, H, thr, blend, x0, y0, sR, sL, sRow, sBrow, sB, sA);
    else cost_split_body<ABL, false>(L, R, A3, d3, nd3, W, H, thr, blend, x0, y0, sR, sL, sRow, sBrow, sB, sA);
}


// =========================================================================================
// K-smooth: P (<=5) Jacobi passes of smoothKernel (MatchLib.cu:1092-1145) and, optionally, the
// box filter (convolutionRows/ColumnsKernelTa, :1593-1697) in one launch.
// =========================================================================================
//
// Tile STX x STY outputs.  The three fields (dx, dy, conf) of tile + halo live in LDS for the whole
// launch: region columns [tileX0-8, tileX0+STX+8), rows [tileY0-7, tileY0+STY+7) -- halo 7 = 5 passes
// + 2 for the box (its outer taps have weight 0 but are still multiplied, as in the reference); the
// column origin is a multiple of 4 so that a thread's "quad" (4 consecutive x) is one ds_read_b128.
// Thread (q, rg) owns quad column q and rows rg, rg+RG, ...; a pass computes into registers,
// barrier, writes back, barrier.  A pass leaves global row 0 / column 0 untouched (ix>0 && iy>0
// guard) and clamps x+1 / y+1 at the image edge.  Out-of-image LDS cells hold the clamped pixel;
// they are refreshed once before the box so that its clamp addressing needs no index logic.
// VAR (development switches, tools/kbench.hip): bit 0 = literal per-plane division, bit 1 = per-pixel border selects
// instead of the replica / repair scheme, bit 2 = west/east neighbours from LDS instead of the neighbouring lanes,
// bit 3 = the box's halo without the box.  Product: VAR = 0.
#ifndef UGSM_SMOOTH_PAD
#define UGSM_SMOOTH_PAD(STX) ((STX) == 112 ? 0 : 4)
#endif
#ifndef SMOOTH_NEWTON
#define SMOOTH_NEWTON 0  // 1: the three quotients of a pixel in binary32 (div3_newton); tools/kbench A/B only
#endif
#ifndef SMOOTH_FUSE_BOX_ROWS
#define SMOOTH_FUSE_BOX_ROWS 0  // 1: measured, no gain at P = 5 (profiles/r05_kbench_smooth_product_form.txt); 0: the box's row pass as a phase of its own after the last pass (rounds 1-4); tools/kbench A/B
#endif
// FIXH: the tile is STY rows high whatever `sty_arg` says -- the height folds into the loop bounds, 2 % faster at level 0 than the
// same kernel with the height in a register (208 against 212.5 us); the launcher picks it whenever the height is STY.
// The passes, the box and the copy-out of ONE tile whose region (tile + halo, clamped onto the image) is in LDS at f0 / f1 / f2 -- shared by
// k_smooth_fused (one tile per workgroup, loaded through registers) and k_smooth_pipe (a workgroup walks tiles; the next tile's region
// arrives by LDS-DMA in a second buffer meanwhile).  Every thread of the workgroup calls it; it contains barriers.
// PIPE (k_smooth_pipe): the wave's LDS-DMA loads of the NEXT tile (issued before this call) are waited for right before this tile's
// global stores are issued -- by then they have long arrived, and the stores themselves are never waited for.
template <int STX, int STY, int NT, int VAR, bool FIXH, bool PIPE = false>
__device__ __forceinline__ void smooth_tile_body(float *const f0, float *const f1, float *const f2, float *__restrict__ o3, const int W, const int H, const int P,
                                                 const int do_box, const int tile_x, const int tile_y, const int sty, const bool pf_tile = false)
{
    constexpr int HX = 8, HY = 7;
    constexpr int RWID = STX + 2 * HX;       // region width (multiple of 4)
    constexpr int LW = RWID + UGSM_SMOOTH_PAD(STX);  // LDS row stride (rows 16-B aligned)
    constexpr int LH = STY + 2 * HY;         // region rows of the TALLEST tile (register arrays and unrolled loops are sized for it)
    constexpr int QW = RWID / 4;             // quad columns
    constexpr int RPW = 64 / QW;             // whole region rows per wave: lane -> (row lane / QW, quad lane % QW), so
                                             // that a quad's west / east neighbours sit in the neighbouring lanes
    constexpr int RG = (NT / 64) * RPW;      // row groups
    constexpr int MAXR = (LH + RG - 1) / RG; // rows per thread per pass
    (void)LH;
    const int LHr = sty + 2 * HY;
    const int tid = threadIdx.x;
    const int tx0 = tile_x * STX, ty0 = tile_y * sty;
    const int x0 = tx0 - HX, y0 = ty0 - HY;  // global coords of LDS (0,0)
    const size_t n = (size_t)W * H;
    const int h = P + ((do_box || (VAR & 8)) ? 2 : 0);  // halo actually needed (VAR & 8: development, the box's halo without the box)
    const int lane = tid & 63;
    const int q = lane % QW, rg = (tid >> 6) * RPW + lane / QW;
    const int c0 = q * 4, gx0 = x0 + c0;
    const bool lane_on = lane < RPW * QW;

    // Image borders without per-pixel selects.  smoothKernel clamps x+1 / y+1 at the last column / row and leaves
    // row 0 / column 0 untouched (MatchLib.cu:1105-1143).  Here every cell of the region is computed alike; the
    // few tiles that touch a border repair it afterwards, under tile-uniform branches:
    //  * east / south clamp: the LDS cell just outside the image (column W, row H) is a replica of its in-image
    //    neighbour -- true after the load, re-established after every write-back (edge_e / edge_s);
    //  * pass-through of row 0 / column 0: their results are replaced by the old values (edge_nw).
    // Cells outside the image otherwise hold whatever the pass produces; no in-image pixel reads them.
    // (VAR & 2, development: the earlier per-pixel selects.)
    const bool edge_e = x0 + RWID > W, edge_s = y0 + LHr > H, edge_nw = x0 <= 0 || y0 <= 0;

    // PRODUCT FORM (round 5; pf_tile, interior tiles of the product kernel): between the passes LDS holds (dx*kappa, dy*kappa, kappa) instead of
    // (dx, dy, kappa).  smoothKernel weights every neighbour's dx and dy with that neighbour's own confidence (MatchLib.cu:1108-1139): the
    // product v*kappa of a pixel enters the stencils of its five neighbours, i.e. it used to be formed five times per pass -- 15
    // multiplications per pixel and pass for the three fields.  Formed ONCE, when the pixel's new value is written back (2 multiplications;
    // kappa*kappa is still formed where it is used), they are the same binary32 numbers -- RN(dx*kappa) of the same two operands -- added in the
    // same order centre, W, E, N, S onto the same leading 0: 7 multiplications instead of 15, bit for bit.  The last pass writes plain values
    // back (the box and the copy-out read those).  Tiles that touch the frame keep the plain form: their pass-through cells (row 0 /
    // column 0) must come out as they went in, and a product cannot be divided back.
    // ROW PASS OF THE BOX FUSED INTO THE LAST JACOBI PASS (round 5; SMOOTH_FUSE_BOX_ROWS): the last pass produces exactly the rows and columns
    // the box's row pass reads (tile +- 2), a quad per lane with the neighbouring quads of the row in the neighbouring lanes -- so the row
    // pass is formed from registers (four DPP moves per plane) when the pass writes back, instead of after another LDS round trip and two
    // more barriers.  Same box5f on the same values.  Interior tiles only: on the frame the box reads clamped replicas that are rebuilt
    // in LDS first.
    const bool fuse_box_rows = SMOOTH_FUSE_BOX_ROWS && VAR == 0 && !PIPE && do_box && P >= 1 && !edge_e && !edge_s && !edge_nw && tx0 - 2 >= 0 && ty0 - 2 >= 0 &&
                               tx0 + STX + 2 <= W && ty0 + sty + 2 <= H;
    auto run_passes = [&](auto pf_tag) {
    constexpr bool PF = decltype(pf_tag)::value;
    for (int p = 1; p <= P; p++) {
        // pass p is needed (and valid) on the region shrunk to halo h-p -- and inside the image: cells above / left of / below /
        // right of it are never read by an in-image pixel (the replica row H and column W are re-established after every
        // write-back), so the tiles on the frame skip them.  (Round 3: a 1741 x 1153 level has 33 tile rows, the last one with ONE
        // image row -- 528 tiles on 512 workgroup slots, i.e. a second round that used to cost as much as the first.)
        const int r_lo = max(HY - (h - p), -y0), r_hi = min(LHr - (HY - (h - p)), H - y0);
        const bool col_on = lane_on && (c0 + 3 >= HX - (h - p)) && (c0 < RWID - (HX - (h - p))) && (c0 + 3 >= -x0) && (c0 < W - x0);
        float nv[MAXR][3][4];
        // one quad-row: the five-point sums and the division from registers, results into nv[u].  LIT = literal
        // division per plane; otherwise the shared-reciprocal form, and the return value says whether every
        // denominator of the quad was in its range (if not, the row is simply redone with LIT).
        auto quad_row = [&](const int u, const int gy, const float (&c4)[3][4], const float (&n4)[3][4], const float (&s4)[3][4],
                            const float (&wl)[3], const float (&er)[3], auto lit_tag) -> bool {
            constexpr bool EDGE = (VAR & 2) != 0, LIT = decltype(lit_tag)::value;
            const bool row_ok = gy > 0 && gy < H;
            const bool south_in = gy + 1 <= H - 1;
            bool ok = true;
#if SMOOTH_NEWTON
            float gmin = 1.0f;
#endif
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const int gx = gx0 + i;
                const bool act = row_ok && gx > 0 && gx < W;
                const bool east_in = gx + 1 <= W - 1;
                float vw[3], ve[3], vs[3];
#pragma unroll
                for (int f = 0; f < 3; f++) {
                    vw[f] = (i == 0) ? wl[f] : c4[f][i > 0 ? i - 1 : 0];
                    const float e_raw = (i == 3) ? er[f] : c4[f][i < 3 ? i + 1 : 3];
                    ve[f] = (!EDGE || east_in) ? e_raw : c4[f][i];
                    vs[f] = (!EDGE || south_in) ? s4[f][i] : c4[f][i];
                }
                const float wc = c4[2][i], ww = vw[2], we = ve[2], wn = n4[2][i], ws = vs[2];
                float sumCorr = 0.0f;
                sumCorr = sumCorr + wc;
                sumCorr = sumCorr + ww;
                sumCorr = sumCorr + we;
                sumCorr = sumCorr + wn;
                sumCorr = sumCorr + ws;
                float acc[3], qf[3];
#pragma unroll
                for (int f = 0; f < 3; f++) {
                    float a = 0.0f;
                    if (PF && f < 2) {  // the neighbours' products, formed when they were written
                        a = c4[f][i] + a;
                        a = vw[f] + a;
                        a = ve[f] + a;
                        a = n4[f][i] + a;
                        a = vs[f] + a;
                    } else {
                        a = c4[f][i] * wc + a;
                        a = vw[f] * ww + a;
                        a = ve[f] * we + a;
                        a = n4[f][i] * wn + a;
                        a = vs[f] * ws + a;
                    }
                    acc[f] = a;
                }
                if constexpr (LIT) {
#pragma unroll
                    for (int f = 0; f < 3; f++) qf[f] = acc[f] / sumCorr;
                } else {
#if SMOOTH_NEWTON  // (probe: timing only -- the guard is per quad-row here and covers the outputs, not the inputs)
                    div3_newton(acc[0], acc[1], acc[2], sumCorr, qf[0], qf[1], qf[2]);
                    gmin = __builtin_elementwise_minimum(gmin, __builtin_elementwise_minimum(__builtin_fabsf(qf[0]), __builtin_fabsf(qf[1])));
#else
                    div3_shared(acc[0], acc[1], acc[2], sumCorr, qf[0], qf[1], qf[2]);
                    ok = ok && div3_shared_ok(sumCorr);
#endif
                }
#pragma unroll
                for (int f = 0; f < 3; f++) nv[u][f][i] = (!EDGE || act) ? qf[f] : c4[f][i];
#ifndef SMOOTH_PIX_ILP
#define SMOOTH_PIX_ILP 1
#endif
                // keep the binary64 temporaries of one pixel at a time (SMOOTH_PIX_ILP = 2, 4: of two / four pixels -- tools/kbench A/B)
                if ((i + 1) % SMOOTH_PIX_ILP == 0) __builtin_amdgcn_sched_barrier(0);
            }
#if SMOOTH_NEWTON
            if constexpr (!LIT) ok = gmin >= 0x1p-40f;
#endif
            return ok;
        };
        // one row of the thread: west / east taps, the quad-row, the rare literal redo, the pass-through cells of the frame
        auto do_row = [&](const int u, const int r, const float (&n4)[3][4], const float (&c4)[3][4], const float (&s4)[3][4]) {
            const int at = r * LW + c0;
            float wl[3], er[3];
            if constexpr (VAR & 4) {
                float t[4];
                ld4(f0 + at - 4, t); wl[0] = t[3]; ld4(f1 + at - 4, t); wl[1] = t[3]; ld4(f2 + at - 4, t); wl[2] = t[3];
                ld4(f0 + at + 4, t); er[0] = t[0]; ld4(f1 + at + 4, t); er[1] = t[0]; ld4(f2 + at + 4, t); er[2] = t[0];
            } else {
                // west / east neighbours from the neighbouring lanes' registers instead of LDS (a narrowed,
                // lane-strided ds_read_b32 there is a 4-way bank conflict).  At q = 0 / QW-1 the value comes
                // from another row: those are region-edge columns, never valid in any pass.
#pragma unroll
                for (int f = 0; f < 3; f++) {
                    wl[f] = lane_below(c4[f][3]);
                    er[f] = lane_above(c4[f][0]);
                }
            }
            bool redo = false;
            const int gy = y0 + r;
            if (col_on) {
                if constexpr (VAR & 1) quad_row(u, gy, c4, n4, s4, wl, er, std::true_type{});
                else redo = !quad_row(u, gy, c4, n4, s4, wl, er, std::false_type{});
            }
            if (__builtin_expect(redo, 0)) {
                // rare: a denominator out of range.  Reload the row (so that nothing has to stay in registers
                // for this path; LDS still holds the previous pass) and divide literally.
                float c4r[3][4], n4r[3][4], s4r[3][4], wlr[3], err[3];
                ld4(f0 + at, c4r[0]); ld4(f1 + at, c4r[1]); ld4(f2 + at, c4r[2]);
                ld4(f0 + at - LW, n4r[0]); ld4(f1 + at - LW, n4r[1]); ld4(f2 + at - LW, n4r[2]);
                ld4(f0 + at + LW, s4r[0]); ld4(f1 + at + LW, s4r[1]); ld4(f2 + at + LW, s4r[2]);
                wlr[0] = f0[at - 1]; wlr[1] = f1[at - 1]; wlr[2] = f2[at - 1];
                err[0] = f0[at + 4]; err[1] = f1[at + 4]; err[2] = f2[at + 4];
                quad_row(u, gy, c4r, n4r, s4r, wlr, err, std::true_type{});
            }
            if (!PF && !(VAR & 2) && edge_nw && col_on) {  // row 0 / column 0 (and anything left / above the image) keeps its value
#pragma unroll
                for (int i = 0; i < 4; i++)
                    if (gy <= 0 || gx0 + i <= 0) {
#pragma unroll
                        for (int f = 0; f < 3; f++) nv[u][f][i] = c4[f][i];
                    }
            }
        };
#pragma unroll
        for (int u = 0; u < MAXR; u++) {
            const int r = r_lo + rg + u * RG;
            if (lane_on && r < r_hi) {
                // every quad of the row loads (the neighbouring lanes' quads feed the west / east taps)
                const int at = r * LW + c0;
                float c4[3][4], n4[3][4], s4[3][4];
                ld4(f0 + at, c4[0]); ld4(f1 + at, c4[1]); ld4(f2 + at, c4[2]);
                ld4(f0 + at - LW, n4[0]); ld4(f1 + at - LW, n4[1]); ld4(f2 + at - LW, n4[2]);
                ld4(f0 + at + LW, s4[0]); ld4(f1 + at + LW, s4[1]); ld4(f2 + at + LW, s4[2]);
                do_row(u, r, n4, c4, s4);
            }
            __builtin_amdgcn_sched_barrier(0);  // one quad-row at a time: interleaving the rows costs 60 more VGPRs
        }
        __syncthreads();
        if (fuse_box_rows && p == P) {
#pragma unroll
            for (int u = 0; u < MAXR; u++) {
                const int r = r_lo + rg + u * RG;
                if (lane_on && r < r_hi) {  // (every lane of the row: the neighbouring quads' values come through DPP)
                    float rb[3][4];
#pragma unroll
                    for (int f = 0; f < 3; f++) {
                        const float x[8] = {lane_below(nv[u][f][2]), lane_below(nv[u][f][3]), nv[u][f][0], nv[u][f][1], nv[u][f][2], nv[u][f][3],
                                            lane_above(nv[u][f][0]), lane_above(nv[u][f][1])};
#pragma unroll
                        for (int i = 0; i < 4; i++) rb[f][i] = box5f(x[i], x[i + 1], x[i + 2], x[i + 3], x[i + 4]);
                    }
                    if (c0 >= HX && c0 < HX + STX) {  // tile columns: what the column pass reads
                        const int at = r * LW + c0;
                        st4(f0 + at, rb[0]); st4(f1 + at, rb[1]); st4(f2 + at, rb[2]);
                    }
                }
            }
        } else
#pragma unroll
        for (int u = 0; u < MAXR; u++) {
            const int r = r_lo + rg + u * RG;
            if (col_on && r < r_hi) {
                const int at = r * LW + c0;
                if (PF && p < P) {  // what the next pass reads: the pixels' weighted terms (formed here, off the division chains)
#pragma unroll
                    for (int i = 0; i < 4; i++) {
                        nv[u][0][i] = nv[u][0][i] * nv[u][2][i];
                        nv[u][1][i] = nv[u][1][i] * nv[u][2][i];
                    }
                }
                st4(f0 + at, nv[u][0]); st4(f1 + at, nv[u][1]); st4(f2 + at, nv[u][2]);
            }
        }
        __syncthreads();
        if (!(VAR & 2) && (edge_e || edge_s)) {  // re-establish the east / south replicas
            if (edge_e) {
                const int cW = W - x0;
                for (int r = tid; r < LHr; r += NT) {
                    f0[r * LW + cW] = f0[r * LW + cW - 1]; f1[r * LW + cW] = f1[r * LW + cW - 1]; f2[r * LW + cW] = f2[r * LW + cW - 1];
                }
            }
            if (edge_s) {
                const int rH = H - y0;
                for (int c = tid; c < RWID; c += NT) {
                    f0[rH * LW + c] = f0[(rH - 1) * LW + c]; f1[rH * LW + c] = f1[(rH - 1) * LW + c]; f2[rH * LW + c] = f2[(rH - 1) * LW + c];
                }
            }
            __syncthreads();
        }
    }
    };  // run_passes
#if defined(SMOOTH_PF_ONLY) && SMOOTH_PF_ONLY  // (timing experiment only: every tile in product form, frame tiles wrong at row 0 / column 0)
    run_passes(std::true_type{});
#else
    if (pf_tile) run_passes(std::true_type{});
    else run_passes(std::false_type{});
#endif

    if (do_box) {
        // refresh the clamped replicas of out-of-image cells within tile+-2 (only edge tiles have any)
        if (tx0 - 2 < 0 || ty0 - 2 < 0 || tx0 + STX + 2 > W || ty0 + sty + 2 > H) {
            for (int it = tid; it < (sty + 4) * (STX + 4); it += NT) {
                const int r = HY - 2 + it / (STX + 4), c = HX - 2 + it % (STX + 4);
                const int gx = x0 + c, gy = y0 + r;
                if (gx < 0 || gx >= W || gy < 0 || gy >= H) {
                    const int src = (clampi(gy, 0, H - 1) - y0) * LW + clampi(gx, 0, W - 1) - x0;
                    f0[r * LW + c] = f0[src]; f1[r * LW + c] = f1[src]; f2[r * LW + c] = f2[src];
                }
            }
            __syncthreads();
        }
        // rows (Ta): tile columns, rows tile-2 .. tile+STY+1, rounded to f32, written back in place
        constexpr int BQ = STX / 4, BRG = NT / BQ, BMAXR = (STY + 4 + BRG - 1) / BRG;
        const int bq = tid % BQ, brg = tid / BQ;
        const int bc0 = HX + bq * 4;
        if (!fuse_box_rows) {  // (else: done by the last pass, from registers)
        float bv[BMAXR][3][4];
#pragma unroll
        for (int u = 0; u < BMAXR; u++) {
            const int r = HY - 2 + brg + u * BRG;
            if (brg < BRG && r < HY + sty + 2) {
                const int at = r * LW + bc0;
#pragma unroll
                for (int f = 0; f < 3; f++) {
                    const float *src = (f == 0 ? f0 : (f == 1 ? f1 : f2)) + at;
                    float v[12];
                    ld4(src - 4, v); ld4(src, v + 4); ld4(src + 4, v + 8);
#pragma unroll
                    for (int i = 0; i < 4; i++) bv[u][f][i] = box5f(v[i + 2], v[i + 3], v[i + 4], v[i + 5], v[i + 6]);
                }
            }
        }
        __syncthreads();
#pragma unroll
        for (int u = 0; u < BMAXR; u++) {
            const int r = HY - 2 + brg + u * BRG;
            if (brg < BRG && r < HY + sty + 2) {
                const int at = r * LW + bc0;
                st4(f0 + at, bv[u][0]); st4(f1 + at, bv[u][1]); st4(f2 + at, bv[u][2]);
            }
        }
        __syncthreads();
        }
        // columns (Ta) into registers, then back to LDS and out with lanes along the rows: a quad-per-lane
        // store touches one 16-B piece per lane (4 instructions per 1-KiB row segment); the copy-out below
        // writes whole contiguous segments
        constexpr int CMAXR = (STY + BRG - 1) / BRG;
        float cv[CMAXR][3][4];
#pragma unroll
        for (int u = 0; u < CMAXR; u++) {
            const int r = HY + brg + u * BRG;
            if (brg < BRG && r < HY + sty) {
                const int at = r * LW + bc0;
#pragma unroll
                for (int f = 0; f < 3; f++) {
                    const float *src = (f == 0 ? f0 : (f == 1 ? f1 : f2)) + at;
                    float a[4], b[4], c[4], d[4], e[4];
                    ld4(src - 2 * LW, a); ld4(src - LW, b); ld4(src, c); ld4(src + LW, d); ld4(src + 2 * LW, e);
#pragma unroll
                    for (int i = 0; i < 4; i++) cv[u][f][i] = box5f(a[i], b[i], c[i], d[i], e[i]);
                }
            }
        }
        if ((W & 3) == 0) {  // rows are 16-byte aligned: the quads leave as they are (a wave's lanes hold consecutive quads of a row)
            if constexpr (PIPE) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
            for (int u = 0; u < CMAXR; u++) {
                const int r = HY + brg + u * BRG;
                const int gx = tx0 + bq * 4, gy = y0 + r;
                if (brg < BRG && r < HY + sty && gx < W && gy < H) {
                    const size_t at = (size_t)gy * W + gx;
#pragma unroll
                    for (int f = 0; f < 3; f++)
                        *reinterpret_cast<float4 *>(o3 + f * n + at) = make_float4(cv[u][f][0], cv[u][f][1], cv[u][f][2], cv[u][f][3]);
                }
            }
            return;
        }
        __syncthreads();
#pragma unroll
        for (int u = 0; u < CMAXR; u++) {
            const int r = HY + brg + u * BRG;
            if (brg < BRG && r < HY + sty) {
                const int at = r * LW + bc0;
                st4(f0 + at, cv[u][0]); st4(f1 + at, cv[u][1]); st4(f2 + at, cv[u][2]);
            }
        }
        __syncthreads();
        if constexpr (PIPE) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        for (int it = tid; it < STX * sty; it += NT) {
            const int r = it / STX, c = it - r * STX;
            const int gx = tx0 + c, gy = ty0 + r;
            if (gx < W && gy < H) {
                const size_t at = (size_t)gy * W + gx;
                const int la = (HY + r) * LW + HX + c;
                o3[at] = f0[la];
                o3[n + at] = f1[la];
                o3[2 * n + at] = f2[la];
            }
        }
    } else {
        if constexpr (PIPE) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        for (int it = tid; it < STX * sty; it += NT) {
            const int r = it / STX, c = it - r * STX;
            const int gx = tx0 + c, gy = ty0 + r;
            if (gx < W && gy < H) {
                const size_t at = (size_t)gy * W + gx;
                const int la = (HY + r) * LW + HX + c;
                o3[at] = f0[la];
                o3[n + at] = f1[la];
                o3[2 * n + at] = f2[la];
            }
        }
    }
}

__device__ int smooth_phase_sleep = 0;  // (development, VAR & 16: tools/kbench mode 20)
#ifndef SMOOTH_FILL_ALL
#define SMOOTH_FILL_ALL 1  // 0: round 4's load phase (cells outside the needed halo stay unwritten); tools/kbench A/B only
#endif
#ifndef SMOOTH_PRODUCT_FORM
#define SMOOTH_PRODUCT_FORM 0  // 1: product form (measured slower, profiles/r05_kbench_smooth_product_form.txt); 0: plain (dx, dy, kappa) in LDS in every tile (rounds 1-4); tools/kbench A/B only
#endif
// OCC (development, tools/kbench mode 20): waves per SIMD the register allocation is held to (0 = the product's NT / 128)
template <int STX, int STY, int NT, int VAR = 0, bool FIXH = false, int OCC = 0>
__global__ __launch_bounds__(NT, (OCC ? OCC : (NT <= 512 ? NT / 128 : 1))) void k_smooth_fused(const float *__restrict__ s3, float *__restrict__ o3, int W, int H, int P, int do_box,
                                                  int tiles_x, int n_tiles, int sty_arg, Batch bt)
{
    if (bt.n > 1) {  // this workgroup's pair of the batch (blockIdx.y)
        s3 = shifted(s3, bt.in[blockIdx.y]);
        o3 = shifted(o3, bt.out[blockIdx.y]);
    }
    if constexpr ((VAR & 16) != 0) {
        // development (tools/kbench mode 20, round 5): PHASE SHIFT.  The two workgroups a CU holds start together and take the same time, so
        // they load together and compute together, launch after launch; the second workgroup of every CU (dispatch order 256 .. 511) waits
        // smooth_phase_sleep x 64 x 127 cycles before it starts, once per launch, so that its successors run half a period out of phase
        const int lin = (int)(blockIdx.y * gridDim.x + blockIdx.x);
        if (lin >= 256 && lin < 512)
            for (int i = 0; i < smooth_phase_sleep; i++) __builtin_amdgcn_s_sleep(127);
    }
    const int sty = FIXH ? STY : sty_arg;
    constexpr int HX = 8, HY = 7;
    constexpr int RWID = STX + 2 * HX;       // region width (multiple of 4)
    constexpr int LW = RWID + UGSM_SMOOTH_PAD(STX);  // LDS row stride (rows 16-B aligned)
    constexpr int LH = STY + 2 * HY;         // region rows of the TALLEST tile (register arrays and unrolled loops are sized for it)
    constexpr int QW = RWID / 4;             // quad columns
    constexpr int RPW = 64 / QW;             // whole region rows per wave: lane -> (row lane / QW, quad lane % QW), so
                                             // that a quad's west / east neighbours sit in the neighbouring lanes
    constexpr int RG = (NT / 64) * RPW;      // row groups
    constexpr int MAXR = (LH + RG - 1) / RG; // rows per thread per pass
    (void)QW; (void)RPW; (void)RG; (void)MAXR;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    // A tile is `sty` <= STY rows high (the host picks the height that fills whole rounds of workgroups: smooth_tile_rows);
    // the region is LHr rows, the LDS planes are that long.
    const int LHr = sty + 2 * HY;
    float *f0 = smem, *f1 = smem + LHr * LW, *f2 = smem + 2 * LHr * LW;

    const int tid = threadIdx.x;
    int tile_x, tile_y;
    xcd_tile(n_tiles, tiles_x, tile_x, tile_y);
    const int tx0 = tile_x * STX, ty0 = tile_y * sty;
    const int x0 = tx0 - HX, y0 = ty0 - HY;  // global coords of LDS (0,0)
    const size_t n = (size_t)W * H;
    const int h = P + ((do_box || (VAR & 8)) ? 2 : 0);  // halo actually needed (VAR & 8: development, the box's halo without the box)
    // product form between the passes (smooth_tile_body): interior tiles of the product kernel
#if defined(SMOOTH_PF_ONLY) && SMOOTH_PF_ONLY
    const bool pf = true;
#else
    const bool pf = SMOOTH_PRODUCT_FORM && VAR == 0 && P >= 1 && x0 > 0 && y0 > 0 && x0 + RWID <= W && y0 + LHr <= H;
#endif

    // ---- load tile + needed halo (clamped onto the image): every global load of the thread is issued
    // before the first LDS store (a rolled loop waits out one HBM round trip per 512 pixels) ------------
    {
        const int r_lo = HY - h, r_hi = LHr - (HY - h);
        constexpr int NLD = (LH * RWID + NT - 1) / NT;
        float v[NLD][3];
#pragma unroll
        for (int u = 0; u < NLD; u++) {
            const int it = tid + u * NT;
            const int r = it / RWID, c = it - r * RWID;
            const bool need = r >= r_lo && r < r_hi && c >= HX - h && c < RWID - (HX - h);
            const int gx = clampi(x0 + c, 0, W - 1), gy = clampi(y0 + r, 0, H - 1);
            // one 32-bit byte offset per pixel against three uniform plane bases (a 64-bit address per load would
            // hold 6 VGPRs per pixel across the whole batch); a plane is < 4 GiB
            const unsigned off = ((unsigned)gy * (unsigned)W + (unsigned)gx) * 4u;
            constexpr float fill = SMOOTH_NEWTON ? 1.0f : 0.0f;
            v[u][0] = need ? *reinterpret_cast<const float *>(reinterpret_cast<const char *>(s3) + off) : fill;
            v[u][1] = need ? *reinterpret_cast<const float *>(reinterpret_cast<const char *>(s3 + n) + off) : fill;
            v[u][2] = need ? *reinterpret_cast<const float *>(reinterpret_cast<const char *>(s3 + 2 * n) + off) : 1.0f;
        }
        // EVERY cell of the region is written, the ones outside the needed halo with (0, 0, confidence 1) (round 5).  A pass works on whole
        // quads: with a halo of 6 or 7 (five passes + the box: every iteration's last launch) the outermost quads of a row straddle the
        // needed region, and their outer pixels -- whose results nobody reads -- used to be computed on never-written LDS: a sum of
        // confidences of 0 there failed div3_shared_ok, so EVERY wave redid its rows with the literal division in the first one or two
        // passes of such a launch: 78 of 274 us at level 0 (tools/kbench mode 20, profiles/r05_kbench_smooth_halo.txt).  A defined,
        // in-range confidence in those cells keeps their (unused) denominators in range; no value any valid pixel reads changes.
#pragma unroll
        for (int u = 0; u < NLD; u++) {
            const int it = tid + u * NT;
            const int r = it / RWID, c = it - r * RWID;
            if (SMOOTH_FILL_ALL ? r < LHr : (r >= r_lo && r < r_hi && c >= HX - h && c < RWID - (HX - h))) {
                f0[r * LW + c] = pf ? v[u][0] * v[u][2] : v[u][0];
                f1[r * LW + c] = pf ? v[u][1] * v[u][2] : v[u][1];
                f2[r * LW + c] = v[u][2];
            }
        }
    }
    __syncthreads();

    smooth_tile_body<STX, STY, NT, VAR, FIXH>(f0, f1, f2, o3, W, H, P, do_box, tile_x, tile_y, sty, pf);
}

#ifdef UGSM_DEV_LIB  // k_smooth_pipe: built, bit-exact, measured 17-20 % SLOWER than k_smooth_fused (profiles/r04_kbench_smooth_pipe.txt) -- libugsm_dev.so / tools only
// =========================================================================================
// K-smooth, pipelined form (round 4): k_smooth_fused alternates a memory phase (tile + halo in, 77 KB; tile out) and a
// compute phase (the passes) per workgroup, and with two workgroups per CU the phases add up instead of overlapping (a launch with zero
// passes takes 82 us at 16 MP, five passes 185 us: DESIGN.md section 4).  Here ONE workgroup of sixteen waves per CU walks tiles it takes
// from a queue and owns TWO tile buffers: while it computes tile k in one, the region of tile k + 1 arrives in the other by LDS-DMA
// (global_load_lds: no registers, no instruction slots of the computing waves beyond the issue), and tile k's stores are issued and
// never waited for.  Same passes, same box, same copy-out (smooth_tile_body): bit-identical.
// MEASURED (tools/kbench mode 17, 16 MP, same box): with zero passes it runs at the chip's copy rate like k_smooth_fused (83 us), but five
// passes take 224 us against 185 us, five passes + box 270 against 226: what the prefetch saves is less than what ONE sixteen-wave
// workgroup per CU loses at its ten to twelve barriers per tile -- with two eight-wave workgroups per CU (k_smooth_fused) a workgroup
// stalled at a barrier leaves the SIMDs to the other one's tile; here all sixteen waves stall together (14.3 us per tile and CU against
// 11.8, for 7.9 us of VALU issue).  Two workgroups per CU with two buffers each would need 4 x 77 KB of LDS.  Not used.
// =========================================================================================
typedef __attribute__((address_space(1))) const void gvoid_c;
typedef __attribute__((address_space(3))) void lvoid;
template <int STY, bool FIXH>
__global__ __launch_bounds__(1024) void k_smooth_pipe(const float *__restrict__ s3, float *__restrict__ o3, int W, int H, int P, int do_box, int tiles_x,
                                                      int n_tiles, int sty_arg, unsigned *__restrict__ queue, Batch bt)
{
    constexpr int STX = 112, NT = 1024, HX = 8, HY = 7, RWID = STX + 2 * HX, LW = RWID;
    static_assert(UGSM_SMOOTH_PAD(112) == 0 && RWID == 128, "an LDS-DMA instruction fills whole region rows: they must be contiguous");
    if (bt.n > 1) {  // this workgroup's pair of the batch (blockIdx.y)
        s3 = shifted(s3, bt.in[blockIdx.y]);
        o3 = shifted(o3, bt.out[blockIdx.y]);
        queue += blockIdx.y;
    }
    const int sty = FIXH ? STY : sty_arg;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    __shared__ int s_next;
    const int LHr = sty + 2 * HY;
    const int plane_f = LHr * LW;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const size_t n = (size_t)W * H;
    // the region of tile t (clamped onto the image) -> buffer b, asynchronously; every wave issues its share
    auto issue = [&](const int t, float *const b) {
        const int tile_y = t / tiles_x, tile_x = t - tile_y * tiles_x;
        const int x0 = tile_x * STX - HX, y0 = tile_y * sty - HY;
        const bool interior = x0 >= 0 && x0 + RWID <= W && y0 >= 0 && y0 + LHr <= H && (W & 3) == 0;
        if (interior) {  // 16 bytes per lane: an instruction fills two region rows (the rows are 16-byte aligned in memory when W is a multiple of 4)
            const int half = (LHr + 1) >> 1;
            for (int j = wave; j < 3 * half; j += NT / 64) {
                const int f = j / half, rp = j - f * half;
                const int r = 2 * rp + (lane >> 5), c = (lane & 31) * 4;
                if (r < LHr) {
                    const float *g = s3 + f * n + (size_t)(y0 + r) * W + (x0 + c);
                    __builtin_amdgcn_global_load_lds((gvoid_c *)g, (lvoid *)(b + f * plane_f + 2 * rp * LW), 16, 0, 0);
                }
            }
        } else {  // the tiles on the frame (and levels whose rows are not 16-byte aligned): 4 bytes per lane from the clamped pixel, half a region row per instruction
            for (int j = wave; j < 6 * LHr; j += NT / 64) {
                const int f = j / (2 * LHr), rem = j - f * 2 * LHr;
                const int r = rem >> 1, c = (rem & 1) * 64 + lane;
                const float *g = s3 + f * n + (size_t)clampi(y0 + r, 0, H - 1) * W + clampi(x0 + c, 0, W - 1);
                __builtin_amdgcn_global_load_lds((gvoid_c *)g, (lvoid *)(b + f * plane_f + r * LW + (rem & 1) * 64), 4, 0, 0);
            }
        }
    };
    // the first two tiles of this workgroup
    if (tid == 0) s_next = (int)atomicAdd(queue, 1u);
    __syncthreads();
    int t = s_next;
    __syncthreads();
    if (t >= n_tiles) return;
    float *const buf0 = smem, *const buf1 = smem + 3 * plane_f;
    issue(t, buf0);
    if (tid == 0) s_next = (int)atomicAdd(queue, 1u);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    int tn = s_next, cur = 0;
    for (;;) {
        if (tn < n_tiles) issue(tn, cur ? buf0 : buf1);
        int got = 0;
        if (tid == 0) got = (int)atomicAdd(queue, 1u);  // the tile after next (its latency hides behind this tile's passes)
        const int tile_y = t / tiles_x, tile_x = t - tile_y * tiles_x;
        float *const f0 = cur ? buf1 : buf0, *const f1 = f0 + plane_f, *const f2 = f1 + plane_f;
        smooth_tile_body<STX, STY, NT, 0, FIXH, true>(f0, f1, f2, o3, W, H, P, do_box, tile_x, tile_y, sty);
        if (tid == 0) s_next = got;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // (every path of the tile body has waited already; the no-store corner cases too now)
        __syncthreads();  // the next tile's region is complete in LDS; nobody reads this tile's buffer any more
        if (tn >= n_tiles) break;
        t = tn;
        tn = s_next;
        cur ^= 1;
    }
}

int smooth_pipe_workgroups = 256;  // persistent workgroups per pair of a launch: one per CU
// queue: one zeroed counter per pair of the launch (the caller clears it on the stream before the launch)
void launch_smooth_pipe(hipStream_t st, const float *s3, float *o3, int W, int H, int passes, int do_box, int tile_rows, unsigned *queue, const Batch *bt)
{
    constexpr int STY = kSmoothTileRowsMax, LWp = 128;
    Batch one{};
    one.n = 1;
    const Batch &B = bt ? *bt : one;
    const int pairs = B.n > 1 ? B.n : 1;
    int sty = tile_rows;
    if (sty < 1 || sty > STY) sty = STY;
    const size_t bytes = 2 * 3 * (size_t)(sty + 14) * LWp * sizeof(float);
    static std::atomic<unsigned long long> attr_mask{0};
    int dev = 0;
    (void)hipGetDevice(&dev);
    const unsigned long long bit = 1ull << (dev & 63);
    if (!(attr_mask.load(std::memory_order_relaxed) & bit)) {
        const int max_bytes = 2 * 3 * (STY + 14) * LWp * (int)sizeof(float);
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&k_smooth_pipe<STY, true>), hipFuncAttributeMaxDynamicSharedMemorySize, max_bytes);
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&k_smooth_pipe<STY, false>), hipFuncAttributeMaxDynamicSharedMemorySize, max_bytes);
        attr_mask.fetch_or(bit, std::memory_order_relaxed);
    }
    const int tiles_x = (W + 111) / 112, n_tiles = tiles_x * ((H + sty - 1) / sty);
    const int wgs = std::min(n_tiles, smooth_pipe_workgroups);
    if (sty == STY) UGSM_LAUNCH((k_smooth_pipe<STY, true>), dim3(wgs, pairs), dim3(1024), bytes, st, s3, o3, W, H, passes, do_box, tiles_x, n_tiles, sty, queue, B);
    else UGSM_LAUNCH((k_smooth_pipe<STY, false>), dim3(wgs, pairs), dim3(1024), bytes, st, s3, o3, W, H, passes, do_box, tiles_x, n_tiles, sty, queue, B);
}
#endif  // UGSM_DEV_LIB

// =========================================================================================
// triangulation of the full-resolution disparity into X, Y, Z planes.
// CdynamicCalibration::get3DPoint, non-foveated branch (src/pointcloud/getPointCloud.cpp:886-949), for
// every pixel: the reference calls it from scalar host loops behind a progress bar (:640-660, :778).
// Purely per-pixel (reads 8 B, writes 12 B): HBM-bound.  The closed form keeps the source's mix of float
// and double term by term (a..j, x, y are floats; pow(v,2.0) is the exact binary64 square; the literal
// 2.0 is a double) -- the expression text is kept identical to the CPU restatement used by the tests, no contraction.
// =========================================================================================
struct Proj {
    double m[12];  // 3x4, row major
};
__device__ __forceinline__ double sq_d(float v) { return (double)v * (double)v; }

// the closed form of get3DPoint (getPointCloud.cpp:908-948) for one left/right correspondence.
// NOTE: this one function follows the reference's expressions term for term, including its variable names
// a..j, x, y -- the formula is a machine-generated closed form whose evaluation order and float/double mix ARE the bit-exactness
// contract (re-associating any term changes the result), so the similarity is unavoidable here and deliberately confined to this
// block; nothing else in the product is written against the reference's text.
__device__ __forceinline__ void tri_point(float x1, float y1, float x2, float y2, const double *P1, const double *P2, float &X, float &Y, float &Z)
{
    float a, b, c, d, e, f, g, h, i, j, x, y;
    a = (float)P1[0];
    b = (float)(P1[2] - x1);
    c = (float)P1[5];
    d = (float)(P1[6] - y1);
    e = (float)(P2[0] - x2 * P2[8]);
    f = (float)(P2[1] - x2 * P2[9]);
    g = (float)(P2[2] - x2 * P2[10]);
    h = (float)(P2[4] - y2 * P2[8]);
    i = (float)(P2[5] - y2 * P2[9]);
    j = (float)(P2[6] - y2 * P2[10]);
    x = (float)(x2 * P2[11] - P2[3]);
    y = (float)(y2 * P2[11] - P2[7]);
    float XUp = (d*f*h - c*g*h - d*e*i + c*e*j)*(-(d*i*x) + c*j*x + d*f*y - c*g*y) +
                sq_d(b)*((f*h - e*i)*(-(i*x) + f*y) + sq_d(c)*(e*x + h*y)) +
                a*b*((-(g*i) + f*j)*(i*x - f*y) + c*d*(f*x + i*y) - sq_d(c)*(g*x + j*y));
    float YUp = (sq_d(b)*(f*h - e*i) + d*(d*f*h - c*g*h - d*e*i + c*e*j))*(h*x - e*y) +
                a*b*((c*d*e + g*h*i - 2.0*f*h*j + e*i*j)*x + (c*d*h + f*g*h - 2.0*e*g*i + e*f*j)*y) +
                sq_d(a)*((g*i - f*j)*(-(j*x) + g*y) + sq_d(d)*(f*x + i*y) - c*d*(g*x + j*y));
    float ZUp = c*(-(d*f*h) + c*g*h + d*e*i - c*e*j)*(h*x - e*y) - a*b*((f*h - e*i)*(-(i*x) + f*y) +
                sq_d(c)*(e*x + h*y)) + sq_d(a)*((g*i - f*j)*(i*x - f*y) - c*d*(f*x + i*y) +
                sq_d(c)*(g*x + j*y));
    float divisor = sq_d(b)*(sq_d(c)*(sq_d(e) + sq_d(h)) + sq_d(f*h - e*i)) +
                    sq_d(d*f*h - c*g*h - d*e*i + c*e*j) - 2.0*a*b*(-(c*d*(e*f + h*i)) +
                    (f*h - e*i)*(-(g*i) + f*j) + sq_d(c)*(e*g + h*j)) + sq_d(a)*
                    (sq_d(d)*(sq_d(f) + sq_d(i)) + sq_d(g*i - f*j) - 2.0*c*d*(f*g + i*j) +
                    sq_d(c)*(sq_d(g) + sq_d(j)));
    X = XUp / divisor;
    Y = YUp / divisor;
    Z = ZUp / divisor;
}

__global__ __launch_bounds__(256) void k_triangulate(const float *__restrict__ dispx, const float *__restrict__ dispy, int W, int H, Proj P1q, Proj P2q,
                                                     float *__restrict__ xyz)
{
    const int xx = blockIdx.x * blockDim.x + threadIdx.x;
    const int yy = blockIdx.y;
    if (xx >= W) return;
    const size_t n = (size_t)W * H, at = (size_t)yy * W + xx;
    float x1, x2, y1, y2;
    x1 = xx;
    y1 = yy;
    x2 = xx + dispx[at];
    y2 = yy + dispy[at];
    float X, Y, Z;
    tri_point(x1, y1, x2, y2, P1q.m, P2q.m, X, Y, Z);
    xyz[at] = X;
    xyz[n + at] = Y;
    xyz[2 * n + at] = Z;
}

// get3DPoint, foveated branch (getPointCloud.cpp:892-903): level src_level of the (F*fovH) x fovW stacks, pixel
// coordinates mapped into the full-resolution frame by mapXcoord / mapYcoord (:387-421).  Those take an int, so the
// right-image coordinate xx + disparity is truncated toward zero before scaling -- kept as in the reference.
__global__ __launch_bounds__(256) void k_triangulate_fovea(const float *__restrict__ stackx, const float *__restrict__ stacky, int fovW, int fovH,
                                                           int src_level, int left_margin, int upper_margin, float scale, Proj P1q, Proj P2q,
                                                           float *__restrict__ xyz)
{
    const int xx = blockIdx.x * blockDim.x + threadIdx.x;
    const int yy = blockIdx.y;
    if (xx >= fovW) return;
    const size_t n = (size_t)fovW * fovH, at = (size_t)yy * fovW + xx;
    const size_t sat = ((size_t)yy + (size_t)fovH * src_level) * fovW + xx;
    const float x1 = (float)left_margin + (float)xx * scale;
    const float y1 = (float)upper_margin + (float)yy * scale;
    const int sx = (int)(xx + stackx[sat]);
    const int sy = (int)(yy + stacky[sat]);
    const float x2 = (float)left_margin + (float)sx * scale;
    const float y2 = (float)upper_margin + (float)sy * scale;
    float X, Y, Z;
    tri_point(x1, y1, x2, y2, P1q.m, P2q.m, X, Y, Z);
    xyz[at] = X;
    xyz[n + at] = Y;
    xyz[2 * n + at] = Z;
}

void launch_triangulate(hipStream_t st, const float *dispx, const float *dispy, int W, int H, const double *P1, const double *P2, float *xyz)
{
    Proj a, b;
    for (int k = 0; k < 12; k++) { a.m[k] = P1[k]; b.m[k] = P2[k]; }
    UGSM_LAUNCH(k_triangulate, dim3((W + 255) / 256, H), dim3(256), 0, st, dispx, dispy, W, H, a, b, xyz);
}

void launch_triangulate_fovea(hipStream_t st, const float *stackx, const float *stacky, int fovW, int fovH, int src_level, int left_margin,
                              int upper_margin, float scale, const double *P1, const double *P2, float *xyz)
{
    Proj a, b;
    for (int k = 0; k < 12; k++) { a.m[k] = P1[k]; b.m[k] = P2[k]; }
    UGSM_LAUNCH(k_triangulate_fovea, dim3((fovW + 255) / 256, fovH), dim3(256), 0, st, stackx, stacky, fovW, fovH, src_level, left_margin,
                       upper_margin, scale, a, b, xyz);
}

// =========================================================================================
// one step of hierarchicalDisparity (MatchGPULib.cpp:2643-2683) -- upsample the coarser
// full-frame field by partsubsampleDispKernel (MatchLib.cu:435-462: dst = s * src[tex((x+.5)/s), tex((y+.5)/s)],
// every channel scaled, confidence included) and paste the finer level's fovea at its crop origin, fused: a
// pasted pixel never computes the upsample it would overwrite.  HBM-bound (12 B written per pixel).
// =========================================================================================
__global__ __launch_bounds__(256) void k_upsample_paste(const float *__restrict__ src3, int W, int H, float *__restrict__ dst3, int W2, int H2,
                                                        const float *__restrict__ fovH_, const float *__restrict__ fovV_, const float *__restrict__ fovC_,
                                                        int fovW, int fovH, int org_x, int org_y)
{
    const int ix = blockIdx.x * blockDim.x + threadIdx.x;
    const int iy = blockIdx.y;
    if (ix >= W2) return;
    const float s = (float)1.41421356;
    const size_t n = (size_t)W * H, n2 = (size_t)W2 * H2, at2 = (size_t)iy * W2 + ix;
    const int fx = ix - org_x, fy = iy - org_y;
    if (fx >= 0 && fx < fovW && fy >= 0 && fy < fovH) {
        const size_t fa = (size_t)fy * fovW + fx;
        dst3[at2] = fovH_[fa];
        dst3[n2 + at2] = fovV_[fa];
        dst3[2 * n2 + at2] = fovC_[fa];
    } else {
        const size_t at = (size_t)tex_index(((float)iy + 0.5f) / s, H) * W + tex_index(((float)ix + 0.5f) / s, W);
        dst3[at2] = s * src3[at];
        dst3[n2 + at2] = s * src3[n + at];
        dst3[2 * n2 + at2] = s * src3[2 * n + at];
    }
}

void launch_upsample_paste(hipStream_t st, const float *src3, int W, int H, float *dst3, int W2, int H2, const float *fovH_, const float *fovV_,
                           const float *fovC_, int fovW, int fovH, int org_x, int org_y)
{
    UGSM_LAUNCH(k_upsample_paste, dim3((W2 + 255) / 256, H2), dim3(256), 0, st, src3, W, H, dst3, W2, H2, fovH_, fovV_, fovC_, fovW, fovH,
                       org_x, org_y);
}

#ifdef UGSM_DEV_LIB  // the probe kernels: in libugsm_dev.so only
// test hook (tests only): poly_fast on arbitrary operands, so that its rarely taken f64 fallback and the
// special values are exercised against the oracle's literal PolyDisparity
__global__ void k_poly_probe(const float *__restrict__ c, const float *__restrict__ l, const float *__restrict__ r, const float *__restrict__ thr,
                             float *__restrict__ delta, float *__restrict__ corr, float *__restrict__ third, int n)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) {
        poly_fast(c[i], l[i], r[i], thr[i], delta[i], corr[i]);
        third[i] = (c[i] >= 0.0f || c[i] != c[i]) ? div3_nonneg(c[i]) : 0.0f;
    }
}
void launch_poly_probe(hipStream_t st, const float *c, const float *l, const float *r, const float *thr, float *delta, float *corr, float *third, int n)
{
    UGSM_LAUNCH(k_poly_probe, dim3((n + 255) / 256), dim3(256), 0, st, c, l, r, thr, delta, corr, third, n);
}

// test hook (tests only): the shared-reciprocal division exactly as k_smooth_fused applies it (fast form,
// range test, literal redo)
__global__ void k_div3_probe(const float *__restrict__ a0, const float *__restrict__ a1, const float *__restrict__ a2, const float *__restrict__ s,
                             float *__restrict__ q0, float *__restrict__ q1, float *__restrict__ q2, int n)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) {
        float x, y, z;
        div3_shared(a0[i], a1[i], a2[i], s[i], x, y, z);
        if (!div3_shared_ok(s[i])) {
            x = a0[i] / s[i];
            y = a1[i] / s[i];
            z = a2[i] / s[i];
        }
        q0[i] = x;
        q1[i] = y;
        q2[i] = z;
    }
}
void launch_div3_probe(hipStream_t st, const float *a0, const float *a1, const float *a2, const float *s, float *q0, float *q1, float *q2, int n)
{
    UGSM_LAUNCH(k_div3_probe, dim3((n + 255) / 256), dim3(256), 0, st, a0, a1, a2, s, q0, q1, q2, n);
}
#endif  // UGSM_DEV_LIB

// ---- launchers ------------------------------------------------------------------------------

void launch_cost_fused(hipStream_t st, Img3 L, Img3 R, const float *A3, const float *d3, float *nd3, int W, int H, float thr, int blend)
{
    const int tiles_x = (W + TX - 1) / TX, n_tiles = tiles_x * ((H + TY - 1) / TY);
    // two threads per quad (k_cost_split): same speed as k_cost_fused on the big levels, 15-25 % shorter launches
    // on the latency-bound coarse ones (12.0 vs 16.3 us at 53x34, 18.7 vs 22.0 us at 615x407)
    UGSM_LAUNCH(k_cost_split<0>, dim3(n_tiles), dim3(512), 0, st, L, R, A3, d3, nd3, W, H, thr, blend, tiles_x, n_tiles);
}

#ifndef UGSM_SMOOTH_MID_NT
#define UGSM_SMOOTH_MID_NT 1024  // one quad-row per thread per pass: the mid/small levels are latency-bound (16.5 vs 17.7 ms per pair)
#endif
#ifndef UGSM_SMOOTH_SMALL_NT
#define UGSM_SMOOTH_SMALL_NT 512
#endif
template <int STX, int STY, int NT>
static void launch_smooth_t(hipStream_t st, const float *s3, float *o3, int W, int H, int passes, int do_box, int sty, const Batch *bt)
{
    Batch one{};
    one.n = 1;
    const Batch &B = bt ? *bt : one;
    const int pairs = B.n > 1 ? B.n : 1;
    constexpr int LW = STX + 16 + UGSM_SMOOTH_PAD(STX), LH = STY + 14;
    constexpr size_t max_bytes = 3 * (size_t)LH * LW * sizeof(float);
    // the attribute is per device: a process may hold contexts on several devices (the launch is made with the context's
    // device current); std::atomic so that contexts driven from different host threads do not race on the mask
    static std::atomic<unsigned long long> attr_mask{0};
    int dev = 0;
    (void)hipGetDevice(&dev);
    const unsigned long long bit = 1ull << (dev & 63);
    if (!(attr_mask.load(std::memory_order_relaxed) & bit)) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&k_smooth_fused<STX, STY, NT, 0, false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)max_bytes);
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&k_smooth_fused<STX, STY, NT, 0, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)max_bytes);
        attr_mask.fetch_or(bit, std::memory_order_relaxed);
    }
    if (sty < 1 || sty > STY) sty = STY;
    const size_t bytes = 3 * (size_t)(sty + 14) * LW * sizeof(float);
    const int tiles_x = (W + STX - 1) / STX, n_tiles = tiles_x * ((H + sty - 1) / sty);
    if (sty == STY) UGSM_LAUNCH((k_smooth_fused<STX, STY, NT, 0, true>), dim3(n_tiles, pairs), dim3(NT), bytes, st, s3, o3, W, H, passes, do_box, tiles_x, n_tiles, sty, B);
    else UGSM_LAUNCH((k_smooth_fused<STX, STY, NT, 0, false>), dim3(n_tiles, pairs), dim3(NT), bytes, st, s3, o3, W, H, passes, do_box, tiles_x, n_tiles, sty, B);
}

// Tile height of the 112-column K-smooth tile for a W x H level.  The kernel's tile may be any height up to kSmoothTileRowsMax (36); two
// workgroups are resident per CU, 512 in all.
//  * Several pairs in flight (`latency` = 0), or a level of many rounds of workgroups (> 3): 36 rows, as always.
//  * A level of a few rounds with the chip to itself: a launch lasts (whole rounds) x (region rows), so a level whose tiles at full
//    height need a few workgroups more than a whole number of rounds pays a round for them (1742 x 1154: 16 x 33 tiles of 36 rows
//    = 1.03 rounds, 51 us in the pipeline; 16 x 58 tiles of 20 rows = two rounds of shorter tiles, 48 us; with a 39-row kernel 16 x 32
//    tiles of 37 rows = one round, 42 us -- but that kernel's unrolled loops cost every 36-row launch 7.7 % more instructions:
//    212 against 208 us at level 0, so the tallest tile stays 36).  The height that minimises rounds x region rows; among equals the
//    fewest tiles.  Worth 0.3 % of a 16 MP pair alone on the chip (tools/ab.py).
int smooth_tile_rows(int W, int H, int latency, int pairs)
{
    constexpr int STX = 112, HMAX = kSmoothTileRowsMax, HDEF = 36, HMIN = 16, SLOTS = 2 * 256;
    const int tiles_x = ((W + STX - 1) / STX) * (pairs > 1 ? pairs : 1);  // (a batched launch: the tiles of all its pairs share the chip)
    const int rows_min = (H + HMAX - 1) / HMAX;
    if (!latency || (long long)tiles_x * ((H + HDEF - 1) / HDEF) > 3 * SLOTS) return HDEF;
    int best = 0;
    long long best_cost = 0, best_tiles = 0;
    for (int n_rows = rows_min; n_rows <= (H + HMIN - 1) / HMIN; n_rows++) {
        const int sty = (H + n_rows - 1) / n_rows;
        if (sty > HMAX || sty < HMIN) continue;
        const long long tiles = (long long)tiles_x * ((H + sty - 1) / sty);
        const long long cost = ((tiles + SLOTS - 1) / SLOTS) * (sty + 14);
        if (!best || cost < best_cost || (cost == best_cost && tiles < best_tiles)) {
            best = sty;
            best_cost = cost;
            best_tiles = tiles;
        }
    }
    return best ? best : std::min(HMAX, std::max(H, 1));
}

// levels of at least this many pixels (and below the 112-column tile's range): 64 x 32 tiles, else 32 x 16.  2^18 since the end of round 3: a
// 615 x 407 level (16 MP's level 6, the fovea windows) is 130 tiles of 64 x 32 on 256 CUs, 520 of 32 x 16 -- one pair alone +0.5 %, the
// foveated stack with four slots +1.5 % (tools/ab.py; 2^19: -0.2 %, 2^16: -0.4 %)
int smooth_mid_min_pixels = 1 << 18;

void launch_smooth_fused(hipStream_t st, const float *s3, float *o3, int W, int H, int passes, int do_box, int tile_rows, const Batch *bt, int tile_class)
{
    // big levels: 112 x (up to 36) tiles, 512 threads, <= 77 KB LDS -> two workgroups per CU (meant to let one's load/store phase overlap
    // the other's passes; measured, the two phases still nearly add up: DESIGN.md section 4).  The region is 128 columns = 32 quads = two whole rows per wave: no idle lanes, and the halo
    // columns every pass recomputes are 12.5 % of the row instead of 20 % (at 16 MP: 5 passes 227 us against 266 us for
    // 64x58, 323 us for the first 64x64 version; 128x64x1024 with one workgroup per CU 406 us);
    // mid levels (from smooth_mid_min_pixels): 64x32; small levels: 32x16 so that the launch is short and the chip still fills
    const size_t px = (size_t)W * H;
    if (tile_rows > 0 || (tile_class == 0 && px >= ((size_t)1 << 19))) launch_smooth_t<112, kSmoothTileRowsMax, 512>(st, s3, o3, W, H, passes, do_box, tile_rows > 0 ? tile_rows : 36, bt);
    else if (tile_class == 2 || (tile_class == 0 && px >= (size_t)smooth_mid_min_pixels)) launch_smooth_t<64, 32, UGSM_SMOOTH_MID_NT>(st, s3, o3, W, H, passes, do_box, 0, bt);
    else launch_smooth_t<32, 16, UGSM_SMOOTH_SMALL_NT>(st, s3, o3, W, H, passes, do_box, 0, bt);
}

// =========================================================================================
// K-pyr: level i+1 / i+2 of the pyramid = zero-padded 5x5 separable blur of the parent sampled at
// floor((x+.5f)*sf) (MatchGPULib.cpp:1071-1096).  The reference blurs the whole parent level and
// then samples it; here the row pass is evaluated only in the sampled columns and the column pass
// only in the sampled rows.  One workgroup = 64x16 outputs of one plane; the parent region it needs
// (<= 133 x 37 for sf <= 2) is staged in LDS.
// =========================================================================================
constexpr int PTX = 64, PTY = 16, PRW = 2 * PTX + 8, PRH = 2 * PTY + 6;

__global__ __launch_bounds__(256) void k_blur_decimate_tiled(const float *__restrict__ src3, int W, int H, float *__restrict__ dst3,
                                                             int W2, int H2, float sf, unsigned *__restrict__ range_bad, int tiles_x, int n_tiles, Batch bt)
{
    __shared__ float sS[PRH * PRW];
    __shared__ float sT[PRH * PTX];
    if (bt.n > 1) {  // this workgroup's image of the batch (blockIdx.y)
        src3 = shifted(src3, bt.in[blockIdx.y]);
        dst3 = shifted(dst3, bt.out[blockIdx.y]);
        if (range_bad) range_bad += bt.cx[blockIdx.y];  // (the pair the image belongs to: the left and the right image of a pair share its range word)
    }
    const int tid = threadIdx.x;
    int tile_x, tile_y;
    xcd_tile(n_tiles, tiles_x, tile_x, tile_y);  // (grid: n_tiles x 1 x 3 planes; the plane only rotates the XCD labels)
    const int ox0 = tile_x * PTX, oy0 = tile_y * PTY;
    const float *src = src3 + (size_t)blockIdx.z * W * H;
    // parent region covered by this tile of outputs (sampling sites are monotone in ix / iy)
    const int ox1 = min(ox0 + PTX, W2) - 1, oy1 = min(oy0 + PTY, H2) - 1;
    const int rx0 = tex_index(((float)ox0 + 0.5f) * sf, W) - 2, ry0 = tex_index(((float)oy0 + 0.5f) * sf, H) - 2;
    const int rw = tex_index(((float)ox1 + 0.5f) * sf, W) + 2 - rx0 + 1, rh = tex_index(((float)oy1 + 0.5f) * sf, H) + 2 - ry0 + 1;
    {   // all of the thread's global loads first, then the LDS stores (a rolled loop pays one HBM round
        // trip per 256 pixels of the region)
        constexpr int NLD = (PRH * PRW + 255) / 256;
        float v[NLD];
#pragma unroll
        for (int u = 0; u < NLD; u++) {
            const int it = tid + u * 256;
            const int r = it / PRW, c = it - r * PRW;
            const int gx = rx0 + c, gy = ry0 + r;
            const bool in = r < rh && c < rw && gx >= 0 && gx < W && gy >= 0 && gy < H;
            v[u] = in ? src[(size_t)gy * W + gx] : 0.0f;  // zero padding (U2/U3)
        }
#pragma unroll
        for (int u = 0; u < NLD; u++) {
            const int it = tid + u * 256;
            if (it < rh * PRW) sS[it] = v[u];
        }
    }
    __syncthreads();
    // row pass at the sampled columns, every region row
    const int lx = tid & (PTX - 1);
    const int ix = ox0 + lx;
    const int cx = tex_index(((float)ix + 0.5f) * sf, W) - rx0;  // region column of the sampling site
    if (ix < W2) {
        for (int r = tid / PTX; r < rh; r += 256 / PTX) {
            const float *p = &sS[r * PRW + cx];
            sT[r * PTX + lx] = tap5(p[-2], p[-1], p[0], p[1], p[2]);
        }
    }
    __syncthreads();
    // column pass at the sampled rows; every value written is checked against the range the guarded division of
    // K-cost relies on (range_ok, ugsm_exact.hpp)
    bool bad = false;
    if (ix < W2) {
        for (int ly = tid / PTX; ly < PTY; ly += 256 / PTX) {
            const int iy = oy0 + ly;
            if (iy < H2) {
                const int cy = tex_index(((float)iy + 0.5f) * sf, H) - ry0;
                const float *p = &sT[cy * PTX + lx];
                const float v = tap5(p[-2 * PTX], p[-PTX], p[0], p[PTX], p[2 * PTX]);
                dst3[(size_t)blockIdx.z * W2 * H2 + (size_t)iy * W2 + ix] = v;
                bad |= !range_ok(v);
            }
        }
    }
    if (bad && range_bad) *range_bad = 1u;
}

// K-pyr for the factor-2 levels (level i+2 from level i, MatchGPULib.cpp:1088-1096: sf = 2.0f, sampling site 2i + 1), streaming form
// (round 4).  The tiled kernel above stages a 133 x 37 parent region per 64 x 16 outputs in LDS with scalar loads and runs at 2-3 TB/s on
// the levels that matter (16 MP: 43 / 26 / 16 us for levels 3 / 4 / 5, a third of a foveated pair's GPU time once the matching is
// batched).  Here ONE WAVE owns a strip of 30 output columns x HS output rows of one plane: lane l holds parent column X0 + l, so a
// parent row is one unit-stride load per lane; ALL 2 HS + 3 parent rows of the strip are requested before the first is used (one memory
// round trip per wave); the row pass is the systolic DPP chain of the marching K-cost (taps added in the reference's order j = -2..2,
// the window centred on column c complete in lane c + 2), evaluated densely and used at the odd columns; the column pass slides a window
// of five row-pass values down the parent rows and emits an output row every second one.  No LDS, no barrier.  Same arithmetic as the
// tiled kernel: zero padding outside the parent (U2/U3), row pass rounded to binary32 before the column pass; level values are >= 0, so
// tap5's leading "0 +" is exact (tap5p).
template <int HS>
__global__ __launch_bounds__(256) void k_blur_decimate2(const float *__restrict__ src3, int W, int H, float *__restrict__ dst3, int W2, int H2,
                                                        unsigned *__restrict__ range_bad, int strips_x, int n_strips, Batch bt)
{
    if (bt.n > 1) {  // this workgroup's image of the batch (blockIdx.y)
        src3 = shifted(src3, bt.in[blockIdx.y]);
        dst3 = shifted(dst3, bt.out[blockIdx.y]);
        if (range_bad) range_bad += bt.cx[blockIdx.y];  // (the pair the image belongs to)
    }
    constexpr int VXO = 30, NR = 2 * HS + 3;  // output columns per strip; parent rows 2 rs - 1 .. 2 (rs + HS - 1) + 3
    const int wv = (int)blockIdx.x * 4 + (int)(threadIdx.x >> 6);  // strip x plane
    if (wv >= 3 * n_strips) return;
    const int plane = wv / n_strips, strip = wv - plane * n_strips;
    const int sy = strip / strips_x, sx = strip - sy * strips_x;
    const int lane = threadIdx.x & 63;
    const int pc = 2 * VXO * sx - 1 + lane;  // parent column of this lane
    const int rs = sy * HS;
    const bool cin = pc >= 0 && pc < W;
    gchar_c *const Sb = uniform_base(src3 + (size_t)plane * W * H);
    const unsigned coff = (unsigned)clampi(pc, 0, W - 1) * 4u, pitch = (unsigned)W * 4u;
    float v[NR];
#pragma unroll
    for (int j = 0; j < NR; j++) {
        const int y = 2 * rs - 1 + j;
        const float t = ld_at(Sb, (unsigned)clampi(y, 0, H - 1) * pitch + coff);
        v[j] = (cin && y >= 0 && y < H) ? t : 0.0f;  // zero padding
    }
    // the window centred on parent column c is complete in lane (c - X0) + 2; output ix samples column 2 ix + 1
    const int ix = VXO * sx + ((lane - 4) >> 1);
    const bool out_lane = lane >= 4 && (lane & 1) == 0 && ix < W2;
    float *const dst = dst3 + (size_t)plane * W2 * H2;
    float h[5];
    bool bad = false;
#pragma unroll
    for (int j = 0; j < NR; j++) {
        // row pass (convolutionRowsKernel, MatchLib.cu:127-134): the partial sum travels one lane to the right per tap
        const float a0 = v[j] * UGSM_G0, a1 = v[j] * UGSM_G1, a2 = v[j] * UGSM_G2;
        const float p2 = lane_below(a0) + a1;
        const float p3 = lane_below(p2) + a2;
        const float p4 = lane_below(p3) + a1;
        h[j % 5] = lane_below(p4) + a0;
        if (j >= 4 && (j & 1) == 0) {  // parent row 2 iy + 3 has arrived: output row iy = rs + (j - 4) / 2
            const int iy = rs + (j - 4) / 2;
            const float o = tap5p(h[(j + 1) % 5], h[(j + 2) % 5], h[(j + 3) % 5], h[(j + 4) % 5], h[j % 5]);
            if (out_lane && iy < H2) {
                dst[(size_t)iy * W2 + ix] = o;
                bad |= !range_ok(o);
            }
        }
    }
    if (bad && range_bad) *range_bad = 1u;
}

// A = colconv_clamp(rowconv_clamp(L^2)) (Square + convolutionRows/ColumnsKernelT, MatchLib.cu:556-578,
// 1461-1565), once per level: it does not depend on the iteration.  64x16 tile, region +2 clamped.
__global__ __launch_bounds__(256) void k_sqblur_tiled(Img3 src, int W, int H, float *__restrict__ dst3, int tiles_x, int n_tiles, Batch bt)
{
    if (bt.n > 1) {  // this workgroup's pair of the batch (blockIdx.y)
        src.p = shifted(src.p, bt.img[blockIdx.y]);
        dst3 = shifted(dst3, bt.out[blockIdx.y]);
    }
    // One workgroup = the 64x16 tile of all three planes: a workgroup of this kernel lives about as long as its global loads
    // take to arrive, so the three planes' loads are in flight together (a third of the workgroups, each with three times the
    // loads outstanding).  Both passes work on quads (4 consecutive x) through 16-byte LDS accesses.  Squares are >= +0 (or NaN):
    // tap5p = tap5.
    constexpr int RW = PTX + 8, RH = PTY + 4;  // region columns 0 .. PTX+3 used; rows 16-byte aligned
    __shared__ __attribute__((aligned(16))) float sS[3][RH * RW];
    __shared__ __attribute__((aligned(16))) float sT[3][RH * PTX];
    const int tid = threadIdx.x;
    int tile_x, tile_y;
    xcd_tile(n_tiles, tiles_x, tile_x, tile_y);  // neighbouring tiles share an XCD's L2: their halo lines are fetched from HBM once
    const int x0 = tile_x * PTX, y0 = tile_y * PTY;
    {
        constexpr int LW = PTX + 4;
        constexpr int NLD = (RH * LW + 255) / 256;
        float v[NLD][3];
#pragma unroll
        for (int u = 0; u < NLD; u++) {
            const int it = min(tid + u * 256, RH * LW - 1);
            const int r = it / LW, c = it - r * LW;
            const size_t at = (size_t)clampi(y0 + r - 2, 0, H - 1) * src.pitch + clampi(x0 + c - 2, 0, W - 1);
#pragma unroll
            for (int k = 0; k < 3; k++) v[u][k] = src.p[k * src.plane + at];
        }
#pragma unroll
        for (int u = 0; u < NLD; u++) {
            const int it = tid + u * 256;
            if (it < RH * LW) {
#pragma unroll
                for (int k = 0; k < 3; k++) sS[k][(it / LW) * RW + (it % LW)] = v[u][k] * v[u][k];
            }
        }
    }
    __syncthreads();
    for (int it = tid; it < 3 * RH * (PTX / 4); it += 256) {  // rows: tile columns 4q .. 4q+3 from region columns 4q .. 4q+7
        const int k = it / (RH * (PTX / 4)), rem = it - k * (RH * (PTX / 4));
        const int r = rem / (PTX / 4), q = rem - r * (PTX / 4);
        float p[8], o[4];
        ld4(&sS[k][r * RW + 4 * q], p);
        ld4(&sS[k][r * RW + 4 * q + 4], p + 4);
#pragma unroll
        for (int i = 0; i < 4; i++) o[i] = tap5p(p[i], p[i + 1], p[i + 2], p[i + 3], p[i + 4]);
        st4(&sT[k][r * PTX + 4 * q], o);
    }
    __syncthreads();
    {   // columns: one quad of one tile row per thread and plane
        const int ly = tid / (PTX / 4), q = tid - ly * (PTX / 4);
        const int gx = x0 + 4 * q, gy = y0 + ly;
        if (gx < W && gy < H) {
#pragma unroll
            for (int k = 0; k < 3; k++) {
                float a[4], b[4], c[4], d[4], e[4], o[4];
                const float *p = &sT[k][ly * PTX + 4 * q];
                ld4(p, a); ld4(p + PTX, b); ld4(p + 2 * PTX, c); ld4(p + 3 * PTX, d); ld4(p + 4 * PTX, e);
#pragma unroll
                for (int i = 0; i < 4; i++) o[i] = tap5p(a[i], b[i], c[i], d[i], e[i]);
                float *const dst = dst3 + (size_t)k * W * H + (size_t)gy * W + gx;
                if ((W & 3) == 0) {  // (then gx + 3 < W and the row segment is 16-byte aligned)
                    *reinterpret_cast<float4 *>(dst) = make_float4(o[0], o[1], o[2], o[3]);
                } else {
                    for (int i = 0; i < 4; i++)
                        if (gx + i < W) dst[i] = o[i];
                }
            }
        }
    }
}

// =========================================================================================
// K-pyr-base: the three finest levels of one image in one pass over the rgb8 input --
//   level 0 = the planar float image (MatchGPULib.cpp:332-338),
//   level 1 = blur(level 0) sampled at floor((i+.5f)*(float)SCALE)   (MatchGPULib.cpp:1071-1087),
//   level 2 = blur(level 0) sampled at floor((i+.5f)*2.0f)           (:1088-1096).
// Separately (k_rgb_planes, then k_blur_decimate_tiled twice) level 0 is written once and read twice: 771 MB of
// traffic per 16 MP image; here the rgb8 tile is read once and the three levels written: 385 MB.  Same arithmetic:
// zero-padded row pass at the sampled columns of every region row (rounded to binary32), then the column pass at the
// sampled rows.  One workgroup = a 64x16 tile of level 0 (+ halo 2); every level-1 / level-2 pixel belongs to the tile
// that contains its sampling site, so each output is written exactly once.
// =========================================================================================
// Where the time goes (ablations, tools/kbench mode 9, 16 MP): the 5-tap passes are bound by LDS instructions, not by arithmetic.
// So the row pass is computed DENSELY, once for both levels, a quad of outputs from two 16-byte LDS reads (64 columns instead of
// the 84 candidate columns of the two levels, and a third of the LDS instructions); only the column pass runs at the sampled
// sites: waves 0-2 own the 52 candidate columns of level 1 (one row phase each), wave 3 the 32 columns of level 2 (two rows at a
// time), so that an output row segment is written by the lanes of one wave.  Level 0 leaves as 16-byte stores.
constexpr int BTX = 64, BTY = 16, BRW = BTX + 8, BRH = BTY + 4, BC1 = 52, BR1 = 16, BC2 = 32, BR2 = 8;
// ABL: development-only ablation mask (tools/kbench.hip mode 9): 1 = no level-1/2 stores, 2 = no column pass, 4 = no row pass, 8 = no level-0 store
template <int ABL = 0>
__global__ __launch_bounds__(256) void k_pyr_base(const uint8_t *__restrict__ rgb, int stride, int W, int H, float *__restrict__ lvl0,
                                                  float *__restrict__ lvl1, int W1, int H1, float *__restrict__ lvl2, int W2, int H2,
                                                  unsigned *__restrict__ range_bad, int tiles_x, int n_tiles, Batch bt, PyrWindow win)
{
    if (bt.n > 1) {  // this workgroup's image of the batch (blockIdx.y): its rgb8 input, its three levels (one offset: they lie in one pyramid)
        const int b = (int)blockIdx.y;
        rgb = shifted(rgb, bt.img[b]);
        lvl0 = shifted(lvl0, bt.out[b]);
        lvl1 = shifted(lvl1, bt.out[b]);
        lvl2 = shifted(lvl2, bt.out[b]);
        if (range_bad) range_bad += bt.cx[b];  // (the pair the image belongs to)
        win.x0 = (int)(bt.in[b] & 0xffffffffll);  // (its fovea window's origin rides in the otherwise unused input-field offset)
        win.y0 = (int)(bt.in[b] >> 32);
    }
    __shared__ __attribute__((aligned(16))) float sS[3][BRH * BRW];  // tile + halo 2: region column c at [c], rows 16-byte aligned
    __shared__ __attribute__((aligned(16))) float sT[3][BRH * BTX];  // row pass of every tile column, every region row
    __shared__ int sRowSite[BR1 + BR2];  // region row of the sampling site of candidate row ly (level 1, then level 2), -1 = not in this tile
    const int tid = threadIdx.x;
    int tile_x, tile_y;
    xcd_tile(n_tiles, tiles_x, tile_x, tile_y);  // neighbouring tiles share an XCD's L2 (halo lines, partially written output lines)
    const int x0 = tile_x * BTX, y0 = tile_y * BTY;
    const float sf1 = (float)1.41421356, sf2 = 2.0f;
    {   // rgb8 -> float planes of tile + halo 2, zero outside the image (the blur's zero padding, U2/U3)
        constexpr int RW = BTX + 4;
        constexpr int NLD = (BRH * RW + 255) / 256;
        float v[NLD][3];
#pragma unroll
        for (int u = 0; u < NLD; u++) {
            const int it = min(tid + u * 256, BRH * RW - 1);
            const int r = it / RW, c = it - r * RW;
            const int gx = x0 - 2 + c, gy = y0 - 2 + r;
            const bool in = gx >= 0 && gx < W && gy >= 0 && gy < H;
            const uint8_t *p = rgb + (size_t)min(max(gy, 0), H - 1) * stride + 3 * min(max(gx, 0), W - 1);
            const float a = (float)p[0], b = (float)p[1], c2 = (float)p[2];
            v[u][0] = in ? a : 0.0f;
            v[u][1] = in ? b : 0.0f;
            v[u][2] = in ? c2 : 0.0f;
        }
#pragma unroll
        for (int u = 0; u < NLD; u++) {
            const int it = tid + u * 256;
            if (it < BRH * RW) {
                const int r = it / RW, c = it - r * RW;
                sS[0][r * BRW + c] = v[u][0];
                sS[1][r * BRW + c] = v[u][1];
                sS[2][r * BRW + c] = v[u][2];
            }
        }
    }
    // candidate outputs of this tile: columns i = ib + lx, rows j = jb + ly; valid when the sampling site lies in the tile.
    // Level 1: the sites in a 64-wide tile are at most 46 consecutive i starting 1..3 above ib1 (52 candidates cover a
    // rounding slip of the float quotient); at most 12 rows, 16 candidates.  Level 2: site = 2i+1, exactly 32 x 8.
    const int ib1 = max((int)((float)x0 / sf1) - 1, 0), jb1 = max((int)((float)y0 / sf1) - 1, 0);
    const int ib2 = x0 / 2, jb2 = y0 / 2;
    if (tid < BR1 + BR2) {
        const bool o1 = tid < BR1;
        const int j = (o1 ? jb1 : jb2) + (o1 ? tid : tid - BR1);
        int site = -1;
        if (j < (o1 ? H1 : H2)) {
            const int sy = tex_index(((float)j + 0.5f) * (o1 ? sf1 : sf2), H);
            if (sy >= y0 && sy < y0 + BTY) site = sy - (y0 - 2);
        }
        sRowSite[tid] = site;
    }
    __syncthreads();
    // Foveated calls (win.w > 0) read level 0 only inside the fovea window (CreateFoveatedPyramid crops after a full build,
    // MatchGPULib.cpp:1128-1190; here the crop is a view, and what no view covers need not exist): the tiles that do not touch the
    // window skip their level-0 store -- 193 MB of the 338 MB this kernel writes per 16 MP image.  Levels 1 and 2 are written whole:
    // levels 3 and 4 are made from them.
    const bool store0 = win.w <= 0 || (x0 < win.x0 + win.w && x0 + BTX > win.x0 && y0 < win.y0 + win.h && y0 + BTY > win.y0);
    if (!(ABL & 8) && store0) {  // level 0: the tile itself; a thread owns 4 consecutive pixels of one row
        const size_t n = (size_t)W * H;
        const int r = tid >> 4, c = (tid & 15) * 4;
        const int gx = x0 + c, gy = y0 + r;
        if (gy < H && gx < W) {
            const size_t at = (size_t)gy * W + gx;
#pragma unroll
            for (int k = 0; k < 3; k++) {
                float q[4];
                ld2(&sS[k][(r + 2) * BRW + c + 2], q);
                ld2(&sS[k][(r + 2) * BRW + c + 4], q + 2);
                if ((W & 3) == 0) {  // (then gx + 3 < W, and every plane row starts 16-byte aligned)
                    *reinterpret_cast<float4 *>(lvl0 + k * n + at) = make_float4(q[0], q[1], q[2], q[3]);
                } else {
                    for (int i = 0; i < 4; i++)
                        if (gx + i < W) lvl0[k * n + at + i] = q[i];
                }
            }
        }
    }
    // row pass, dense: tile columns 4q .. 4q+3 of region row r from region columns 4q .. 4q+7 (level-0 values are >= 0: tap5p = tap5
    // without its "0 +")
    if constexpr (!(ABL & 4)) {
        for (int it = tid; it < 3 * BRH * (BTX / 4); it += 256) {
            const int k = it / (BRH * (BTX / 4)), rem = it - k * (BRH * (BTX / 4));
            const int r = rem / (BTX / 4), q = rem - r * (BTX / 4);
            float p[8], o[4];
            ld4(&sS[k][r * BRW + 4 * q], p);
            ld4(&sS[k][r * BRW + 4 * q + 4], p + 4);
#pragma unroll
            for (int i = 0; i < 4; i++) o[i] = tap5p(p[i], p[i + 1], p[i + 2], p[i + 3], p[i + 4]);
            st4(&sT[k][r * BTX + 4 * q], o);
        }
    }
    __syncthreads();
    // column pass at the sampled sites (level 0 holds the integers 0..255: always inside range_ok; levels 1 and 2 are checked)
    const int wave = tid >> 6, lane = tid & 63;
    const bool one = wave < 3;
    static_assert(BC1 <= 64 && 2 * BC2 == 64, "waves 0-2: one level-1 row per step; wave 3: two level-2 rows per step");
    const int lx = one ? lane : (lane % BC2);
    const int phase = one ? wave : (lane / BC2), nphase = one ? 3 : 2;
    const int ci = (one ? ib1 : ib2) + lx;
    int tcol = -1;  // tile column of this candidate column's sampling site, -1 = not in this tile
    if ((!one || lane < BC1) && ci < (one ? W1 : W2)) {
        const int site = tex_index(((float)ci + 0.5f) * (one ? sf1 : sf2), W);
        if (site >= x0 && site < x0 + BTX) tcol = site - x0;
    }
    bool bad = false;
    if (tcol >= 0 && !(ABL & 2)) {
        float *const dst = one ? lvl1 : lvl2;
        const int Wd = one ? W1 : W2;
        const size_t nd = (size_t)Wd * (one ? H1 : H2);
        const int nrow = one ? BR1 : BR2, jb = one ? jb1 : jb2, tb = one ? 0 : BR1;
        for (int ly = phase; ly < nrow; ly += nphase) {
            const int cy = sRowSite[tb + ly];
            if (cy >= 0) {
                const size_t at = (size_t)(jb + ly) * Wd + ci;
#pragma unroll
                for (int k = 0; k < 3; k++) {
                    const float *p = &sT[k][cy * BTX + tcol];
                    const float v = tap5p(p[-2 * BTX], p[-BTX], p[0], p[BTX], p[2 * BTX]);
                    if (!(ABL & 1) || v == 12345.678f) dst[k * nd + at] = v;
                    bad |= !range_ok(v);
                }
            }
        }
    }
    if (bad && range_bad) *range_bad = 1u;
}

// K-pyr-base, streaming form (round 4): the same three levels from the same rgb8 input with the same arithmetic, without LDS.  The tiled
// kernel above is bound by its LDS passes (ablations, tools/kbench mode 9: 44 us of 132 without them at 16 MP); here ONE WAVE owns a strip of
// 60 image columns and marches down HS rows: lane l holds column X0 + l, a row is three byte loads per lane; the dense row pass of every
// channel is the systolic DPP chain of the marching K-cost (taps in the reference's order, the window centred on column c complete in lane
// c + 2); the column pass slides a window of five row-pass values per channel down the rows and is evaluated only at the rows that are a
// sampling site of level 1 or level 2 -- both levels sample the SAME blurred image (MatchGPULib.cpp:1071-1096), so one value serves both.
// A lane's column is fixed for the strip: whether it is a sampling site of level 1 (floor((i + .5f) * (float)SCALE)) or of level 2
// (2 i + 1), and which output column it feeds, is worked out once; the row's sites are wave-uniform.  Level 0 leaves from the loaded values.
__device__ __forceinline__ int pyr_site_index(const int pos, const float sf, const int n_src, const int n_dst)  // i with tex_index((i + .5f) * sf, n_src) == pos, or -1
{
    const int i0 = (int)((float)pos / sf);
#pragma unroll
    for (int d = -2; d <= 2; d++) {
        const int i = i0 + d;
        if (i >= 0 && i < n_dst && tex_index(((float)i + 0.5f) * sf, n_src) == pos) return i;
    }
    return -1;
}
template <int HS>
__global__ __launch_bounds__(256) void k_pyr_base_march(const uint8_t *__restrict__ rgb, int stride, int W, int H, float *__restrict__ lvl0,
                                                        float *__restrict__ lvl1, int W1, int H1, float *__restrict__ lvl2, int W2, int H2,
                                                        unsigned *__restrict__ range_bad, int strips_x, int n_strips, Batch bt, PyrWindow win)
{
    if (bt.n > 1) {  // this workgroup's image of the batch (blockIdx.y)
        const int b = (int)blockIdx.y;
        rgb = shifted(rgb, bt.img[b]);
        lvl0 = shifted(lvl0, bt.out[b]);
        lvl1 = shifted(lvl1, bt.out[b]);
        lvl2 = shifted(lvl2, bt.out[b]);
        if (range_bad) range_bad += bt.cx[b];
        win.x0 = (int)(bt.in[b] & 0xffffffffll);
        win.y0 = (int)(bt.in[b] >> 32);
    }
    constexpr int VXS = 60;
    const int wv = (int)blockIdx.x * 4 + (int)(threadIdx.x >> 6);
    if (wv >= n_strips) return;
    const int sy = wv / strips_x, sx = wv - sy * strips_x;
    const int lane = threadIdx.x & 63;
    const int pc = sx * VXS - 2 + lane;        // the column whose pixel this lane holds
    const int cc = pc - 2;                     // ... and the column whose row-pass window is complete in this lane
    const int y0 = sy * HS;
    const int y1 = min(y0 + HS, H);            // centre rows y0 .. y1 - 1
    const float sf1 = (float)1.41421356, sf2 = 2.0f;
    const bool cin = pc >= 0 && pc < W;
    const bool own = lane >= 2 && lane < 2 + VXS && pc < W;  // columns sx * 60 .. + 59: this lane stores their level-0 pixels
    const bool centre = lane >= 4 && cc < W;   // lanes 4 .. 63 hold the windows of columns sx * 60 .. + 59
    const int i1 = centre ? pyr_site_index(cc, sf1, W, W1) : -1;
    const int i2 = (centre && (cc & 1) && (cc >> 1) < W2 && tex_index(((float)(cc >> 1) + 0.5f) * sf2, W) == cc) ? (cc >> 1) : -1;
    // level 0 is stored where the call reads it: everywhere (win.w <= 0) or in the strips that touch the fovea window
    const bool store0 = win.w <= 0 || (sx * VXS < win.x0 + win.w && sx * VXS + VXS > win.x0 && y0 < win.y0 + win.h && y1 > win.y0);
    const size_t n0 = (size_t)W * H, n1 = (size_t)W1 * H1, n2 = (size_t)W2 * H2;
    const uint8_t *const col = rgb + 3 * (size_t)clampi(pc, 0, W - 1);
    auto load = [&](const int y, unsigned (&b)[3]) {
        const uint8_t *p = col + (size_t)clampi(y, 0, H - 1) * stride;
        b[0] = p[0];
        b[1] = p[1];
        b[2] = p[2];
    };
    float w[3][5];
#pragma unroll
    for (int k = 0; k < 3; k++)
#pragma unroll
        for (int u = 0; u < 5; u++) w[k][u] = 0.0f;
    bool bad = false;
    unsigned bcur[3], bnx1[3], bnx2[3];
    load(y0 - 2, bcur);
    load(y0 - 1, bnx1);
    // the next level-1 row whose sampling site lies at or below y0, and that site (the sites increase strictly: a row is the site of one j at most)
    int jn = max((int)((float)y0 / sf1) - 2, 0);
    int sn = tex_index(((float)jn + 0.5f) * sf1, H);
    while (sn < y0 && jn < H1) {
        jn++;
        sn = tex_index(((float)jn + 0.5f) * sf1, H);
    }
    for (int y = y0 - 2; y < y1 + 2; y++) {
        load(y + 2, bnx2);  // two rows ahead of the arithmetic
        const bool yin = y >= 0 && y < H;
        const int cr = y - 2;  // the row whose column window is complete once row y is in
        // (wave-uniform) is cr a sampling row of level 1 / level 2?
        int j1 = -1;
        if (cr >= y0 && cr < y1 && jn < H1 && cr == sn) {
            j1 = jn;
            jn++;
            sn = tex_index(((float)jn + 0.5f) * sf1, H);
        }
        const int j2 = (cr >= y0 && cr < y1 && (cr & 1) && (cr >> 1) < H2) ? (cr >> 1) : -1;
#pragma unroll
        for (int k = 0; k < 3; k++) {
            const float v = (cin && yin) ? (float)bcur[k] : 0.0f;  // zero padding (U2/U3)
            if (store0 && own && y >= y0 && y < y1) lvl0[k * n0 + (size_t)y * W + pc] = v;
            // row pass (level-0 values are >= 0: tap5p = tap5 without its "0 +"), the partial sum travels one lane to the right per tap
            const float a0 = v * UGSM_G0, a1 = v * UGSM_G1, a2 = v * UGSM_G2;
            const float p2 = lane_below(a0) + a1;
            const float p3 = lane_below(p2) + a2;
            const float p4 = lane_below(p3) + a1;
            w[k][0] = w[k][1];
            w[k][1] = w[k][2];
            w[k][2] = w[k][3];
            w[k][3] = w[k][4];
            w[k][4] = lane_below(p4) + a0;
        }
        if (j1 >= 0 || j2 >= 0) {
#pragma unroll
            for (int k = 0; k < 3; k++) {
                const float o = tap5p(w[k][0], w[k][1], w[k][2], w[k][3], w[k][4]);
                if (j1 >= 0 && i1 >= 0) {
                    lvl1[k * n1 + (size_t)j1 * W1 + i1] = o;
                    bad |= !range_ok(o);
                }
                if (j2 >= 0 && i2 >= 0) {
                    lvl2[k * n2 + (size_t)j2 * W2 + i2] = o;
                    bad |= !range_ok(o);
                }
            }
        }
#pragma unroll
        for (int k = 0; k < 3; k++) {
            bcur[k] = bnx1[k];
            bnx1[k] = bnx2[k];
        }
    }
    if (bad && range_bad) *range_bad = 1u;
}
int pyr_base_streaming = 1;  // (development: UGSM_PYR_BASE_STREAM=0 -> the LDS-tiled k_pyr_base)

void launch_pyr_base(hipStream_t st, const uint8_t *rgb, int stride, int W, int H, float *lvl0, float *lvl1, int W1, int H1, float *lvl2, int W2,
                     int H2, unsigned *range_bad, const Batch *bt, PyrWindow win)
{
    Batch one{};
    one.n = 1;
    const Batch &B = bt ? *bt : one;
    // Streaming form for the foveated calls only (win.w > 0: level 0 is stored in the window's strips alone).  Where level 0 is written
    // whole -- 193 of 338 MB per 16 MP image -- the tiled kernel's aligned 16-byte stores win: 132 against 151 us per image, 16 MP full mode
    // 183.7 against 179.4 pairs/s; foveated batches 880 -> 903 pairs/s with it (tools/ab.py, same box).  UGSM_PYR_BASE_STREAM=2: everywhere.
    if ((pyr_base_streaming == 1 && win.w > 0) || pyr_base_streaming == 2) {
        constexpr int HS = 32;
        const int strips_x = (W + 59) / 60, n_strips = strips_x * ((H + HS - 1) / HS);
        UGSM_LAUNCH(k_pyr_base_march<HS>, dim3((n_strips + 3) / 4, B.n > 1 ? B.n : 1), dim3(256), 0, st, rgb, stride, W, H, lvl0, lvl1, W1, H1, lvl2, W2,
                           H2, range_bad, strips_x, n_strips, B, win);
        return;
    }
    const int tiles_x = (W + BTX - 1) / BTX, n_tiles = tiles_x * ((H + BTY - 1) / BTY);
    UGSM_LAUNCH(k_pyr_base<0>, dim3(n_tiles, B.n > 1 ? B.n : 1), dim3(256), 0, st, rgb, stride, W, H, lvl0, lvl1, W1, H1, lvl2, W2, H2, range_bad, tiles_x, n_tiles, B, win);
}

int blur_decimate_streaming = 1;  // (development: UGSM_PYR_STREAM=0 -> the tiled kernel for the factor-2 levels too)
long long blur_decimate_streaming_min = 0;  // (development: UGSM_PYR_STREAM_MIN: launches of fewer output pixels keep the tiled kernel)
// bt (optional): bt->n IMAGES in one launch -- image j reads src3 + in[j], writes dst3 + out[j] and reports into range_bad[cx[j]]
void launch_blur_decimate(hipStream_t st, const float *src3, int W, int H, float *dst3, int W2, int H2, float sf, unsigned *range_bad, const Batch *bt,
                          long long stream_min)
{
    Batch one{};
    one.n = 1;
    const Batch &B = bt ? *bt : one;
    const int images = B.n > 1 ? B.n : 1;
#ifdef UGSM_DEV_LIB  // (the product's pyramid only ever asks for sqrt 2 and 2; the one-kernel-per-stage fallback lives in libugsm_dev.so)
    if (sf > 2.0f || sf < 1.0f) {  // region bound assumes 1 <= sf <= 2 (the reference uses sqrt2 and 2)
        for (int j = 0; j < images; j++) {
            const float *src = images > 1 ? reinterpret_cast<const float *>(reinterpret_cast<const char *>(src3) + B.in[j]) : src3;
            float *dst = images > 1 ? reinterpret_cast<float *>(reinterpret_cast<char *>(dst3) + B.out[j]) : dst3;
            launch_blur_decimate_ref(st, src, W, H, dst, W2, H2, sf);
            if (range_bad) launch_range_scan(st, dst, 3 * (size_t)W2 * H2, range_bad + (images > 1 ? B.cx[j] : 0));
        }
        return;
    }
#endif
    const long long out_px = (long long)W2 * H2 * images;
    if (sf == 2.0f && blur_decimate_streaming && out_px >= std::max(blur_decimate_streaming_min, stream_min)) {  // every level from the third on: the streaming form
        const int hs = out_px >= 400000 ? 16 : (out_px >= 40000 ? 8 : 4);  // short strips where there are few: a launch lasts as long as one wave
        const int strips_x = (W2 + 29) / 30, n_strips = strips_x * ((H2 + hs - 1) / hs);
        const dim3 grid((3 * n_strips + 3) / 4, images);
        if (hs == 16) UGSM_LAUNCH(k_blur_decimate2<16>, grid, dim3(256), 0, st, src3, W, H, dst3, W2, H2, range_bad, strips_x, n_strips, B);
        else if (hs == 8) UGSM_LAUNCH(k_blur_decimate2<8>, grid, dim3(256), 0, st, src3, W, H, dst3, W2, H2, range_bad, strips_x, n_strips, B);
        else UGSM_LAUNCH(k_blur_decimate2<4>, grid, dim3(256), 0, st, src3, W, H, dst3, W2, H2, range_bad, strips_x, n_strips, B);
        return;
    }
    const int tiles_x = (W2 + PTX - 1) / PTX, n_tiles = tiles_x * ((H2 + PTY - 1) / PTY);
    UGSM_LAUNCH(k_blur_decimate_tiled, dim3(n_tiles, images, 3), dim3(256), 0, st, src3, W, H, dst3, W2, H2, sf, range_bad, tiles_x, n_tiles, B);
}

void launch_sqblur_clamp(hipStream_t st, Img3 src, int W, int H, float *dst3, const Batch *bt)
{
    Batch one{};
    one.n = 1;
    const Batch &B = bt ? *bt : one;
    const int tiles_x = (W + PTX - 1) / PTX, n_tiles = tiles_x * ((H + PTY - 1) / PTY);
    UGSM_LAUNCH(k_sqblur_tiled, dim3(n_tiles, B.n > 1 ? B.n : 1), dim3(256), 0, st, src, W, H, dst3, tiles_x, n_tiles, B);
}

}  // namespace ugsm
