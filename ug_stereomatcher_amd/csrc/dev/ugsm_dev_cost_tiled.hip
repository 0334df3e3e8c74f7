// dev/ugsm_dev_cost_tiled.hip -- libugsm_dev.so only: round 1's K-cost, LDS-tiled (k_cost_split).  The product runs the marching forms
// (ugsm_kernels_march.hip, ugsm_kernels_march4.hip) and the coarse-level form (ugsm_kernels_small.hip) on every level; this kernel is
// what ugsm_config.march_min_pixels < 0 selects -- kept as a second, structurally different statement of the same iteration that the
// tests compare the product's kernels with, bit for bit.  ~1080 VALU lane-instructions per pixel-iteration against ~720.
//
// One matcher iteration's warp + 5-shift squared-NCC cost (3 channels) + parabola + confidence blend + disparity update, one launch.
// Citations: /root/reference/src/gpu_matcher/<file>:<line>.
#include "../ugsm_exact.hpp"
#include "../ugsm_launch.hpp"

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


// -----------------------------------------------------------------------------------------
// k_cost_split: TWO threads per quad (512-thread workgroup): the LDS footprint per tile -- not registers -- caps the
// workgroups per CU, so the waves per LDS byte are doubled by splitting each quad's work by correlation shift:
//   role 0 (threads 0..255)   = shifts (-1,0), (+1,0), pixels 0,1 of shift (0,0), parabola x
//   role 1 (threads 256..511) = shifts (0,-1), (0,+1), pixels 2,3 of shift (0,0), the R'^2 row pass, parabola y
// (the B column pass is shared by both).  The roles are wave-uniform (no divergence).  They meet once per tile:
// each publishes its two pixels of Q(0,0), then the x / y parabola results (LDS exchanges in the dead sRow planes).
// Where the time goes (SQ counters, profiles/): 1 880 VALU instructions per wave, 0.23 per SIMD cycle at 4 waves/SIMD,
// about three quarters of the issue capacity at the measured ~3.3 cycles per instruction.
constexpr int kSplitWaves = 4;  // waves per SIMD the register allocation aims at (2 workgroups per CU)
// The body of k_cost_split.  INTERIOR: the tile and its halo of 3 lie inside the image, so the address clamps of P0, the
// zero-padding and validity selects, the clamped B fetches of P3 and the store bounds are compiled out (a workgroup-uniform
// choice made by the kernel below; about 95 % of the tiles of a 16 MP level).
template <bool INTERIOR>
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
    }

    // ---- epilogue: parabola x (role 0) / y (role 1) in the compute mapping, hand-over through LDS, then
    // update + blend + coalesced stores with lanes along the rows ---------------------------------------
    __syncthreads();  // every P3 is done with sRow: planes 0..3 become hand-over buffers
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
}

__global__ __launch_bounds__(512, kSplitWaves) void k_cost_split(Img3 L, Img3 R, const float *__restrict__ A3, const float *__restrict__ d3,
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
    const bool interior = x0 >= 3 && y0 >= 3 && x0 + TX + 3 <= W && y0 + TY + 3 <= H;
    if (interior) cost_split_body<true>(L, R, A3, d3, nd3, W, H, thr, blend, x0, y0, sR, sL, sRow, sBrow, sB, sA);
    else cost_split_body<false>(L, R, A3, d3, nd3, W, H, thr, blend, x0, y0, sR, sL, sRow, sBrow, sB, sA);
}

void launch_cost_fused(hipStream_t st, Img3 L, Img3 R, const float *A3, const float *d3, float *nd3, int W, int H, float thr, int blend)
{
    const int tiles_x = (W + TX - 1) / TX, n_tiles = tiles_x * ((H + TY - 1) / TY);
    UGSM_LAUNCH(k_cost_split, dim3(n_tiles), dim3(512), 0, st, L, R, A3, d3, nd3, W, H, thr, blend, tiles_x, n_tiles);
}

}  // namespace ugsm
