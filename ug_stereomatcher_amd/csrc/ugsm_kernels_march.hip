// ugsm_kernels_march.hip -- K-cost for the large levels as a marching (row-streaming) kernel.
//
// Same arithmetic as the CPU oracle (and as libugsm_dev.so's LDS-tiled k_cost_split), bit for bit: one matcher iteration's
// warp + 5-shift squared-NCC cost over 3 channels + parabola + confidence blend + disparity update
// (matchlevel, /root/reference/src/gpu_matcher/MatchGPULib.cpp:1745-2250 and the MatchLib.cu kernels cited below).
//
// Layout: ONE WAVE owns a strip of 64 image columns (one pixel per lane, lanes along the scanline) and marches
// down the rows.  Nothing goes through LDS and there is no barrier:
//   * horizontal neighbours (the +-1 shifts, the 5-tap row passes, B at x+-1) are the neighbouring lanes' registers,
//     read by DPP wave shifts that the compiler folds into the consuming v_add/v_mul;
//   * the 5-tap COLUMN passes are transposed-form FIR filters: a row's three tap products h*g0, h*g1, h*g2 are formed
//     once and added into four running partial sums per image; because the reference adds its taps in row order
//     j = -2..2 (MatchLib.cu:127-134, 1484-1487) the arrival order of the rows IS the reference's summation order,
//     so the partial sums hold exactly the reference's intermediate values.  7 VALU per pixel and pass
//     (3 products shared between the outputs they feed + 4 adds) instead of 9;
//   * global loads run two rows ahead ((dx,dy)) / one row ahead (the warped gather of R, L, A) of the arithmetic.
// Per pixel-iteration this is ~720 VALU lane-instructions against ~1080 in the LDS-tiled kernel (SQ_INSTS_VALU, profiles/),
// with the halo recomputed only 6 rows per strip (tile: 6 rows per 28) and 6 columns per 64.
//
// Border semantics without cross-lane fix-ups: a lane whose pixel lies outside the image computes
// the warp at the CLAMPED pixel, which is what a clamp-addressed fetch of R' returns (texture clamp, MatchLib.cu:56-60),
// so R' is edge-replicated by construction; L is taken as zero outside (zero-padded smem convolution of the products,
// SURVEY 9 U2/U3); only the five B fetches at clamped positions need a select, in the strips that touch the frame.
//
// What was measured and not kept -- two pixels per lane (spills at any occupancy that pays), FMA-contracted convolutions (slower, and no
// parity claim), row passes chain after chain instead of in lockstep, row clamps in the interior strips, s_setprio: docs/HISTORY.md,
// profiles/r05_kbench_march_issue.txt.
#include "ugsm_exact.hpp"
#include "ugsm_launch.hpp"
#include <type_traits>

namespace ugsm {

// Pixels a lane holds.  The per-lane values stay arrays of NP: rewriting them as scalars -- the same operations -- makes the compiler schedule the
// row step differently and the kernel 2.5-4 % slower (profiles/r06_kbench_march_scalar.txt).  This kernel's code generation follows the shape
// of its source: tools/isa_dump.py tells whether an edit moved its instruction stream, tools/kbench mode 2 what that costs.
constexpr int NP = 1;

// value of the lane below / above (lane 0 / lane 63 read 0: those lanes hold strip halo whose results are dropped)
__device__ __forceinline__ float shr1(float v)  // from lane - 1
{
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x138, 0xf, 0xf, true));
}
__device__ __forceinline__ float shl1(float v)  // from lane + 1
{
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x130, 0xf, 0xf, true));
}

// the value of `v` at the pixel column next to the lane's own: OFF = -1 or +1
template <int OFF>
__device__ __forceinline__ float nbr(const float (&v)[NP])
{
    static_assert(OFF == -1 || OFF == 1, "");
    return OFF < 0 ? shr1(v[0]) : shl1(v[0]);
}

// 5-tap row pass over the lane-distributed row p, taps added in the reference's order j = -2..2.
// Products are >= +0 so the reference's leading "0 +" is exact (tap5p, ugsm_exact.hpp).
__device__ __forceinline__ void rowconv5(const float (&p)[NP], float (&out)[NP])
{
    // the partial sum travels one lane to the right per tap (systolic), so every add takes its left operand through DPP and no
    // separate lane move is needed; the finished sum of the window centred on column c arrives in lane c + 2 (March::SKEW):
    // everything downstream of a row pass lives two lanes right of its pixel
    const float a0 = p[0] * UGSM_G0, a1 = p[0] * UGSM_G1, a2 = p[0] * UGSM_G2;
    const float p2 = shr1(a0) + a1;
    const float p3 = shr1(p2) + a2;
    const float p4 = shr1(p3) + a1;
    out[0] = shr1(p4) + a0;
}

// The N systolic row passes of one colour channel -- R'^2 and the five products -- advance in LOCKSTEP, stage by stage, instead of chain
// after chain: between a chain's VALU result and the DPP read of it one stage later lie N - 1 independent instructions, so the two wait
// states of the VALU-write -> DPP-read hazard need no s_nop (978 -> 191 in the kernel; profiles/r05_kbench_march_issue.txt).
template <int N>
__device__ __forceinline__ void rowconv5_lockstep(const float (&v)[N], float (&out)[N])
{
    float a0[N], a1[N], a2[N], t[N];
#pragma unroll
    for (int c = 0; c < N; c++) {
        a0[c] = v[c] * UGSM_G0;
        a1[c] = v[c] * UGSM_G1;
        a2[c] = v[c] * UGSM_G2;
    }
#pragma unroll
    for (int c = 0; c < N; c++) t[c] = shr1(a0[c]) + a1[c];
#pragma unroll
    for (int c = 0; c < N; c++) t[c] = shr1(t[c]) + a2[c];
#pragma unroll
    for (int c = 0; c < N; c++) t[c] = shr1(t[c]) + a1[c];
#pragma unroll
    for (int c = 0; c < N; c++) out[c] = shr1(t[c]) + a0[c];
}

// One step of the transposed-form 5-tap column pass: `h` is the row-pass value of the row that has just arrived;
// s[0..3] hold the partial sums of the four output rows still open.  Returns the sum of the row that closes (two rows up).
__device__ __forceinline__ float colstep5(float (&s)[4], const float h)
{
    const float a0 = h * UGSM_G0, a1 = h * UGSM_G1, a2 = h * UGSM_G2;
    const float out = s[3] + a0;
    s[3] = s[2] + a1;
    s[2] = s[1] + a2;
    s[1] = s[0] + a1;
    s[0] = a0;
    return out;
}

constexpr int kMarchWaves = 3;  // waves per SIMD the register allocation aims at
struct March {
    static constexpr int COLS = 64;       // columns a wave holds
    static constexpr int VX = COLS - 6;   // columns it produces (halo 3 on both sides)
    static constexpr int SKEW = 2;        // lanes between a pixel and the results of its row passes (rowconv5)
    static constexpr int ORG = 0;         // first output column of strip 0
};

// loads of one image row into the lane-distributed form: uniform plane base + the lanes' 32-bit byte offsets
__device__ __forceinline__ void ld_row(gchar_c *base, const unsigned (&off)[NP], float (&o)[NP])
{
#pragma unroll
    for (int j = 0; j < NP; j++) o[j] = ld_at(base, off[j]);
}

// tex_index (ugsm_device.hpp) without branches: floor, clamp to [0, n-1] in float, convert.  v_med3_f32 returns
// min3 of its operands when one of them is a NaN, and v_min_f32 ignores a quiet NaN, so NaN -> 0 as in tex_index;
// +-Inf and values beyond the int range are clamped before the conversion.  (tests: test_march_wild_disparities)
__device__ __forceinline__ int tex_index_nb(const float coord, const float nm1)
{
    return (int)__builtin_amdgcn_fmed3f(floorf(coord), 0.0f, nm1);
}

// what the load pipeline holds for one row r: (dx,dy)(r+1), the gathered R'(r), L(r-1), A(r-3) and the strip's own (dx,dy,conf)(r-3)
struct MarchRow {
    float dx[NP], dy[NP];
    float R[3][NP], L[3][NP], A[3][NP], O[3][NP];
};

// SEED: d3 is the coarser level's field (sm.Ws x sm.Hs) and this is the level's first iteration: every (dx, dy, conf) the
// iteration reads is subsampleDispKernel's value (MatchLib.cu:372-401, k_seed in ugsm_kernels_ref.hip) formed on the fly,
// SCALE * coarse[floor((x + cx + .5f) * sf), floor((y + cy + .5f) * sf)] with the product in binary64 -- the seeded field is never
// written to memory (at 16 MP: 290 MB and a 95 us launch per level saved for ~3 % more arithmetic in this one iteration).
template <bool EDGE, bool FAST, bool SEED = false>
__device__ __forceinline__ void cost_march_body(const Img3 &L, const Img3 &R, const float *__restrict__ A3, const float *__restrict__ d3,
                                                float *__restrict__ nd3, const int W, const int H, const float thr, const int blend,
                                                const int X0, const int xs, const int xe, const int ys, const int ye, const SeedMap sm = SeedMap{0, 0, 0, 0})
{
    const int lane = threadIdx.x & 63;
    const size_t n = (size_t)W * H;
    const size_t nD = SEED ? (size_t)sm.Ws * sm.Hs : n;  // plane size of the field d3 points at
    gchar_c *const Lb[3] = {uniform_base(L.p), uniform_base(L.p + L.plane), uniform_base(L.p + 2 * L.plane)};
    gchar_c *const Rb[3] = {uniform_base(R.p), uniform_base(R.p + R.plane), uniform_base(R.p + 2 * R.plane)};
    gchar_c *const Ab[3] = {uniform_base(A3), uniform_base(A3 + n), uniform_base(A3 + 2 * n)};
    gchar_c *const Db[3] = {uniform_base(d3), uniform_base(d3 + nD), uniform_base(d3 + 2 * nD)};
    gchar_c *const Nb[3] = {uniform_base(nd3), uniform_base(nd3 + n), uniform_base(nd3 + 2 * n)};

    // per-lane column constants.  px: the pixel whose R', L and products the lane holds; po = px - SKEW: the pixel whose row-pass
    // results (N, B), A, own (dx,dy,conf) and output it holds
    constexpr int SKEW = March::SKEW;
    int px[NP], po[NP];
    unsigned coff[NP], coffo[NP];  // byte offsets of the (clamped) columns px / po inside a row
    float xc[NP];                  // warp x coordinate of the (clamped) pixel centre
    bool cin[NP], stv[NP];
#pragma unroll
    for (int j = 0; j < NP; j++) {
        px[j] = X0 + NP * lane + j;
        po[j] = px[j] - SKEW;
        const int pc = EDGE ? clampi(px[j], 0, W - 1) : px[j];
        cin[j] = !EDGE || (px[j] >= 0 && px[j] < W);
        coff[j] = (unsigned)pc * 4u;
        coffo[j] = (unsigned)(EDGE || SKEW ? clampi(po[j], 0, W - 1) : po[j]) * 4u;
        xc[j] = (float)pc + 0.5f;
        stv[j] = po[j] >= xs && po[j] < xe;
    }
    const float seed_sf = (float)(1 / UGSM_SCALE);
    unsigned scol[NP], scolo[NP];  // SEED: byte offsets of the coarse columns the (clamped) columns px / po sample
    if constexpr (SEED) {
#pragma unroll
        for (int j = 0; j < NP; j++) {
            scol[j] = (unsigned)tex_index(((float)(clampi(px[j], 0, W - 1) + sm.cx) + 0.5f) * seed_sf, sm.Ws) * 4u;
            scolo[j] = (unsigned)tex_index(((float)(clampi(po[j], 0, W - 1) + sm.cx) + 0.5f) * seed_sf, sm.Ws) * 4u;
        }
    }
    auto seedv = [&](const float v) -> float {
        if constexpr (SEED) return (float)(UGSM_SCALE * (double)v);
        else return v;
    };
    const unsigned pitchW = (unsigned)W * 4u, pitchL = (unsigned)L.pitch * 4u;
    const float wm1 = (float)(W - 1), hm1 = (float)(H - 1);
    // rows are clamped (scalar ops) only in the strips that touch the frame: an interior strip keeps six rows clear of it (`interior`,
    // k_cost_march)
    auto rowc = [&](int r) { return !EDGE ? r : min(max(r, 0), H - 1); };
    auto row_off = [&](const int r, const unsigned pitch, unsigned (&off)[NP], const bool skewed = false) {
        const unsigned ro = (unsigned)rowc(r) * pitch;
#pragma unroll
        for (int j = 0; j < NP; j++) off[j] = ro + (skewed ? coffo[j] : coff[j]);
    };

    // row offsets into the field d3 points at: the level's own rows, or (SEED) the coarse rows they sample
    auto d_off = [&](const int r, unsigned (&off)[NP], const bool skewed) {
        if constexpr (SEED) {
            const unsigned ro = (unsigned)tex_index(((float)(rowc(r) + sm.cy) + 0.5f) * seed_sf, sm.Hs) * ((unsigned)sm.Ws * 4u);
#pragma unroll
            for (int j = 0; j < NP; j++) off[j] = ro + (skewed ? scolo[j] : scol[j]);
        } else {
            row_off(r, pitchW, off, skewed);
        }
    };
    auto load_d = [&](const int r, float (&dx)[NP], float (&dy)[NP]) {
        unsigned off[NP];
        d_off(r, off, false);
        ld_row(Db[0], off, dx);
        ld_row(Db[1], off, dy);
    };
    // warpAbyB (MatchLib.cu:510-515): R'[x,y] = tex(R, x + 0.5 + dx, y + 0.5 + dy) at the clamped pixel
    auto gather = [&](const int r, const float (&dx)[NP], const float (&dy)[NP], float (&o)[3][NP]) {
        const float yc = (float)rowc(r) + 0.5f;
#pragma unroll
        for (int j = 0; j < NP; j++) {
            const int sx = tex_index_nb(xc[j] + seedv(dx[j]), wm1);
            const int sy = tex_index_nb(yc + seedv(dy[j]), hm1);
            const unsigned off = (__umul24((unsigned)sy, (unsigned)R.pitch) + (unsigned)sx) * 4u;
#pragma unroll
            for (int k = 0; k < 3; k++) o[k][j] = ld_at(Rb[k], off);
        }
    };
    auto load_L = [&](const int r, float (&o)[3][NP]) {
        unsigned off[NP];
        row_off(r, pitchL, off);
#pragma unroll
        for (int k = 0; k < 3; k++) ld_row(Lb[k], off, o[k]);
    };
    auto load_AO = [&](const int r, float (&a)[3][NP], float (&od)[3][NP]) {
        unsigned off[NP];
        row_off(r, pitchW, off, true);
#pragma unroll
        for (int k = 0; k < 3; k++) ld_row(Ab[k], off, a[k]);
        unsigned offd[NP];
        d_off(r, offd, true);
#pragma unroll
        for (int k = 0; k < 3; k++) ld_row(Db[k], offd, od[k]);
    };

    // ---- state carried down the rows --------------------------------------------------------------------------
    float Rm1[3][NP], Rm2[3][NP];  // R'(r-1), R'(r-2)
    float Bm1[3][NP], Bm2[3][NP];  // B(r-3), B(r-4)
    float aB[3][NP][4];            // open partial sums of the B column pass
    float aN[3][5][NP][4];         // open partial sums of the five product column passes
#pragma unroll
    for (int k = 0; k < 3; k++)
#pragma unroll
        for (int j = 0; j < NP; j++) {
            Rm1[k][j] = Rm2[k][j] = Bm1[k][j] = Bm2[k][j] = 0.0f;
#pragma unroll
            for (int u = 0; u < 4; u++) {
                aB[k][j][u] = 0.0f;
#pragma unroll
                for (int s = 0; s < 5; s++) aN[k][s][j][u] = 0.0f;
            }
        }

    // One row step.  `cur` holds row r's loads (issued during the previous step); the next row's loads are issued into
    // `nxt` before the arithmetic.  The two sets swap roles from step to step (the row loop is unrolled by two), so a
    // loaded register is never copied: a copy would have to wait for its load and would end the prefetch.
    auto step = [&](const int r, MarchRow &cur, MarchRow &nxt) {
        load_d(r + 2, nxt.dx, nxt.dy);
        gather(r + 1, cur.dx, cur.dy, nxt.R);
        load_L(r, nxt.L);
        load_AO(r - 2, nxt.A, nxt.O);

        // ---- arithmetic on R'(r), L(r-1), A(r-3) ------------------------------------------------------------
        const int y = r - 1, o = r - 3;
        const bool do_prod = r >= ys - 1;
        const bool do_out = o >= ys && o < ye;  // (the row loop may run one step past the strip: two steps per trip)
        const bool yin = !EDGE || (y >= 0 && y < H);
        float Q[5][NP];
#pragma unroll
        for (int k = 0; k < 3; k++) {
            float rc[NP], sq[NP], hb[NP], bnew[NP];
#pragma unroll
            for (int j = 0; j < NP; j++) {
                rc[j] = cur.R[k][j];
                sq[j] = rc[j] * rc[j];  // Square, MatchLib.cu:569-570
            }
            if (!do_prod) {  // (the strip's first rows: only B is due)
                rowconv5(sq, hb);  // convolutionRowsKernelT / ColumnsKernelT on R'^2 (clamp): B
#pragma unroll
                for (int j = 0; j < NP; j++) bnew[j] = colstep5(aB[k][j], hb[j]);  // = B(r-2)
            }
            if (do_prod) {
                float l[NP], p[5][NP], Nv[5][NP];
#pragma unroll
                for (int j = 0; j < NP; j++) l[j] = (cin[j] && yin) ? cur.L[k][j] : 0.0f;
#pragma unroll
                for (int j = 0; j < NP; j++) {  // CompareMove, MatchLib.cu:622-624
                    p[0][j] = l[j] * nbr<-1>(Rm1[k]);  // shift (-1, 0)
                    p[1][j] = l[j] * nbr<+1>(Rm1[k]);  // shift (+1, 0)
                    p[2][j] = l[j] * Rm2[k][j];               // shift (0, -1)
                    p[3][j] = l[j] * rc[j];                   // shift (0, +1)
                    p[4][j] = l[j] * Rm1[k][j];               // shift (0, 0)
                }
                {   // the six row passes of the channel in lockstep: B (convolutionRowsKernelT / ColumnsKernelT on R'^2, clamp) and N_s(r-3)
                    // (convolutionRowsKernel / ColumnsKernel on the products, zero padded)
                    const float v6[6] = {sq[0], p[0][0], p[1][0], p[2][0], p[3][0], p[4][0]};
                    float h6[6];
                    rowconv5_lockstep<6>(v6, h6);
                    bnew[0] = colstep5(aB[k][0], h6[0]);
#pragma unroll
                    for (int s = 0; s < 5; s++) Nv[s][0] = colstep5(aN[k][s][0], h6[s + 1]);
                }
                if (do_out) {
#pragma unroll
                    for (int j = 0; j < NP; j++) {
                        const float a = cur.A[k][j], bc = Bm1[k][j];
                        float bl = nbr<-1>(Bm1[k]), br = nbr<+1>(Bm1[k]), bu = Bm2[k][j], bd = bnew[j];
                        if constexpr (EDGE) {  // B at the clamped position (MatchLib.cu:676-679)
                            bl = (po[j] <= 0) ? bc : bl;
                            br = (po[j] >= W - 1) ? bc : br;
                            bu = (o <= 0) ? bc : bu;
                            bd = (o >= H - 1) ? bc : bd;
                        }
                        const float q[5] = {ncc2_t<FAST>(Nv[0][j], a, bl), ncc2_t<FAST>(Nv[1][j], a, br), ncc2_t<FAST>(Nv[2][j], a, bu),
                                            ncc2_t<FAST>(Nv[3][j], a, bd), ncc2_t<FAST>(Nv[4][j], a, bc)};
#pragma unroll
                        for (int s = 0; s < 5; s++) {  // MatchGPULib.cpp:2033-2070: q0 ; q1+q0 ; ((q0+q1)+q2)/3
                            if (k == 0) Q[s][j] = q[s];
                            else if (k == 1) Q[s][j] = q[s] + Q[s][j];
                            else Q[s][j] = div3_nonneg(Q[s][j] + q[s]);
                        }
                    }
                }
            }
#pragma unroll
            for (int j = 0; j < NP; j++) {
                Bm2[k][j] = Bm1[k][j];
                Bm1[k][j] = bnew[j];
                Rm2[k][j] = Rm1[k][j];
                Rm1[k][j] = rc[j];
            }
        }
        if (do_out) {
            // PolyDisparity x / y, corr product, update, confidence blend (MatchGPULib.cpp:2129-2250)
            float ndx[NP], ndy[NP], nkp[NP];
#pragma unroll
            for (int j = 0; j < NP; j++) {
                float ddx, ddy, rx, ry;
                poly_fast(Q[4][j], Q[0][j], Q[1][j], thr, ddx, rx);
                poly_fast(Q[4][j], Q[2][j], Q[3][j], thr, ddy, ry);
                float kap = ry * rx;
                if (blend) kap = blend_conf(seedv(cur.O[2][j]), kap);
                ndx[j] = seedv(cur.O[0][j]) + ddx;
                ndy[j] = seedv(cur.O[1][j]) + ddy;
                nkp[j] = kap;
            }
            const unsigned ro = (unsigned)o * pitchW;
            if (stv[0]) {
                st_at(Nb[0], ro + coffo[0], ndx[0]);
                st_at(Nb[1], ro + coffo[0], ndy[0]);
                st_at(Nb[2], ro + coffo[0], nkp[0]);
            }
        }
    };

    // ---- prologue of the load pipeline, then the row loop (two steps per trip) ---------------------------------
    int r = ys - 3;
    const int r_end = ye + 2;  // last R' row any output of the strip needs
    MarchRow P0, P1;
    {
        float d0x[NP], d0y[NP];
        load_d(r, d0x, d0y);
        load_d(r + 1, P0.dx, P0.dy);
        load_L(r - 1, P0.L);
        load_AO(r - 3, P0.A, P0.O);
        gather(r, d0x, d0y, P0.R);
    }
    for (; r <= r_end; r += 2) {
        step(r, P0, P1);
        step(r + 1, P1, P0);
    }
}

// A workgroup is MARCH_WPB independent waves (they never synchronise): single-wave workgroups are admitted only 8 per CU
// (2 waves per SIMD; measured: wave lifetime 2/3 of the launch at a nominal 3 waves per SIMD), four-wave workgroups reach the
// occupancy the register allocation allows.
constexpr int MARCH_WPB = 4;
// strip of this wave: blocks are dealt to the XCDs as contiguous bands of strips (xcd_tile), a block's waves take consecutive strips
__device__ __forceinline__ bool march_strip(const int n_strips, const int strips_x, int &sx, int &sy)
{
    const int n_blocks = (n_strips + MARCH_WPB - 1) / MARCH_WPB;
    int bx, by;
    xcd_tile(n_blocks, n_blocks, bx, by);  // (one row of blocks: bx = the remapped block index)
    const int strip = bx * MARCH_WPB + (int)(threadIdx.x >> 6);
    if (strip >= n_strips) return false;
    sy = strip / strips_x;
    sx = strip - sy * strips_x;
    return true;
}
// AGE CLASSES (round 3).  A SIMD holds up to three strips' waves, and the instruction arbiter serves them oldest first: with
// equal strips the first-dispatched wave runs at full single-wave speed and finishes at 0.57 of the launch, the second at 0.75,
// and the last one spends the final quarter alone on its SIMD at the issue rate of one wave (round 2: 380 k / 500 k / 670 k cycles at
// 16 MP).  Waves that are to finish TOGETHER need strips in proportion to their speed.  The launcher therefore splits the
// workgroups into `ncls` classes by dispatch order (cls = blockIdx / cls_blocks: the dispatcher deals workgroups breadth-first, one
// per CU and round, so class c is the c-th wave on its SIMD) and gives class c strips of hc[c] rows; a class's strips tile the
// image in its own row bands: group g of Hg = sum(hc) rows holds one strip of every class, class c at row offset sum(hc[<c]).
// hc[1] == 0: one class, the uniform strips of round 2.
struct StripClasses {
    int ncls, cls_blocks;  // classes; workgroups per class (a multiple of 8: the XCD remap works inside a class)
    int hc[3];             // strip rows of class c
};
__device__ __forceinline__ bool march_strip_cls(const StripClasses &sc, const int strips_x, const int n_groups, const int H, int &sx, int &ys, int &ye)
{
    const int cls = (int)blockIdx.x / sc.cls_blocks;
    const int jb = (int)blockIdx.x - cls * sc.cls_blocks;
    int bx, by;
    xcd_tile_at(jb, sc.cls_blocks, sc.cls_blocks, bx, by);
    const int strip = bx * MARCH_WPB + (int)(threadIdx.x >> 6);
    if (strip >= strips_x * n_groups) return false;
    const int g = strip / strips_x;
    sx = strip - g * strips_x;
    const int Hg = sc.hc[0] + sc.hc[1] + sc.hc[2];
    const int off = cls == 0 ? 0 : (cls == 1 ? sc.hc[0] : sc.hc[0] + sc.hc[1]);
    ys = g * Hg + off;
    ye = min(ys + sc.hc[cls], H);
    return ys < H;
}
// grid: one wave (64 threads) per strip of March::VX columns x Hs rows; strips dealt to the XCDs as contiguous bands
__global__ __launch_bounds__(64 * MARCH_WPB, kMarchWaves) void k_cost_march(Img3 L, Img3 R, const float *__restrict__ A3, const float *__restrict__ d3,
                                                                      float *__restrict__ nd3, int W, int H, float thr, int blend, int strips_x,
                                                                      int n_strips, int Hs, const unsigned *__restrict__ range_bad, SeedMap sm, StripClasses sc,
                                                                      Batch bt)
{
    if (bt.n > 1) {  // this workgroup's pair of the batch (blockIdx.y)
        const int b = (int)blockIdx.y;
        L.p = shifted(L.p, bt.img[b]);
        R.p = shifted(R.p, bt.img[b]);
        A3 = shifted(A3, bt.in[b]);
        d3 = shifted(d3, bt.in[b]);
        nd3 = shifted(nd3, bt.out[b]);
        if (range_bad) range_bad += b;
        sm.cx = bt.cx[b];
        sm.cy = bt.cy[b];
    }
    int sx, sy, ys, ye;
    if (sc.ncls > 1) {  // (kernel-uniform) strips by age class; n_strips = strips per class group count x strips_x is passed as Hs = groups
        if (!march_strip_cls(sc, strips_x, Hs, H, sx, ys, ye)) return;
    } else {
        if (!march_strip(n_strips, strips_x, sx, sy)) return;
        ys = sy * Hs;
        ye = min(ys + Hs, H);
    }
    const int xs = sx * March::VX + March::ORG;
    const int xe = min(xs + March::VX, W);
    const int X0 = xs - 3;
    // interior: every pixel a lane holds lies inside the image, and so does every row the strip loads (ys - 6 .. ye + 5: no row clamp)
    const bool interior = X0 >= 0 && X0 + March::COLS <= W && ys >= 6 && ye <= H - 6;
    // range-guarded division (ugsm_exact.hpp) when the pyramid builder found every value of the pair in range
    const bool fast = range_bad != nullptr && __builtin_amdgcn_readfirstlane((int)*range_bad) == 0;
    if (sm.Ws > 0) {  // first iteration of a level, seeded from the coarser level's field (kernel-uniform)
        if (fast) {
            if (interior) cost_march_body<false, true, true>(L, R, A3, d3, nd3, W, H, thr, blend, X0, xs, xe, ys, ye, sm);
            else cost_march_body<true, true, true>(L, R, A3, d3, nd3, W, H, thr, blend, X0, xs, xe, ys, ye, sm);
        } else {
            if (interior) cost_march_body<false, false, true>(L, R, A3, d3, nd3, W, H, thr, blend, X0, xs, xe, ys, ye, sm);
            else cost_march_body<true, false, true>(L, R, A3, d3, nd3, W, H, thr, blend, X0, xs, xe, ys, ye, sm);
        }
        return;
    }
    if (fast) {
        if (interior) cost_march_body<false, true>(L, R, A3, d3, nd3, W, H, thr, blend, X0, xs, xe, ys, ye);
        else cost_march_body<true, true>(L, R, A3, d3, nd3, W, H, thr, blend, X0, xs, xe, ys, ye);
    } else {
        if (interior) cost_march_body<false, false>(L, R, A3, d3, nd3, W, H, thr, blend, X0, xs, xe, ys, ye);
        else cost_march_body<true, false>(L, R, A3, d3, nd3, W, H, thr, blend, X0, xs, xe, ys, ye);
    }
}

// Strip height of the marching K-cost.  Every strip is resident at once when there are at most 1024 x w of them (256 CUs x 4 SIMDs,
// w waves per SIMD, w <= what the register allocation allows); a launch then lasts as long as one strip: (rows + halo and prologue)
// row steps, and a step takes longer the more waves share the SIMD.  Measured (tools/kbench mode 10, 0.5-8 Mpx): 1.57 us per step and
// 4.3 extra steps at one wave per SIMD, 1.83 us and 7.3 at two, 2.6 us and 6.5 at three.  Take the shortest strips that fit for each w
// and keep the w that finishes first: large levels end up at the full occupancy with tall strips (6 halo rows recomputed per strip
// matter there), levels around 1 Mpx at one or two waves per SIMD with 10-18 rows.
// `throughput` (several pairs in flight: other pairs' kernels share the SIMDs whatever this launch does): the tallest strips that
// are still all resident -- the fewest halo rows.
int march_strip_rows(int W, int H, int throughput, int pairs)
{
    const int strips_x = ((W - March::ORG + March::VX - 1) / March::VX) * (pairs > 1 ? pairs : 1);  // (a batched launch: the strips of all its pairs share the chip)
    static const float t_step[3] = {1.57f, 1.83f, 2.6f}, extra[3] = {4.3f, 7.3f, 6.5f};
    const int max_w = kMarchWaves;
    float best = 0.0f;
    int Hs = 6;
    for (int w = throughput ? (max_w < 3 ? max_w : 3) : 1; w <= max_w && w <= 3; w++) {
        const int sy = (256 * 4 * w / strips_x) > 0 ? (256 * 4 * w / strips_x) : 1;  // strips per column of strips that still fit
        int h = (H + sy - 1) / sy;
        if (h < 6) h = 6;
        const float t = t_step[w - 1] * ((float)h + extra[w - 1]);
        if (best == 0.0f || t < best) {
            best = t;
            Hs = h;
        }
    }
    return Hs;
}

// Share of a strip group's rows (per mille) that the first- and the second-dispatched wave of a SIMD take; the third takes the
// rest.  {0, 0} = uniform strips.  (tools/kbench mode 12 sweeps it; UGSM_MARCH_AGE="p0,p1" under UGSM_DEV=1.)
// Measured (MI355X, same box, profiles/r03_kbench_age_strips.txt): 16 MP 297-303 -> 272-279 us, 8 MP 161 -> 146-150 us, 4 MP 81-84 -> 75-76 us
// per launch, bit-identical output; wave lifetimes on a SIMD 378 k / 495 k / 645 k cycles (uniform) -> 520 k / 544 k / 562 k (the
// clock the chip holds falls from 2.23 to 2.16 GHz with it: a denser instruction stream, MI355X_MICROARCH.md "DVFS give-back").
// A pair alone gains 1 % (108.4 -> 109.5 pairs/s); with four pairs in flight nothing changes (161.7 -> 161.6): there the SIMD slots
// the finished first waves leave behind are taken by the other pairs' kernels anyway.
int march_age_permille[2] = {470, 340};
static void launch_cost_march_t(hipStream_t st, Img3 L, Img3 R, const float *A3, const float *d3, float *nd3, int W, int H, float thr, int blend,
                                int rows, const unsigned *range_bad, SeedMap sm = SeedMap{0, 0, 0, 0}, const Batch *bt = nullptr)
{
    Batch one{};
    one.n = 1;
    const Batch &B = bt ? *bt : one;
    const int pairs = B.n > 1 ? B.n : 1;
    const int strips_x = (W - March::ORG + March::VX - 1) / March::VX;
    // rows: > 0 a fixed strip height; 0 the latency heights and strips by age class; -1 the latency heights, no age classes; -2 the
    // throughput heights, no age classes; -3 the throughput heights and age classes
    const bool age = (rows == 0 || rows == -3) && pairs == 1, tput = rows == -2 || rows == -3;  // (age classes count on blockIdx.x being the dispatch order)
    int Hs = rows > 0 ? rows : march_strip_rows(W, H, tput, pairs);
    int strips_y = (H + Hs - 1) / Hs;
    int n_strips = strips_x * strips_y;
    int n_blocks = (n_strips + MARCH_WPB - 1) / MARCH_WPB;
    StripClasses sc{1, 0, {Hs, 0, 0}};
    // three waves per SIMD (the large levels: every strip resident at once, 3 x 1024 of them): strips by age class.  One dispatch
    // round = one workgroup per CU, so a class is exactly `cus` workgroups (blockIdx / cus = the wave's rank on its SIMD); the
    // strips of a class must fit into them (4 x cus strips), which fixes the number of strip groups and with it the group height.
    if (age && march_age_permille[0] > 0 && n_strips > 2 * 1024 && n_strips <= 3 * 1024 + strips_x) {
        const int cus = 256;
        const int n_groups = (MARCH_WPB * cus) / strips_x;
        const int Hg = n_groups > 0 ? (H + n_groups - 1) / n_groups : 0;
        const int h0 = (Hg * march_age_permille[0] + 500) / 1000, h1 = (Hg * march_age_permille[1] + 500) / 1000;
        const int h2 = Hg - h0 - h1;
        if (n_groups > 0 && h0 >= 6 && h1 >= 6 && h2 >= 6) {
            sc = StripClasses{3, cus, {h0, h1, h2}};
            n_blocks = 3 * cus;
            Hs = (H + Hg - 1) / Hg;  // (the kernel's Hs argument carries the group count in class mode)
            n_strips = 3 * strips_x * Hs;
        }
    }
    UGSM_LAUNCH(k_cost_march, dim3(n_blocks, pairs), dim3(64 * MARCH_WPB), 0, st, L, R, A3, d3, nd3, W, H, thr, blend, strips_x, n_strips, Hs, range_bad, sm, sc, B);
}

void launch_cost_march_seeded(hipStream_t st, Img3 L, Img3 R, const float *A3, const float *coarse3, SeedMap sm, float *nd3, int W, int H, float thr,
                              int blend, int rows, const unsigned *range_bad, const Batch *bt)
{
    launch_cost_march_t(st, L, R, A3, coarse3, nd3, W, H, thr, blend, rows, range_bad, sm, bt);
}

void launch_cost_march(hipStream_t st, Img3 L, Img3 R, const float *A3, const float *d3, float *nd3, int W, int H, float thr, int blend, int rows,
                       const unsigned *range_bad, const Batch *bt)
{
    launch_cost_march_t(st, L, R, A3, d3, nd3, W, H, thr, blend, rows, range_bad, SeedMap{0, 0, 0, 0}, bt);
}

}  // namespace ugsm
