// ugsm_kernels_march4.hip -- K-cost, marching form with the colour channels side by side: the latency form for the levels whose
// launch lasts as long as ONE strip (0.15 - 2.5 Mpx when a pair has the chip to itself).
//
// Same arithmetic as k_cost_march (ugsm_kernels_march.hip), k_cost_split and the CPU oracle, bit for bit: one matcher iteration's
// warp + 5-shift squared-NCC cost over 3 channels + parabola + confidence blend + disparity update
// (matchlevel, /root/reference/src/gpu_matcher/MatchGPULib.cpp:1745-2250 and the MatchLib.cu kernels cited below).
//
// Why.  In k_cost_march one wave owns a strip and does everything for a row step: 3 channels x (6 row passes + 6 column passes
// + 5 divisions) + the parabolas, ~660 VALU instructions.  A wave alone on its SIMD issues one instruction per ~4.3 cycles
// (profiles/r03_valubench.txt), so a step takes ~1.5 us whatever else the chip does, and a level of 0.25 - 2 Mpx -- too few strips to give every
// SIMD two or three waves -- takes (rows + halo + prologue) x 1.5 us: 17 - 50 us per launch at 16 MP's levels 6 - 3, 34 launches,
// 1.3 ms of a 9.1 ms pair (profiles/r03_level_breakdown.txt).  Here a strip belongs to a WORKGROUP of four waves:
//   waves 0, 1, 2  one colour channel each: the warped gather of R_k, R'^2 and the five products L_k R'_k[x+s], their row and column
//                  passes (systolic DPP row passes, transposed-form column passes, exactly as in k_cost_march) and the five
//                  quotients q_{s,k}, which go to LDS;
//   wave 3         the epilogue of the row before: Q_s = ((q_s0 + q_s1) + q_s2) / 3 (MatchGPULib.cpp:2033-2070), the two parabolas,
//                  the confidence blend, the update and the stores.
// One barrier per row step; the q rows are double-buffered by step parity, so the epilogue wave reads step r's quotients while the
// channel waves already write step r+1's.  A step is then ~230 instructions of one wave instead of ~660.
//
// One pixel per lane, literal float contract, range-guarded division when the pair's pyramids are in range (as k_cost_march); the
// level's seeding rides on the first launch (SEED) as there.
#include "ugsm_exact.hpp"
#include "ugsm_launch.hpp"

namespace ugsm {

namespace m4 {

constexpr int VX = 58, SKEW = 2;  // output columns of a strip (64 lanes, halo 3 on both sides); lanes between a pixel and its row-pass results

__device__ __forceinline__ float shr1(float v)  // from lane - 1 (lane 0 reads 0)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x138, 0xf, 0xf, true));
}
__device__ __forceinline__ float shl1(float v)  // from lane + 1 (lane 63 reads 0)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x130, 0xf, 0xf, true));
}
// 5-tap row pass, taps added in the reference's order j = -2..2 (MatchLib.cu:127-134, 1484-1487); the partial sum travels one lane to
// the right per tap, so the sum of the window centred on column c arrives in lane c + 2 (SKEW)
__device__ __forceinline__ float rowconv5(const float p)
{
    const float a0 = p * UGSM_G0, a1 = p * UGSM_G1, a2 = p * UGSM_G2;
    const float p2 = shr1(a0) + a1;
    const float p3 = shr1(p2) + a2;
    const float p4 = shr1(p3) + a1;
    return shr1(p4) + a0;
}
// the six row passes of a channel wave in lockstep (as rowconv5_lockstep in ugsm_kernels_march.hip): a channel wave is alone on its SIMD
// on the levels this kernel serves, so every s_nop of the VALU-write -> DPP-read hazard would be an idle issue slot
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
// one step of the transposed-form 5-tap column pass: returns the sum of the row that closes (two rows up)
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
__device__ __forceinline__ int tex_index_nb(const float coord, const float nm1)  // tex_index without branches (ugsm_kernels_march.hip)
{
    return (int)__builtin_amdgcn_fmed3f(floorf(coord), 0.0f, nm1);
}

// what a lane needs to know about its column, and the row / seed offset helpers shared by the two roles
template <bool EDGE, bool SEED>
struct Cols {
    int px, po;              // the pixel whose R', L and products the lane holds; po = px - SKEW: row-pass results, A, O, output
    unsigned coff, coffo;    // byte offsets of the (clamped) columns px / po inside a row
    unsigned scol, scolo;    // SEED: byte offsets of the coarse columns they sample
    float xc;
    bool cin, stv;
    int W, H;
    SeedMap sm;
    __device__ __forceinline__ Cols(const int X0, const int W_, const int H_, const int xs, const int xe, const SeedMap sm_) : W(W_), H(H_), sm(sm_)
    {
        const int lane = threadIdx.x & 63;
        px = X0 + lane;
        po = px - SKEW;
        const int pc = EDGE ? clampi(px, 0, W - 1) : px;
        cin = !EDGE || (px >= 0 && px < W);
        coff = (unsigned)pc * 4u;
        coffo = (unsigned)clampi(po, 0, W - 1) * 4u;
        xc = (float)pc + 0.5f;
        stv = po >= xs && po < xe;
        scol = scolo = 0;
        if constexpr (SEED) {
            const float sf = (float)(1 / UGSM_SCALE);
            scol = (unsigned)tex_index(((float)(clampi(px, 0, W - 1) + sm.cx) + 0.5f) * sf, sm.Ws) * 4u;
            scolo = (unsigned)tex_index(((float)(clampi(po, 0, W - 1) + sm.cx) + 0.5f) * sf, sm.Ws) * 4u;
        }
    }
    __device__ __forceinline__ int rowc(const int r) const { return min(max(r, 0), H - 1); }
    // offset of row r of the field d3 points at: the level's own row, or (SEED) the coarse row it samples
    __device__ __forceinline__ unsigned d_off(const int r, const bool skewed) const
    {
        if constexpr (SEED) {
            const float sf = (float)(1 / UGSM_SCALE);
            return (unsigned)tex_index(((float)(rowc(r) + sm.cy) + 0.5f) * sf, sm.Hs) * ((unsigned)sm.Ws * 4u) + (skewed ? scolo : scol);
        } else {
            return (unsigned)rowc(r) * ((unsigned)W * 4u) + (skewed ? coffo : coff);
        }
    }
};
template <bool SEED>
__device__ __forceinline__ float seedv(const float v)  // subsampleDispKernel's value (MatchLib.cu:393-394): the product in binary64
{
    if constexpr (SEED) return (float)(UGSM_SCALE * (double)v);
    else return v;
}

constexpr int Q_FLOATS = 2 * 3 * 5 * 64;  // [step parity][channel][shift][lane]
__device__ __forceinline__ int q_at(const int buf, const int k, const int s, const int lane) { return ((buf * 3 + k) * 5 + s) * 64 + lane; }

// ---- a channel wave ------------------------------------------------------------------------------------------------------
struct ChRow {
    float dx, dy, R, L, A;
};
template <bool EDGE, bool FAST, bool SEED>
__device__ __forceinline__ void channel_wave(const int k, const Img3 &L, const Img3 &R, const float *__restrict__ A3, const float *__restrict__ d3,
                                             const int W, const int H, const int X0, const int xs, const int xe, const int ys, const int ye,
                                             const SeedMap sm, float *__restrict__ q_lds)
{
    const int lane = threadIdx.x & 63;
    const size_t n = (size_t)W * H;
    const size_t nD = SEED ? (size_t)sm.Ws * sm.Hs : n;
    gchar_c *const Lb = uniform_base(L.p + (size_t)k * L.plane);
    gchar_c *const Rb = uniform_base(R.p + (size_t)k * R.plane);
    gchar_c *const Ab = uniform_base(A3 + (size_t)k * n);
    gchar_c *const Dx = uniform_base(d3), *const Dy = uniform_base(d3 + nD);
    const Cols<EDGE, SEED> c(X0, W, H, xs, xe, sm);
    const unsigned pitchW = (unsigned)W * 4u, pitchL = (unsigned)L.pitch * 4u;
    const float wm1 = (float)(W - 1), hm1 = (float)(H - 1);

    auto load_d = [&](const int r, float &dx, float &dy) {
        const unsigned off = c.d_off(r, false);
        dx = ld_at(Dx, off);
        dy = ld_at(Dy, off);
    };
    // warpAbyB (MatchLib.cu:510-515): R'[x,y] = tex(R, x + 0.5 + dx, y + 0.5 + dy) at the clamped pixel
    auto gather = [&](const int r, const float dx, const float dy) -> float {
        const float yc = (float)c.rowc(r) + 0.5f;
        const int sx = tex_index_nb(c.xc + seedv<SEED>(dx), wm1);
        const int sy = tex_index_nb(yc + seedv<SEED>(dy), hm1);
        return ld_at(Rb, (__umul24((unsigned)sy, (unsigned)R.pitch) + (unsigned)sx) * 4u);
    };
    auto load_L = [&](const int r) -> float { return ld_at(Lb, (unsigned)c.rowc(r) * pitchL + c.coff); };
    auto load_A = [&](const int r) -> float { return ld_at(Ab, (unsigned)c.rowc(r) * pitchW + c.coffo); };

    float Rm1 = 0.0f, Rm2 = 0.0f, Bm1 = 0.0f, Bm2 = 0.0f;  // R'(r-1), R'(r-2), B(r-3), B(r-4)
    float aB[4] = {0.0f, 0.0f, 0.0f, 0.0f}, aN[5][4];
#pragma unroll
    for (int s = 0; s < 5; s++)
#pragma unroll
        for (int u = 0; u < 4; u++) aN[s][u] = 0.0f;

    auto step = [&](const int r, const int buf, const ChRow &cur, ChRow &nxt) {
        load_d(r + 2, nxt.dx, nxt.dy);
        nxt.R = gather(r + 1, cur.dx, cur.dy);
        nxt.L = load_L(r);
        nxt.A = load_A(r - 2);
        // ---- arithmetic on R'(r), L(r-1), A(r-3) ----
        const int y = r - 1, o = r - 3;
        const bool do_prod = r >= ys - 1;
        const bool do_out = o >= ys && o < ye;
        const bool yin = !EDGE || (y >= 0 && y < H);
        const float rc = cur.R;
        const float sq = rc * rc;                  // Square, MatchLib.cu:569-570
        float bnew;                                // = B(r-2): convolutionRowsKernelT / ColumnsKernelT on R'^2 (clamp)
        if (!do_prod) bnew = colstep5(aB, rowconv5(sq));  // (the strip's first rows: only B is due)
        if (do_prod) {
            const float l = (c.cin && yin) ? cur.L : 0.0f;
            float p[5];                            // CompareMove, MatchLib.cu:622-624
            p[0] = l * shr1(Rm1);                  // shift (-1, 0)
            p[1] = l * shl1(Rm1);                  // shift (+1, 0)
            p[2] = l * Rm2;                        // shift (0, -1)
            p[3] = l * rc;                         // shift (0, +1)
            p[4] = l * Rm1;                        // shift (0, 0)
            float Nv[5];
            {   // the six row passes in lockstep (no s_nop between a stage and the DPP read of it): B, and N_s(r-3) --
                // convolutionRowsKernel / ColumnsKernel on the products (zero padded)
                const float v6[6] = {sq, p[0], p[1], p[2], p[3], p[4]};
                float h6[6];
                rowconv5_lockstep<6>(v6, h6);
                bnew = colstep5(aB, h6[0]);
#pragma unroll
                for (int s = 0; s < 5; s++) Nv[s] = colstep5(aN[s], h6[s + 1]);
            }
            if (do_out) {
                const float a = cur.A, bc = Bm1;
                float bl = shr1(Bm1), br = shl1(Bm1), bu = Bm2, bd = bnew;
                if constexpr (EDGE) {  // B at the clamped position (MatchLib.cu:676-679)
                    bl = (c.po <= 0) ? bc : bl;
                    br = (c.po >= W - 1) ? bc : br;
                    bu = (o <= 0) ? bc : bu;
                    bd = (o >= H - 1) ? bc : bd;
                }
                q_lds[q_at(buf, k, 0, lane)] = ncc2_t<FAST>(Nv[0], a, bl);
                q_lds[q_at(buf, k, 1, lane)] = ncc2_t<FAST>(Nv[1], a, br);
                q_lds[q_at(buf, k, 2, lane)] = ncc2_t<FAST>(Nv[2], a, bu);
                q_lds[q_at(buf, k, 3, lane)] = ncc2_t<FAST>(Nv[3], a, bd);
                q_lds[q_at(buf, k, 4, lane)] = ncc2_t<FAST>(Nv[4], a, bc);
            }
        }
        Bm2 = Bm1;
        Bm1 = bnew;
        Rm2 = Rm1;
        Rm1 = rc;
        __syncthreads();  // the epilogue wave may read this step's quotients
    };

    int r = ys - 3;
    const int r_end = ye + 2;
    ChRow P0, P1;
    {
        float d0x, d0y;
        load_d(r, d0x, d0y);
        load_d(r + 1, P0.dx, P0.dy);
        P0.L = load_L(r - 1);
        P0.A = load_A(r - 3);
        P0.R = gather(r, d0x, d0y);
    }
    for (; r <= r_end; r += 2) {
        step(r, 0, P0, P1);
        step(r + 1, 1, P1, P0);
    }
}

// ---- the epilogue wave ---------------------------------------------------------------------------------------------------
template <bool EDGE, bool SEED>
__device__ __forceinline__ void epilogue_wave(const float *__restrict__ d3, float *__restrict__ nd3, const int W, const int H, const float thr, const int blend,
                                              const int X0, const int xs, const int xe, const int ys, const int ye, const SeedMap sm,
                                              const float *__restrict__ q_lds)
{
    const int lane = threadIdx.x & 63;
    const size_t n = (size_t)W * H;
    const size_t nD = SEED ? (size_t)sm.Ws * sm.Hs : n;
    gchar_c *const Db[3] = {uniform_base(d3), uniform_base(d3 + nD), uniform_base(d3 + 2 * nD)};
    gchar_c *const Nb[3] = {uniform_base(nd3), uniform_base(nd3 + n), uniform_base(nd3 + 2 * n)};
    const Cols<EDGE, SEED> c(X0, W, H, xs, xe, sm);
    const unsigned pitchW = (unsigned)W * 4u;
    auto load_O = [&](const int r, float (&od)[3]) {
        const unsigned off = c.d_off(r, true);
#pragma unroll
        for (int f = 0; f < 3; f++) od[f] = ld_at(Db[f], off);
    };
    auto step = [&](const int r, const int buf, const float (&cur)[3], float (&nxt)[3]) {
        load_O(r - 2, nxt);
        __syncthreads();  // the channel waves have written this step's quotients
        const int o = r - 3;
        if (o >= ys && o < ye) {
            float Q[5];
#pragma unroll
            for (int s = 0; s < 5; s++) {  // MatchGPULib.cpp:2033-2070: q0 ; q1 + q0 ; ((q0 + q1) + q2) / 3
                const float q0 = q_lds[q_at(buf, 0, s, lane)], q1 = q_lds[q_at(buf, 1, s, lane)], q2 = q_lds[q_at(buf, 2, s, lane)];
                Q[s] = div3_nonneg((q1 + q0) + q2);
            }
            // PolyDisparity x / y, corr product, update, confidence blend (MatchGPULib.cpp:2129-2250)
            float ddx, ddy, rx, ry;
            poly_fast(Q[4], Q[0], Q[1], thr, ddx, rx);
            poly_fast(Q[4], Q[2], Q[3], thr, ddy, ry);
            float kap = ry * rx;
            if (blend) kap = blend_conf(seedv<SEED>(cur[2]), kap);
            const float ndx = seedv<SEED>(cur[0]) + ddx, ndy = seedv<SEED>(cur[1]) + ddy;
            if (c.stv) {
                const unsigned at = (unsigned)o * pitchW + c.coffo;
                st_at(Nb[0], at, ndx);
                st_at(Nb[1], at, ndy);
                st_at(Nb[2], at, kap);
            }
        }
    };
    int r = ys - 3;
    const int r_end = ye + 2;
    float O0[3], O1[3];
    load_O(r - 3, O0);
    for (; r <= r_end; r += 2) {
        step(r, 0, O0, O1);
        step(r + 1, 1, O1, O0);
    }
}

}  // namespace m4

// grid: one workgroup of four waves per strip of m4::VX columns x Hs rows; strips dealt to the XCDs as contiguous bands
// (the register allocation is made for 4 waves per SIMD: 72 VGPRs)
__global__ __launch_bounds__(256, 4) void k_cost_march4(Img3 L, Img3 R, const float *__restrict__ A3, const float *__restrict__ d3, float *__restrict__ nd3,
                                                       int W, int H, float thr, int blend, int strips_x, int n_strips, int Hs,
                                                       const unsigned *__restrict__ range_bad, SeedMap sm, Batch bt)
{
    __shared__ float q_lds[m4::Q_FLOATS];
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
    int sx, sy;
    xcd_tile(n_strips, strips_x, sx, sy);
    const int xs = sx * m4::VX, ys = sy * Hs;
    const int xe = min(xs + m4::VX, W), ye = min(ys + Hs, H);
    const int X0 = xs - 3;
    // interior: every pixel a lane holds lies inside the image, and so do the product rows ys-2 .. ye+1 and the rows ys-1 .. ye of the B fetches
    const bool interior = X0 >= 0 && X0 + 64 <= W && ys >= 2 && ye <= H - 2;
    const bool fast = range_bad != nullptr && __builtin_amdgcn_readfirstlane((int)*range_bad) == 0;
    const bool seeded = sm.Ws > 0;
    const int role = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    // (every branch below is workgroup-uniform except the role, and both roles run the same number of row steps = barriers)
#define UGSM_M4_DISPATCH(EDGE, SEED)                                                                                                  \
    do {                                                                                                                              \
        if (role == 3) m4::epilogue_wave<EDGE, SEED>(d3, nd3, W, H, thr, blend, X0, xs, xe, ys, ye, sm, q_lds);                       \
        else if (fast) m4::channel_wave<EDGE, true, SEED>(role, L, R, A3, d3, W, H, X0, xs, xe, ys, ye, sm, q_lds);                   \
        else m4::channel_wave<EDGE, false, SEED>(role, L, R, A3, d3, W, H, X0, xs, xe, ys, ye, sm, q_lds);                            \
    } while (0)
    if (seeded) {
        if (interior) UGSM_M4_DISPATCH(false, true);
        else UGSM_M4_DISPATCH(true, true);
    } else {
        if (interior) UGSM_M4_DISPATCH(false, false);
        else UGSM_M4_DISPATCH(true, false);
    }
#undef UGSM_M4_DISPATCH
}

// Strip height: every strip resident at once (four workgroups of four waves per CU: one wave of every workgroup per SIMD), the
// shortest strips that still fit -- a launch lasts (rows + halo + prologue) row steps.
int march4_strip_rows(int W, int H, int pairs)
{
    const int strips_x = ((W + m4::VX - 1) / m4::VX) * (pairs > 1 ? pairs : 1);  // (a batched launch: the strips of all its pairs share the chip)
    const int per_col = (4 * 256) / strips_x > 0 ? (4 * 256) / strips_x : 1;
    int h = (H + per_col - 1) / per_col;
    return h < 6 ? 6 : h;
}

void launch_cost_march4(hipStream_t st, Img3 L, Img3 R, const float *A3, const float *d3, float *nd3, int W, int H, float thr, int blend, int rows,
                        const unsigned *range_bad, SeedMap sm, const Batch *bt)
{
    Batch one{};
    one.n = 1;
    const Batch &B = bt ? *bt : one;
    const int pairs = B.n > 1 ? B.n : 1;
    const int strips_x = (W + m4::VX - 1) / m4::VX;
    const int Hs = rows > 0 ? rows : march4_strip_rows(W, H, pairs);
    const int strips_y = (H + Hs - 1) / Hs;
    const int n_strips = strips_x * strips_y;
    UGSM_LAUNCH(k_cost_march4, dim3(n_strips, pairs), dim3(256), 0, st, L, R, A3, d3, nd3, W, H, thr, blend, strips_x, n_strips, Hs, range_bad, sm, B);
}

}  // namespace ugsm
