// ugsm_kernels_small.hip -- latency forms of K-cost and K-smooth for the coarse pyramid levels.
//
// Below ~0.2 Mpx a level has fewer tiles than the chip has CUs and one matcher iteration is two DEPENDENT launches whose
// duration is the critical path of a single tile (round 2, tools/level_breakdown.py: the LDS-tiled k_cost_split 14 us, k_smooth_fused 11 us per launch
// from 54 x 36 up to 436 x 289, 22 iterations per level, 7 such levels = 4 of the 11 ms a pair took when it was alone on the GPU).
// These kernels do the same arithmetic -- same helpers, same operation order, bit for bit (tests/test_gpu_small.py) -- with the
// tile's chain cut short instead of its instruction count:
//   k_cost_small   the three colour channels of a tile run side by side in three thread groups (own LDS images each) instead of
//                  three barrier-separated rounds; tile 16 x 12; the global loads go straight to LDS.
//   k_smooth_small one thread per pixel of tile + halo 7, double-buffered fields (one barrier per Jacobi pass instead of two),
//                  tile 18 x 4, 18 x 10 or 18 x 18.
// Citations: /root/reference/src/gpu_matcher/<file>:<line>.
#include "ugsm_exact.hpp"
#include "ugsm_launch.hpp"

namespace ugsm {

// =========================================================================================
// k_cost_small
// =========================================================================================
// Geometry of one channel group = the k_cost_split tile (dev/ugsm_dev_cost_tiled.hip) at TXS x TYS: two roles of RT threads, a thread
// owns a quad (4 consecutive x) of one row-pass row, lanes walk down the rows (conflict-free ds_read_b128: every row stride is
// an odd number of quads for TXS a multiple of 8).
template <int TXS, int TYS>
struct CostSmall {
    static constexpr int QX = TXS / 4, PR = TYS + 4;
    static constexpr int RT = PR * QX;  // threads per role
    static constexpr int CT = 2 * RT;   // threads per channel group
    static constexpr int NT = 3 * CT;
    static constexpr int SR_W = TXS + 20, SR_H = TYS + 6, SR_OX = 8;
    static constexpr int SL_W = TXS + 12, SL_H = TYS + 4, SL_OX = 4;
    static constexpr int ROW_W = TXS + 4, ROW_H = TYS + 4;
    static constexpr int SB_W = TXS + 12, SB_Q = (TXS + 8) / 4, SBROW_H = TYS + 6, SB_H = TYS + 2, SB_OX = 4;
    static constexpr int IDX_W = TXS + 6, IDX_H = TYS + 6;
    // LDS floats of one channel, in this order; [sR | sL | sBrow] doubles as the group's hand-over area after P2.5
    static constexpr int O_SR = 0, O_SL = O_SR + SR_H * SR_W, O_SBROW = O_SL + SL_H * SL_W, O_SB = O_SBROW + SBROW_H * SB_W;
    static constexpr int O_SROW = O_SB + SB_H * SB_W, O_SA = O_SROW + 5 * ROW_H * ROW_W, CH_FLOATS = O_SA + TYS * ROW_W;
    static constexpr int X_FLOATS = 10 * CT;  // hand-over: 10 floats per thread
    static_assert(RT % 64 == 0, "roles must be whole waves");
    static_assert(TXS % 8 == 0, "row strides must be an odd number of quads");
    static_assert(X_FLOATS <= O_SB, "hand-over area must fit in the dead sR/sL/sBrow images");
    static_assert((O_SL % 4) == 0 && (O_SBROW % 4) == 0 && (O_SB % 4) == 0 && (O_SROW % 4) == 0 && (O_SA % 4) == 0 && (CH_FLOATS % 4) == 0, "16-byte alignment");
    static_assert(SB_H * SB_Q <= CT, "P2.5 is one item per thread");
    static_assert(NT <= 1024, "workgroup size");
};

template <int TXS, int TYS, bool INTERIOR>
__device__ __forceinline__ void cost_small_body(const Img3 &L, const Img3 &R, const float *__restrict__ A3, const float *__restrict__ d3,
                                                float *__restrict__ nd3, const int W, const int H, const float thr, const int blend, const int x0,
                                                const int y0, float *__restrict__ smem)
{
    using G = CostSmall<TXS, TYS>;
    const int tid = threadIdx.x;
    const int k = tid / G::CT, ct = tid - k * G::CT;  // channel group (wave-uniform), thread within it
    const int role = ct / G::RT, t = ct - role * G::RT;
    const int trow = t % G::PR, qx = t / G::PR;
    const int cx = qx * 4;
    const int gy = y0 + trow, gx0 = x0 + cx;
    const bool live = INTERIOR ? (trow < TYS) : (trow < TYS && gy < H && gx0 < W);
    const size_t n = (size_t)W * H;
    float *const sm = smem + k * G::CH_FLOATS;
    float *const sR = sm + G::O_SR, *const sL = sm + G::O_SL, *const sBrow = sm + G::O_SBROW, *const sB = sm + G::O_SB;
    float *const sRow = sm + G::O_SROW, *const sA = sm + G::O_SA;

    // ---- P0: the group's own channel of tile + halo, global -> registers -> LDS.  (dx, dy) of the halo-3 region is read by all
    // three groups (same lines, L2 hits): sharing the warp addresses through LDS would cost a barrier in front of the gathers.
    constexpr int NR = (G::IDX_H * G::IDX_W + G::CT - 1) / G::CT;
    constexpr int NL = (G::SL_H * (TXS + 4) + G::CT - 1) / G::CT;
    constexpr int NA = (TXS * TYS + G::CT - 1) / G::CT;
    constexpr int NO = (TXS * TYS + G::NT - 1) / G::NT;
    gchar_c *const Lb = uniform_base(L.p + (size_t)k * L.plane);
    gchar_c *const Rb = uniform_base(R.p + (size_t)k * R.plane);
    gchar_c *const Ab = uniform_base(A3 + (size_t)k * n);
    gchar_c *const Db[3] = {uniform_base(d3), uniform_base(d3 + n), uniform_base(d3 + 2 * n)};
    float ddx[NR], ddy[NR], lv[NL], av[NA], od[3][NO];
    int gxh[NR], gyh[NR];
#pragma unroll
    for (int u = 0; u < NR; u++) {  // (dx, dy) at the pixels whose warped fetch the tile needs
        const int it = min(ct + u * G::CT, G::IDX_H * G::IDX_W - 1);
        const int r = it / G::IDX_W, c = it - r * G::IDX_W;
        gxh[u] = INTERIOR ? x0 + c - 3 : clampi(x0 + c - 3, 0, W - 1);
        gyh[u] = INTERIOR ? y0 + r - 3 : clampi(y0 + r - 3, 0, H - 1);
        const unsigned off = ((unsigned)gyh[u] * (unsigned)W + (unsigned)gxh[u]) * 4u;
        ddx[u] = ld_at(Db[0], off);
        ddy[u] = ld_at(Db[1], off);
    }
#pragma unroll
    for (int u = 0; u < NL; u++) {
        const int it = min(ct + u * G::CT, G::SL_H * (TXS + 4) - 1);
        const int r = it / (TXS + 4), c = it - r * (TXS + 4);  // c: tile column + 2
        const int gxl = x0 + c - 2, gyl = y0 + r - 2;
        const bool in = INTERIOR || (gxl >= 0 && gxl < W && gyl >= 0 && gyl < H);
        const unsigned off = INTERIOR ? ((unsigned)gyl * (unsigned)L.pitch + (unsigned)gxl) * 4u
                                      : ((unsigned)clampi(gyl, 0, H - 1) * (unsigned)L.pitch + (unsigned)clampi(gxl, 0, W - 1)) * 4u;
        const float v = ld_at(Lb, off);
        lv[u] = in ? v : 0.0f;  // zero padding of the smem convolution
    }
#pragma unroll
    for (int u = 0; u < NA; u++) {
        const int it = ct + u * G::CT;
        const int r = it / TXS, c = it - r * TXS;
        const bool in = it < TXS * TYS && (INTERIOR || (x0 + c < W && y0 + r < H));
        const unsigned off = in ? ((unsigned)(y0 + r) * (unsigned)W + (unsigned)(x0 + c)) * 4u : 0u;
        const float v = ld_at(Ab, off);
        av[u] = in ? v : 1.0f;
    }
#pragma unroll
    for (int u = 0; u < NO; u++) {  // the tile's own (dx, dy, conf) for the update at the end, in the store mapping
        const int it = tid + u * G::NT;
        const int r = it / TXS, c = it - r * TXS;
        const bool in = it < TXS * TYS && (INTERIOR || (x0 + c < W && y0 + r < H));
        const unsigned off = in ? ((unsigned)(y0 + r) * (unsigned)W + (unsigned)(x0 + c)) * 4u : 0u;
#pragma unroll
        for (int f = 0; f < 3; f++) {
            const float v = ld_at(Db[f], off);
            od[f][u] = in ? v : 0.0f;
        }
    }
    {
        float rv[NR];
#pragma unroll
        for (int u = 0; u < NR; u++) {  // warped source (warpAbyB, MatchLib.cu:510-515)
            const int sx = tex_index(((float)gxh[u] + 0.5f) + ddx[u], W);
            const int sy = tex_index(((float)gyh[u] + 0.5f) + ddy[u], H);
            rv[u] = ld_at(Rb, (unsigned)(sy * R.pitch + sx) * 4u);
        }
#pragma unroll
        for (int u = 0; u < NL; u++) {
            const int it = ct + u * G::CT;
            if (it < G::SL_H * (TXS + 4)) {
                const int r = it / (TXS + 4), c = it - r * (TXS + 4);
                sL[r * G::SL_W + (c - 2 + G::SL_OX)] = lv[u];
            }
        }
#pragma unroll
        for (int u = 0; u < NA; u++) {
            const int it = ct + u * G::CT;
            if (it < TXS * TYS) sA[(it / TXS) * G::ROW_W + (it % TXS)] = av[u];
        }
#pragma unroll
        for (int u = 0; u < NR; u++) {
            const int it = ct + u * G::CT;
            if (it < G::IDX_H * G::IDX_W) {
                const int r = it / G::IDX_W, c = it - r * G::IDX_W;
                sR[r * G::SR_W + (c - 3 + G::SR_OX)] = rv[u];
            }
        }
    }
    __syncthreads();

    // ---- P2: row passes (CompareMove + convolutionRowsKernel; Square + convolutionRowsKernelT) ------------------------------
    {
        const int r = trow;  // 0 .. PR-1 <-> tile row r-2
        float l[12];
        const float *pl = &sL[r * G::SL_W + cx - 4 + G::SL_OX];
        ld4(pl, l); ld4(pl + 4, l + 4); ld4(pl + 8, l + 8);
        const float *pr = &sR[(r + 1) * G::SR_W + cx - 4 + G::SR_OX];  // sR row index = tile row + 3
        float o[4];
        if (role == 0) {
            float rc[12];
            ld4(pr, rc); ld4(pr + 4, rc + 4); ld4(pr + 8, rc + 8);
            float p0[8], p1[8], p4[8];  // arrays hold tile columns cx-4 .. cx+7; products at columns cx-2 .. cx+5
#pragma unroll
            for (int j = 0; j < 8; j++) {
                const float lvj = l[j + 2];
                p0[j] = lvj * rc[j + 1];  // shift (-1, 0)
                p1[j] = lvj * rc[j + 3];  // shift (+1, 0)
                p4[j] = lvj * rc[j + 2];  // shift (0, 0)
            }
#pragma unroll
            for (int i = 0; i < 4; i++) o[i] = tap5p(p0[i], p0[i + 1], p0[i + 2], p0[i + 3], p0[i + 4]);
            st4(&sRow[(0 * G::ROW_H + r) * G::ROW_W + cx], o);
#pragma unroll
            for (int i = 0; i < 4; i++) o[i] = tap5p(p1[i], p1[i + 1], p1[i + 2], p1[i + 3], p1[i + 4]);
            st4(&sRow[(1 * G::ROW_H + r) * G::ROW_W + cx], o);
#pragma unroll
            for (int i = 0; i < 4; i++) o[i] = tap5p(p4[i], p4[i + 1], p4[i + 2], p4[i + 3], p4[i + 4]);
            st4(&sRow[(4 * G::ROW_H + r) * G::ROW_W + cx], o);
        } else {
            float ru[12], rd[12];
            ld4(pr - G::SR_W, ru); ld4(pr - G::SR_W + 4, ru + 4); ld4(pr - G::SR_W + 8, ru + 8);
            ld4(pr + G::SR_W, rd); ld4(pr + G::SR_W + 4, rd + 4); ld4(pr + G::SR_W + 8, rd + 8);
            float p2[8], p3[8];
#pragma unroll
            for (int j = 0; j < 8; j++) {
                const float lvj = l[j + 2];
                p2[j] = lvj * ru[j + 2];  // shift (0, -1)
                p3[j] = lvj * rd[j + 2];  // shift (0, +1)
            }
#pragma unroll
            for (int i = 0; i < 4; i++) o[i] = tap5p(p2[i], p2[i + 1], p2[i + 2], p2[i + 3], p2[i + 4]);
            st4(&sRow[(2 * G::ROW_H + r) * G::ROW_W + cx], o);
#pragma unroll
            for (int i = 0; i < 4; i++) o[i] = tap5p(p3[i], p3[i + 1], p3[i + 2], p3[i + 3], p3[i + 4]);
            st4(&sRow[(3 * G::ROW_H + r) * G::ROW_W + cx], o);
        }
        // row pass of R'^2 on rows tile+halo3, quad columns -4 .. TXS+3: shared by both roles
        for (int it = ct; it < G::SBROW_H * G::SB_Q; it += G::CT) {
            const int q = it / G::SBROW_H, rr = it - q * G::SBROW_H;
            const int cb = q * 4 - 4;
            float v[12];
            const float *pq = &sR[rr * G::SR_W + cb - 4 + G::SR_OX];
            ld4(pq, v); ld4(pq + 4, v + 4); ld4(pq + 8, v + 8);
            float sq[8];
#pragma unroll
            for (int j = 0; j < 8; j++) sq[j] = v[j + 2] * v[j + 2];
#pragma unroll
            for (int i = 0; i < 4; i++) o[i] = tap5p(sq[i], sq[i + 1], sq[i + 2], sq[i + 3], sq[i + 4]);
            st4(&sBrow[rr * G::SB_W + cb + G::SB_OX], o);
        }
    }
    __syncthreads();
    // ---- P2.5: column pass of R'^2 -> B on tile+halo1 -------------------------------------------------------------------------
    if (ct < G::SB_H * G::SB_Q) {
        const int q = ct / G::SB_H, r = ct - q * G::SB_H;  // r: tile row + 1
        float a[4], b[4], c[4], d[4], e[4], o[4];
        const float *pb = &sBrow[r * G::SB_W + q * 4];
        ld4(pb, a); ld4(pb + G::SB_W, b); ld4(pb + 2 * G::SB_W, c); ld4(pb + 3 * G::SB_W, d); ld4(pb + 4 * G::SB_W, e);
#pragma unroll
        for (int i = 0; i < 4; i++) o[i] = tap5p(a[i], b[i], c[i], d[i], e[i]);
        st4(&sB[r * G::SB_W + q * 4], o);
    }
    __syncthreads();
    // ---- P3: column pass of the products + MoveCorrelation (MatchLib.cu:681-687) for this channel -------------------------
    float q4a[4] = {0.0f, 0.0f, 0.0f, 0.0f}, q4b[4] = {0.0f, 0.0f, 0.0f, 0.0f}, q2[2] = {0.0f, 0.0f};
    if (live) {
        const float *pb = &sB[(trow + 1) * G::SB_W + cx - 4 + G::SB_OX];
        float a4[4];
        ld4(&sA[trow * G::ROW_W + cx], a4);
        auto colpass = [&](int s, float *N) {
            float r0[4], r1[4], r2[4], r3[4], r4[4];
            const float *ps = &sRow[(s * G::ROW_H + trow) * G::ROW_W + cx];  // rows trow .. trow+4 <-> tile rows trow-2 .. trow+2
            ld4(ps, r0); ld4(ps + G::ROW_W, r1); ld4(ps + 2 * G::ROW_W, r2); ld4(ps + 3 * G::ROW_W, r3); ld4(ps + 4 * G::ROW_W, r4);
#pragma unroll
            for (int i = 0; i < 4; i++) N[i] = tap5p(r0[i], r1[i], r2[i], r3[i], r4[i]);
        };
        // shift (0,0): each role takes two pixels of the quad (role 0: 0,1; role 1: 2,3)
        auto half4 = [&](const int hsel, const float *bq) {
            float r0[2], r1[2], r2[2], r3[2], r4[2];
            const float *ps = &sRow[(4 * G::ROW_H + trow) * G::ROW_W + cx + 2 * hsel];
            ld2(ps, r0); ld2(ps + G::ROW_W, r1); ld2(ps + 2 * G::ROW_W, r2); ld2(ps + 3 * G::ROW_W, r3); ld2(ps + 4 * G::ROW_W, r4);
#pragma unroll
            for (int j = 0; j < 2; j++) q2[j] = ncc2_nn(tap5p(r0[j], r1[j], r2[j], r3[j], r4[j]), a4[2 * hsel + j], bq[2 * hsel + j]);
        };
        float N[4];
        if (role == 0) {
            float bc[12];
            ld4(pb, bc); ld4(pb + 4, bc + 4); ld4(pb + 8, bc + 8);  // columns cx-4 .. cx+7, pixel i at [i+4]
            colpass(0, N);
#pragma unroll
            for (int i = 0; i < 4; i++) q4a[i] = ncc2_nn(N[i], a4[i], (!INTERIOR && gx0 + i == 0) ? bc[i + 4] : bc[i + 3]);
            colpass(1, N);
#pragma unroll
            for (int i = 0; i < 4; i++) q4b[i] = ncc2_nn(N[i], a4[i], (!INTERIOR && gx0 + i >= W - 1) ? bc[i + 4] : bc[i + 5]);
            half4(0, bc + 4);
        } else {
            float bm[4], bu[4], bd[4];
            ld4(pb + 4, bm); ld4(pb - G::SB_W + 4, bu); ld4(pb + G::SB_W + 4, bd);
            const bool top = !INTERIOR && (gy == 0), bot = !INTERIOR && (gy == H - 1);
            colpass(2, N);
#pragma unroll
            for (int i = 0; i < 4; i++) q4a[i] = ncc2_nn(N[i], a4[i], top ? bm[i] : bu[i]);
            colpass(3, N);
#pragma unroll
            for (int i = 0; i < 4; i++) q4b[i] = ncc2_nn(N[i], a4[i], bot ? bm[i] : bd[i]);
            half4(1, bm);
        }
    }
    // channels 1 and 2 hand their quotients to channel 0 through their own (dead) sR/sL/sBrow images
    if (k > 0) {
        float *const xa = sm, *const xb = sm + 4 * G::CT, *const xc = sm + 8 * G::CT;
        st4(xa + 4 * ct, q4a);
        st4(xb + 4 * ct, q4b);
        *reinterpret_cast<float2 *>(xc + 2 * ct) = make_float2(q2[0], q2[1]);
    }
    __syncthreads();
    float *const s0 = smem;  // channel 0's images: sRow planes become the hand-over buffers of the epilogue
    float *const xq = s0 + G::O_SROW + (0 * G::ROW_H + trow) * G::ROW_W + cx;
    float Qa[4], Qb[4];
    if (k == 0) {
        // mean over the channels in the reference's order: ((q0 + q1) + q2) / 3 (MatchGPULib.cpp:2168-2187)
        float u4[4], v4[4], u2[2], v2[2];
        const float *const m1 = smem + G::CH_FLOATS, *const m2 = smem + 2 * G::CH_FLOATS;
        ld4(m1 + 4 * ct, u4); ld4(m2 + 4 * ct, v4);
#pragma unroll
        for (int i = 0; i < 4; i++) Qa[i] = div3_nonneg((u4[i] + q4a[i]) + v4[i]);
        ld4(m1 + 4 * G::CT + 4 * ct, u4); ld4(m2 + 4 * G::CT + 4 * ct, v4);
#pragma unroll
        for (int i = 0; i < 4; i++) Qb[i] = div3_nonneg((u4[i] + q4b[i]) + v4[i]);
        ld2(m1 + 8 * G::CT + 2 * ct, u2); ld2(m2 + 8 * G::CT + 2 * ct, v2);
        if (live) {  // Q(0,0): each role publishes its two pixels, both parabolas need the four
            xq[2 * role] = div3_nonneg((u2[0] + q2[0]) + v2[0]);
            xq[2 * role + 1] = div3_nonneg((u2[1] + q2[1]) + v2[1]);
        }
    }
    __syncthreads();
    if (k == 0 && live) {
        float c4[4], dd[4], rho[4];
        ld4(xq, c4);
#pragma unroll
        for (int i = 0; i < 4; i++) poly_fast(c4[i], Qa[i], Qb[i], thr, dd[i], rho[i]);  // x: shifts (-1,0),(+1,0); y: (0,-1),(0,+1)
        st4(s0 + G::O_SROW + ((1 + role) * G::ROW_H + trow) * G::ROW_W + cx, dd);   // plane 1: delta x, plane 2: delta y
        st4(s0 + G::O_SROW + ((3 + role) * G::ROW_H + trow) * G::ROW_W + cx, rho);  // plane 3: rho x,  plane 4: rho y
    }
    __syncthreads();
    gchar_c *const Nb[3] = {uniform_base(nd3), uniform_base(nd3 + n), uniform_base(nd3 + 2 * n)};
    const float *const sX = s0 + G::O_SROW;
#pragma unroll
    for (int u = 0; u < NO; u++) {
        const int it = tid + u * G::NT;
        const int r = it / TXS, c = it - r * TXS;
        const int gxo = x0 + c, gyo = y0 + r;
        if (it < TXS * TYS && (INTERIOR || (gxo < W && gyo < H))) {
            const unsigned off = ((unsigned)gyo * (unsigned)W + (unsigned)gxo) * 4u;
            const float ddx_ = sX[(1 * G::ROW_H + r) * G::ROW_W + c], ddy_ = sX[(2 * G::ROW_H + r) * G::ROW_W + c];
            float kap = sX[(4 * G::ROW_H + r) * G::ROW_W + c] * sX[(3 * G::ROW_H + r) * G::ROW_W + c];  // rho_y * rho_x
            if (blend) kap = blend_conf(od[2][u], kap);
            st_at(Nb[0], off, od[0][u] + ddx_);
            st_at(Nb[1], off, od[1][u] + ddy_);
            st_at(Nb[2], off, kap);
        }
    }
}

template <int TXS, int TYS>
__global__ __launch_bounds__((CostSmall<TXS, TYS>::NT), 4) void k_cost_small(Img3 L, Img3 R, const float *__restrict__ A3, const float *__restrict__ d3,
                                                                        float *__restrict__ nd3, int W, int H, float thr, int blend, int tiles_x, int n_tiles,
                                                                        Batch bt)
{
    extern __shared__ __attribute__((aligned(16))) float smem_cs[];
    if (bt.n > 1) {  // this workgroup's pair of the batch (blockIdx.y)
        const int b = (int)blockIdx.y;
        L.p = shifted(L.p, bt.img[b]);
        R.p = shifted(R.p, bt.img[b]);
        A3 = shifted(A3, bt.in[b]);
        d3 = shifted(d3, bt.in[b]);
        nd3 = shifted(nd3, bt.out[b]);
    }
    int tile_x, tile_y;
    xcd_tile(n_tiles, tiles_x, tile_x, tile_y);
    const int x0 = tile_x * TXS, y0 = tile_y * TYS;
    const bool interior = x0 >= 3 && y0 >= 3 && x0 + TXS + 3 <= W && y0 + TYS + 3 <= H;
    if (interior) cost_small_body<TXS, TYS, true>(L, R, A3, d3, nd3, W, H, thr, blend, x0, y0, smem_cs);
    else cost_small_body<TXS, TYS, false>(L, R, A3, d3, nd3, W, H, thr, blend, x0, y0, smem_cs);
}

template <int TXS, int TYS>
static void launch_cost_small_t(hipStream_t st, Img3 L, Img3 R, const float *A3, const float *d3, float *nd3, int W, int H, float thr, int blend, const Batch *bt)
{
    Batch one{};
    one.n = 1;
    const Batch &B = bt ? *bt : one;
    using G = CostSmall<TXS, TYS>;
    const int tiles_x = (W + TXS - 1) / TXS, n_tiles = tiles_x * ((H + TYS - 1) / TYS);
    constexpr size_t bytes = 3 * (size_t)G::CH_FLOATS * sizeof(float);
    static_assert(bytes <= 64 * 1024, "stays under the default dynamic LDS limit");
    UGSM_LAUNCH((k_cost_small<TXS, TYS>), dim3(n_tiles, B.n > 1 ? B.n : 1), dim3(G::NT), bytes, st, L, R, A3, d3, nd3, W, H, thr, blend, tiles_x, n_tiles, B);
}

void launch_cost_small(hipStream_t st, Img3 L, Img3 R, const float *A3, const float *d3, float *nd3, int W, int H, float thr, int blend, const Batch *bt)
{
    launch_cost_small_t<16, 12>(st, L, R, A3, d3, nd3, W, H, thr, blend, bt);
}

// =========================================================================================
// k_smooth_small: P (<= 5) Jacobi passes of smoothKernel (MatchLib.cu:1092-1145) + the 3x3 box (convolutionRows/ColumnsKernelTa,
// :1593-1697), same arithmetic as k_smooth_fused (ugsm_kernels_smooth.hip), one THREAD PER PIXEL of the tile + halo 7 region.
// =========================================================================================
// Region 32 columns x RH rows -> tile 18 x (RH - 14).  A pixel's own fields stay in registers across the passes, west / east come
// from the neighbouring lanes (a region row is half a wave), north / south from LDS; the fields are double-buffered there, so a
// pass costs one barrier.  Image borders are per-pixel selects (smoothKernel leaves row 0 / column 0 untouched and clamps x+1 / y+1
// at the last column / row); the box reads its taps at clamped coordinates.
template <int RH>
__global__ __launch_bounds__(32 * RH) void k_smooth_small(const float *__restrict__ s3, float *__restrict__ o3, int W, int H, int P, int do_box,
                                                          int tiles_x, int n_tiles, Batch bt)
{
    constexpr int RW = 32, HALO = 7, STX = RW - 2 * HALO, STY = RH - 2 * HALO, NT = RW * RH;
    __shared__ float buf[2][3][NT];
    if (bt.n > 1) {  // this workgroup's pair of the batch (blockIdx.y)
        s3 = shifted(s3, bt.in[blockIdx.y]);
        o3 = shifted(o3, bt.out[blockIdx.y]);
    }
    const int tid = threadIdx.x;
    const int c = tid & (RW - 1), r = tid >> 5;
    int tile_x, tile_y;
    xcd_tile(n_tiles, tiles_x, tile_x, tile_y);
    const int x0 = tile_x * STX - HALO, y0 = tile_y * STY - HALO;  // global coordinates of region cell (0, 0)
    const int gx = x0 + c, gy = y0 + r;
    const size_t n = (size_t)W * H;
    const int h = P + (do_box ? 2 : 0);                             // halo actually needed
    const int inset = min(min(c, RW - 1 - c), min(r, RH - 1 - r));  // distance from the region's edge
    gchar_c *const Sb[3] = {uniform_base(s3), uniform_base(s3 + n), uniform_base(s3 + 2 * n)};

    float v[3];
    {
        const unsigned off = ((unsigned)clampi(gy, 0, H - 1) * (unsigned)W + (unsigned)clampi(gx, 0, W - 1)) * 4u;
        const bool need = inset >= HALO - h;
#pragma unroll
        for (int f = 0; f < 3; f++) {
            v[f] = need ? ld_at(Sb[f], off) : 0.0f;
            buf[0][f][tid] = v[f];
        }
    }
    __syncthreads();

    const bool act = gy > 0 && gy < H && gx > 0 && gx < W;
    const bool east_in = gx + 1 <= W - 1, south_in = gy + 1 <= H - 1;
    int cur = 0;
    for (int p = 1; p <= P; p++) {
        const bool on = inset >= HALO - (h - p);  // pass p is needed (and valid) on the region shrunk to halo h - p
        float vw[3], ve[3];
#pragma unroll
        for (int f = 0; f < 3; f++) {  // (at c = 0 / 31 the value comes from another row: region-edge columns, never valid in any pass)
            vw[f] = lane_below(v[f]);
            ve[f] = lane_above(v[f]);
        }
        if (on) {
            float vn[3], vs[3];
#pragma unroll
            for (int f = 0; f < 3; f++) {
                vn[f] = buf[cur][f][tid - RW];
                const float s_raw = buf[cur][f][tid + RW];
                vs[f] = south_in ? s_raw : v[f];
                ve[f] = east_in ? ve[f] : v[f];
            }
            const float wc = v[2], ww = vw[2], we = ve[2], wn = vn[2], ws = vs[2];
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
                a = v[f] * wc + a;
                a = vw[f] * ww + a;
                a = ve[f] * we + a;
                a = vn[f] * wn + a;
                a = vs[f] * ws + a;
                acc[f] = a;
            }
            div3_shared(acc[0], acc[1], acc[2], sumCorr, qf[0], qf[1], qf[2]);
            const bool lit = !div3_shared_ok(sumCorr);
            if (__builtin_amdgcn_ballot_w64(lit) != 0) {  // rare: a denominator outside the shared reciprocal's range
                asm volatile("; literal division of smoothKernel" ::: "memory");
                if (lit) {
#pragma unroll
                    for (int f = 0; f < 3; f++) qf[f] = acc[f] / sumCorr;
                }
            }
#pragma unroll
            for (int f = 0; f < 3; f++) {
                v[f] = act ? qf[f] : v[f];
                buf[cur ^ 1][f][tid] = v[f];
            }
        }
        __syncthreads();
        cur ^= 1;
    }

    const bool in_tile = c >= HALO && c < HALO + STX && r >= HALO && r < HALO + STY && gx < W && gy < H;
    gchar_c *const Ob[3] = {uniform_base(o3), uniform_base(o3 + n), uniform_base(o3 + 2 * n)};
    const unsigned ooff = in_tile ? ((unsigned)gy * (unsigned)W + (unsigned)gx) * 4u : 0u;
    if (do_box) {
        // rows (Ta): tile columns, in-image rows of tile -2 .. tile + STY + 1, rounded to f32 into the other buffer
        const bool rowp = c >= HALO && c < HALO + STX && r >= HALO - 2 && r < HALO + STY + 2 && gx < W && gy >= 0 && gy < H;
        if (rowp) {
            int cj[5];
#pragma unroll
            for (int j = 0; j < 5; j++) cj[j] = r * RW + clampi(gx + j - 2, 0, W - 1) - x0;
#pragma unroll
            for (int f = 0; f < 3; f++) {
                const float *b = buf[cur][f];
                buf[cur ^ 1][f][tid] = box5f(b[cj[0]], b[cj[1]], b[cj[2]], b[cj[3]], b[cj[4]]);
            }
        }
        __syncthreads();
        if (in_tile) {  // columns (Ta)
            int rj[5];
#pragma unroll
            for (int j = 0; j < 5; j++) rj[j] = (clampi(gy + j - 2, 0, H - 1) - y0) * RW + c;
#pragma unroll
            for (int f = 0; f < 3; f++) {
                const float *b = buf[cur ^ 1][f];
                st_at(Ob[f], ooff, box5f(b[rj[0]], b[rj[1]], b[rj[2]], b[rj[3]], b[rj[4]]));
            }
        }
    } else if (in_tile) {
#pragma unroll
        for (int f = 0; f < 3; f++) st_at(Ob[f], ooff, v[f]);
    }
}

template <int RH>
static void launch_smooth_small_t(hipStream_t st, const float *s3, float *o3, int W, int H, int passes, int do_box, const Batch *bt)
{
    constexpr int STX = 18, STY = RH - 14;
    const int tiles_x = (W + STX - 1) / STX, n_tiles = tiles_x * ((H + STY - 1) / STY);
    Batch one{};
    one.n = 1;
    const Batch &B = bt ? *bt : one;
    UGSM_LAUNCH((k_smooth_small<RH>), dim3(n_tiles, B.n > 1 ? B.n : 1), dim3(32 * RH), 0, st, s3, o3, W, H, passes, do_box, tiles_x, n_tiles, B);
}

void launch_smooth_small(hipStream_t st, const float *s3, float *o3, int W, int H, int passes, int do_box, int rh, const Batch *bt)
{
    if (passes < 0 || passes > 5) {  // the halo of 7 covers five passes + the box (callers split longer runs, enqueue_smooth)
        launch_smooth_fused(st, s3, o3, W, H, passes, do_box, 0, bt);
        return;
    }
    if (rh == 18) launch_smooth_small_t<18>(st, s3, o3, W, H, passes, do_box, bt);
    else if (rh == 24) launch_smooth_small_t<24>(st, s3, o3, W, H, passes, do_box, bt);
    else launch_smooth_small_t<32>(st, s3, o3, W, H, passes, do_box, bt);
}

}  // namespace ugsm
