// ugsm_kernels_pyr.hip -- the image pyramid and what is computed once per level.
//
// K-pyr-base (k_pyr_base / k_pyr_base_march): rgb8 -> levels 0, 1, 2 in one pass;  K-pyr (k_blur_decimate2 / k_blur_decimate_tiled): the
// further levels, blur evaluated only at the sampled sites;  K-sq (k_sqblur_tiled): A = G_clamp * (L^2);  k_range_scan.
// Replaces CreatePyramidFromImage and its per-channel upload / blur / download / subsample round trips
// (/root/reference/src/gpu_matcher/MatchGPULib.cpp:1033-1125, :866-1031).  Same IEEE operations in the same order as the CPU oracle.
#include "ugsm_exact.hpp"
#include "ugsm_launch.hpp"
#include <algorithm>

namespace ugsm {

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
// Where the time goes (ablations, round 2, 16 MP): the 5-tap passes are bound by LDS instructions, not by arithmetic.
// So the row pass is computed DENSELY, once for both levels, a quad of outputs from two 16-byte LDS reads (64 columns instead of
// the 84 candidate columns of the two levels, and a third of the LDS instructions); only the column pass runs at the sampled
// sites: waves 0-2 own the 52 candidate columns of level 1 (one row phase each), wave 3 the 32 columns of level 2 (two rows at a
// time), so that an output row segment is written by the lanes of one wave.  Level 0 leaves as 16-byte stores.
constexpr int BTX = 64, BTY = 16, BRW = BTX + 8, BRH = BTY + 4, BC1 = 52, BR1 = 16, BC2 = 32, BR2 = 8;
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
    if (store0) {  // level 0: the tile itself; a thread owns 4 consecutive pixels of one row
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
    if (tcol >= 0) {
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
                    dst[k * nd + at] = v;
                    bad |= !range_ok(v);
                }
            }
        }
    }
    if (bad && range_bad) *range_bad = 1u;
}

// K-pyr-base, streaming form (round 4): the same three levels from the same rgb8 input with the same arithmetic, without LDS.  The tiled
// kernel above is bound by its LDS passes (ablations, round 2: 44 us of 132 without them at 16 MP); here ONE WAVE owns a strip of
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
    UGSM_LAUNCH(k_pyr_base, dim3(n_tiles, B.n > 1 ? B.n : 1), dim3(256), 0, st, rgb, stride, W, H, lvl0, lvl1, W1, H1, lvl2, W2, H2, range_bad, tiles_x, n_tiles, B, win);
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
    // (1 <= sf <= 2: the tiled kernel's region bound assumes it; the pyramid asks for sqrt 2 and 2, MatchGPULib.cpp:1071-1096)
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

// range_bad[0] = 1 if any of the `count` floats at p is outside range_ok (ugsm_exact.hpp); the caller zeroes the word first.
// The pyramid kernels make this check as they write a level; this pass serves the stage-level test entry points, which
// receive their planes ready-made.
__global__ __launch_bounds__(256) void k_range_scan(const float *__restrict__ p, size_t count, unsigned *__restrict__ range_bad)
{
    bool bad = false;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < count; i += (size_t)gridDim.x * 256) bad |= !range_ok(p[i]);
    if (bad) *range_bad = 1u;
}
void launch_range_scan(hipStream_t st, const float *p, size_t count, unsigned *range_bad)
{
    const size_t blocks = (count + 255) / 256;
    UGSM_LAUNCH(k_range_scan, dim3((unsigned)(blocks < 4096 ? (blocks ? blocks : 1) : 4096)), dim3(256), 0, st, p, count, range_bad);
}

}  // namespace ugsm
