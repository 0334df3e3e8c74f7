// ugsm_kernels_smooth.hip -- K-smooth: up to 5 confidence-weighted Jacobi passes + the 3x3 box in one LDS-tiled launch.
//
// Arithmetic is bit-identical to the one-stage-per-kernel path (dev/ugsm_dev_stages.hip) and to the CPU oracle: same IEEE operations in
// the same order, no contraction (see ugsm_device.hpp).  No MFMA: a 5-point stencil.  The coarse levels' form is k_smooth_small
// (ugsm_kernels_small.hip).  Citations: /root/reference/src/gpu_matcher/<file>:<line>.
#include "ugsm_exact.hpp"
#include "ugsm_launch.hpp"
#include <algorithm>
#include <atomic>
#include <type_traits>

namespace ugsm {

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
// What was measured and not kept -- a pipelined form with LDS-DMA prefetch, a marching form, the products v * kappa kept in LDS, the
// box's row pass fused into the last pass, binary32 Newton quotients, three workgroups per CU: docs/HISTORY.md, profiles/r04_* / r05_kbench_smooth_*.
constexpr int smooth_pad(int stx) { return stx == 112 ? 0 : 4; }  // floats added to an LDS row (rows stay 16-byte aligned)
// The passes, the box and the copy-out of ONE tile whose region (tile + halo, clamped onto the image) is in LDS at f0 / f1 / f2.  Every
// thread of the workgroup calls it; it contains barriers.
// FIXH: the tile is STY rows high whatever `sty` says -- the height folds into the loop bounds, 2 % faster at level 0 than the
// same kernel with the height in a register (208 against 212.5 us); the launcher picks it whenever the height is STY.
template <int STX, int STY, int NT, bool FIXH>
__device__ __forceinline__ void smooth_tile_body(float *const f0, float *const f1, float *const f2, float *__restrict__ o3, const int W, const int H, const int P,
                                                 const int do_box, const int tile_x, const int tile_y, const int sty)
{
    constexpr int HX = 8, HY = 7;
    constexpr int RWID = STX + 2 * HX;       // region width (multiple of 4)
    constexpr int LW = RWID + smooth_pad(STX);  // LDS row stride (rows 16-B aligned)
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
    const int h = P + (do_box ? 2 : 0);  // halo actually needed
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
    const bool edge_e = x0 + RWID > W, edge_s = y0 + LHr > H, edge_nw = x0 <= 0 || y0 <= 0;

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
        auto quad_row = [&](const int u, const float (&c4)[3][4], const float (&n4)[3][4], const float (&s4)[3][4],
                            const float (&wl)[3], const float (&er)[3], auto lit_tag) -> bool {
            constexpr bool LIT = decltype(lit_tag)::value;
            bool ok = true;
#pragma unroll
            for (int i = 0; i < 4; i++) {
                float vw[3], ve[3], vs[3];
#pragma unroll
                for (int f = 0; f < 3; f++) {
                    vw[f] = (i == 0) ? wl[f] : c4[f][i > 0 ? i - 1 : 0];
                    ve[f] = (i == 3) ? er[f] : c4[f][i < 3 ? i + 1 : 3];
                    vs[f] = s4[f][i];
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
                    a = c4[f][i] * wc + a;
                    a = vw[f] * ww + a;
                    a = ve[f] * we + a;
                    a = n4[f][i] * wn + a;
                    a = vs[f] * ws + a;
                    acc[f] = a;
                }
                if constexpr (LIT) {
#pragma unroll
                    for (int f = 0; f < 3; f++) qf[f] = acc[f] / sumCorr;
                } else {
                    div3_shared(acc[0], acc[1], acc[2], sumCorr, qf[0], qf[1], qf[2]);
                    ok = ok && div3_shared_ok(sumCorr);
                }
#pragma unroll
                for (int f = 0; f < 3; f++) nv[u][f][i] = qf[f];
                __builtin_amdgcn_sched_barrier(0);  // keep the binary64 temporaries of one pixel at a time
            }
            return ok;
        };
        // one row of the thread: west / east taps, the quad-row, the rare literal redo, the pass-through cells of the frame
        auto do_row = [&](const int u, const int r, const float (&n4)[3][4], const float (&c4)[3][4], const float (&s4)[3][4]) {
            const int at = r * LW + c0;
            float wl[3], er[3];
            // west / east neighbours from the neighbouring lanes' registers instead of LDS (a narrowed,
            // lane-strided ds_read_b32 there is a 4-way bank conflict).  At q = 0 / QW-1 the value comes
            // from another row: those are region-edge columns, never valid in any pass.
#pragma unroll
            for (int f = 0; f < 3; f++) {
                wl[f] = lane_below(c4[f][3]);
                er[f] = lane_above(c4[f][0]);
            }
            bool redo = false;
            const int gy = y0 + r;
            if (col_on) redo = !quad_row(u, c4, n4, s4, wl, er, std::false_type{});
            if (__builtin_expect(redo, 0)) {
                // rare: a denominator out of range.  Reload the row (so that nothing has to stay in registers
                // for this path; LDS still holds the previous pass) and divide literally.
                float c4r[3][4], n4r[3][4], s4r[3][4], wlr[3], err[3];
                ld4(f0 + at, c4r[0]); ld4(f1 + at, c4r[1]); ld4(f2 + at, c4r[2]);
                ld4(f0 + at - LW, n4r[0]); ld4(f1 + at - LW, n4r[1]); ld4(f2 + at - LW, n4r[2]);
                ld4(f0 + at + LW, s4r[0]); ld4(f1 + at + LW, s4r[1]); ld4(f2 + at + LW, s4r[2]);
                wlr[0] = f0[at - 1]; wlr[1] = f1[at - 1]; wlr[2] = f2[at - 1];
                err[0] = f0[at + 4]; err[1] = f1[at + 4]; err[2] = f2[at + 4];
                quad_row(u, c4r, n4r, s4r, wlr, err, std::true_type{});
            }
            if (edge_nw && col_on) {  // row 0 / column 0 (and anything left / above the image) keeps its value
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
#pragma unroll
        for (int u = 0; u < MAXR; u++) {
            const int r = r_lo + rg + u * RG;
            if (col_on && r < r_hi) {
                const int at = r * LW + c0;
                st4(f0 + at, nv[u][0]); st4(f1 + at, nv[u][1]); st4(f2 + at, nv[u][2]);
            }
        }
        __syncthreads();
        if (edge_e || edge_s) {  // re-establish the east / south replicas
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

template <int STX, int STY, int NT, bool FIXH = false>
__global__ __launch_bounds__(NT, (NT <= 512 ? NT / 128 : 1)) void k_smooth_fused(const float *__restrict__ s3, float *__restrict__ o3, int W, int H, int P, int do_box,
                                                  int tiles_x, int n_tiles, int sty_arg, Batch bt)
{
    if (bt.n > 1) {  // this workgroup's pair of the batch (blockIdx.y)
        s3 = shifted(s3, bt.in[blockIdx.y]);
        o3 = shifted(o3, bt.out[blockIdx.y]);
    }
    const int sty = FIXH ? STY : sty_arg;
    constexpr int HX = 8, HY = 7;
    constexpr int RWID = STX + 2 * HX;       // region width (multiple of 4)
    constexpr int LW = RWID + smooth_pad(STX);  // LDS row stride (rows 16-B aligned)
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
    const int h = P + (do_box ? 2 : 0);  // halo actually needed

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
            v[u][0] = need ? *reinterpret_cast<const float *>(reinterpret_cast<const char *>(s3) + off) : 0.0f;
            v[u][1] = need ? *reinterpret_cast<const float *>(reinterpret_cast<const char *>(s3 + n) + off) : 0.0f;
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
            if (r < LHr) {
                f0[r * LW + c] = v[u][0];
                f1[r * LW + c] = v[u][1];
                f2[r * LW + c] = v[u][2];
            }
        }
    }
    __syncthreads();

    smooth_tile_body<STX, STY, NT, FIXH>(f0, f1, f2, o3, W, H, P, do_box, tile_x, tile_y, sty);
}

constexpr int kSmoothMidNT = 1024;  // 64 x 32 tiles: one quad-row per thread per pass -- the mid levels are latency-bound (16.5 vs 17.7 ms per pair)
constexpr int kSmoothSmallNT = 512;
template <int STX, int STY, int NT>
static void launch_smooth_t(hipStream_t st, const float *s3, float *o3, int W, int H, int passes, int do_box, int sty, const Batch *bt)
{
    Batch one{};
    one.n = 1;
    const Batch &B = bt ? *bt : one;
    const int pairs = B.n > 1 ? B.n : 1;
    constexpr int LW = STX + 16 + smooth_pad(STX), LH = STY + 14;
    constexpr size_t max_bytes = 3 * (size_t)LH * LW * sizeof(float) + 65536;  // (+ what smooth_lds_extra_bytes may ask for)
    // the attribute is per device: a process may hold contexts on several devices (the launch is made with the context's
    // device current); std::atomic so that contexts driven from different host threads do not race on the mask
    static std::atomic<unsigned long long> attr_mask{0};
    int dev = 0;
    (void)hipGetDevice(&dev);
    const unsigned long long bit = 1ull << (dev & 63);
    if (!(attr_mask.load(std::memory_order_relaxed) & bit)) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&k_smooth_fused<STX, STY, NT, false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)max_bytes);
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&k_smooth_fused<STX, STY, NT, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)max_bytes);
        attr_mask.fetch_or(bit, std::memory_order_relaxed);
    }
    if (sty < 1 || sty > STY) sty = STY;
    const size_t bytes = 3 * (size_t)(sty + 14) * LW * sizeof(float) + (size_t)std::min(std::max(smooth_lds_extra_bytes, 0), 65536);
    const int tiles_x = (W + STX - 1) / STX, n_tiles = tiles_x * ((H + sty - 1) / sty);
    if (sty == STY) UGSM_LAUNCH((k_smooth_fused<STX, STY, NT, true>), dim3(n_tiles, pairs), dim3(NT), bytes, st, s3, o3, W, H, passes, do_box, tiles_x, n_tiles, sty, B);
    else UGSM_LAUNCH((k_smooth_fused<STX, STY, NT, false>), dim3(n_tiles, pairs), dim3(NT), bytes, st, s3, o3, W, H, passes, do_box, tiles_x, n_tiles, sty, B);
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
// (development: UGSM_SMOOTH_LDS_EXTRA -- LDS bytes a launch asks for beyond its tile's, i.e. fewer workgroups per CU: 10 000 puts the
// 112 x 36 tile at one workgroup per CU instead of two; the placement experiment of DESIGN.md section 4.3)
int smooth_lds_extra_bytes = 0;

void launch_smooth_fused(hipStream_t st, const float *s3, float *o3, int W, int H, int passes, int do_box, int tile_rows, const Batch *bt, int tile_class)
{
    // big levels: 112 x (up to 36) tiles, 512 threads, <= 77 KB LDS -> two workgroups per CU (meant to let one's load/store phase overlap
    // the other's passes; measured, the two phases still nearly add up: DESIGN.md section 4).  The region is 128 columns = 32 quads = two whole rows per wave: no idle lanes, and the halo
    // columns every pass recomputes are 12.5 % of the row instead of 20 % (at 16 MP: 5 passes 227 us against 266 us for
    // 64x58, 323 us for the first 64x64 version; 128x64x1024 with one workgroup per CU 406 us);
    // mid levels (from smooth_mid_min_pixels): 64x32; small levels: 32x16 so that the launch is short and the chip still fills
    const size_t px = (size_t)W * H;
    if (tile_rows > 0 || (tile_class == 0 && px >= ((size_t)1 << 19))) launch_smooth_t<112, kSmoothTileRowsMax, 512>(st, s3, o3, W, H, passes, do_box, tile_rows > 0 ? tile_rows : 36, bt);
    else if (tile_class == 2 || (tile_class == 0 && px >= (size_t)smooth_mid_min_pixels)) launch_smooth_t<64, 32, kSmoothMidNT>(st, s3, o3, W, H, passes, do_box, 0, bt);
    else launch_smooth_t<32, 16, kSmoothSmallNT>(st, s3, o3, W, H, passes, do_box, 0, bt);
}

}  // namespace ugsm
