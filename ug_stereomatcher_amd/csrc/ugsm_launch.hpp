// ugsm_launch.hpp -- host-visible launchers of the gfx950 kernels.
#pragma once

#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stddef.h>
#include <stdint.h>

namespace ugsm {

// Per-launch timing (ugsm_config.profile_events): when the submitting thread has a probe set, the next launch carries the probe's two
// events IN its dispatch (hipExtLaunchKernelGGL: the kernel's own begin and end timestamps, what rocprofv3's kernel trace reports), not
// two hipEventRecord markers around it, whose interval also holds the gaps to the neighbouring launches (6 us per launch on this stack:
// 8 % of the dominant kernel's mean duration; profiles/r04_trace_summary.md).  Almost every bracket of the runtime holds ONE launch; where
// a bracket holds several (the development fallback of launch_blur_decimate: reference kernel + range scan), the first launch carries
// the start event and every launch re-records the stop event, so the interval runs from the first kernel's begin to the LAST kernel's
// end and no launch of the bracket goes untimed; n counts them.
struct LaunchProbe {
    hipEvent_t a, b;
    int n;
};
inline thread_local LaunchProbe *g_probe = nullptr;
#define UGSM_LAUNCH(kern, grid, block, shmem, st, ...)                                                                \
    do {                                                                                                              \
        ::ugsm::LaunchProbe *probe_ = ::ugsm::g_probe;                                                                \
        if (probe_) hipExtLaunchKernelGGL(kern, grid, block, shmem, st, probe_->n++ == 0 ? probe_->a : nullptr, probe_->b, 0, __VA_ARGS__); \
        else hipLaunchKernelGGL(kern, grid, block, shmem, st, __VA_ARGS__);                                           \
    } while (0)

// A 3-plane float image view: plane k starts at p + k*plane, rows are `pitch` floats
// apart.  Pyramid levels are dense (pitch == width); fovea crops of the pyramid
// (CreateFoveatedPyramid, MatchGPULib.cpp:1171-1185) are views into the full level,
// not copies -- kernels clamp / zero-pad at the view's logical W x H.
struct Img3 {
    const float *p;
    int pitch;
    size_t plane;
};

// Where a level's first iteration finds its starting field when the seeding is fused into K-cost: the coarser level's
// (dx, dy, conf), Ws x Hs, sampled as subsampleDispKernel does (MatchLib.cu:372-401) with the fovea crop offset (cx, cy).
struct SeedMap {
    int Ws, Hs, cx, cy;
};

// B pairs of ONE size in one launch (round 4: the batch dimension).  Every launch of a level of 615 x 407 pixels and below lasts as long
// as one tile's or one strip's chain, whatever the chip could do beside it; with B pairs marching through the levels in lockstep the
// same launch carries B times the work: gridDim.y (gridDim.z / 3 where y is the image row) = pairs, and the kernel shifts its pointers
// to its pair's planes before it does anything else.  Offsets are BYTES relative to the pointers of the launch, which are pair 0's.
constexpr int kMaxBatch = 16;  // = UGSM_MAX_BATCH (include/ugsm.h)
struct Batch {
    int n;                     // pairs in the launch; <= 1: one pair, nothing below is read
    int cx[kMaxBatch], cy[kMaxBatch];  // seeded launches: SeedMap.cx / cy of pair b (fovea windows of different pairs sit at different places)
    long long img[kMaxBatch];  // pair b's image planes: the L and R views (K-cost, A = G * L^2), the rgb8 image (k_pyr_base), a level of the pyramid (k_blur_decimate, k_copy_view)
    long long in[kMaxBatch];   // pair b's input fields: d3 / coarse3 / s3, and A3
    long long out[kMaxBatch];  // pair b's output: nd3 / o3 / dst (a slot buffer, or the caller's)
};

// ---- plain per-pixel kernels (ugsm_kernels_aux.hip) ---------------------------------------
void launch_seed(hipStream_t st, const float *src3, int Ws, int Hs, float *dst3, int Wd, int Hd, int cx, int cy, const Batch *bt = nullptr);
void launch_copy_view(hipStream_t st, Img3 src, int W, int H, float *dst, size_t dst_plane, int dst_pitch, const Batch *bt = nullptr);
// rgb8 -> planar float level 0 (MatchGPULib.cpp:332-338) where k_pyr_base does not apply: pyramids of fewer than three levels, kernel_path 1
void launch_rgb_planes(hipStream_t st, const uint8_t *rgb, int stride, int W, int H, float *planes);
// LR-consistency check (north_star; no reference counterpart): zeroes the confidence of left3 where right3 does not point back within tau
void launch_lr_check(hipStream_t st, float *left3, const float *right3, int W, int H, float tau, unsigned long long *marked);
// X, Y, Z planes from the full-resolution (dx, dy) and the two 3x4 projection matrices
void launch_triangulate(hipStream_t st, const float *dispx, const float *dispy, int W, int H, const double *P1, const double *P2, float *xyz);
void launch_triangulate_fovea(hipStream_t st, const float *stackx, const float *stacky, int fovW, int fovH, int src_level, int left_margin,
                              int upper_margin, float scale, const double *P1, const double *P2, float *xyz);
void launch_upsample_paste(hipStream_t st, const float *src3, int W, int H, float *dst3, int W2, int H2, const float *fovH_, const float *fovV_,
                           const float *fovC_, int fovW, int fovH, int org_x, int org_y);
// S_dx, S_dy, C of weightedDifference (MatchGPULib.cpp:1336-1437) into out3; rowsum = 3*H doubles of scratch
void launch_weighted_difference(hipStream_t st, const float *newd3, const float *oldd3, int W, int H, double *rowsum, double *out3);

// ---- libugsm_dev.so only (UGSM_DEV_LIB; csrc/dev/): kernel_path 1 (one kernel per reference stage), round 1's LDS-tiled K-cost
// (ugsm_config.march_min_pixels < 0) and the probe kernels.  The product build has no-op stand-ins so that the runtime reads the same;
// ugsm_create refuses the configurations that would reach them and the probe entry points are not compiled in.
#ifdef UGSM_DEV_LIB
constexpr bool kDevLib = true;
void launch_blur_decimate_ref(hipStream_t st, const float *src3, int W, int H, float *dst3, int W2, int H2, float sf);
void launch_sqblur_clamp_ref(hipStream_t st, Img3 src, int W, int H, float *dst3);
void launch_warp_ref(hipStream_t st, Img3 R, const float *d3, int W, int H, float *Rw3);
void launch_cost_ref(hipStream_t st, Img3 L, const float *Rw3, const float *A3, const float *B3, const float *d3, float *nd3,
                     int W, int H, float thr, int blend, float *dbg8);
void launch_smooth_pass_ref(hipStream_t st, const float *s3, float *o3, int W, int H);
void launch_box_ref(hipStream_t st, const float *s3, float *o3, int W, int H);
// One iteration's warp + cost + parabola + update, LDS-tiled (k_cost_split; no batch index)
void launch_cost_fused(hipStream_t st, Img3 L, Img3 R, const float *A3, const float *d3, float *nd3, int W, int H, float thr, int blend);
// test hooks: the product kernels' exact shortcuts (range-guarded division, shared-reciprocal division, parabola fast path, x/3) on arbitrary operands
void launch_div_probe(hipStream_t st, const float *n, const float *d, float *q, int count);
void launch_div3_probe(hipStream_t st, const float *a0, const float *a1, const float *a2, const float *s, float *q0, float *q1, float *q2, int n);
void launch_poly_probe(hipStream_t st, const float *c, const float *l, const float *r, const float *thr, float *delta, float *corr, float *third, int n);
#else
constexpr bool kDevLib = false;
inline void launch_blur_decimate_ref(hipStream_t, const float *, int, int, float *, int, int, float) {}
inline void launch_sqblur_clamp_ref(hipStream_t, Img3, int, int, float *) {}
inline void launch_warp_ref(hipStream_t, Img3, const float *, int, int, float *) {}
inline void launch_cost_ref(hipStream_t, Img3, const float *, const float *, const float *, const float *, float *, int, int, float, int, float *) {}
inline void launch_smooth_pass_ref(hipStream_t, const float *, float *, int, int) {}
inline void launch_box_ref(hipStream_t, const float *, float *, int, int) {}
inline void launch_cost_fused(hipStream_t, Img3, Img3, const float *, const float *, float *, int, int, float, int) {}
#endif

// ---- K-cost ----------------------------------------------------------------------------------
// One iteration's warp + cost + parabola + update as a marching kernel (ugsm_kernels_march.hip): one wave per strip of columns, no LDS.
// rows = strip height (> 0 fixed; 0 / -1 / -2 / -3: launch_cost_march_t's modes).
void launch_cost_march(hipStream_t st, Img3 L, Img3 R, const float *A3, const float *d3, float *nd3, int W, int H, float thr, int blend, int rows,
                       const unsigned *range_bad, const Batch *bt = nullptr);
// First iteration of a level with the seeding fused in: coarse3 = the coarser level's field (never materialised at this level's size)
void launch_cost_march_seeded(hipStream_t st, Img3 L, Img3 R, const float *A3, const float *coarse3, SeedMap sm, float *nd3, int W, int H, float thr,
                              int blend, int rows, const unsigned *range_bad, const Batch *bt = nullptr);
// strips by age class (ugsm_kernels_march.hip): share (per mille) of a strip group's rows for the first / second wave of a SIMD; {0, 0} = uniform
extern int march_age_permille[2];
// strip height the marching K-cost picks for a W x H level; host only
int march_strip_rows(int W, int H, int throughput = 0, int pairs = 1);
// The same iteration with the three colour channels on three waves of a workgroup and the epilogue on a fourth (ugsm_kernels_march4.hip): the
// latency form for levels whose launch lasts as long as one strip.  sm.Ws > 0: the level's first iteration, seeded from the coarser field d3.
void launch_cost_march4(hipStream_t st, Img3 L, Img3 R, const float *A3, const float *d3, float *nd3, int W, int H, float thr, int blend, int rows,
                        const unsigned *range_bad, SeedMap sm, const Batch *bt = nullptr);
int march4_strip_rows(int W, int H, int pairs = 1);
// The same iteration for the coarse levels (ugsm_kernels_small.hip): the three channels of a 16 x 12 tile side by side, built for the
// latency of one tile rather than for throughput.
void launch_cost_small(hipStream_t st, Img3 L, Img3 R, const float *A3, const float *d3, float *nd3, int W, int H, float thr, int blend, const Batch *bt = nullptr);

// ---- K-smooth --------------------------------------------------------------------------------
// `passes` Jacobi smoothing passes (+ the 3x3 box when do_box) in one LDS-tiled launch (ugsm_kernels_smooth.hip).
// tile_rows: > 0 = the 112-column tile at this height (1..kSmoothTileRowsMax; smooth_tile_rows picks it); 0 = the tile class by the
// level's size (112 x 36 from 0.5 Mpx, 64 x 32 from 0.26 Mpx, else 32 x 16).
constexpr int kSmoothTileRowsMax = 36;  // (39 rows still fit two workgroups per CU -- 3 x 53 x 128 floats = 81 408 B of LDS -- but the kernel's
                                        // unrolled load / box loops, sized for the tallest tile, then cost every 36-row launch 7.7 % more instructions)
// tile_class (batched launches: the class by what the launch holds, not by one pair's level): 0 = by the level's size, 1 = 32 x 16, 2 = 64 x 32
void launch_smooth_fused(hipStream_t st, const float *s3, float *o3, int W, int H, int passes, int do_box, int tile_rows = 0, const Batch *bt = nullptr,
                         int tile_class = 0);
int smooth_tile_rows(int W, int H, int whole_rounds, int pairs = 1);
extern int smooth_mid_min_pixels;  // (development: UGSM_SMOOTH_MID_MIN)
extern int smooth_lds_extra_bytes; // (development: UGSM_SMOOTH_LDS_EXTRA)
// `passes` (<= 5) Jacobi passes (+ box) for the coarse levels (ugsm_kernels_small.hip): one thread per pixel of an 18 x (rh - 14) tile + halo 7 (rh = 18, 24 or 32)
void launch_smooth_small(hipStream_t st, const float *s3, float *o3, int W, int H, int passes, int do_box, int rh, const Batch *bt = nullptr);

// ---- pyramid, A planes (ugsm_kernels_pyr.hip) ---------------------------------------------------
// range_bad (device word, may be null = unknown): 0 when every pyramid value of the pair passed range_ok (ugsm_exact.hpp),
// which lets K-cost use the range-guarded division; launch_range_scan ORs the check of `count` floats into it.
void launch_range_scan(hipStream_t st, const float *p, size_t count, unsigned *range_bad);
extern int blur_decimate_streaming;  // (development: UGSM_PYR_STREAM)
extern int pyr_base_streaming;       // (development: UGSM_PYR_BASE_STREAM)
extern long long blur_decimate_streaming_min;  // (development: UGSM_PYR_STREAM_MIN)
// zero-padded blur evaluated at the decimation sites
// (range_bad: every level value written is checked as it is produced; may be null)
// stream_min: launches of fewer output pixels keep the LDS-tiled kernel for a factor-2 level too (0 = the streaming kernel k_blur_decimate2 always)
void launch_blur_decimate(hipStream_t st, const float *src3, int W, int H, float *dst3, int W2, int H2, float sf, unsigned *range_bad, const Batch *bt = nullptr,
                          long long stream_min = 0);
// The part of level 0 a foveated call reads: the fovea window (w x h at (x0, y0), level-0 pixels); w <= 0: everything (full mode).  In a
// batched launch the origin of image b's window is bt->in[b] = (y0 << 32) | x0.
struct PyrWindow {
    int x0, y0, w, h;
};
void launch_pyr_base(hipStream_t st, const uint8_t *rgb, int stride, int W, int H, float *lvl0, float *lvl1, int W1, int H1, float *lvl2, int W2,
                     int H2, unsigned *range_bad, const Batch *bt = nullptr, PyrWindow win = PyrWindow{0, 0, 0, 0});
void launch_sqblur_clamp(hipStream_t st, Img3 src, int W, int H, float *dst3, const Batch *bt = nullptr);

}  // namespace ugsm
