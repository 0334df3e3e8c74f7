/*
 * ugsm.h -- C-ABI of libugsm.so: the MI355X (gfx950) pyramidal dense stereo matcher
 * that replaces ug_stereomatcher's MatchGPULib + MatchLib.cu behind the
 * UG_matcher_gpu node / GetDisparitiesGPU.srv.
 *
 * Plain C, plain pointers and sizes; no torch, OpenCV or ROS types.  File:line
 * citations are relative to /root/reference/src/gpu_matcher/ and name the reference
 * interface each entry point replaces.  The reference's own C boundary
 * (25 one-kernel `extern "C"` wrappers, MatchLib_common.h:35-74 and
 * MatchGPULib.cpp:45-247) is deliberately NOT replicated: that granularity is what
 * forces ~130 launches per iteration.  INTEGRATION.md shows the MatchGPULib shim and
 * the node-side binding.
 *
 * Threading: calls on one ctx must be serialised by the caller (the reference node
 * is a single-threaded ros::spin, UG_GPU_matcher.cpp:749-752).  Different slots of a
 * ctx run concurrently on the device.  No entry point calls exit(); every failure is
 * a status code (reference: checkCudaErrors -> exit(EXIT_FAILURE)).
 */
#ifndef UGSM_H
#define UGSM_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* libugsm.so is built with -fvisibility=hidden: what this header declares is everything the library exports (nm -D lists ugsm_* only). */
#if defined(__GNUC__)
#pragma GCC visibility push(default)
#endif

#define UGSM_ABI_VERSION 6  /* 6: the kernel choices follow what is in flight, not ugsm_config.slots: ugsm_plan_level takes `alone`, ugsm_plan_level_in_frame
                               is gone, ugsm_level_plan.latency_policy is .alone; ugsm_enqueue_* returns UGSM_OK once the pair is accepted (a failed
                               CALL is reported through ugsm_completion.status only); the fovea shard carries a status word (a rank that fails still
                               reaches the exchange); march_min_pixels < 0 and march_smooth are libugsm_dev.so's
                               5: the queue (ugsm_enqueue_*, ugsm_flush, ugsm_next_done, ugsm_queue_depth, ugsm_queue_plan, ugsm_poll; UGSM_PENDING / UGSM_EMPTY),
                               RCCL inside the library (ugsm_shard_*), hidden visibility for everything else
                               4: ugsm_config grew (batch, stream_priority), ugsm_submit_full_batch, ugsm_submit_foveated_batch; kernel_path 1,
                               march_smooth and the probe entry points moved to libugsm_dev.so (include/ugsm_dev.h)
                               3: ugsm_config grew (lr_check_threshold, streams), ugsm_stage_lr_check, ugsm_slot_stream, ugsm_last_lr_marked,
                               ugsm_submit_full_host, ugsm_submit_foveated_host, ugsm_plan_level_in_frame */

/* status codes */
#define UGSM_OK                0
#define UGSM_ERR_BAD_ARG       1  /* null pointer, non-positive size, bad slot/level, image above 2^28 pixels */
#define UGSM_ERR_SIZE_MISMATCH 2  /* left/right differ, or stride < 3*W (ref: unchecked, MatchGPULib.cpp:315-323) */
#define UGSM_ERR_TOO_SMALL     3  /* a pyramid level would be < 1 px (ref: zero-size malloc, MatchGPULib.cpp:1247) */
#define UGSM_ERR_NO_DEVICE     4  /* no HIP device / HIP runtime unusable */
#define UGSM_ERR_DEVICE        5  /* a HIP call failed; see ugsm_last_error */
#define UGSM_ERR_NOMEM         6
#define UGSM_ERR_STATE         7  /* e.g. fine phase without coarse phase */
#define UGSM_PENDING           8  /* not an error: ugsm_poll / ugsm_next_done(block = 0) -- the work asked about has not finished yet */
#define UGSM_EMPTY             9  /* not an error: ugsm_next_done -- every pair enqueued so far has been reported */
#define UGSM_ERR_PEER          10 /* the fovea shard: another rank failed its part of the step (this rank's result is not valid), or no rank answered
                                     within the deadline and the communicator was aborted; ugsm_last_error says which */

#define UGSM_MAX_LEVELS 32
#define UGSM_MAX_BATCH 16  /* pairs per ugsm_submit_*_batch call */

typedef struct ugsm_ctx ugsm_ctx;

/* Replaces the compile-time/argv configuration of the reference:
 *   device        <- "-device=N" parsed by findCudaDevice, MatchGPULib.cpp:254
 *   levels        <- MAX_LEVEL, MatchLib_common.h:13 (14)
 *   fovea_levels  <- foveatelevel = argv[2] or 7, MatchGPULib.cpp:259-264
 *   slots         <- pairs in flight (one HIP stream each); the reference has 1
 *   kernel_path   <- 0: fused gfx950 kernels (default, and the only path of libugsm.so); 1: one-stage-per-kernel
 *                    path kept for A/B parity checks (same results bit for bit) -- in libugsm_dev.so only since ABI 4,
 *                    libugsm.so answers UGSM_ERR_BAD_ARG
 *   march_*       <- which levels run the marching form of the cost kernel (same results bit for bit) */
typedef struct ugsm_config {
    int device;
    int levels;
    int fovea_levels;
    int slots;
    int kernel_path;
    int profile_events; /* slot 0 times its launches with HIP events carried in the dispatch (the kernel's own begin and end): 1 = the cost kernel only, 2 = every kernel class */
    int march_min_pixels; /* levels of at least this many pixels run K-cost as the marching kernel (one wave per strip of
                             columns, no LDS); 0 = default threshold (0.4 Mpx per launch; in effect 3 Mpx: the channel-parallel form
                             k_cost_march4 takes the levels below first); < 0 = never: round 1's LDS-tiled k_cost_split everywhere --
                             libugsm_dev.so only since ABI 6 (libugsm.so answers UGSM_ERR_BAD_ARG) */
    int march_np;         /* ignored since ABI 3 (kept for layout): the two-pixels-per-lane development form of the marching kernel
                             is no longer in the library (tools/kbench.hip instantiates it) */
    int march_rows;       /* tuning / tests: strip height of the marching kernel (0 = automatic) */
    int march_smooth;     /* ignored since ABI 6 (kept for layout): the marching K-smooth, bit-identical and measured slower than the
                             LDS-tiled one (docs/HISTORY.md), is no longer built */
    float early_exit_threshold; /* SURVEY 8f row f-4, OFF at 0 (default): when > 0, a level stops iterating as soon as the
                             confidence-weighted mean change of dx and of dy between two iterations is below it
                             (differenceIterations / weightedDifference, MatchGPULib.cpp:1323-1437 -- dead code in the
                             reference, whose results this option therefore leaves; one host round trip per iteration) */
    int small_max_pixels; /* levels of at most this many pixels run K-cost / K-smooth in their latency forms (ugsm_kernels_small.hip:
                             channel-parallel 16 x 12 tiles, one thread per pixel; same results bit for bit); 0 = default
                             threshold (0.15 Mpx for a call that has the chip to itself, 50 k pixels for one that shares it), < 0 = never */
    float lr_check_threshold; /* LR-consistency check, OFF at 0 (default).  Named by the north star; THE REFERENCE HAS NONE (no
                             right-to-left pass in MatchLib.cu / MatchGPULib.cpp), so any value > 0 leaves the reference's results:
                             full mode only (ugsm_match_full / ugsm_submit_full), the pair is matched a second time with the images
                             exchanged, and the confidence of every left pixel whose match (x + dx, y + dy) in the right-to-left
                             field does not point back within this many pixels, in x or in y, is set to 0 (dx, dy unchanged).
                             Doubles the matching work of a call. */
    int streams;          /* HIP streams the slots' work is dealt onto; 0 (default) = one per slot.  With fewer streams than slots, slot i
                             enqueues on the stream of slot i % streams: several pairs QUEUED per stream.  The chip runs four hardware
                             queues well and no more, so a throughput host uses streams = 4 and slots = 8: a
                             stream's next pair is already enqueued when the one before it ends.  ugsm_wait(slot) still waits for
                             that slot's pair only. */
    int batch;            /* pairs per ugsm_submit_*_batch call the context expects (1 .. UGSM_MAX_BATCH; 0 = 1), and the size of the calls the
                             queue forms (ugsm_enqueue_*).  A slot's buffers are sized for it on first use (slots x batch x 1.35 GB at 16 MP,
                             ugsm_context_device_bytes) so that no reallocation lands between calls of different sizes; if that much memory
                             cannot be had the slot is sized for the call at hand instead, and a call that fits by itself still runs
                             (UGSM_ERR_NOMEM only if even that fails; a refused call leaves no buffer behind).  ugsm_plan_level reports the
                             kernels of a call of this many pairs. */
    int stream_priority;  /* HIP priority of the slots' streams.  0 (default): a pool of their own -- slots 0-3 at the GREATEST priority,
                             4-7 at the least, the rest at the process default: HIP deals streams onto 4 hardware queues PER PRIORITY
                             LEVEL, and two streams on one queue run strictly one after the other, so slots that share the default
                             pool with the host application's streams (the null stream any hipMemcpy uses, for a start) end up three
                             to a queue: 129 instead of 165 pairs/s at 16 MP.  The price: the library's kernels
                             are scheduled ahead of the host application's other GPU work, and a second context in the process (or
                             another process on the card) that does the same shares those four queues.  1: every slot at the
                             process default -- opt out, for a host whose own GPU work must not be outranked, or that runs several
                             contexts; 2: every slot at the greatest priority; 3: every slot at the least. */
} ugsm_config;

void ugsm_default_config(ugsm_config *cfg);
int ugsm_abi_version(void);
/* 0 in libugsm.so; 1 in libugsm_dev.so, the same sources built with the development kernels (include/ugsm_dev.h) */
int ugsm_is_dev_library(void);
const char *ugsm_status_string(int status);

/* MatchGPULib::MatchGPULib(argc, argv), MatchGPULib.cpp:251-265.  Owns all device
 * memory; reused across calls (the reference allocates/frees per level and resets
 * the device per call, :400).  Buffers grow on demand to the largest size seen. */
int ugsm_create(const ugsm_config *cfg, ugsm_ctx **out);
void ugsm_destroy(ugsm_ctx *ctx);
const char *ugsm_last_error(const ugsm_ctx *ctx);

/* ---- geometry / schedule (pure host; usable without a GPU) ---------------------- */

/* matching(): w[i+1] = (int)(w[i]/1.41421356), MatchGPULib.cpp:1224-1228 */
int ugsm_level_dims(int W, int H, int levels, int *w, int *h);
/* matchlevel(): mi = i>5 ? 22 : 2(i+1), MatchGPULib.cpp:1741 */
int ugsm_level_iterations(int level);
/* matchlevel(): realSmoothtime 10 for the two finest levels else 5, :2257-2261 */
int ugsm_level_smooth_passes(int level);
/* matchlevel(): clamp annealing, :1673 + :2299-2306; out[mi] */
int ugsm_threshold_schedule(int mi, float *out);
/* initStack()/getFoveaWidth()/getFoveaHeight(), MatchGPULib.cpp:406-426,268-274 */
int ugsm_fovea_dims(int W, int H, int levels, int fovea_levels, int *fovW, int *fovH);
/* Sum over levels of iterations x pixels; fovea_levels==0 => full-resolution mode */
long long ugsm_pixel_iterations(int W, int H, int levels, int fovea_levels);

/* Which kernels a W x H level runs under `cfg` (NULL = defaults), for maintainers and the host tests; results never depend on it.
 * ONE thing besides the level's size decides: whether the call has the chip to itself (`alone` != 0) or shares it with other calls.
 * The library answers that per call from what is in flight when the call is submitted -- nothing unfinished on any other slot and no
 * pairs waiting behind it in the queue -- so the blocking entry points (ugsm_match_*: the node's service call and its one-at-a-time
 * topic path) are alone whatever cfg->slots says, and the calls of a burst are not.  A call alone gets every launch as SHORT as possible
 * (nothing else fills the CUs a launch leaves idle: the coarse-level latency kernels up to 0.15 Mpx on their smallest tiles, K-smooth tile
 * heights that fill whole rounds of workgroups, the right pyramid and the A planes on a side stream -- an idle neighbour slot's stream,
 * borrowed for the call; a one-stream context has one stream more for it); a call that shares the
 * chip gets every launch doing little redundant work (latency kernels up to 50 k pixels only, on 18 x 18 tiles; one stream).
 * cost_kernel / smooth_kernel: 0 = LDS-tiled (k_smooth_fused; as a cost kernel: k_cost_split, libugsm_dev.so only), 1 = marching
 * (k_cost_march), 2 = coarse-level latency form (k_cost_small / k_smooth_small), 3 = one kernel per reference stage (kernel_path 1),
 * 4 (cost_kernel only) = channel-parallel marching form (k_cost_march4);
 * smooth_rh: region height of k_smooth_small (18, 24 or 32; else 0); strip_rows: rows per strip of the marching K-cost (else 0);
 * seed_fused: 1 if the level's seeding rides on its first K-cost launch; smooth_tile_rows: height of k_smooth_fused's 112-column
 * tile where that tile is used (else 0).  With cfg->batch > 1 the plan is that of a call of cfg->batch pairs (ugsm_submit_*_batch): every
 * threshold is compared with what the LAUNCH holds, pairs_per_launch x the level. */
typedef struct ugsm_level_plan {
    int cost_kernel, smooth_kernel, smooth_rh, strip_rows, seed_fused, smooth_tile_rows, alone;
    int pairs_per_launch;  /* ABI 4: cfg->batch where a call of that many pairs runs this level as one launch for all of them, else 1 */
} ugsm_level_plan;
int ugsm_plan_level(const ugsm_config *cfg, int alone, int W, int H, ugsm_level_plan *out);

/* ---- the service path: host buffers in, host buffers out ------------------------ */

/* MatchGPULib::match(L, R, 0), MatchGPULib.cpp:303-403, as used by
 * GPU_matcher::disparitySrv (UG_GPU_matcher.cpp:645-658) and mainRoutine (:423-442).
 * rgbL/rgbR: rgb8 rows of `stride` bytes (cv::Mat::step, :318).  dispH/dispV/dispC:
 * caller-allocated H*W float32 planes (the 32FC1 payloads of dispH/dispV/dispC). */
int ugsm_match_full(ugsm_ctx *ctx, const uint8_t *rgbL, const uint8_t *rgbR, int W, int H,
                    int stride, float *dispH, float *dispV, float *dispC);

/* MatchGPULib::matchStack / matchStackPyramid, MatchGPULib.cpp:429-700, plus the
 * node's stack packing (UG_GPU_matcher.cpp:293-320 disparity stacks, :203-226
 * pyramid stacks).  stackH/V/C: (fovea_levels*fovH) x fovW float32, level 0 (finest)
 * first.  pyrL/pyrR (may be NULL): (fovea_levels*3*fovH) x fovW, rows ordered
 * [level][channel][row].  off_x/off_y: fovea-centre offset from the image centre in
 * level-0 pixels; (0,0) is the reference's centred fovea (:1173-1176). */
int ugsm_match_foveated(ugsm_ctx *ctx, const uint8_t *rgbL, const uint8_t *rgbR, int W, int H,
                        int stride, int off_x, int off_y, float *stackH, float *stackV,
                        float *stackC, float *pyrL, float *pyrR);

/* MatchGPULib::match(L, R, fov == 1) (MatchGPULib.cpp:354-360): foveated matching followed by
 * hierarchicalDisparity (:2589-2701) -- one full-resolution (dx, dy, conf) field whose centre window comes from
 * the fine fovea levels and whose periphery from the coarser ones.  Host buffers as ugsm_match_full.
 * The reference node never takes this path (UG_GPU_matcher.cpp:421-423,644-645); SURVEY 8f row f-3. */
int ugsm_match_foveated_full(ugsm_ctx *ctx, const uint8_t *rgbL, const uint8_t *rgbR, int W, int H,
                             int stride, int off_x, int off_y, float *outH, float *outV, float *outC);

/* ---- throughput path: device buffers, asynchronous, one slot = one stream ------- */

/* Same computation as ugsm_match_full on device-resident inputs/outputs.
 * d_out: 3 contiguous H*W planes (dx, dy, conf).  Returns after enqueueing. */
int ugsm_submit_full(ugsm_ctx *ctx, int slot, const uint8_t *d_rgbL, const uint8_t *d_rgbR,
                     int W, int H, int stride, float *d_out);
/* Same as ugsm_match_foveated on device buffers; d_stack: 3 x (F*fovH) x fovW. */
int ugsm_submit_foveated(ugsm_ctx *ctx, int slot, const uint8_t *d_rgbL, const uint8_t *d_rgbR,
                         int W, int H, int stride, int off_x, int off_y, float *d_stack,
                         float *d_pyrL, float *d_pyrR);
/* ugsm_match_full without the wait, on any slot: for a host that keeps several pairs in flight from PAGE-LOCKED memory (SURVEY 8d:
 * "end-to-end from pinned host memory").  Every buffer -- both images and the three result planes -- must be page-locked
 * (ugsm_host_alloc, hipHostMalloc or hipHostRegister), else UGSM_ERR_BAD_ARG; the uploads, the match and the three downloads are
 * enqueued on the slot's stream and the call returns; ugsm_wait(slot) before the planes are read or the slot is used again. */
int ugsm_submit_full_host(ugsm_ctx *ctx, int slot, const uint8_t *rgbL, const uint8_t *rgbR, int W, int H,
                          int stride, float *dispH, float *dispV, float *dispC);
/* The same for ugsm_match_foveated (pyrL / pyrR may be NULL; page-locked if given). */
int ugsm_submit_foveated_host(ugsm_ctx *ctx, int slot, const uint8_t *rgbL, const uint8_t *rgbR, int W, int H,
                              int stride, int off_x, int off_y, float *stackH, float *stackV,
                              float *stackC, float *pyrL, float *pyrR);
/* B pairs per call (round 4) -- BASELINE configs[4] is "a batch of 8 x 16 MP foveated pairs"; the reference's loop is 176 strictly
 * sequential iterations per pair (MatchGPULib.cpp:1741-1743), and on every level of 615 x 407 pixels and below each of them is a launch
 * that lasts as long as ONE tile's or strip's chain whatever the rest of the chip could do.  The n pairs of a batch (1 <= n <=
 * UGSM_MAX_BATCH, all W x H) march through the levels in lockstep on the slot's stream: every level of at most 9 Mpx is ONE launch
 * for all of them (a pair index in every kernel's grid), larger levels fill the chip pair by pair and are launched so.  Same
 * arithmetic, same results bit for bit as n single calls.  d_rgbL / d_rgbR / d_out (d_stack, d_pyrL, d_pyrR): HOST arrays of n
 * DEVICE pointers, buffers laid out as for ugsm_submit_full / ugsm_submit_foveated; off_x / off_y: n window offsets (NULL = centred);
 * d_pyrL / d_pyrR may be NULL.  The slot holds the whole batch: ugsm_wait(slot) waits for all n pairs.  Contexts with
 * early_exit_threshold or lr_check_threshold set (and kernel_path 1) run the pairs one after the other. */
int ugsm_submit_full_batch(ugsm_ctx *ctx, int slot, int n, const uint8_t *const *d_rgbL, const uint8_t *const *d_rgbR,
                           int W, int H, int stride, float *const *d_out);
int ugsm_submit_foveated_batch(ugsm_ctx *ctx, int slot, int n, const uint8_t *const *d_rgbL, const uint8_t *const *d_rgbR,
                               int W, int H, int stride, const int *off_x, const int *off_y, float *const *d_stack,
                               float *const *d_pyrL, float *const *d_pyrR);
/* The same from PAGE-LOCKED host memory (as ugsm_submit_full_host / ugsm_submit_foveated_host): the uploads of all n pairs, one batched
 * match and the downloads of all results are enqueued on the slot's stream and the call returns; ugsm_wait(slot) before the results are read
 * or the slot is used again.  Every buffer page-locked, else UGSM_ERR_BAD_ARG.  (No pyramid stacks in the foveated form.) */
int ugsm_submit_full_batch_host(ugsm_ctx *ctx, int slot, int n, const uint8_t *const *rgbL, const uint8_t *const *rgbR,
                                int W, int H, int stride, float *const *dispH, float *const *dispV, float *const *dispC);
int ugsm_submit_foveated_batch_host(ugsm_ctx *ctx, int slot, int n, const uint8_t *const *rgbL, const uint8_t *const *rgbR,
                                    int W, int H, int stride, const int *off_x, const int *off_y, float *const *stackH,
                                    float *const *stackV, float *const *stackC);
int ugsm_wait(ugsm_ctx *ctx, int slot);
/* ugsm_wait on every slot, in order; every slot is waited for whatever the ones before it answered, the first failure is the one returned. */
int ugsm_wait_all(ugsm_ctx *ctx);
/* The HIP stream `slot` enqueues on (a hipStream_t, returned as a plain pointer): lets a host that owns other streams -- the
 * RCCL collective of the fovea shard -- order them after the slot's work ON THE DEVICE (record an event on this stream, make the
 * other stream wait for it) instead of blocking in ugsm_wait.  The stream stays owned by the context. */
int ugsm_slot_stream(ugsm_ctx *ctx, int slot, void **hip_stream);
/* Non-blocking ugsm_wait: UGSM_OK when everything enqueued on `slot` has finished (its launch statistics are harvested, as by
 * ugsm_wait), UGSM_PENDING while it has not. */
int ugsm_poll(ugsm_ctx *ctx, int slot);

/* ---- the queue: the library owns the slots (round 5) -------------------------------------------------------------------------------
 *
 * What a host needs for THROUGHPUT is not "a call of n pairs on slot s" but "here is another pair; tell me when it is done": which
 * pairs share a call, which slot takes it and when a slot is free again is the library's business.  (Rounds 3-4 kept that logic in the
 * benchmark harness -- bench.py's plan_calls / run / submit -- where no user of this header could reach it.)  The reference node's
 * topic path (UG_GPU_matcher.cpp:126-185,414-494: one blocking match() per synchronised image pair inside a single-threaded
 * ros::spin, :749-752) becomes enqueue-on-arrival + publish-on-completion with `frames_in_flight` pairs outstanding
 * (ros/UG_GPU_matcher_ugsm.cpp, ug_stereomatcher_amd/service.py); results are reported strictly in the order the pairs were enqueued.
 *
 * ugsm_enqueue_* appends one pair to the context's backlog and returns; it never waits for that pair.  Calls are formed from the
 * backlog by one rule ("batch what has piled up"):
 *   - a call goes out as soon as `target` pairs of one kind (mode, memory kind, W x H, stride) wait, where target = ugsm_config.batch,
 *     except for the first `slots` calls of a burst (after ugsm_create, and after every flush), which are staggered -- call c takes
 *     ceil(batch (c + 2) / (slots + 1)) pairs (4, 5, 7, 8 for batch 8 on four slots) so that the slots do not march through the pyramid
 *     levels in phase from a drained pipe ("The queue");
 *   - ugsm_flush, or a blocking ugsm_next_done, declares that nothing more is coming for now: whatever waits goes out in calls of at
 *     most `target` pairs as slots come free, without waiting for a call to fill.  A host that wants every frame started at once calls
 *     ugsm_flush after every ugsm_enqueue_*: calls then hold one pair while slots are free and grow by themselves under load;
 *   - a pair of another kind than the ones waiting sends those out first (a call holds pairs of one kind).
 * Slots are used in rotation; a free slot is preferred, else the call waits (inside ugsm_enqueue_*: back-pressure) for the slot that
 * holds the oldest call.  At most (slots + 1) x batch pairs are outstanding -- enqueued and not yet reported by ugsm_next_done -- at
 * any time: a host that recycles (slots + 1) x batch result buffers in enqueue order never overwrites a result it has not been told
 * about.  A host that lets completions pile up unfetched until that many are outstanding gets UGSM_ERR_STATE from ugsm_enqueue_*
 * (nothing is enqueued; fetch with ugsm_next_done and try again).
 * The library makes progress only inside its own entry points (no thread of its own): calls go out and completions are noticed during
 * ugsm_enqueue_*, ugsm_flush and ugsm_next_done.  While pairs are outstanding the slots belong to the queue: the slot-level entry points
 * (ugsm_submit_*, ugsm_match_*, ugsm_stage_*) answer UGSM_ERR_STATE until every pair has been reported.
 * Results are identical, bit for bit, to single calls of ugsm_submit_full / ugsm_submit_foveated on the same inputs, however the pairs
 * were grouped. */
typedef struct ugsm_completion {
    uint64_t tag;          /* the host's name for the pair (ugsm_enqueue_*) */
    int status;            /* UGSM_OK, or the status of the library call the pair went out in */
    int slot;              /* slot that ran it */
    int call_pairs;        /* pairs of that call ... */
    int reserved;
    long long call_index;  /* ... and its running number since ugsm_create */
    long long done_ns;     /* CLOCK_MONOTONIC, nanoseconds, when the library noticed the call complete */
    float *result[5];      /* ugsm_enqueue_*_managed only (else NULL): page-locked planes owned by the library, valid until the NEXT
                              ugsm_next_done on this context -- full mode: dispH, dispV, dispC (H x W each); foveated: stackH, stackV, stackC
                              ((F fovH) x fovW each), then the L and R pyramid stacks ((F 3 fovH) x fovW) if they were asked for */
} ugsm_completion;

/* Return value of every ugsm_enqueue_*: UGSM_OK = the pair is ACCEPTED -- it will be reported by ugsm_next_done exactly once, whatever
 * happens to the call it goes out in; anything else = the pair is REJECTED and nothing was enqueued (bad arguments, buffers that are not
 * page-locked, UGSM_ERR_STATE when (slots + 1) x batch pairs are outstanding, UGSM_ERR_NOMEM for the library's own bookkeeping or staging).
 * A library call that fails -- e.g. UGSM_ERR_NOMEM when the slot's buffers cannot be had -- is reported through ugsm_completion.status of
 * each of its pairs and nowhere else: the call an enqueue happens to send may hold OTHER pairs than the one just appended.  A failed call's
 * pairs are reported only after whatever the call did put on the slot's stream has drained, so a completion always means that the pair's
 * input and result buffers are no longer in use.
 * Device buffers (as ugsm_submit_full / ugsm_submit_foveated; d_pyrL / d_pyrR may be NULL).  Inputs and outputs must stay valid and
 * untouched until the pair's tag has been reported by ugsm_next_done. */
int ugsm_enqueue_full(ugsm_ctx *ctx, const uint8_t *d_rgbL, const uint8_t *d_rgbR, int W, int H, int stride, float *d_out, uint64_t tag);
int ugsm_enqueue_foveated(ugsm_ctx *ctx, const uint8_t *d_rgbL, const uint8_t *d_rgbR, int W, int H, int stride, int off_x, int off_y,
                          float *d_stack, float *d_pyrL, float *d_pyrR, uint64_t tag);
/* PAGE-LOCKED host buffers (as ugsm_submit_full_host / ugsm_submit_foveated_host; UGSM_ERR_BAD_ARG if any is not): uploads, match
 * and downloads are enqueued with the call the pair goes out in.  Same lifetime rule. */
int ugsm_enqueue_full_host(ugsm_ctx *ctx, const uint8_t *rgbL, const uint8_t *rgbR, int W, int H, int stride, float *dispH, float *dispV,
                           float *dispC, uint64_t tag);
int ugsm_enqueue_foveated_host(ugsm_ctx *ctx, const uint8_t *rgbL, const uint8_t *rgbR, int W, int H, int stride, int off_x, int off_y,
                               float *stackH, float *stackV, float *stackC, float *pyrL, float *pyrR, uint64_t tag);
/* Any host memory in, library-owned results out: the images are copied into page-locked staging memory of the context BEFORE the call
 * returns (the caller's buffers -- a ROS message's payload -- may be freed at once), the results land in page-locked planes the library
 * lends to the host through ugsm_completion.result.  This is what the node's topic path uses: no page-locked memory to manage, no
 * result planes to allocate per frame (the reference mallocs and frees 193 MB per 16 MP frame, UG_GPU_matcher.cpp:414-418,487-489).
 * want_pyramids: also return the L / R fovea pyramid stacks (matchStackPyramid, MatchGPULib.cpp:534-700). */
int ugsm_enqueue_full_managed(ugsm_ctx *ctx, const uint8_t *rgbL, const uint8_t *rgbR, int W, int H, int stride, uint64_t tag);
int ugsm_enqueue_foveated_managed(ugsm_ctx *ctx, const uint8_t *rgbL, const uint8_t *rgbR, int W, int H, int stride, int off_x, int off_y,
                                  int want_pyramids, uint64_t tag);
/* "Nothing more is coming for now": every pair waiting at this moment goes out as slots allow, without waiting for its call to fill.
 * Never blocks: pairs no slot is free for stay queued and go out inside a later ugsm_enqueue_* / ugsm_next_done. */
int ugsm_flush(ugsm_ctx *ctx);
/* The oldest pair not yet reported.  UGSM_OK: *out filled (out->status tells how its call went).  block = 0: UGSM_PENDING if that pair
 * has not finished (or has not gone out yet), UGSM_EMPTY if there is none.  block != 0: implies ugsm_flush, waits for the pair;
 * UGSM_EMPTY only if nothing is outstanding. */
int ugsm_next_done(ugsm_ctx *ctx, ugsm_completion *out, int block);
/* Pairs waiting in the backlog, pairs in calls that are in flight (as far as the library has noticed), completions not yet fetched. */
int ugsm_queue_depth(ugsm_ctx *ctx, int *waiting, int *in_flight, int *unreported);
/* Host only: the calls the queue of a context created with `cfg` (NULL = defaults) forms from a burst of n_pairs pairs of one kind
 * enqueued back to back from idle and then flushed -- sizes[0 .. return value) (cap entries at most are written; the return value is
 * the number of calls, or -1 for bad arguments).  20 pairs, batch 8, four slots: 4, 5, 7, 4. */
int ugsm_queue_plan(const ugsm_config *cfg, int n_pairs, int *sizes, int cap);

/* Fovea sharding over several GPUs (north-star; no reference counterpart: the
 * reference has one centred fovea on one GPU).  coarse: pyramids + levels
 * top..F-1 on the full frame; d_state receives level F-1's (dx,dy,conf),
 * 3*fovH*fovW floats -- the 3 MB object broadcast over RCCL.  fine: levels
 * F-2..0 for the window at (off_x, off_y) from a (possibly received) d_state;
 * needs the pair's WHOLE pyramids in the slot, i.e. ugsm_submit_pyramids first (UGSM_ERR_STATE otherwise).  The one-shot foveated
 * calls (ugsm_match_foveated, ugsm_submit_foveated[_batch|_host]) know their windows when they build the pyramids and store level 0
 * only inside them (CreateFoveatedPyramid crops after a full build, MatchGPULib.cpp:1128-1190; here 193 MB per 16 MP image are never
 * written), so their pyramids do NOT serve a later ugsm_submit_fovea_fine at another offset. */
int ugsm_submit_pyramids(ugsm_ctx *ctx, int slot, const uint8_t *d_rgbL, const uint8_t *d_rgbR,
                         int W, int H, int stride);
int ugsm_submit_fovea_coarse(ugsm_ctx *ctx, int slot, float *d_state);
int ugsm_submit_fovea_fine(ugsm_ctx *ctx, int slot, const float *d_state, int off_x, int off_y,
                           float *d_stack);

/* ---- the fovea shard with its exchange inside the library (round 5): RCCL on the slot's own stream ---------------------------------
 *
 * One process (or one context) per GPU; the contexts of a shard form ONE RCCL communicator.  ugsm_submit_fovea_shard enqueues, on the
 * slot's stream and nowhere else: the pair's pyramids; on rank `src_rank` the coarse full-frame levels top .. F-1
 * (ugsm_submit_fovea_coarse); ncclBroadcast of level F-1's (dx, dy, conf) -- 3 x fovH x fovW floats, 3.0 MB at 16 MP -- from src_rank;
 * the fine levels F-2 .. 0 of THIS rank's window (off_x, off_y) into d_stack (ugsm_submit_fovea_fine).  No event, no second stream,
 * no host synchronisation: stream order is the whole protocol, and the call returns after enqueueing (ugsm_wait / ugsm_poll as usual).
 * Every rank must make the same sequence of shard calls on the same slot numbers (collectives match by order).  With the centre window
 * the result equals ugsm_submit_foveated's, bit for bit.  The reference has one centred fovea on one GPU (MatchGPULib.cpp:1173-1176),
 * seeded from level F-1 (:1230-1240, :1283-1293); the window offset and the shard are this build's (BASELINE.json north_star).
 * librccl.so.1 is loaded on first use (dlopen: a host that never shards does not pay for a 570 MB library); UGSM_ERR_NO_DEVICE when it
 * cannot be found, UGSM_ERR_DEVICE + ugsm_last_error for RCCL failures.
 *
 * WHEN A RANK FAILS (ABI 6).  A collective is a promise to the other ranks, so the library keeps it whatever happens locally:
 *   - what every rank refuses alike (bad arguments, bad geometry, a context whose queue is busy) is refused BEFORE anything is enqueued, on
 *     every rank, and no collective goes out;
 *   - a rank whose pyramids or coarse phase fail afterwards (UGSM_ERR_NOMEM, UGSM_ERR_DEVICE) STILL takes part in the broadcast, then returns
 *     its status from ugsm_submit_fovea_shard.  The state the source sends carries a status word: where the source failed, every other rank's
 *     ugsm_wait / ugsm_poll on that slot answers UGSM_ERR_PEER (ugsm_last_error names the rank and its status; d_stack is not valid).  A
 *     non-source rank that fails harms nobody: the others' results are valid.  The communicator stays usable: the next step runs;
 *   - a rank that never reaches the exchange (a crashed process; a rank that could not even allocate the 3 MB state buffer) cannot be told
 *     from a slow one, so there is a deadline: ugsm_shard_set_timeout(ctx, ms).  A step still unfinished `ms` after its submission makes
 *     ugsm_wait abort the communicator (ncclCommAbort), and answer UGSM_ERR_PEER; every later shard call answers UGSM_ERR_STATE until the host
 *     has called ugsm_shard_finalize and ugsm_shard_init again (on every surviving rank, with a new id).  0 (default): no deadline.
 * ugsm_shard_gather is a collective like the others: after a failed step every rank still makes the gather call its protocol has. */
#define UGSM_SHARD_ID_BYTES 128  /* sizeof(ncclUniqueId) */
/* Rank 0 makes an id (ncclGetUniqueId) and hands its 128 bytes to the other ranks by whatever means the host has (a ROS parameter, a
 * file, MPI, torch.distributed's store): out-of-band, once. */
int ugsm_shard_unique_id(void *id128);
/* One process per GPU: joins the communicator `id128` names as rank `rank` of `world` (ncclCommInitRank).  Collective: returns when
 * all ranks have joined.  A context belongs to at most one communicator (UGSM_ERR_STATE otherwise). */
int ugsm_shard_init(ugsm_ctx *ctx, const void *id128, int rank, int world);
/* One process that owns several GPUs: ctxs[0 .. n) (each created on its own device) become ranks 0 .. n-1 of one communicator
 * (ncclCommInitAll).  The shard calls of the n contexts must then come from ONE HOST THREAD PER CONTEXT (a collective waits for its
 * peers; the library cannot wrap a step in ncclGroupStart / ncclGroupEnd because a grouped broadcast is only launched at the group's
 * end, i.e. after the fine phase that must follow it on the stream). */
int ugsm_shard_init_all(ugsm_ctx *const *ctxs, int n);
int ugsm_shard_rank(const ugsm_ctx *ctx, int *rank, int *world);
/* Number of ranks RCCL itself counts: an ncclAllReduce(sum) of one 1 per rank on slot 0's stream, waited for.  = world when the
 * communicator really spans that many ranks. */
int ugsm_shard_count_ranks(ugsm_ctx *ctx, int *ranks);
int ugsm_submit_fovea_shard(ugsm_ctx *ctx, int slot, const uint8_t *d_rgbL, const uint8_t *d_rgbR, int W, int H, int stride, int off_x,
                            int off_y, float *d_stack, int src_rank);
/* Deadline of a shard step, in milliseconds from its submission (0 = none); see "when a rank fails" above. */
int ugsm_shard_set_timeout(ugsm_ctx *ctx, long long milliseconds);
/* Optional: every rank's fovea stack (3 x (F fovH) x fovW floats, 21 MB at 16 MP) to the one consumer rank -- ncclSend on the others,
 * ncclRecv x (world - 1) + one device copy on dst_rank, on the slot's stream (ordered after the slot's shard call).  d_all (dst_rank
 * only, else NULL): world x stack_floats. */
int ugsm_shard_gather(ugsm_ctx *ctx, int slot, const float *d_stack, long long stack_floats, float *d_all, int dst_rank);
/* Leaves the communicator (ncclCommDestroy); ugsm_destroy does it too. */
int ugsm_shard_finalize(ugsm_ctx *ctx);

/* ---- next row: triangulation of the full-resolution disparity ---------- */

/* CdynamicCalibration::get3DPoint, non-foveated branch (src/pointcloud/getPointCloud.cpp:886-949),
 * for every pixel of the match result instead of the node's scalar host loops (:640-660, :778).
 * d_dispx/d_dispy: H*W device planes (e.g. planes 0 and 1 of ugsm_submit_full's d_out); P1, P2:
 * the 3x4 projection matrices of calL.xml/calR.xml, row-major doubles on the HOST; d_xyz: 3 device
 * planes X, Y, Z.  Enqueued on `slot`'s stream (ordered after a submit on the same slot). */
int ugsm_triangulate(ugsm_ctx *ctx, int slot, const float *d_dispx, const float *d_dispy, int W, int H,
                     const double *P1, const double *P2, float *d_xyz);

/* Row f-1, foveated branch.  Where level `src_level` of the fovea stack sits in level `dest_level` of the
 * full pyramid and the coordinate scale between them: CdynamicCalibration::left_marginOf_in /
 * upper_marginOf_in / mapXcoord (src/pointcloud/getPointCloud.cpp:387-484).  Host only. */
int ugsm_fovea_mapping(int W, int H, int src_level, int dest_level, int *left_margin, int *upper_margin,
                       float *scale);
/* get3DPoint with foveated == 1 (getPointCloud.cpp:892-903) for every pixel of level `src_level` of the
 * (F*fovH) x fovW device stacks d_stackx / d_stacky (the layout ugsm_match_foveated returns);
 * d_xyz: X, Y, Z planes of fovH x fovW floats.  Asynchronous on `slot`, like ugsm_triangulate. */
int ugsm_triangulate_fovea(ugsm_ctx *ctx, int slot, const float *d_stackx, const float *d_stacky, int fovW,
                           int fovH, int src_level, int left_margin, int upper_margin, float scale,
                           const double *P1, const double *P2, float *d_xyz);

/* Row f-3: MatchGPULib::hierarchicalDisparity (MatchGPULib.cpp:2589-2701, kernel MatchLib.cu:435-462):
 * one full-resolution (dx, dy, conf) field from the foveated stacks -- the coarsest fovea level (the whole
 * frame) upsampled level by level (x SCALE, every channel), each finer fovea pasted at its window.
 * d_stackH/V/C: device, (fovea_levels*fovH) x fovW each; d_out3: device, 3 planes W x H.
 * off_x/off_y as passed to ugsm_match_foveated.  Asynchronous on `slot`; ugsm_wait(ctx, slot) to finish. */
int ugsm_reconstruct_full(ugsm_ctx *ctx, int slot, const float *d_stackH, const float *d_stackV,
                          const float *d_stackC, int W, int H, int off_x, int off_y, float *d_out3);

/* ---- stage-level entry points (tests only; device pointers; synchronous) -------- */
/* (the probes of the kernels' exact arithmetic shortcuts -- ugsm_stage_poly_probe, ugsm_stage_div3_probe, ugsm_stage_div_probe -- are
 * declared in include/ugsm_dev.h and exported by libugsm_dev.so only) */

/* CreatePyramidFromImage, MatchGPULib.cpp:1033-1125: builds the pyramid of one rgb8
 * image in `slot`'s left pyramid and copies level `level` (3 planes) to d_out3. */
int ugsm_stage_pyramid(ugsm_ctx *ctx, const uint8_t *d_rgb, int W, int H, int stride, int level,
                       float *d_out3);
/* matchlevel, MatchGPULib.cpp:1662-2489: iterations m_from..m_to of a level with mi
 * iterations and S smoothing passes.  d_dbg8 (may be NULL): 5 Q planes + dx',dy',kappa
 * before smoothing, of the last iteration run (kernel_path 1 only). */
int ugsm_stage_iterate(ugsm_ctx *ctx, const float *d_L3, const float *d_R3, float *d_d3, int W,
                       int H, int mi, int S, int is_top, int m_from, int m_to, float *d_dbg8);
/* subsampleDisp / foveatedsubsampleDisp, MatchGPULib.cpp:1526-1655 */
int ugsm_stage_seed(ugsm_ctx *ctx, const float *d_src3, int W, int H, float *d_dst3, int W2, int H2,
                    int Wup, int Hup, int crop_x, int crop_y);
/* S smoothing passes (+ box if do_box), MatchGPULib.cpp:2257-2412, in place */
int ugsm_stage_smooth(ugsm_ctx *ctx, float *d_d3, int W, int H, int passes, int do_box);

/* Row f-4: weightedDifference (MatchGPULib.cpp:1336-1437) of two device (dx, dy, conf) fields, weights = the new field's
 * conf: out2[0] = dx, out2[1] = dy (host).  Fixed-order binary64 sums, identical to the CPU restatement. */
int ugsm_stage_weighted_difference(ugsm_ctx *ctx, const float *d_new3, const float *d_old3, int W, int H, float *out2);
/* The LR-consistency check (ugsm_config.lr_check_threshold; no reference counterpart) on two device (dx, dy, conf) fields: zeroes
 * d_left3's confidence where d_right3 does not point back within tau; *marked (host, may be NULL) = number of pixels marked. */
int ugsm_stage_lr_check(ugsm_ctx *ctx, float *d_left3, const float *d_right3, int W, int H, float tau, long long *marked);
/* Pixels the LR check of the last full-mode call on `slot` marked (-1: the call ran without the check; a batched call, which such a
 * context runs pair by pair: of its last pair).  Valid after ugsm_wait. */
long long ugsm_last_lr_marked(ugsm_ctx *ctx, int slot);
/* Iterations each level of the last call on `slot` actually ran (early_exit_threshold > 0 can stop a level early);
 * per_level[UGSM_MAX_LEVELS], -1 for levels not run.  ugsm_stage_iterate records its count at index 0. */
int ugsm_last_iterations(ugsm_ctx *ctx, int slot, int *per_level);

/* ---- instrumentation ------------------------------------------------------------ */

typedef struct ugsm_kernel_stat {
    char name[48];
    int level;              /* pyramid level the launches belong to; -1: none (stage entry points, copies) */
    int reserved;
    long long launches;
    double total_ms;        /* sum of HIP-event durations (profile_events, slot 0) */
    double pixel_launches;  /* sum over launches of pixels processed */
} ugsm_kernel_stat;

/* One entry per (kernel, pyramid level) with at least one harvested launch (launches are harvested by ugsm_wait).
 * Fills up to `cap` entries; returns the number of entries there are. */
int ugsm_get_kernel_stats(ugsm_ctx *ctx, ugsm_kernel_stat *out, int cap);
int ugsm_reset_kernel_stats(ugsm_ctx *ctx);
/* Changes ugsm_config.profile_events of a live context (0 off, 1 cost kernels, 2 every kernel); takes effect for
 * launches enqueued afterwards.  bench.py times its throughput region with events off and reads kernel durations
 * from a separate single-pair pass. */
int ugsm_set_profile_events(ugsm_ctx *ctx, int mode);

/* Device memory the context holds right now in its slots' buffers (pyramids, fields, staging; they grow on demand and are kept):
 * about 1.35 GB per pair of a call and slot at 16 MP, i.e. slots x batch x 1.35 GB once every slot has run a full-size call
 * (four slots, batch 8: 43 GB; batch 16: 86 GB of the 288).  -1 for a null context. */
long long ugsm_context_device_bytes(const ugsm_ctx *ctx);

/* Device-memory helpers so a C/C++ host (the ROS node) needs no HIP headers. */
int ugsm_dev_alloc(ugsm_ctx *ctx, void **d_ptr, long long bytes);
int ugsm_dev_free(ugsm_ctx *ctx, void *d_ptr);
/* Page-locked host memory (optional): images and result planes placed here make the host<->device copies
 * of ugsm_match_full / ugsm_match_foveated plain DMA (the reference's node mallocs and frees its planes,
 * UG_GPU_matcher.cpp:414-418; a node that adopts these two calls keeps them for the life of the context). */
int ugsm_host_alloc(ugsm_ctx *ctx, void **h_ptr, long long bytes);
int ugsm_host_free(ugsm_ctx *ctx, void *h_ptr);
int ugsm_copy_to_device(ugsm_ctx *ctx, void *d_dst, const void *h_src, long long bytes);
int ugsm_copy_to_host(ugsm_ctx *ctx, void *h_dst, const void *d_src, long long bytes);

#if defined(__GNUC__)
#pragma GCC visibility pop
#endif
#ifdef __cplusplus
}
#endif
#endif /* UGSM_H */
