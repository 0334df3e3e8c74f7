// ugsm_runtime.cpp -- host runtime + C-ABI of libugsm.so.
//
// Replaces the host orchestration of /root/reference/src/gpu_matcher/MatchGPULib.cpp:
// a persistent context owns every device buffer (the reference mallocs/frees ~23 buffers
// per level, re-uploads the image planes every iteration and resets the device per call),
// the pyramid and the disparity state never leave HBM, and each pair in flight has its own
// HIP stream ("slot") so the launch-bound coarse levels of one pair overlap the
// bandwidth/VALU-bound fine levels of another.
#include "../../include/ugsm.h"
#ifdef UGSM_DEV_LIB
#include "../../include/ugsm_dev.h"
#endif
#include "ugsm_device.hpp"
#include "ugsm_internal.hpp"
#include "ugsm_launch.hpp"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <condition_variable>
#include <functional>
#include <memory>
#include <mutex>
#include <string>
#include <system_error>
#include <thread>
#include <unordered_map>
#include <vector>

using namespace ugsm;

namespace {

constexpr double kScale = 1.41421356;  // MatchLib_common.h:15

enum KClass { KC_COST = 0, KC_SMOOTH, KC_SQBLUR, KC_PYR, KC_SEED, KC_WARP, KC_BOX, KC_MISC, KC_COST_MARCH, KC_PYR_BASE, KC_COST_SMALL, KC_SMOOTH_SMALL, KC_COST_MARCH4, KC_COUNT };
const char *kClassName[2][KC_COUNT] = {
    {"k_cost_split", "k_smooth_fused", "k_sqblur_tiled", "k_blur_decimate_tiled", "k_seed", "-", "-", "misc", "k_cost_march", "k_pyr_base", "k_cost_small",
     "k_smooth_small", "k_cost_march4"},
    {"k_cost_ref", "k_smooth_pass", "k_sqblur_clamp", "k_blur_decimate", "k_seed", "k_warp", "k_box", "misc", "-", "-", "-", "-", "-"}};
constexpr int kNoLevel = UGSM_MAX_LEVELS;  // stats cell of launches that belong to no pyramid level
struct StatCell {
    long long launches = 0;
    double total_ms = 0, pixel_launches = 0;
};

struct EvRec {
    int kclass;
    int level;
    double pixels;
    hipEvent_t a, b;
};

struct Slot {
    hipStream_t st = nullptr;
    int W = 0, H = 0, levels = 0;
    int w[UGSM_MAX_LEVELS], h[UGSM_MAX_LEVELS];
    size_t off[UGSM_MAX_LEVELS];  // float offset of level i inside pyrL/pyrR
    float *pyrL = nullptr, *pyrR = nullptr;
    size_t pyr_cap = 0;  // floats
    // Batch (round 4): the call in flight matches `nb` pairs of one size in lockstep.  Every per-pair buffer of the slot holds nb copies:
    // pair b's pyramids at pyrL / pyrR + b * pyr_stride, its level buffers (A, d0, d1) at + b * lvl_stride, its range word at range_bad + b.
    int nb = 1;
    size_t pyr_stride = 0, lvl_stride = 0;  // floats
    float *A = nullptr, *Rw = nullptr, *B = nullptr, *d0 = nullptr, *d1 = nullptr;
    // Side stream of the slot (round 3): the right image's upload and pyramid, and then A = G_clamp * L^2 of every full-frame level
    // (MatchGPULib.cpp:1866-1875, once per level), run there beside the left pyramid and the coarse levels' iterations -- launches
    // that last microseconds and depend on nothing but the left pyramid.  The main stream joins it through events: ev_R before the
    // first level, ev_A[i] before level i's first K-cost launch, so a wait on the main stream still covers everything.
    // WHICH stream that is (round 6): a call uses a side stream only when it has the chip to itself (`alone`), i.e. when every other slot
    // is idle -- so with two or more streams in the context the side stream of a slot is A NEIGHBOUR SLOT'S OWN STREAM, borrowed for the call:
    // a context holds no stream that calls in flight leave idle, and a lone call's two streams sit on two hardware queues whatever way HIP
    // deals streams onto queues.  A context with one stream gets one stream more (owns_st2).
    hipStream_t st2 = nullptr;
    bool owns_st2 = false;
    // set where a call puts work on st2, cleared where a ugsm_submit_* returns success (every piece of that work has then been joined by
    // the main stream).  Still set at ugsm_wait / ugsm_poll = the submit failed part-way: the side stream is waited for as well, so that a
    // reported call never leaves work behind that reads the caller's images.
    bool forked = false;
    int prio = 0;                  // HIP priority of `st` (a side stream of the slot's own is created in the same pool)
    bool owns_st = true;           // false: `st` is the stream of slot (index % streams): several slots queue their pairs on one stream
    hipEvent_t ev_done = nullptr;  // recorded at the end of every ugsm_submit_*: what ugsm_wait waits for when the stream is shared
    bool done_recorded = false;
    bool busy = false;   // a call has been enqueued here and neither ugsm_wait nor ugsm_poll has seen it finish (get_slot / slot_idle)
    bool alone = true;   // the call in flight had the chip to itself when it was submitted (call_alone): what the kernel choices follow
    hipEvent_t ev_in = nullptr, ev_L = nullptr, ev_R = nullptr, ev_A[UGSM_MAX_LEVELS] = {};
    float *lr = nullptr;     // LR check: the right-to-left field of level 0 (3 planes) + one 8-byte counter behind it
    size_t lr_cap = 0;
    unsigned long long *lr_host = nullptr;  // page-locked copy of the counter
    bool lr_ran = false;
    float *Apyr = nullptr;   // A of level i at Apyr + off[i] (same layout as the pyramids)
    size_t apyr_cap = 0;
    int a_from = -1;         // levels a_from .. top have their A in Apyr (ev_A recorded); -1: none (A is computed in line)
    size_t lvl_cap = 0;  // floats per 3-plane level buffer
    uint8_t *rgbL = nullptr, *rgbR = nullptr;
    size_t rgb_cap = 0;
    float *d2 = nullptr;            // third (dx,dy,conf) buffer: only with early_exit_threshold > 0 (the previous iteration's field must survive)
    size_t d2_cap = 0;
    double *wd_rows = nullptr;      // row f-4 scratch: 3 doubles per row + 3 totals
    size_t wd_cap = 0;
    double *wd_host = nullptr;      // page-locked, 3 doubles
    int iters_run[UGSM_MAX_LEVELS];
    unsigned *range_bad = nullptr;  // device word: 0 while every pyramid value of the pair in this slot passed range_ok (ugsm_exact.hpp)
    float *hout = nullptr;  // device staging for host-API outputs
    size_t hout_cap = 0;
    float *hpin = nullptr;  // page-locked host staging for host-API outputs that land in pageable caller memory
    size_t hpin_cap = 0;
    hipEvent_t out_ev[12] = {};  // one per piece of the pageable copy-out (copy_out_planes)
    // out of memory for the configured batch (prepare_slot): the capacity that failed, for which geometry, and the context's release
    // count then -- the attempt is not repeated until something has been released or a call needs that much
    int nomem_pairs = 0;
    size_t nomem_tot = 0;
    long long nomem_epoch = 0;
    bool have_pyr = false;
    bool have_coarse = false;
    bool range_known = false;
    int cur_level = kNoLevel;  // pyramid level the launches being enqueued belong to (statistics only)  // range_bad describes the images the next run_level works on
    std::vector<EvRec> pending;
    std::vector<hipEvent_t> pool;
};

// A persistent team of host threads for the service path's page touching and staging copies (the first version
// created ~29 std::threads per call; a failed creation -- EAGAIN under a thread or cgroup limit -- threw through an extern "C"
// entry point with joinable threads on the stack, i.e. std::terminate in the caller's process).  Workers are created once, on the
// first service call; creation failures are caught and the team simply stays smaller (down to the calling thread alone).
class HostTeam {
public:
    explicit HostTeam(unsigned want)
    {
        for (unsigned t = 1; t < want; t++) {
            try {
                workers_.emplace_back([this, t] { loop(t); });
            } catch (const std::system_error &) {
                break;  // no more threads to be had: run with what there is
            }
        }
    }
    ~HostTeam()
    {
        {
            std::lock_guard<std::mutex> lk(m_);
            stop_ = true;
            gen_++;
        }
        cv_.notify_all();
        for (std::thread &t : workers_) t.join();
    }
    unsigned size() const { return (unsigned)workers_.size() + 1; }
    // f(member, members) on every member; the caller is member 0.  Returns when all are done.  f must not throw.
    void run(const std::function<void(unsigned, unsigned)> &f)
    {
        const unsigned n = size();
        if (n == 1) {
            f(0, 1);
            return;
        }
        {
            std::lock_guard<std::mutex> lk(m_);
            job_ = &f;
            pending_ = n - 1;
            gen_++;
        }
        cv_.notify_all();
        f(0, n);
        std::unique_lock<std::mutex> lk(m_);
        done_.wait(lk, [this] { return pending_ == 0; });
        job_ = nullptr;
    }

private:
    void loop(unsigned me)
    {
        unsigned seen = 0;
        for (;;) {
            const std::function<void(unsigned, unsigned)> *job;
            {
                std::unique_lock<std::mutex> lk(m_);
                cv_.wait(lk, [&] { return gen_ != seen; });
                seen = gen_;
                if (stop_) return;
                job = job_;
            }
            if (job) (*job)(me, size());
            {
                std::lock_guard<std::mutex> lk(m_);
                if (--pending_ == 0) done_.notify_one();
            }
        }
    }
    std::vector<std::thread> workers_;
    std::mutex m_;
    std::condition_variable cv_, done_;
    const std::function<void(unsigned, unsigned)> *job_ = nullptr;
    unsigned gen_ = 0, pending_ = 0;
    bool stop_ = false;
};

}  // namespace

struct ugsm_ctx {
    std::unique_ptr<HostTeam> team;  // service path only; created on first use (host_team below)
    ugsm_config cfg;
    std::vector<Slot> slots;
    std::string err;
    StatCell cells[KC_COUNT][UGSM_MAX_LEVELS + 1];
    int fuse_seed = 1, small_mask = 3, small_rh_force = 0;  // development overrides (apply_dev_env); the defaults are the product
    int two_streams = 0;  // a call alone forks onto a side stream (every context; UGSM_TWO_STREAMS=0 under UGSM_DEV=1: never)
    int march4_lo = -1, march4_hi = -1;  // development override of k_cost_march4's pixel range (use_march4; -1 = by the mode; 0, 0 = never)
    int force_alone = -1;  // development override of call_alone(): 1 = every call is taken to be alone on the chip, 0 = none is
    int streams = 1;      // streams the slots' work is dealt onto: slot i enqueues on the stream of slot i % streams (ugsm_config.streams)
    int march_mode = 0;   // strip heights of k_cost_march when cfg.march_rows == 0 (launch_cost_march's `rows`: 0, -1, -2, -3)
    int smooth_big_min = 0;  // development override: levels of at least this many pixels run k_smooth_fused on its 112-column tile (0 = by the mode)
    int smooth_rows = 0;  // height of k_smooth_fused's 112-column tile: 0 = by policy (smooth_rows_for), > 0 fixed, -1 / -2 = the latency / throughput rule
    long long batch_max_px = 0;  // development override of kBatchMaxPixels (batch_level): levels up to this size go through a batched call as one launch; < 0 = none
    CtxHooks hooks;  // the queue (ugsm_queue.cpp) and the RCCL shard (ugsm_shard.cpp): layers over the slot API
    long long dev_bytes = 0;    // device memory held by the slots' growable buffers (grow); ugsm_context_device_bytes
    std::unordered_map<const void *, size_t> dev_allocs;  // ... buffer by buffer, so that every release is accounted whatever path it takes
    long long release_epoch = 0;  // counts the releases of tracked device buffers (untrack): "memory may have come free since"
    bool counted_live = false;  // this context counts in g_live_contexts (ugsm_create got as far as handing it out)
    long long mem_limit = 0;    // development (UGSM_MEM_LIMIT_MB under UGSM_DEV=1): grow refuses to go past it -- the UGSM_ERR_NOMEM path without exhausting a GPU
};

namespace ugsm {
CtxHooks &ctx_hooks(ugsm_ctx *ctx) { return ctx->hooks; }
const ugsm_config &ctx_config(const ugsm_ctx *ctx) { return ctx->cfg; }
void *ctx_slot_stream(ugsm_ctx *ctx, int slot) { return (ctx && slot >= 0 && slot < (int)ctx->slots.size()) ? (void *)ctx->slots[(size_t)slot].st : nullptr; }
int ctx_fail(ugsm_ctx *ctx, int status, const char *what)
{
    if (ctx && what) ctx->err = what;
    return status;
}
}  // namespace ugsm

namespace {

#define HIPCHK(ctx, call)                                                                             \
    do {                                                                                              \
        hipError_t e__ = (call);                                                                      \
        if (e__ != hipSuccess) {                                                                      \
            char b__[512];                                                                            \
            snprintf(b__, sizeof b__, "%s failed: %s (%s:%d)", #call, hipGetErrorString(e__), __FILE__, __LINE__); \
            (ctx)->err = b__;                                                                         \
            return UGSM_ERR_DEVICE;                                                                   \
        }                                                                                             \
    } while (0)

#define UCHK(call)                       \
    do {                                 \
        int r__ = (call);                \
        if (r__ != UGSM_OK) return r__;  \
    } while (0)

constexpr long long kMaxPixels = 1LL << 28;  // 268 Mpx: 4 B x pixels of one plane stays below 2^31

int level_dims(int W, int H, int levels, int *w, int *h)
{
    if (W < 1 || H < 1 || levels < 1 || levels > UGSM_MAX_LEVELS) return UGSM_ERR_BAD_ARG;
    if ((long long)W * H > kMaxPixels) return UGSM_ERR_BAD_ARG;  // the kernels address a plane by 32-bit byte offsets
    if (H > 65535) return UGSM_ERR_BAD_ARG;  // several launchers put the image row on grid.y (HIP: at most 65535)
    w[0] = W;
    h[0] = H;
    for (int i = 0; i < levels - 1; i++) {  // MatchGPULib.cpp:1224-1228
        w[i + 1] = (int)(w[i] / kScale);
        h[i + 1] = (int)(h[i] / kScale);
    }
    for (int i = 0; i < levels; i++)
        if (w[i] < 1 || h[i] < 1) return UGSM_ERR_TOO_SMALL;
    return UGSM_OK;
}

int level_iterations(int i) { return (i > 5) ? 22 : ((i + 1) * 2); }  // MatchGPULib.cpp:1741
int level_smooth(int i) { return (i < 2) ? 10 : 5; }                  // MatchGPULib.cpp:2257-2261

void threshold_schedule(int mi, float *out)
{
    float threshold = 1.0f;  // MatchGPULib.cpp:1673
    for (int m = 1; m <= mi; m++) {
        out[m - 1] = threshold;
        if (m % 2 == 0) {  // :2299-2306
            if ((mi / 2 - m / 2) < 7)
                threshold = (float)(((mi / 2 - m / 2) - 1) * ((1 - 0.1) / (mi / 2 - 1.0)) + 0.1);
            else
                threshold = 1.0f;
        }
    }
}

struct FoveaGeom {
    int fw, fh, Wup, Hup;
    int ox[UGSM_MAX_LEVELS], oy[UGSM_MAX_LEVELS];  // crop origin at level lev < F-1
    int cx[UGSM_MAX_LEVELS], cy[UGSM_MAX_LEVELS];  // seed-crop origin for transition lev+1 -> lev
};

// Reference (off = 0): MatchGPULib.cpp:1143-1146,1173-1176 (pyramid crop) and :1612-1615 (seed crop).
// The offset generalisation is this build's (DESIGN.md "Fovea windows").
void fovea_geometry(const int *w, const int *h, int F, int off_x, int off_y, FoveaGeom &g)
{
    g.fw = w[F - 1];
    g.fh = h[F - 1];
    g.Wup = w[F - 2];
    g.Hup = h[F - 2];
    int ex[UGSM_MAX_LEVELS], ey[UGSM_MAX_LEVELS];
    ex[F - 1] = ey[F - 1] = 0;
    for (int lev = F - 2; lev >= 0; lev--) {
        int ccx = w[lev] / 2 - g.fw / 2, ccy = h[lev] / 2 - g.fh / 2;
        int ox = ccx + (int)lrint(off_x / pow(kScale, lev));
        int oy = ccy + (int)lrint(off_y / pow(kScale, lev));
        ox = std::min(std::max(ox, 0), w[lev] - g.fw);
        oy = std::min(std::max(oy, 0), h[lev] - g.fh);
        g.ox[lev] = ox;
        g.oy[lev] = oy;
        ex[lev] = ox - ccx;
        ey[lev] = oy - ccy;
    }
    for (int lev = F - 2; lev >= 0; lev--) {
        int lx = g.Wup / 2 - g.fw / 2 + ex[lev] - (int)lrint(kScale * ex[lev + 1]);
        int ly = g.Hup / 2 - g.fh / 2 + ey[lev] - (int)lrint(kScale * ey[lev + 1]);
        g.cx[lev] = std::min(std::max(lx, 0), g.Wup - g.fw);
        g.cy[lev] = std::min(std::max(ly, 0), g.Hup - g.fh);
    }
}

// ---- buffers ---------------------------------------------------------------------------

void untrack(ugsm_ctx *ctx, const void *p)
{
    auto it = ctx->dev_allocs.find(p);
    if (it == ctx->dev_allocs.end()) return;
    ctx->dev_bytes -= (long long)it->second;
    ctx->dev_allocs.erase(it);
    ctx->release_epoch++;
}

template <class T>
int grow(ugsm_ctx *ctx, T *&p, size_t &cap, size_t need)
{
    if (need <= cap) return UGSM_OK;
    if (p) {
        untrack(ctx, p);
        HIPCHK(ctx, hipFree(p));
    }
    p = nullptr;
    cap = 0;
    hipError_t e = (ctx->mem_limit > 0 && ctx->dev_bytes + (long long)(need * sizeof(T)) > ctx->mem_limit) ? hipErrorOutOfMemory
                                                                                                          : hipMalloc((void **)&p, need * sizeof(T));
    if (e != hipSuccess) {
        (void)hipGetLastError();  // (the failed allocation must not poison the next launch check)
        char b[256];
        snprintf(b, sizeof b, "hipMalloc of %zu bytes failed: %s (the context holds %lld bytes)", need * sizeof(T), hipGetErrorString(e), ctx->dev_bytes);
        ctx->err = b;
        p = nullptr;
        return UGSM_ERR_NOMEM;
    }
    cap = need;
    ctx->dev_bytes += (long long)(need * sizeof(T));
    ctx->dev_allocs[p] = need * sizeof(T);
    return UGSM_OK;
}

// (dx,dy,conf) ping-pong, A = G*L^2 and (kernel_path 1 only) R' and B: 3-plane buffers sized
// for the largest level seen.
int ensure_level_bufs(ugsm_ctx *ctx, Slot &s, size_t lvl)
{
    if (lvl <= s.lvl_cap) return UGSM_OK;
    // all-or-nothing: a failure part-way (out of memory) leaves every buffer freed and the capacity at zero, never a
    // capacity that some buffer does not have
    const size_t old = s.lvl_cap;
    s.lvl_cap = 0;
    int st = UGSM_OK;
    float **bufs[5] = {&s.A, &s.d0, &s.d1, &s.Rw, &s.B};
    const int nb = ctx->cfg.kernel_path == 1 ? 5 : 3;
    for (int k = 0; k < nb && st == UGSM_OK; k++) {
        size_t c = old;
        st = grow(ctx, *bufs[k], c, lvl);
    }
    if (st != UGSM_OK) {
        for (int k = 0; k < nb; k++) {
            if (*bufs[k]) {
                untrack(ctx, *bufs[k]);
                (void)hipFree(*bufs[k]);
            }
            *bufs[k] = nullptr;
        }
        return st;
    }
    s.lvl_cap = lvl;
    return UGSM_OK;
}

// the slot's pyramids and level buffers for `cap_pairs` pairs (both pyramids or neither; level buffers all or nothing)
int alloc_slot_buffers(ugsm_ctx *ctx, Slot &s, int cap_pairs, size_t tot, size_t lvl)
{
    const size_t pyr_need = cap_pairs > 1 ? s.pyr_stride * cap_pairs : tot;
    if (pyr_need > s.pyr_cap) {
        // both or neither: a failure leaves the slot without pyramids and the capacity at zero (never a capacity one of them lacks)
        size_t capL = s.pyr_cap, capR = s.pyr_cap;
        s.pyr_cap = 0;
        int st = grow(ctx, s.pyrL, capL, pyr_need);
        if (st == UGSM_OK) st = grow(ctx, s.pyrR, capR, pyr_need);
        if (st != UGSM_OK) {
            for (float **b : {&s.pyrL, &s.pyrR}) {
                if (*b) {
                    untrack(ctx, *b);
                    (void)hipFree(*b);
                }
                *b = nullptr;
            }
            return st;
        }
        s.pyr_cap = pyr_need;
    }
    return ensure_level_bufs(ctx, s, cap_pairs > 1 ? s.lvl_stride * cap_pairs : lvl);
}

int prepare_slot(ugsm_ctx *ctx, Slot &s, int W, int H, int nb = 1)
{
    const int levels = ctx->cfg.levels;
    int w[UGSM_MAX_LEVELS], h[UGSM_MAX_LEVELS];
    UCHK(level_dims(W, H, levels, w, h));
    if (nb < 1 || nb > kMaxBatch) return UGSM_ERR_BAD_ARG;
    size_t tot = 0;
    for (int i = 0; i < levels; i++) {
        s.w[i] = w[i];
        s.h[i] = h[i];
        s.off[i] = tot;
        tot += 3 * (size_t)w[i] * h[i];
    }
    s.W = W;
    s.H = H;
    s.levels = levels;
    s.nb = nb;
    for (int i = 0; i < UGSM_MAX_LEVELS; i++) s.iters_run[i] = -1;
    s.have_pyr = false;
    s.have_coarse = false;
    s.a_from = -1;
    const size_t lvl = 3 * (size_t)W * H;
    // (pairs of a batch start on 256-byte boundaries: the kernels' 16-byte accesses stay aligned whatever the level sizes add up to)
    s.pyr_stride = (tot + 63) & ~(size_t)63;
    s.lvl_stride = (lvl + 63) & ~(size_t)63;
    // sized for the batch the context was created for (ugsm_config.batch) even when this call brings fewer pairs: a host that batches
    // what has piled up alternates between call sizes, and a reallocation (hipFree + hipMalloc of gigabytes: a device-wide
    // synchronisation) must not land in the middle of its stream of calls
    const int want_pairs = std::max(nb, std::min(std::max(ctx->cfg.batch, 1), kMaxBatch));
    // ... but a context created for batches of eight on a card that has room for four still serves the calls that fit: if the buffers for
    // the configured batch cannot be had, size them for this call alone (they grow again when a larger call comes and memory allows)
    int st = UGSM_OK;
    for (int cap_pairs : {want_pairs, nb}) {
        // (the configured batch did not fit last time, nothing has been released since and this call does not need it: straight to the call's
        // own size -- no hipFree / failing hipMalloc / hipFree / hipMalloc of gigabytes, each a device-wide synchronisation, per call)
        if (cap_pairs > nb && s.nomem_pairs > 0 && cap_pairs >= s.nomem_pairs && s.nomem_tot == tot && s.nomem_epoch == ctx->release_epoch) continue;
        st = alloc_slot_buffers(ctx, s, cap_pairs, tot, lvl);
        if (st != UGSM_ERR_NOMEM) break;
        // out of memory: what the attempt (or an earlier, larger call) left in the slot goes back before anything else is tried, so that a
        // refused call never keeps memory it cannot use (the slot's stream has drained its earlier work by then: hipFree waits for it)
        const std::string why = ctx->err;
        float **bufs[7] = {&s.pyrL, &s.pyrR, &s.A, &s.d0, &s.d1, &s.Rw, &s.B};
        for (float **b : bufs) {
            if (*b) {
                untrack(ctx, *b);
                (void)hipFree(*b);
            }
            *b = nullptr;
        }
        s.pyr_cap = 0;
        s.lvl_cap = 0;
        ctx->err = why;
        s.nomem_pairs = cap_pairs;
        s.nomem_tot = tot;
        s.nomem_epoch = ctx->release_epoch;  // (after this attempt's own releases)
        if (cap_pairs == nb) break;
    }
    return st;
}

// ---- event bookkeeping -----------------------------------------------------------------

struct Timer {
    ugsm_ctx *ctx;
    Slot *s;
    bool on;
    EvRec rec;
    LaunchProbe probe;
    Timer(ugsm_ctx *c, Slot *sl, int slot_idx, int kclass, double pixels) : ctx(c), s(sl)
    {
        // 1: only the dominant (cost) kernel is bracketed -- two events per launch are not free (a 16 MP pair
        // has ~750 launches; bracketing all of them costs slot 0 about 20 %); 2: every kernel class
        const bool cost_class = kclass == KC_COST || kclass == KC_COST_MARCH || kclass == KC_COST_SMALL || kclass == KC_COST_MARCH4;
        on = slot_idx == 0 && (c->cfg.profile_events >= 2 || (c->cfg.profile_events == 1 && cost_class));
        if (!on) return;
        rec.kclass = kclass;
        rec.level = sl->cur_level;
        rec.pixels = pixels;
        for (hipEvent_t *e : {&rec.a, &rec.b}) {
            if (!s->pool.empty()) {
                *e = s->pool.back();
                s->pool.pop_back();
            } else if (hipEventCreate(e) != hipSuccess) {
                on = false;
                return;
            }
        }
        probe = LaunchProbe{rec.a, rec.b, 0};  // the bracket's launch carries the two events in its dispatch (ugsm_launch.hpp)
        g_probe = &probe;
    }
    ~Timer()
    {
        if (!on) return;
        g_probe = nullptr;
        if (probe.n == 0) {  // (nothing was launched inside the bracket)
            s->pool.push_back(rec.a);
            s->pool.push_back(rec.b);
            return;
        }
        s->pending.push_back(rec);
    }
};

void harvest(ugsm_ctx *ctx, Slot &s)
{
    for (EvRec &r : s.pending) {
        float ms = 0;
        if (hipEventElapsedTime(&ms, r.a, r.b) == hipSuccess) {
            StatCell &c = ctx->cells[r.kclass][r.level];
            c.launches += 1;
            c.total_ms += ms;
            c.pixel_launches += r.pixels;
        }
        s.pool.push_back(r.a);
        s.pool.push_back(r.b);
    }
    s.pending.clear();
}

// ---- development overrides ---------------------------------------------------------------
// tools/ and tests/ steer the per-level kernel choice through environment variables.  They are read ONLY when UGSM_DEV=1 is set
// as well, so that a production process never changes behaviour because of a stray variable, and in ONE place, shared by
// ugsm_create and ugsm_plan_level (which therefore reports what a context created under the same environment launches).
struct DevKnobs {
    int fuse_seed = 1;       // UGSM_FUSE_SEED=0: seed every level with its own launch
    int small_mask = 3;      // UGSM_SMALL_MASK: bit 0 = k_cost_small, bit 1 = k_smooth_small
    int small_rh_force = 0;  // UGSM_SMALL_RH: region height of k_smooth_small whatever the level size (18, 24 or 32)
    int force_alone = -1;    // UGSM_ALONE=1 / 0: the kernel choices of a call that has the chip to itself / that shares it, whatever is in flight
    int two_streams = -1;    // UGSM_TWO_STREAMS=0: no call forks onto a side stream (default 1: a call that is alone does)
    char side_prio = 0;         // UGSM_SIDE_PRIO=h|n|l|s: a side stream of its own for every slot, at that priority (s: the slot's); default: borrowed (ugsm_create)
    char stream_prio[65] = "";  // UGSM_STREAM_PRIO: one letter per slot, h / n / l = greatest / default / least stream priority (slot_stream_priority)
    int march_mode = 0;      // UGSM_MARCH_MODE=0,-1,-2,-3: launch_cost_march's strip-height / age-class mode (default 0: latency heights, strips by age class)
    long long batch_max_px = 0;  // UGSM_BATCH_MAX_PIXELS: levels up to this many pixels are ONE launch for all pairs of a batched call (default kBatchMaxPixels; < 0: none)
    int smooth_big_min = 0;  // UGSM_SMOOTH_BIG_MIN: pixel count from which K-smooth uses the 112-column tile (default 2^19)
    int smooth_rows = 0;     // UGSM_SMOOTH_ROWS: tile height of the large levels' K-smooth (1..39), -1 / -2 = the latency / throughput rule whatever the slots
    int march4_lo = -1, march4_hi = -1;  // UGSM_MARCH4=lo,hi: pixel range of k_cost_march4 (0,0 = never; default: march4_default_range)
};
bool dev_env_on()
{
    const char *e = getenv("UGSM_DEV");
    return e && e[0] == '1';
}
// set_globals: also apply the process-wide tuning variables (UGSM_SMOOTH_MID_MIN, UGSM_PYR_STREAM*, UGSM_MARCH_AGE).  They are CREATE-TIME,
// PROCESS-WIDE settings: ugsm_create applies them only while no other context of the process is alive (g_live_contexts, under
// g_globals_mutex), so that creating a second context never changes the kernels of a live one, from whatever thread; the
// host-only ugsm_plan_level never does.
std::mutex g_globals_mutex;
int g_live_contexts = 0;
void apply_dev_env(ugsm_config &cfg, DevKnobs &k, bool set_globals)
{
    if (!dev_env_on()) return;
    auto geti = [](const char *name, int &dst) {
        if (const char *e = getenv(name)) dst = atoi(e);
    };
    geti("UGSM_KERNEL_PATH", cfg.kernel_path);
    geti("UGSM_MARCH_MIN_PIXELS", cfg.march_min_pixels);
    geti("UGSM_MARCH_ROWS", cfg.march_rows);
    geti("UGSM_SMALL_MAX_PIXELS", cfg.small_max_pixels);
    geti("UGSM_SMALL_MASK", k.small_mask);
    geti("UGSM_FUSE_SEED", k.fuse_seed);
    geti("UGSM_TWO_STREAMS", k.two_streams);
    geti("UGSM_SMOOTH_ROWS", k.smooth_rows);
    geti("UGSM_MARCH_MODE", k.march_mode);
    if (const char *e = getenv("UGSM_STREAM_PRIO")) snprintf(k.stream_prio, sizeof k.stream_prio, "%s", e);
    if (const char *e = getenv("UGSM_SIDE_PRIO")) k.side_prio = e[0];
    if (const char *e = getenv("UGSM_ALONE")) k.force_alone = e[0] == '1' ? 1 : (e[0] == '0' ? 0 : -1);
    geti("UGSM_SMOOTH_BIG_MIN", k.smooth_big_min);
    if (const char *e = getenv("UGSM_BATCH_MAX_PIXELS")) k.batch_max_px = atoll(e);
    if (set_globals) {  // the process-wide tuning variables: back to their defaults first, so that a variable a test has removed stops acting
        smooth_mid_min_pixels = 1 << 18;
        smooth_lds_extra_bytes = 0;
        blur_decimate_streaming = 1;
        pyr_base_streaming = 1;
        blur_decimate_streaming_min = 0;
        march_age_permille[0] = 470;
        march_age_permille[1] = 340;
        geti("UGSM_SMOOTH_MID_MIN", smooth_mid_min_pixels);
        geti("UGSM_SMOOTH_LDS_EXTRA", smooth_lds_extra_bytes);
        geti("UGSM_PYR_STREAM", blur_decimate_streaming);
        geti("UGSM_PYR_BASE_STREAM", pyr_base_streaming);
        if (const char *e = getenv("UGSM_PYR_STREAM_MIN")) blur_decimate_streaming_min = atoll(e);
    }
    if (const char *e = set_globals ? getenv("UGSM_MARCH_AGE") : nullptr) {
        int a = 0, b = 0;
        if (sscanf(e, "%d,%d", &a, &b) == 2 && a >= 0 && b >= 0 && a + b < 1000) {
            march_age_permille[0] = a;
            march_age_permille[1] = b;
        }
    }
    if (const char *e = getenv("UGSM_MARCH4")) {
        int a = 0, b = 0;
        if (sscanf(e, "%d,%d", &a, &b) == 2 && a >= 0 && b >= a) {
            k.march4_lo = a;
            k.march4_hi = b;
        }
    }
    int rh = 0;
    geti("UGSM_SMALL_RH", rh);
    if (rh == 18 || rh == 24 || rh == 32) k.small_rh_force = rh;
}

// The choices of a context that are not in ugsm_config: the development overrides (the defaults are with the policy functions below).
void set_policy(ugsm_ctx *c, const DevKnobs &k)
{
    c->small_mask = k.small_mask;
    c->fuse_seed = k.fuse_seed;
    c->small_rh_force = k.small_rh_force;
    c->smooth_rows = k.smooth_rows;
    c->smooth_big_min = k.smooth_big_min > 0 ? k.smooth_big_min : 0;
    // strips by age class, every context: +8 % on a level-0 launch alone on the chip, +1.2 % on a pair alone; with four pairs in
    // flight it costs 0.4 % (163.2 against 163.7 pairs/s) -- kept on there too, so that a kernel measured alone is the kernel that ran
    c->march_mode = k.march_mode <= 0 ? k.march_mode : 0;
    c->march4_lo = k.march4_lo;
    c->march4_hi = k.march4_hi;
    c->force_alone = k.force_alone;
    c->batch_max_px = k.batch_max_px;
}

// ---- stages ----------------------------------------------------------------------------

// The pairs of one launch: b0 .. b0 + n - 1 of the batch in the slot (n = 1: an ordinary launch of pair b0).
struct Grp {
    int b0, n;
};
// f(Grp) for the whole batch in one launch (`batched`) or pair by pair
template <class F>
void for_groups(int nb, bool batched, F &&f)
{
    if (batched && nb > 1) {
        f(Grp{0, nb});
        return;
    }
    for (int b = 0; b < nb; b++) f(Grp{b, 1});
}
inline long long byte_diff(const void *a, const void *b) { return (long long)(reinterpret_cast<intptr_t>(a) - reinterpret_cast<intptr_t>(b)); }
// The Batch argument of a group's launch.  Inputs and outputs are the slot's level buffers (pair b at + b * lvl_stride) unless
// `outs` names per-pair destinations (the callers' buffers); `views`: the L / R views of the pairs (fovea windows sit at different
// places); `seeds`: their seed-crop origins.  All arrays are indexed by the pair's number in the batch.
Batch make_batch(const Slot &s, Grp g, const Img3 *views = nullptr, const SeedMap *seeds = nullptr, float *const *outs = nullptr)
{
    Batch b{};
    b.n = g.n;
    for (int j = 0; j < g.n; j++) {
        b.in[j] = (long long)(j * s.lvl_stride * sizeof(float));
        b.out[j] = outs ? byte_diff(outs[g.b0 + j], outs[g.b0]) : b.in[j];
        b.img[j] = views ? byte_diff(views[g.b0 + j].p, views[g.b0].p) : 0;
        b.cx[j] = seeds ? seeds[g.b0 + j].cx : 0;
        b.cy[j] = seeds ? seeds[g.b0 + j].cy : 0;
    }
    return b;
}

// CreatePyramidFromImage (MatchGPULib.cpp:1033-1125) for the left OR the right image of every pair of the call: rgb[b] -> pyr + b * pyr_stride.
// The images of a batch go through every pyramid kernel together (one launch per level for all of them).
// What a foveated call reads of level 0: every pair's fovea window (the level-0 crop of fovea_geometry); k_pyr_base then stores level 0
// only where a window lies.  Full-mode calls and ugsm_submit_pyramids (window not known yet) pass none.
struct FoveaWin {
    int w = 0, h = 0;
    int x0[kMaxBatch], y0[kMaxBatch];
};
int build_pyramids(ugsm_ctx *ctx, Slot &s, int si, const uint8_t *const *rgb, int stride, float *pyr, hipStream_t stream = nullptr, const FoveaWin *win = nullptr)
{
    const hipStream_t pst = stream ? stream : s.st;  // (launches on the side stream are not bracketed by events: Timer records on s.st)
    const int levels = s.levels, nb = s.nb;
    const bool ref = ctx->cfg.kernel_path == 1;
    // levels 0, 1, 2 in one pass over the rgb8 input (k_pyr_base); the one-stage-per-kernel path keeps the three launches
    const bool base = !ref && levels >= 3;
    Batch bt{};  // image j of the launch = pair j: its rgb8 input, its pyramid, its range word
    bt.n = nb;
    for (int j = 0; j < nb; j++) {
        bt.img[j] = byte_diff(rgb[j], rgb[0]);
        bt.in[j] = bt.out[j] = (long long)(j * s.pyr_stride * sizeof(float));
        bt.cx[j] = j;
    }
    Batch bt0 = bt;  // k_pyr_base: the input-field offsets carry the pairs' window origins instead
    if (win)
        for (int j = 0; j < nb; j++) bt0.in[j] = ((long long)win->y0[j] << 32) | (unsigned)win->x0[j];
    const PyrWindow pw = win ? PyrWindow{win->x0[0], win->y0[0], win->w, win->h} : PyrWindow{0, 0, 0, 0};
    const Batch *const pb = nb > 1 ? &bt : nullptr;
    // One full-mode pair alone on the chip: the streaming factor-2 kernel's many short workgroups on the side stream get
    // in the way of the main stream's latency-bound launches -- 112.3 pairs/s with it on every level, 114.4 with it on the launches of
    // >= 1.5 M outputs only, 114.1 without it (tools/ab.py, same box); foveated calls and everything with more in flight gain from it
    // on every level (a lone foveated pair +1.8 %, batches of eight +3.3 %).
    const long long stream_min = (s.alone && nb == 1 && !win) ? 1500000 : 0;
    s.cur_level = 0;
    if (base) {
        Timer t(ctx, &s, si, KC_PYR_BASE, (double)s.W * s.H * nb);
        launch_pyr_base(pst, rgb[0], stride, s.W, s.H, pyr + s.off[0], pyr + s.off[1], s.w[1], s.h[1], pyr + s.off[2], s.w[2], s.h[2], s.range_bad,
                        nb > 1 ? &bt0 : nullptr, pw);
    } else {
        // (pyramids of fewer than three levels, and kernel_path 1: image by image)
        for (int b = 0; b < nb; b++) {
            Timer t(ctx, &s, si, KC_MISC, (double)s.W * s.H);
            launch_rgb_planes(pst, rgb[b], stride, s.W, s.H, pyr + b * s.pyr_stride + s.off[0]);
        }
    }
    // CreatePyramidFromImage, MatchGPULib.cpp:1063-1106: level 1 from level 0 (sf=(float)SCALE),
    // level i+2 from level i (sf=2.0f).  Levels are produced in dependency order.
    for (int i = 0; i < levels; i++) {
        if (i == 0 && levels > 1 && !base) {
            s.cur_level = 1;
            Timer t(ctx, &s, si, KC_PYR, (double)s.w[1] * s.h[1] * nb);
            float sf = (float)kScale;
            if (ref) launch_blur_decimate_ref(pst, pyr + s.off[0], s.w[0], s.h[0], pyr + s.off[1], s.w[1], s.h[1], sf);
            else launch_blur_decimate(pst, pyr + s.off[0], s.w[0], s.h[0], pyr + s.off[1], s.w[1], s.h[1], sf, s.range_bad, pb, stream_min);
        }
        if (i + 2 < levels && !(base && i == 0)) {
            s.cur_level = i + 2;
            Timer t(ctx, &s, si, KC_PYR, (double)s.w[i + 2] * s.h[i + 2] * nb);
            float sf = (float)(0.000 + (int)(kScale * kScale + 0.5));  // :1090
            if (ref) launch_blur_decimate_ref(pst, pyr + s.off[i], s.w[i], s.h[i], pyr + s.off[i + 2], s.w[i + 2], s.h[i + 2], sf);
            else launch_blur_decimate(pst, pyr + s.off[i], s.w[i], s.h[i], pyr + s.off[i + 2], s.w[i + 2], s.h[i + 2], sf, s.range_bad, pb, stream_min);
        }
    }
    s.cur_level = kNoLevel;
    HIPCHK(ctx, hipGetLastError());
    return UGSM_OK;
}

// ---- per-level kernel choices ---------------------------------------------------------------------------------------
// Which kernel runs a level is decided by the level's size -- compared with what the LAUNCH holds: pairs x W x H, pairs = the batch where
// the level is batched, else 1 -- and by ONE question: does the call have the chip to itself (Slot::alone; call_alone below)?  A call alone
// wants every launch SHORT: nothing else fills the CUs a launch leaves idle.  A call that shares the chip wants every launch to do LITTLE
// REDUNDANT WORK: what it wastes, the others could have used.  Five choices ask the question (marked ALONE below, plus the pyramid's
// streaming threshold in build_pyramids and the side stream); the rest is six numbers.  The A/B behind each: DESIGN.md section 4.
// (Rounds 3-5 switched between two whole threshold sets by ugsm_config.slots and the frame size; forced against each other they differ by
// 0.35 % where bench.py measures, and a lone 16 MP call on a four-slot context lost 11.6 % to the wrong one: VERDICT r05 #1.)
constexpr long long kBatchMaxPixels = 9000000;  // levels up to here go through a batched call as ONE launch for all its pairs (level 0 of a 16 MP pair: pair by pair)
constexpr int kSmallMaxPixelsAlone = 150000;    // the coarse levels' latency kernels (ugsm_kernels_small.hip) up to here for a call alone ...
constexpr int kSmallMaxPixelsShared = 50000;    // ... and up to here on a shared chip (above it k_cost_march4 and the tiled K-smooth redo less)
constexpr int kMarch4MaxPixels = 3000000;       // above them k_cost_march4 (four waves per strip: wins while a launch lasts as long as one strip) up to here
constexpr int kMarchMinPixels = 400000;         // k_cost_march from here -- asked after k_cost_march4, so in effect above its range
constexpr int kSmoothBigMinPixels = 1 << 19;    // k_smooth_fused's 112-column tile from here (1.39 x the tile in halo work, against 1.8 x for 64 x 32)

bool batch_level(const ugsm_ctx *ctx, int W, int H)
{
    const long long thr = ctx->batch_max_px > 0 ? ctx->batch_max_px : (ctx->batch_max_px < 0 ? 0 : kBatchMaxPixels);
    return (long long)W * H <= thr;
}
// pairs a launch of a W x H level of the call in `s` holds
int launch_pairs(const ugsm_ctx *ctx, const Slot &s, int W, int H) { return (s.nb > 1 && batch_level(ctx, W, H)) ? s.nb : 1; }

int small_max_px(const ugsm_config &cfg, bool alone)
{
    if (cfg.small_max_pixels < 0) return 0;
    return cfg.small_max_pixels > 0 ? cfg.small_max_pixels : (alone ? kSmallMaxPixelsAlone : kSmallMaxPixelsShared);
}
bool use_march4(const ugsm_ctx *ctx, int W, int H, bool alone, int pairs = 1)
{
    const long long px = (long long)W * H * pairs;
    if (ctx->cfg.kernel_path == 1) return false;
    if (ctx->march4_hi >= 0) return ctx->march4_hi > 0 && px >= ctx->march4_lo && px <= ctx->march4_hi;  // (development override)
    return px > small_max_px(ctx->cfg, alone) && px <= kMarch4MaxPixels;
}
bool use_march(const ugsm_ctx *ctx, int W, int H, int pairs = 1)  // (ugsm_config.march_min_pixels moves the threshold: tests run every level through it)
{
    const int thr = ctx->cfg.march_min_pixels;
    return thr >= 0 && (long long)W * H * pairs >= (thr > 0 ? thr : kMarchMinPixels);
}
int march_rows_arg(const ugsm_ctx *ctx) { return ctx->cfg.march_rows > 0 ? ctx->cfg.march_rows : ctx->march_mode; }

// The latency kernels' K-smooth region height (0 = not a level of theirs).  On a shared chip 18 x 18 tiles (3.2 x the tile in halo work; the
// 18 x 4 / 18 x 10 tiles redo 8 x / 4.6 x: free on an idle chip, 3.6 % / 9.2 % of the throughput with four pairs in flight); a call ALONE
// takes the smallest tile that still gives every workgroup a CU of its own, or nearly.
int small_rh(const ugsm_ctx *ctx, int W, int H, bool alone, int pairs = 1)
{
    const long long px = (long long)W * H * pairs;
    if (ctx->cfg.small_max_pixels < 0 || ctx->cfg.kernel_path == 1 || px > small_max_px(ctx->cfg, alone) || use_march(ctx, W, H, pairs)) return 0;
    if (ctx->small_rh_force) return ctx->small_rh_force;
    return !alone ? 32 : (px <= 36000 ? 18 : (px <= 80000 ? 24 : 32));
}

// Seeding (subsampleDisp, MatchGPULib.cpp:1526-1590) rides on the level's first K-cost launch when that is a marching kernel: the seeded field is
// never written.  Not with the early exit (the field before the first iteration is compared against), not on the one-stage-per-kernel path.
bool fuse_seed(const ugsm_ctx *ctx, int W, int H, bool alone, int pairs = 1)
{
    return ctx->fuse_seed && ctx->cfg.kernel_path != 1 && !(ctx->cfg.early_exit_threshold > 0.0f) && (use_march(ctx, W, H, pairs) || use_march4(ctx, W, H, alone, pairs));
}

// k_smooth_fused's tile: > 0 = the 112-column tile at this height -- a call ALONE picks the height that fills whole rounds of workgroups
// (smooth_tile_rows), else 36 --; 0 = the tile class by the level's size (64 x 32 from 0.26 Mpx, else 32 x 16)
int smooth_rows_for(const ugsm_ctx *ctx, int W, int H, bool alone, int pairs = 1)
{
    if ((long long)W * H * pairs < (ctx->smooth_big_min > 0 ? ctx->smooth_big_min : kSmoothBigMinPixels)) return 0;
    if (pairs > 1 && W < 100) return 0;  // (a level narrower than the 112-column tile: the smaller tile classes waste fewer lanes)
    if (ctx->smooth_rows > 0) return std::min(ctx->smooth_rows, kSmoothTileRowsMax);
    return smooth_tile_rows(W, H, ctx->smooth_rows == -1 ? 1 : (ctx->smooth_rows == -2 ? 0 : (alone ? 1 : 0)), pairs);
}
// ... and the tile class of the smaller levels of a batched launch: by what the launch holds
int smooth_class_for(int W, int H, int pairs)
{
    return pairs <= 1 ? 0 : (((long long)W * H * pairs >= smooth_mid_min_pixels && W >= 48 && H >= 24) ? 2 : 1);
}

// S Jacobi passes + the 3x3 box (MatchGPULib.cpp:2257-2412) for every pair of the call.  On return `a` holds the
// result and `b` is scratch (base pointers: pair k's buffers lie at + k * lvl_stride).
// final_out (optional, fused path): per-pair destinations; the last launch writes there instead of into `b`, and `a` is then not meaningful.
int enqueue_smooth(ugsm_ctx *ctx, Slot &s, int si, float *&a, float *&b, int W, int H, int S, bool do_box, float *const *final_out = nullptr)
{
    const double px = (double)W * H;
    if (ctx->cfg.kernel_path == 1) {  // (never batched)
        for (int j = 0; j < S; j++) {
            Timer t(ctx, &s, si, KC_SMOOTH, px);
            launch_smooth_pass_ref(s.st, a, b, W, H);
            std::swap(a, b);
        }
        if (do_box) {
            Timer t(ctx, &s, si, KC_BOX, px);
            launch_box_ref(s.st, a, b, W, H);
            std::swap(a, b);
        }
    } else {
        const int pairs = launch_pairs(ctx, s, W, H);
        int left = S;
        do {
            int p = std::min(left, 5);
            left -= p;
            if (p == 0 && !do_box) break;
            const int rh = (ctx->small_mask & 2) ? small_rh(ctx, W, H, s.alone, pairs) : 0;
            const bool box_now = do_box && left == 0;
            const bool to_final = left == 0 && final_out;
            for_groups(s.nb, pairs > 1, [&](Grp g) {
                Timer t(ctx, &s, si, rh ? KC_SMOOTH_SMALL : KC_SMOOTH, px * g.n);
                const float *src = a + g.b0 * s.lvl_stride;
                float *dst = to_final ? final_out[g.b0] : b + g.b0 * s.lvl_stride;
                const Batch bt = make_batch(s, g, nullptr, nullptr, to_final ? final_out : nullptr);
                const Batch *pb = g.n > 1 ? &bt : nullptr;
                if (rh) launch_smooth_small(s.st, src, dst, W, H, p, box_now, rh, pb);
                else launch_smooth_fused(s.st, src, dst, W, H, p, box_now, smooth_rows_for(ctx, W, H, s.alone, g.n), pb, smooth_class_for(W, H, g.n));
            });
            if (!to_final) std::swap(a, b);
        } while (left > 0);
    }
    return UGSM_OK;
}

// Row f-4 (MatchGPULib.cpp:1323-1437): S_dx / C and S_dy / C of two device fields, synchronously (one host round trip).
int weighted_difference(ugsm_ctx *ctx, Slot &s, const float *newd3, const float *oldd3, int W, int H, float out2[2])
{
    const size_t need = 3 * (size_t)H + 3;
    if (need > s.wd_cap) {
        if (s.wd_rows) HIPCHK(ctx, hipFree(s.wd_rows));
        s.wd_rows = nullptr;
        s.wd_cap = 0;
        HIPCHK(ctx, hipMalloc((void **)&s.wd_rows, need * sizeof(double)));
        s.wd_cap = need;
    }
    if (!s.wd_host) HIPCHK(ctx, hipHostMalloc((void **)&s.wd_host, 3 * sizeof(double), hipHostMallocDefault));
    double *tot = s.wd_rows + 3 * (size_t)H;
    launch_weighted_difference(s.st, newd3, oldd3, W, H, s.wd_rows, tot);
    HIPCHK(ctx, hipMemcpyAsync(s.wd_host, tot, 3 * sizeof(double), hipMemcpyDeviceToHost, s.st));
    HIPCHK(ctx, hipStreamSynchronize(s.st));
    out2[0] = (float)(s.wd_host[0] / s.wd_host[2]);
    out2[1] = (float)(s.wd_host[1] / s.wd_host[2]);
    return UGSM_OK;
}

// matchlevel (MatchGPULib.cpp:1662-2489), iterations m_from..m_to, for every pair of the call (s.nb; in lockstep).  Lv / Rv: the pairs'
// views of the level (s.nb entries).  cur holds (dx,dy,conf) on entry and on exit; other is scratch of the same size (base pointers:
// pair k's fields lie at + k * lvl_stride).
// final_out (optional, fused path only): per-pair destinations where the last iteration leaves its result instead of the ping-pong buffer
// (saves the device-to-device copy of the finished level); cur/other are then not meaningful afterwards.
// seed (optional, see fuse_seed; s.nb entries): `cur` holds the COARSER level's field (seed->Ws x seed->Hs) and iteration m_from reads its
// starting field through the seeding map instead of from a materialised seeded field.
// A_pre (optional, single pairs only): A = G_clamp * L^2 of this level, already computed (on the slot's side stream; the caller has made
// the main stream wait for it); otherwise it is computed here, into s.A.
int run_level(ugsm_ctx *ctx, Slot &s, int si, const Img3 *Lv, const Img3 *Rv, int W, int H, int mi, int S, bool is_top, int m_from,
              int m_to, float *&cur, float *&other, float *dbg8, float *const *final_out = nullptr, const SeedMap *seed = nullptr,
              const float *A_pre = nullptr)
{
    const bool ref = ctx->cfg.kernel_path == 1;
    const double px = (double)W * H;
    std::vector<float> thr((size_t)std::max(mi, 1));
    threshold_schedule(mi, thr.data());
    // row f-4, opt-in: stop the level when the confidence-weighted mean change of dx and dy falls below the threshold.  The
    // previous iteration's field has to survive the smoothing ping-pong, hence a third buffer.
    const float eps = ctx->cfg.early_exit_threshold;
    const bool early = eps > 0.0f;
    if ((early || ref) && s.nb != 1) return UGSM_ERR_STATE;  // (the batch entry points run such contexts pair by pair)
    float *third = nullptr;
    if (early) {
        const size_t lvl = 3 * (size_t)W * H;
        UCHK(grow(ctx, s.d2, s.d2_cap, std::max(lvl, s.lvl_cap)));
        // the three field buffers rotate through the levels: the spare one is whichever is neither `cur` nor `other` right now
        third = (s.d0 != cur && s.d0 != other) ? s.d0 : ((s.d1 != cur && s.d1 != other) ? s.d1 : s.d2);
        final_out = nullptr;
    }
    int ran = 0;
    const int pairs = launch_pairs(ctx, s, W, H);
    const bool batched = pairs > 1;
    const float *const A3 = A_pre ? A_pre : s.A;  // (pair k's A at + k * lvl_stride)
    if (!A_pre) {   // A = G_clamp * L^2 does not depend on the iteration: once per level.
        for_groups(s.nb, batched, [&](Grp g) {
            Timer t(ctx, &s, si, KC_SQBLUR, px * g.n);
            if (ref) {
                launch_sqblur_clamp_ref(s.st, Lv[g.b0], W, H, s.A);
            } else {
                const Batch bt = make_batch(s, g, Lv);
                launch_sqblur_clamp(s.st, Lv[g.b0], W, H, s.A + g.b0 * s.lvl_stride, g.n > 1 ? &bt : nullptr);
            }
        });
    }
    const bool march4 = !ref && use_march4(ctx, W, H, s.alone, pairs);
    const bool march = !ref && use_march(ctx, W, H, pairs);
    const bool small = !ref && (ctx->small_mask & 1) && small_rh(ctx, W, H, s.alone, pairs) != 0;
    // (k_cost_split, the LDS-tiled form a level falls back to when the marching kernels are switched off, has no batch index: pair by pair)
    const bool cost_batched = batched && (march4 || march || small);
    for (int m = m_from; m <= m_to; m++) {
        const int blend = !(is_top && m == 1);  // MatchGPULib.cpp:2223
        if (ref) {
            {
                Timer t(ctx, &s, si, KC_WARP, px);
                launch_warp_ref(s.st, Rv[0], cur, W, H, s.Rw);
            }
            {
                Timer t(ctx, &s, si, KC_SQBLUR, px);
                launch_sqblur_clamp_ref(s.st, Img3{s.Rw, W, (size_t)W * H}, W, H, s.B);
            }
            Timer t(ctx, &s, si, KC_COST, px);
            launch_cost_ref(s.st, Lv[0], s.Rw, A3, s.B, cur, other, W, H, thr[m - 1], blend, (m == m_to) ? dbg8 : nullptr);
        } else {
            const bool seeded = seed && m == m_from;
            for_groups(s.nb, cost_batched, [&](Grp g) {
                Timer t(ctx, &s, si, march4 ? KC_COST_MARCH4 : (march ? KC_COST_MARCH : (small ? KC_COST_SMALL : KC_COST)), px * g.n);
                const size_t fo = g.b0 * s.lvl_stride;
                const Img3 L = Lv[g.b0], R = Rv[g.b0];
                const Batch bt = make_batch(s, g, Lv, seeded ? seed : nullptr);
                const Batch *pb = g.n > 1 ? &bt : nullptr;
                const unsigned *rb = s.range_known ? s.range_bad + g.b0 : nullptr;
                if (march4)
                    launch_cost_march4(s.st, L, R, A3 + fo, cur + fo, other + fo, W, H, thr[m - 1], blend, 0, rb, seeded ? seed[g.b0] : SeedMap{0, 0, 0, 0}, pb);
                else if (march && seeded)
                    launch_cost_march_seeded(s.st, L, R, A3 + fo, cur + fo, seed[g.b0], other + fo, W, H, thr[m - 1], blend, march_rows_arg(ctx), rb, pb);
                else if (march) launch_cost_march(s.st, L, R, A3 + fo, cur + fo, other + fo, W, H, thr[m - 1], blend, march_rows_arg(ctx), rb, pb);
                else if (small) launch_cost_small(s.st, L, R, A3 + fo, cur + fo, other + fo, W, H, thr[m - 1], blend, pb);
                else launch_cost_fused(s.st, L, R, A3 + fo, cur + fo, other + fo, W, H, thr[m - 1], blend);
            });
        }
        ran = m;
        if (early) {
            float *a = other, *b = third;  // cur (the field before this iteration) stays untouched
            UCHK(enqueue_smooth(ctx, s, si, a, b, W, H, S, true, nullptr));
            float dif[2] = {0.0f, 0.0f};
            if (m < m_to) UCHK(weighted_difference(ctx, s, a, cur, W, H, dif));
            float *old = cur;
            cur = a;  // the new field; b and the old field are the scratch pair of the next iteration
            other = b;
            third = old;
            if (m < m_to && dif[0] < eps && dif[1] < eps) break;  // differenceIterations: both below the threshold
            continue;
        }
        float *a = other, *b = cur;
        UCHK(enqueue_smooth(ctx, s, si, a, b, W, H, S, true, (m == m_to && !ref) ? final_out : nullptr));
        if (m == m_to && !ref && final_out) break;
        cur = a;
        other = b;
    }
    s.iters_run[s.cur_level == kNoLevel ? 0 : s.cur_level] = ran;
    HIPCHK(ctx, hipGetLastError());
    return UGSM_OK;
}

Img3 level_view(const Slot &s, const float *pyr, int lev, int ox, int oy)
{
    return Img3{pyr + s.off[lev] + (size_t)oy * s.w[lev] + ox, s.w[lev], (size_t)s.w[lev] * s.h[lev]};
}
// the full-frame views of level `lev` for every pair of the call
void full_views(const Slot &s, const float *pyr, int lev, Img3 *out)
{
    for (int b = 0; b < s.nb; b++) out[b] = level_view(s, pyr + b * s.pyr_stride, lev, 0, 0);
}

// Whether this call may use the slot's side stream: single pairs that have the chip to themselves (with four pairs in flight eight
// streams on the four hardware queues serialise what one stream per pair lets overlap: -13 %), the fused path, no event brackets (the
// statistics belong to one stream).
bool side_stream_ok(const ugsm_ctx *ctx, const Slot &s)
{
    return s.nb == 1 && s.alone && s.st2 != nullptr && ctx->cfg.kernel_path != 1 && ctx->cfg.profile_events == 0 && ctx->two_streams;
}
// A = G_clamp * L^2 of the full-frame levels a_from .. top on the side stream, beside whatever the main stream does next (the right
// pyramid's join, the coarse levels' iterations); run_level takes them through level_A, which makes the main stream wait for each.
int enqueue_side_A(ugsm_ctx *ctx, Slot &s, int a_from)
{
    if (a_from < 0 || ctx->cfg.early_exit_threshold > 0.0f) return UGSM_OK;
    size_t tot = s.off[s.levels - 1] + 3 * (size_t)s.w[s.levels - 1] * s.h[s.levels - 1];
    UCHK(grow(ctx, s.Apyr, s.apyr_cap, tot));
    HIPCHK(ctx, hipEventRecord(s.ev_L, s.st));  // (the left pyramid is complete on the main stream)
    s.forked = true;
    HIPCHK(ctx, hipStreamWaitEvent(s.st2, s.ev_L, 0));
    for (int i = s.levels - 1; i >= a_from; i--) {  // coarsest first: that is the order the levels need them in
        launch_sqblur_clamp(s.st2, level_view(s, s.pyrL, i, 0, 0), s.w[i], s.h[i], s.Apyr + s.off[i]);
        HIPCHK(ctx, hipEventRecord(s.ev_A[i], s.st2));
    }
    s.a_from = a_from;
    return UGSM_OK;
}

// The pyramids of every pair of the call (s.nb = nb pairs; rgbL / rgbR: nb device pointers).
// a_from: the levels a_from .. top get their A = G_clamp * L^2 precomputed on the side stream (full mode: 0; foveated: F-1, the
// fine levels work on crops whose A is clamped at the crop's own border and is computed in line); < 0: none.
int enqueue_pyramids(ugsm_ctx *ctx, Slot &s, int si, const uint8_t *const *d_rgbL, const uint8_t *const *d_rgbR, int nb, int W, int H, int stride, int a_from = -1,
                     const FoveaWin *win = nullptr)
{
    if (!d_rgbL || !d_rgbR || nb < 1 || nb > kMaxBatch) return UGSM_ERR_BAD_ARG;
    for (int b = 0; b < nb; b++)
        if (!d_rgbL[b] || !d_rgbR[b]) return UGSM_ERR_BAD_ARG;
    if (stride < 3 * W) return UGSM_ERR_SIZE_MISMATCH;
    UCHK(prepare_slot(ctx, s, W, H, nb));
    // the pyramid kernels of the fused path check every value they write (level 0 holds the integers 0..255)
    s.range_known = ctx->cfg.kernel_path != 1 && s.range_bad != nullptr;
    if (s.range_known) HIPCHK(ctx, hipMemsetAsync(s.range_bad, 0, sizeof(unsigned) * nb, s.st));
    // One stream, as in rounds 1 and 2: the one-stage-per-kernel path, a batch, and whenever launches are bracketed by events (the
    // statistics belong to one stream).  Otherwise fork: R's pyramid and the A planes on the side stream.
    const bool fork = side_stream_ok(ctx, s);
    s.a_from = -1;
    if (!fork) {
        UCHK(build_pyramids(ctx, s, si, d_rgbL, stride, s.pyrL, nullptr, win));
        UCHK(build_pyramids(ctx, s, si, d_rgbR, stride, s.pyrR, nullptr, win));
        s.have_pyr = nb == 1 && !win;  // (the fovea-shard entry points work on single pairs, and on whole pyramids)
        return UGSM_OK;
    }
    // the side stream starts after everything enqueued on this slot so far (the previous pair still reads pyrR and Apyr)
    HIPCHK(ctx, hipEventRecord(s.ev_in, s.st));
    s.forked = true;
    HIPCHK(ctx, hipStreamWaitEvent(s.st2, s.ev_in, 0));
    UCHK(build_pyramids(ctx, s, si, d_rgbR, stride, s.pyrR, s.st2, win));
    HIPCHK(ctx, hipEventRecord(s.ev_R, s.st2));
    UCHK(build_pyramids(ctx, s, si, d_rgbL, stride, s.pyrL, nullptr, win));
    if (a_from >= 0) UCHK(enqueue_side_A(ctx, s, a_from));
    HIPCHK(ctx, hipStreamWaitEvent(s.st, s.ev_R, 0));
    HIPCHK(ctx, hipGetLastError());
    s.have_pyr = !win;
    return UGSM_OK;
}
int enqueue_pyramids(ugsm_ctx *ctx, Slot &s, int si, const uint8_t *d_rgbL, const uint8_t *d_rgbR, int W, int H, int stride, int a_from = -1,
                     const FoveaWin *win = nullptr)
{
    return enqueue_pyramids(ctx, s, si, &d_rgbL, &d_rgbR, 1, W, H, stride, a_from, win);
}

// the windows of a foveated call's pairs (n offsets), for enqueue_pyramids
int fovea_windows(const ugsm_ctx *ctx, int W, int H, int n, const int *off_x, const int *off_y, FoveaWin &win)
{
    const int F = ctx->cfg.fovea_levels;
    int w[UGSM_MAX_LEVELS], h[UGSM_MAX_LEVELS];
    UCHK(level_dims(W, H, ctx->cfg.levels, w, h));
    if (F < 2 || F > ctx->cfg.levels || n < 1 || n > kMaxBatch) return UGSM_ERR_BAD_ARG;
    for (int b = 0; b < n; b++) {
        FoveaGeom g;
        fovea_geometry(w, h, F, off_x[b], off_y[b], g);
        win.w = g.fw;
        win.h = g.fh;
        win.x0[b] = g.ox[0];
        win.y0[b] = g.oy[0];
    }
    return UGSM_OK;
}

// A of full-frame level i if it was precomputed on the side stream (the main stream is made to wait for it here), else null
const float *level_A(ugsm_ctx *ctx, Slot &s, int i)
{
    if (s.a_from < 0 || i < s.a_from) return nullptr;
    if (hipStreamWaitEvent(s.st, s.ev_A[i], 0) != hipSuccess) return nullptr;
    (void)ctx;
    return s.Apyr + s.off[i];
}

// matching() with foveatedmatching==0, MatchGPULib.cpp:1196-1318, for the s.nb pairs of the call; d_out: their result buffers.
// swap: the images exchanged (the right-to-left match of the LR check): the right pyramid is the "left" image; A is then computed
// in line (the side stream's A planes belong to the left image).
int enqueue_full(ugsm_ctx *ctx, Slot &s, int si, float *const *d_out, bool swap = false)
{
    const int levels = s.levels, nb = s.nb;
    const float *const pL = swap ? s.pyrR : s.pyrL, *const pR = swap ? s.pyrL : s.pyrR;
    float *cur = s.d0, *other = s.d1;
    const int top = levels - 1;
    for (int b = 0; b < nb; b++)  // U1: zero seed
        HIPCHK(ctx, hipMemsetAsync(cur + b * s.lvl_stride, 0, sizeof(float) * 3 * (size_t)s.w[top] * s.h[top], s.st));
    SeedMap sm[kMaxBatch];
    Img3 Lv[kMaxBatch], Rv[kMaxBatch];
    bool seeded = false;
    for (int i = top; i >= 0; i--) {
        s.cur_level = i;
        const int mi = level_iterations(i);
        // the finest level's last smoothing launch writes the caller's buffer directly (no 193 MB device copy at 16 MP)
        const bool direct = i == 0 && ctx->cfg.kernel_path != 1 && level_smooth(0) > 0 && !(ctx->cfg.early_exit_threshold > 0.0f);
        full_views(s, pL, i, Lv);
        full_views(s, pR, i, Rv);
        UCHK(run_level(ctx, s, si, Lv, Rv, s.w[i], s.h[i], mi, level_smooth(i), i == top, 1, mi, cur, other, nullptr, direct ? d_out : nullptr,
                       seeded ? sm : nullptr, (swap || nb > 1) ? nullptr : level_A(ctx, s, i)));
        if (direct) return UGSM_OK;
        seeded = false;
        if (i > 0) {
            const int np = launch_pairs(ctx, s, s.w[i - 1], s.h[i - 1]);
            if (fuse_seed(ctx, s.w[i - 1], s.h[i - 1], s.alone, np)) {  // the next level's first K-cost launch reads `cur` through the seeding map
                for (int b = 0; b < nb; b++) sm[b] = SeedMap{s.w[i], s.h[i], 0, 0};
                seeded = true;
            } else {
                for_groups(nb, np > 1, [&](Grp g) {
                    Timer t(ctx, &s, si, KC_SEED, (double)s.w[i - 1] * s.h[i - 1] * g.n);
                    const Batch bt = make_batch(s, g);
                    launch_seed(s.st, cur + g.b0 * s.lvl_stride, s.w[i], s.h[i], other + g.b0 * s.lvl_stride, s.w[i - 1], s.h[i - 1], 0, 0, g.n > 1 ? &bt : nullptr);
                });
                std::swap(cur, other);
            }
        }
    }
    for (int b = 0; b < nb; b++)
        HIPCHK(ctx, hipMemcpyAsync(d_out[b], cur + b * s.lvl_stride, sizeof(float) * 3 * (size_t)s.W * s.H, hipMemcpyDeviceToDevice, s.st));
    return UGSM_OK;
}
int enqueue_full(ugsm_ctx *ctx, Slot &s, int si, float *d_out, bool swap = false) { return enqueue_full(ctx, s, si, &d_out, swap); }

// Full mode with the optional LR-consistency check (ugsm_config.lr_check_threshold; no reference counterpart): the match, the match
// with the images exchanged into the slot's own buffer, then one kernel that zeroes the inconsistent confidences of d_out.  (Single pairs.)
int enqueue_full_lr(ugsm_ctx *ctx, Slot &s, int si, float *d_out)
{
    s.lr_ran = false;
    UCHK(enqueue_full(ctx, s, si, d_out));
    const float tau = ctx->cfg.lr_check_threshold;
    if (!(tau > 0.0f)) return UGSM_OK;
    const size_t n3 = 3 * (size_t)s.W * s.H;
    UCHK(grow(ctx, s.lr, s.lr_cap, n3 + 4));
    if (!s.lr_host) HIPCHK(ctx, hipHostMalloc((void **)&s.lr_host, sizeof(unsigned long long), hipHostMallocDefault));
    unsigned long long *cnt = reinterpret_cast<unsigned long long *>(s.lr + ((n3 + 1) & ~(size_t)1));  // (8-byte aligned)
    HIPCHK(ctx, hipMemsetAsync(cnt, 0, sizeof *cnt, s.st));
    UCHK(enqueue_full(ctx, s, si, s.lr, true));
    {
        Timer t(ctx, &s, si, KC_MISC, (double)s.W * s.H);
        launch_lr_check(s.st, d_out, s.lr, s.W, s.H, tau, cnt);
    }
    HIPCHK(ctx, hipMemcpyAsync(s.lr_host, cnt, sizeof *cnt, hipMemcpyDeviceToHost, s.st));
    HIPCHK(ctx, hipGetLastError());
    s.lr_ran = true;
    return UGSM_OK;
}

// a batched plane copy: pair j of group g from src0 + j * lvl_stride (a level buffer) or from its view, to its own destination
Batch copy_batch(const Slot &s, Grp g, const Img3 *views, float *const *dsts, size_t dst_off)
{
    Batch b{};
    b.n = g.n;
    for (int j = 0; j < g.n; j++) {
        b.img[j] = views ? byte_diff(views[g.b0 + j].p, views[g.b0].p) : (long long)(j * s.lvl_stride * sizeof(float));
        b.out[j] = byte_diff(dsts[g.b0 + j] + dst_off, dsts[g.b0] + dst_off);
    }
    return b;
}

// matching() with foveatedmatching==1 (MatchGPULib.cpp:1230-1294), split at level F-1; d_state: one buffer per pair of the call.
int enqueue_fovea_coarse(ugsm_ctx *ctx, Slot &s, int si, float *const *d_state)
{
    const int levels = s.levels, F = ctx->cfg.fovea_levels, nb = s.nb;
    if (F < 2 || F > levels) return UGSM_ERR_BAD_ARG;
    float *cur = s.d0, *other = s.d1;
    const int top = levels - 1;
    for (int b = 0; b < nb; b++)
        HIPCHK(ctx, hipMemsetAsync(cur + b * s.lvl_stride, 0, sizeof(float) * 3 * (size_t)s.w[top] * s.h[top], s.st));
    SeedMap sm[kMaxBatch];
    Img3 Lv[kMaxBatch], Rv[kMaxBatch];
    bool seeded = false;
    for (int i = top; i >= F - 1; i--) {
        s.cur_level = i;
        const int mi = level_iterations(i);
        full_views(s, s.pyrL, i, Lv);
        full_views(s, s.pyrR, i, Rv);
        UCHK(run_level(ctx, s, si, Lv, Rv, s.w[i], s.h[i], mi, level_smooth(i), i == top, 1, mi, cur, other, nullptr, nullptr, seeded ? sm : nullptr,
                       nb > 1 ? nullptr : level_A(ctx, s, i)));
        seeded = false;
        if (i > F - 1) {
            const int np = launch_pairs(ctx, s, s.w[i - 1], s.h[i - 1]);
            if (fuse_seed(ctx, s.w[i - 1], s.h[i - 1], s.alone, np)) {
                for (int b = 0; b < nb; b++) sm[b] = SeedMap{s.w[i], s.h[i], 0, 0};
                seeded = true;
            } else {
                for_groups(nb, np > 1, [&](Grp g) {
                    Timer t(ctx, &s, si, KC_SEED, (double)s.w[i - 1] * s.h[i - 1] * g.n);
                    const Batch bt = make_batch(s, g);
                    launch_seed(s.st, cur + g.b0 * s.lvl_stride, s.w[i], s.h[i], other + g.b0 * s.lvl_stride, s.w[i - 1], s.h[i - 1], 0, 0, g.n > 1 ? &bt : nullptr);
                });
                std::swap(cur, other);
            }
        }
    }
    const size_t fn = (size_t)s.w[F - 1] * s.h[F - 1];
    if (nb == 1) {
        HIPCHK(ctx, hipMemcpyAsync(d_state[0], cur, sizeof(float) * 3 * fn, hipMemcpyDeviceToDevice, s.st));
    } else {
        const Grp g{0, nb};
        const Batch bt = copy_batch(s, g, nullptr, d_state, 0);
        launch_copy_view(s.st, Img3{cur, s.w[F - 1], fn}, s.w[F - 1], s.h[F - 1], d_state[0], fn, s.w[F - 1], &bt);
    }
    s.have_coarse = nb == 1;
    return UGSM_OK;
}
int enqueue_fovea_coarse(ugsm_ctx *ctx, Slot &s, int si, float *d_state) { return enqueue_fovea_coarse(ctx, s, si, &d_state); }

// The fine phase for the s.nb pairs of the call: per-pair states, window offsets and destinations (d_pyrL / d_pyrR may be null, and so may
// their entries).
int enqueue_fovea_fine(ugsm_ctx *ctx, Slot &s, int si, const float *const *d_state, const int *off_x, const int *off_y, float *const *d_stack,
                       float *const *d_pyrL, float *const *d_pyrR)
{
    const int levels = s.levels, F = ctx->cfg.fovea_levels, nb = s.nb;
    if (F < 2 || F > levels) return UGSM_ERR_BAD_ARG;
    FoveaGeom g[kMaxBatch];
    for (int b = 0; b < nb; b++) fovea_geometry(s.w, s.h, F, off_x[b], off_y[b], g[b]);
    const int fw = g[0].fw, fh = g[0].fh;
    const size_t fn = (size_t)fw * fh;
    const int np = launch_pairs(ctx, s, fw, fh);  // (every level of the fine phase is a window of this size)
    float *cur = s.d0, *other = s.d1;
    for (int b = 0; b < nb; b++)
        HIPCHK(ctx, hipMemcpyAsync(cur + b * s.lvl_stride, d_state[b], sizeof(float) * 3 * fn, hipMemcpyDeviceToDevice, s.st));
    // a stack row block / a pyramid-stack block for every pair: one launch for the batch
    auto copy_out = [&](const Img3 *views, const float *field, float *const *dsts, size_t dst_off, size_t dst_plane) {
        for_groups(nb, np > 1, [&](Grp gr) {
            const Batch bt = copy_batch(s, gr, views, dsts, dst_off);
            const Img3 src = views ? views[gr.b0] : Img3{field + gr.b0 * s.lvl_stride, fw, fn};
            launch_copy_view(s.st, src, fw, fh, dsts[gr.b0] + dst_off, dst_plane, fw, gr.n > 1 ? &bt : nullptr);
        });
    };
    // stack row block F-1 = the whole level F-1 (UG_GPU_matcher.cpp:293-303)
    copy_out(nullptr, cur, d_stack, (size_t)(F - 1) * fn, (size_t)F * fn);
    SeedMap sm[kMaxBatch];
    Img3 Lv[kMaxBatch], Rv[kMaxBatch];
    for (int i = F - 2; i >= 0; i--) {
        s.cur_level = i;
        // foveatedsubsampleDisp, MatchGPULib.cpp:1595-1655
        for (int b = 0; b < nb; b++) {
            sm[b] = SeedMap{fw, fh, g[b].cx[i], g[b].cy[i]};
            Lv[b] = level_view(s, s.pyrL + b * s.pyr_stride, i, g[b].ox[i], g[b].oy[i]);
            Rv[b] = level_view(s, s.pyrR + b * s.pyr_stride, i, g[b].ox[i], g[b].oy[i]);
        }
        const bool seeded = fuse_seed(ctx, fw, fh, s.alone, np);
        if (!seeded) {
            for_groups(nb, np > 1, [&](Grp gr) {
                Timer t(ctx, &s, si, KC_SEED, (double)fn * gr.n);
                const Batch bt = make_batch(s, gr, nullptr, sm);
                launch_seed(s.st, cur + gr.b0 * s.lvl_stride, fw, fh, other + gr.b0 * s.lvl_stride, fw, fh, sm[gr.b0].cx, sm[gr.b0].cy, gr.n > 1 ? &bt : nullptr);
            });
            std::swap(cur, other);
        }
        const int mi = level_iterations(i);
        UCHK(run_level(ctx, s, si, Lv, Rv, fw, fh, mi, level_smooth(i), false, 1, mi, cur, other, nullptr, nullptr, seeded ? sm : nullptr));
        copy_out(nullptr, cur, d_stack, (size_t)i * fn, (size_t)F * fn);
    }
    // pyramid stacks as the node publishes them (UG_GPU_matcher.cpp:203-213): [level][channel][row]
    for (int side = 0; side < 2; side++) {
        float *const *dsts = side == 0 ? d_pyrL : d_pyrR;
        if (!dsts) continue;
        bool all = true, any = false;
        for (int b = 0; b < nb; b++) {
            all = all && dsts[b] != nullptr;
            any = any || dsts[b] != nullptr;
        }
        if (!any) continue;
        const float *pyr = side == 0 ? s.pyrL : s.pyrR;
        for (int k = 0; k < F; k++) {
            for (int b = 0; b < nb; b++) {
                const int ox = (k < F - 1) ? g[b].ox[k] : 0, oy = (k < F - 1) ? g[b].oy[k] : 0;
                Lv[b] = level_view(s, pyr + b * s.pyr_stride, k, ox, oy);
            }
            if (all) {
                copy_out(Lv, nullptr, dsts, (size_t)k * 3 * fn, fn);
            } else {
                for (int b = 0; b < nb; b++)
                    if (dsts[b]) launch_copy_view(s.st, Lv[b], fw, fh, dsts[b] + (size_t)k * 3 * fn, fn, fw);
            }
        }
    }
    HIPCHK(ctx, hipGetLastError());
    return UGSM_OK;
}
int enqueue_fovea_fine(ugsm_ctx *ctx, Slot &s, int si, const float *d_state, int off_x, int off_y, float *d_stack, float *d_pyrL, float *d_pyrR)
{
    return enqueue_fovea_fine(ctx, s, si, &d_state, &off_x, &off_y, &d_stack, d_pyrL ? &d_pyrL : nullptr, d_pyrR ? &d_pyrR : nullptr);
}

// Has everything enqueued on the slot finished?  (Never blocks; a slot found idle stops counting as busy.)
bool slot_idle(Slot &s)
{
    if (!s.busy) return true;
    const hipError_t e = s.done_recorded ? hipEventQuery(s.ev_done) : hipStreamQuery(s.st);
    if (e == hipErrorNotReady) return false;
    if (e != hipSuccess) (void)hipGetLastError();  // (the slot's own ugsm_wait / ugsm_poll reports it)
    s.busy = false;
    return true;
}
// Does the call about to be enqueued on `slot` have the chip to itself?  It does when nothing is unfinished on any other slot of the
// context and the dispatcher has not said that more calls follow (the queue, ugsm_queue.cpp: pairs wait behind this call, or a call
// filled up by itself -- a host that submits faster than the chip matches).  The blocking entry points (ugsm_match_*: the node's service
// call, UG_GPU_matcher.cpp:497-694, and its one-at-a-time topic path, :126-185) are therefore alone whatever ugsm_config.slots says; a
// burst through the slot API is alone for its first call only.  What is in flight decides -- not how many slots the context was created
// with.
bool call_alone(ugsm_ctx *ctx, int slot)
{
    if (ctx->force_alone >= 0) return ctx->force_alone == 1;
    if (ctx->hooks.queue_calling && ctx->hooks.queue_more) return false;
    for (int i = 0; i < (int)ctx->slots.size(); i++)
        if (i != slot && !slot_idle(ctx->slots[i])) return false;
    return true;
}

int get_slot(ugsm_ctx *ctx, int slot, Slot **out, bool enqueues = true)
{
    if (!ctx) return UGSM_ERR_BAD_ARG;
    if (slot < 0 || slot >= (int)ctx->slots.size()) {
        ctx->err = "slot out of range";
        return UGSM_ERR_BAD_ARG;
    }
    if (enqueues && ctx->hooks.queue_busy && !ctx->hooks.queue_calling) {
        ctx->err = "pairs enqueued with ugsm_enqueue_* are outstanding: the slots belong to the queue until ugsm_next_done has reported them all";
        return UGSM_ERR_STATE;
    }
    *out = &ctx->slots[slot];
    if (enqueues) {
        (*out)->alone = call_alone(ctx, slot);
        (*out)->busy = true;
        (*out)->done_recorded = false;  // whatever this call enqueues comes after the slot's last completion mark
    }
    return UGSM_OK;
}

// End of a ugsm_submit_*: the slot's completion mark.  With several slots on one stream (ugsm_config.streams < slots) ugsm_wait waits for
// this event, not for the pairs other slots have queued behind it on the same stream.
int mark_done(ugsm_ctx *ctx, Slot &s)
{
    s.forked = false;  // (the call was enqueued whole: the main stream waits for everything it put on the side stream)
    if (ctx->streams >= (int)ctx->slots.size()) return UGSM_OK;
    HIPCHK(ctx, hipEventRecord(s.ev_done, s.st));
    s.done_recorded = true;
    return UGSM_OK;
}

int stage_in(ugsm_ctx *ctx, Slot &s, const uint8_t *rgbL, const uint8_t *rgbR, int W, int H, int stride)
{
    if (!rgbL || !rgbR || W < 1 || H < 1) return UGSM_ERR_BAD_ARG;
    if (stride < 3 * W) return UGSM_ERR_SIZE_MISMATCH;
    const size_t bytes = (size_t)stride * H;
    if (bytes > s.rgb_cap) {
        size_t c = s.rgb_cap;
        UCHK(grow(ctx, s.rgbL, c, bytes));
        UCHK(grow(ctx, s.rgbR, s.rgb_cap, bytes));
    }
    HIPCHK(ctx, hipMemcpyAsync(s.rgbL, rgbL, bytes, hipMemcpyHostToDevice, s.st));
    // the right image goes up on the side stream, where its pyramid is built: the transfer (0.9 ms at 16 MP) then runs under the
    // left pyramid instead of in front of it.  The side stream first waits for what the slot did before.
    if (s.alone && s.st2 && ctx->cfg.kernel_path != 1 && ctx->cfg.profile_events == 0 && ctx->two_streams) {  // (single pairs: as side_stream_ok)
        HIPCHK(ctx, hipEventRecord(s.ev_in, s.st));
        s.forked = true;
        HIPCHK(ctx, hipStreamWaitEvent(s.st2, s.ev_in, 0));
        HIPCHK(ctx, hipMemcpyAsync(s.rgbR, rgbR, bytes, hipMemcpyHostToDevice, s.st2));
    } else {
        HIPCHK(ctx, hipMemcpyAsync(s.rgbR, rgbR, bytes, hipMemcpyHostToDevice, s.st));
    }
    return UGSM_OK;
}

// Result planes into caller memory.  Page-locked destinations (ugsm_host_alloc, or memory the caller registered) take the
// device-to-host copies directly.  Pageable ones go through a page-locked staging buffer and a team of host threads: the
// reference node allocates fresh result planes for every call (UG_GPU_matcher.cpp:414-418), and the first touch of 193 MB of
// fresh pages by the single copy thread of a pageable hipMemcpy costs more than the whole match (30.7 ms against 16.6 ms per
// 16 MP pair); spread over the team, and with plane k+1's transfer running under plane k's copy, it costs ~2 ms.
bool is_pinned(const void *p)
{
    hipPointerAttribute_t a;
    if (hipPointerGetAttributes(&a, p) != hipSuccess) {
        (void)hipGetLastError();  // an ordinary malloc'd pointer is "invalid value" to the runtime: clear the sticky error
        return false;
    }
    return a.type == hipMemoryTypeHost;
}

// The context's host team (UGSM_COPY_THREADS members, default min(8, hardware threads); 1 = the calling thread only).  Never throws.
HostTeam *host_team(ugsm_ctx *ctx)
{
    if (!ctx->team) {
        const char *e = getenv("UGSM_COPY_THREADS");
        unsigned n = e ? (unsigned)atoi(e) : std::min(8u, std::max(1u, std::thread::hardware_concurrency()));
        n = std::min(std::max(1u, n), 64u);
        try {
            ctx->team.reset(new HostTeam(n));
        } catch (...) {  // (allocation failure of the team object itself)
            return nullptr;
        }
    }
    return ctx->team.get();
}

constexpr size_t kTeamMinBytes = 1u << 22;  // below 4 MB a single memcpy / page walk is faster than waking the team

void team_copy(ugsm_ctx *ctx, void *dst, const void *src, size_t bytes)
{
    HostTeam *team = bytes >= kTeamMinBytes ? host_team(ctx) : nullptr;
    if (!team || team->size() == 1) {
        memcpy(dst, src, bytes);
        return;
    }
    team->run([=](unsigned t, unsigned n) {
        const size_t chunk = ((bytes / n) + 4095) & ~(size_t)4095;
        const size_t off = (size_t)t * chunk;
        if (off < bytes) memcpy((char *)dst + off, (const char *)src + off, std::min(chunk, bytes - off));
    });
}

// First touch of the caller's (possibly fresh) result pages by the host team WHILE the GPU is still matching: every page of
// dst[0..2] gets one byte written (the planes are overwritten in full afterwards), so the page faults are off the critical path.
void prefault_planes(ugsm_ctx *ctx, float *const dst[3], size_t plane_floats)
{
    const size_t pb = plane_floats * sizeof(float);
    if (pb < kTeamMinBytes) return;  // (a small call: the copy itself touches the pages)
    if (is_pinned(dst[0]) && is_pinned(dst[1]) && is_pinned(dst[2])) return;
    auto touch = [=](unsigned t, unsigned n) {
        for (int k = 0; k < 3; k++) {
            volatile char *p = (volatile char *)dst[k];
            const size_t lo = pb * t / n, hi = pb * (t + 1) / n;
            for (size_t o = (lo + 4095) & ~(size_t)4095; o < hi; o += 4096) p[o] = 0;
            if (t == 0 && pb) p[0] = 0;
        }
    };
    HostTeam *team = host_team(ctx);
    if (team) team->run(touch);
    else touch(0, 1);
}

int copy_out_planes(ugsm_ctx *ctx, Slot &s, const float *d_src, size_t plane_floats, float *const dst[3])
{
    const size_t pb = plane_floats * sizeof(float);
    if (is_pinned(dst[0]) && is_pinned(dst[1]) && is_pinned(dst[2])) {
        for (int k = 0; k < 3; k++) HIPCHK(ctx, hipMemcpyAsync(dst[k], d_src + k * plane_floats, pb, hipMemcpyDeviceToHost, s.st));
        return UGSM_OK;
    }
    if (3 * plane_floats > s.hpin_cap) {
        if (s.hpin) HIPCHK(ctx, hipHostFree(s.hpin));
        s.hpin = nullptr;
        s.hpin_cap = 0;
        HIPCHK(ctx, hipHostMalloc((void **)&s.hpin, 3 * pb, hipHostMallocDefault));
        s.hpin_cap = 3 * plane_floats;
    }
    // Every plane in PIECES: a piece's copy into the caller's pages runs under the next piece's transfer, so what is left exposed behind
    // the last transfer is one piece's copy, not one plane's (a quarter plane at 16 MP: 0.25 ms instead of 1 ms).
    const int pieces = pb >= 4 * kTeamMinBytes ? 4 : 1;
    const size_t pf = ((plane_floats + pieces - 1) / pieces + 1023) & ~(size_t)1023;  // floats per piece (4 KiB multiples)
    auto piece = [&](int i, size_t &off, size_t &len) {
        const int k = i / pieces, c = i - k * pieces;
        const size_t lo = std::min((size_t)c * pf, plane_floats), hi = std::min(lo + pf, plane_floats);
        off = (size_t)k * plane_floats + lo;
        len = hi - lo;
        return k;
    };
    for (int i = 0; i < 3 * pieces; i++) {
        size_t off, len;
        piece(i, off, len);
        if (!s.out_ev[i]) HIPCHK(ctx, hipEventCreateWithFlags(&s.out_ev[i], hipEventDisableTiming));
        if (len) HIPCHK(ctx, hipMemcpyAsync(s.hpin + off, d_src + off, len * sizeof(float), hipMemcpyDeviceToHost, s.st));
        HIPCHK(ctx, hipEventRecord(s.out_ev[i], s.st));
    }
    for (int i = 0; i < 3 * pieces; i++) {
        size_t off, len;
        const int k = piece(i, off, len);
        HIPCHK(ctx, hipEventSynchronize(s.out_ev[i]));
        if (len) team_copy(ctx, (char *)dst[k] + (off - (size_t)k * plane_floats) * sizeof(float), s.hpin + off, len * sizeof(float));
    }
    return UGSM_OK;
}

}  // namespace

namespace ugsm {
void ctx_host_copy(ugsm_ctx *ctx, void *dst, const void *src, size_t bytes) { team_copy(ctx, dst, src, bytes); }
bool host_pinned(const void *p) { return is_pinned(p); }
bool dev_env() { return dev_env_on(); }
}  // namespace ugsm

// =========================================================================================
// C-ABI
// =========================================================================================
extern "C" {

void ugsm_default_config(ugsm_config *cfg)
{
    if (!cfg) return;
    memset(cfg, 0, sizeof *cfg);
    cfg->device = 0;
    cfg->levels = 14;       // MAX_LEVEL, MatchLib_common.h:13
    cfg->fovea_levels = 7;  // MatchGPULib.cpp:263
    cfg->slots = 1;
    cfg->kernel_path = 0;
    cfg->profile_events = 0;
}

int ugsm_abi_version(void) { return UGSM_ABI_VERSION; }
int ugsm_is_dev_library(void) { return kDevLib ? 1 : 0; }

const char *ugsm_status_string(int st)
{
    switch (st) {
    case UGSM_OK: return "ok";
    case UGSM_ERR_BAD_ARG: return "bad argument";
    case UGSM_ERR_SIZE_MISMATCH: return "size mismatch";
    case UGSM_ERR_TOO_SMALL: return "image too small for the requested number of pyramid levels";
    case UGSM_ERR_NO_DEVICE: return "no HIP device";
    case UGSM_ERR_DEVICE: return "HIP runtime error";
    case UGSM_ERR_NOMEM: return "out of device memory";
    case UGSM_ERR_STATE: return "call sequence error";
    case UGSM_PENDING: return "not finished yet";
    case UGSM_EMPTY: return "nothing outstanding";
    case UGSM_ERR_PEER: return "a peer rank of the fovea shard failed or did not answer";
    default: return "unknown status";
    }
}

int ugsm_create(const ugsm_config *cfg_in, ugsm_ctx **out)
{
    if (!out) return UGSM_ERR_BAD_ARG;
    *out = nullptr;
    ugsm_config cfg;
    if (cfg_in) cfg = *cfg_in;
    else ugsm_default_config(&cfg);
    DevKnobs knobs;
    {
        std::lock_guard<std::mutex> lk(g_globals_mutex);
        apply_dev_env(cfg, knobs, g_live_contexts == 0);  // (nothing unless UGSM_DEV=1)
    }
    {   // the kernels carry the Gaussian taps as literals (ugsm_device.hpp); they must be the numbers the reference computes at
        // start-up: five float literals divided by their float sum (MatchGPULib.cpp:761-774)
        const float lit[5] = {0.0816475f, 0.218507f, 0.303281f, 0.218507f, 0.0816475f};
        volatile float sum = 0.0f;
        for (float v : lit) sum = sum + v;
        const float g[3] = {lit[0] / sum, lit[1] / sum, lit[2] / sum}, k[3] = {UGSM_G0, UGSM_G1, UGSM_G2};
        if (memcmp(g, k, sizeof g) != 0) return UGSM_ERR_STATE;
    }
    if (cfg.levels < 1 || cfg.levels > UGSM_MAX_LEVELS || cfg.slots < 1 || cfg.slots > 64 || cfg.kernel_path < 0 ||
        cfg.kernel_path > 1 || cfg.fovea_levels < 0 || cfg.fovea_levels > cfg.levels || !(cfg.lr_check_threshold >= 0.0f) || cfg.streams < 0 ||
        cfg.batch < 0 || cfg.batch > UGSM_MAX_BATCH || cfg.stream_priority < 0 || cfg.stream_priority > 3)
        return UGSM_ERR_BAD_ARG;
    if (cfg.kernel_path == 1 && !kDevLib) return UGSM_ERR_BAD_ARG;       // the one-kernel-per-stage path lives in libugsm_dev.so
    if (cfg.march_min_pixels < 0 && !kDevLib) return UGSM_ERR_BAD_ARG;  // ... and so does round 1's LDS-tiled K-cost
    static_assert(UGSM_MAX_BATCH == kMaxBatch, "include/ugsm.h and ugsm_launch.hpp disagree on the batch size");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1) return UGSM_ERR_NO_DEVICE;
    if (cfg.device < 0 || cfg.device >= ndev) return UGSM_ERR_BAD_ARG;
    if (hipSetDevice(cfg.device) != hipSuccess) return UGSM_ERR_NO_DEVICE;
    ugsm_ctx *ctx = new ugsm_ctx();
    ctx->cfg = cfg;
    set_policy(ctx, knobs);
    if (dev_env_on() && getenv("UGSM_MEM_LIMIT_MB")) ctx->mem_limit = atoll(getenv("UGSM_MEM_LIMIT_MB")) << 20;
    // The side stream pays when a pair is alone on the chip (115.6 against 113.9 pairs/s at 16 MP: the right pyramid and the A planes run
    // beside the left pyramid and the coarse levels; profiles/r06_ab_alone.txt).  With four pairs in flight USING theirs it loses 13 % (136
    // against 157 pairs/s): eight streams on the four hardware queues serialise kernels that one stream per pair lets overlap.  So every
    // slot has one, and only a call that is alone on the chip uses it (side_stream_ok).
    ctx->two_streams = knobs.two_streams >= 0 ? knobs.two_streams : 1;
    ctx->slots.resize(cfg.slots);
    int prio_least = 0, prio_greatest = 0;
    (void)hipDeviceGetStreamPriorityRange(&prio_least, &prio_greatest);
    // ugsm_config.streams: fewer streams than slots = several pairs QUEUED per stream (slot i on the stream of slot i % streams): the
    // next pair of a stream is already enqueued when the one before it ends, where a slot of its own stream idles while the host
    // notices the end and enqueues ~600 launches (1.6 ms of a pair's 24 ms in flight at 16 MP)
    ctx->streams = cfg.streams > 0 ? std::min(cfg.streams, cfg.slots) : cfg.slots;
    for (int si = 0; si < cfg.slots; si++) {
        Slot &s = ctx->slots[si];
        if (si >= ctx->streams) {
            s.st = ctx->slots[si % ctx->streams].st;
            s.owns_st = false;
        }
        // Every slot's stream on a hardware queue of its own.  HIP deals a process's streams onto GPU_MAX_HW_QUEUES (4) hardware
        // queues PER PRIORITY LEVEL, least-used first, and two streams that share a queue run their kernels strictly one after
        // the other (tools/queue_probe).  At the default priority the host application's streams -- the null stream any hipMemcpy
        // uses, for a start -- hold queues of the same pool, and four slots then land on three queues: 128 pairs/s instead of 165
        // at 16 MP (tools/ab.py; round 2 measured 161 only because its idle side streams happened to push the slots apart).  So:
        // slots 0-3 at the greatest priority (a pool the application is unlikely to use), slots 4-7 at the least, the rest at the
        // default.  Equal priority among the first four.
        // ugsm_config.stream_priority: 1 = everything at the process default (opt out), 2 / 3 = everything at the greatest / least
        const char by_cfg = cfg.stream_priority == 1 ? 'n' : (cfg.stream_priority == 2 ? 'h' : (cfg.stream_priority == 3 ? 'l' : (si < 4 ? 'h' : (si < 8 ? 'l' : 'n'))));
        const char pc = knobs.stream_prio[0] ? (si < (int)strlen(knobs.stream_prio) ? knobs.stream_prio[si] : 'n') : by_cfg;
        const int prio = pc == 'h' ? prio_greatest : (pc == 'l' ? prio_least : 0);
        bool ok = (!s.owns_st || hipStreamCreateWithPriority(&s.st, hipStreamNonBlocking, prio) == hipSuccess) &&
                  hipMalloc((void **)&s.range_bad, 64) == hipSuccess && hipEventCreateWithFlags(&s.ev_done, hipEventDisableTiming) == hipSuccess;
        s.prio = prio;
        for (hipEvent_t *e : {&s.ev_in, &s.ev_L, &s.ev_R}) ok = ok && hipEventCreateWithFlags(e, hipEventDisableTiming) == hipSuccess;
        for (int i = 0; i < cfg.levels; i++) ok = ok && hipEventCreateWithFlags(&s.ev_A[i], hipEventDisableTiming) == hipSuccess;
        if (!ok) {
            ugsm_destroy(ctx);
            return UGSM_ERR_DEVICE;
        }
    }
    // The side streams.  A call forks onto one only when it is alone on the chip -- every other slot idle -- so a slot BORROWS the stream of
    // a neighbour: no stream is created that calls in flight would leave idle.  What decided this (profiles/r06_ab_side_streams.txt,
    // r06_ab_lone_call.txt): side streams of the slots' own, created between the slots' streams in another priority pool, cost a context with
    // several calls in flight 28-38 % though never used (which queues the slots' streams land on depends on what is created around them);
    // created after them they cost such a context nothing in either pool, but eight streams on the pool's four hardware queues left
    // a lone call on a four-slot context 2-3 % behind the same call on a one-slot context (one of a slot's two streams shares a queue with a
    // neighbour's).  Borrowed, a lone call's two streams are two slots' streams: two hardware queues wherever the slots' streams have one each.
    // One stream in the whole context (one slot, or ugsm_config.streams = 1): one more stream, created here, serves every slot.
    // UGSM_SIDE_PRIO (development) asks for the streams of rounds 3-6a: one per slot, created for slot 1, 2, ..., 0, in the pool named.
    for (int k = 0; k < cfg.slots && ctx->two_streams; k++) {
        if (!knobs.side_prio) {
            Slot &s = ctx->slots[k];
            if (ctx->streams >= 2) {
                // (the slot BEFORE it in the rotation: a burst's first call is alone and forks -- the burst's second call then goes to the
                // next slot, whose stream holds no borrowed work in front of it)
                s.st2 = ctx->slots[(k % ctx->streams + ctx->streams - 1) % ctx->streams].st;
                continue;
            }
            if (k > 0) {
                s.st2 = ctx->slots[0].st2;
                continue;
            }
        }
        Slot &s = ctx->slots[knobs.side_prio ? (k + 1) % cfg.slots : k];
        const int side = knobs.side_prio == 'h' ? prio_greatest : (knobs.side_prio == 'l' ? prio_least : (knobs.side_prio == 'n' ? 0 : s.prio));
        if (hipStreamCreateWithPriority(&s.st2, hipStreamNonBlocking, side) != hipSuccess) {
            ugsm_destroy(ctx);
            return UGSM_ERR_DEVICE;
        }
        s.owns_st2 = true;
    }
    {
        std::lock_guard<std::mutex> lk(g_globals_mutex);
        g_live_contexts++;
        ctx->counted_live = true;
    }
    *out = ctx;
    return UGSM_OK;
}

void ugsm_destroy(ugsm_ctx *ctx)
{
    if (!ctx) return;
    if (ctx->counted_live) {
        std::lock_guard<std::mutex> lk(g_globals_mutex);
        g_live_contexts--;
    }
    (void)hipSetDevice(ctx->cfg.device);
    for (Slot &s : ctx->slots) {  // (the layers' state refers to work on the slots' streams: drain them first)
        if (s.st) (void)hipStreamSynchronize(s.st);
        if (s.st2) (void)hipStreamSynchronize(s.st2);
    }
    if (ctx->hooks.queue && ctx->hooks.queue_free) ctx->hooks.queue_free(ctx, ctx->hooks.queue);
    if (ctx->hooks.shard && ctx->hooks.shard_free) ctx->hooks.shard_free(ctx, ctx->hooks.shard);
    ctx->hooks = CtxHooks{};
    for (Slot &s : ctx->slots) {
        if (s.st) (void)hipStreamSynchronize(s.st);
        if (s.st2) (void)hipStreamSynchronize(s.st2);
        harvest(ctx, s);
        for (hipEvent_t e : s.pool) (void)hipEventDestroy(e);
        for (void *p : {(void *)s.pyrL, (void *)s.pyrR, (void *)s.A, (void *)s.Rw, (void *)s.B, (void *)s.d0, (void *)s.d1,
                        (void *)s.rgbL, (void *)s.rgbR, (void *)s.hout, (void *)s.range_bad, (void *)s.d2, (void *)s.wd_rows})
            if (p) (void)hipFree(p);
        if (s.wd_host) (void)hipHostFree(s.wd_host);
        if (s.hpin) (void)hipHostFree(s.hpin);
        for (hipEvent_t e : s.out_ev)
            if (e) (void)hipEventDestroy(e);
        for (hipEvent_t e : {s.ev_in, s.ev_L, s.ev_R})
            if (e) (void)hipEventDestroy(e);
        for (hipEvent_t e : s.ev_A)
            if (e) (void)hipEventDestroy(e);
        if (s.Apyr) (void)hipFree(s.Apyr);
        if (s.lr) (void)hipFree(s.lr);
        if (s.lr_host) (void)hipHostFree(s.lr_host);
        if (s.st2 && s.owns_st2) (void)hipStreamDestroy(s.st2);
        if (s.st && s.owns_st) (void)hipStreamDestroy(s.st);
        if (s.ev_done) (void)hipEventDestroy(s.ev_done);
    }
    delete ctx;
}

const char *ugsm_last_error(const ugsm_ctx *ctx) { return ctx ? ctx->err.c_str() : "null context"; }

int ugsm_level_dims(int W, int H, int levels, int *w, int *h)
{
    if (!w || !h) return UGSM_ERR_BAD_ARG;
    return level_dims(W, H, levels, w, h);
}
int ugsm_level_iterations(int level) { return level < 0 ? 0 : level_iterations(level); }
int ugsm_level_smooth_passes(int level) { return level < 0 ? 0 : level_smooth(level); }

int ugsm_plan_level(const ugsm_config *cfg_in, int alone, int W, int H, ugsm_level_plan *out)
{
    if (!out || W < 1 || H < 1) return UGSM_ERR_BAD_ARG;
    ugsm_ctx probe;  // host-only: the same policy functions the launch path calls, on a context that owns no device state
    if (cfg_in) probe.cfg = *cfg_in;
    else ugsm_default_config(&probe.cfg);
    {   // ... under the same development overrides ugsm_create would apply in this process (none unless UGSM_DEV=1)
        DevKnobs knobs;
        apply_dev_env(probe.cfg, knobs, false);
        set_policy(&probe, knobs);
    }
    memset(out, 0, sizeof *out);
    if (probe.cfg.kernel_path == 1) {
        out->cost_kernel = out->smooth_kernel = 3;
        out->pairs_per_launch = 1;
        return UGSM_OK;
    }
    // a context created for batches (ugsm_config.batch) is asked about a call of that many pairs: a level of at most kBatchMaxPixels is one
    // launch for all of them, and the thresholds are compared with what the launch holds
    const int nb = std::min(std::max(probe.cfg.batch, 1), kMaxBatch);
    const bool al = probe.force_alone >= 0 ? probe.force_alone == 1 : alone != 0;
    out->alone = al ? 1 : 0;
    const int pairs = (nb > 1 && batch_level(&probe, W, H)) ? nb : 1;
    const bool march4 = use_march4(&probe, W, H, al, pairs);  // (asked first, as in run_level)
    const bool march = use_march(&probe, W, H, pairs);
    const int rh = small_rh(&probe, W, H, al, pairs);
    out->cost_kernel = march4 ? 4 : (march ? 1 : ((rh && (probe.small_mask & 1)) ? 2 : 0));
    out->smooth_kernel = (rh && (probe.small_mask & 2)) ? 2 : 0;
    out->smooth_rh = (probe.small_mask & 2) ? rh : 0;
    out->strip_rows = march4 ? march4_strip_rows(W, H, pairs)
                             : (march ? (probe.cfg.march_rows > 0 ? probe.cfg.march_rows : march_strip_rows(W, H, probe.march_mode <= -2, pairs)) : 0);
    out->seed_fused = fuse_seed(&probe, W, H, al, pairs) ? 1 : 0;
    out->smooth_tile_rows = out->smooth_kernel == 0 ? smooth_rows_for(&probe, W, H, al, pairs) : 0;
    out->pairs_per_launch = pairs;
    return UGSM_OK;
}
int ugsm_threshold_schedule(int mi, float *out)
{
    if (mi < 1 || !out) return UGSM_ERR_BAD_ARG;
    threshold_schedule(mi, out);
    return UGSM_OK;
}
int ugsm_fovea_dims(int W, int H, int levels, int F, int *fovW, int *fovH)
{
    int w[UGSM_MAX_LEVELS], h[UGSM_MAX_LEVELS];
    if (!fovW || !fovH || F < 1 || F > levels) return UGSM_ERR_BAD_ARG;
    UCHK(level_dims(W, H, levels, w, h));
    *fovW = w[F - 1];  // MatchGPULib.cpp:419-420
    *fovH = h[F - 1];
    return UGSM_OK;
}
long long ugsm_pixel_iterations(int W, int H, int levels, int F)
{
    int w[UGSM_MAX_LEVELS], h[UGSM_MAX_LEVELS];
    if (level_dims(W, H, levels, w, h) != UGSM_OK) return -1;
    long long tot = 0;
    for (int i = 0; i < levels; i++) {
        long long px = (F > 0 && i < F - 1) ? (long long)w[F - 1] * h[F - 1] : (long long)w[i] * h[i];
        tot += px * level_iterations(i);
    }
    return tot;
}

int ugsm_submit_pyramids(ugsm_ctx *ctx, int slot, const uint8_t *d_rgbL, const uint8_t *d_rgbR, int W, int H, int stride)
{
    Slot *s;
    UCHK(get_slot(ctx, slot, &s));
    HIPCHK(ctx, hipSetDevice(ctx->cfg.device));
    // (no A planes here: the ranks of a fovea shard that only run the fine phase never use them; a coarse phase that follows computes
    // them on the side stream itself -- ADVICE r03)
    UCHK(enqueue_pyramids(ctx, *s, slot, d_rgbL, d_rgbR, W, H, stride, -1));
    return mark_done(ctx, *s);
}

int ugsm_submit_full(ugsm_ctx *ctx, int slot, const uint8_t *d_rgbL, const uint8_t *d_rgbR, int W, int H, int stride, float *d_out)
{
    Slot *s;
    UCHK(get_slot(ctx, slot, &s));
    if (!d_out) return UGSM_ERR_BAD_ARG;
    HIPCHK(ctx, hipSetDevice(ctx->cfg.device));
    UCHK(enqueue_pyramids(ctx, *s, slot, d_rgbL, d_rgbR, W, H, stride, 0));
    UCHK(enqueue_full_lr(ctx, *s, slot, d_out));
    return mark_done(ctx, *s);
}

int ugsm_submit_fovea_coarse(ugsm_ctx *ctx, int slot, float *d_state)
{
    Slot *s;
    UCHK(get_slot(ctx, slot, &s));
    if (!d_state) return UGSM_ERR_BAD_ARG;
    if (!s->have_pyr) return UGSM_ERR_STATE;
    HIPCHK(ctx, hipSetDevice(ctx->cfg.device));
    if (s->a_from < 0 && side_stream_ok(ctx, *s) && ctx->cfg.fovea_levels >= 2) UCHK(enqueue_side_A(ctx, *s, ctx->cfg.fovea_levels - 1));
    UCHK(enqueue_fovea_coarse(ctx, *s, slot, d_state));
    return mark_done(ctx, *s);
}

int ugsm_submit_fovea_fine(ugsm_ctx *ctx, int slot, const float *d_state, int off_x, int off_y, float *d_stack)
{
    Slot *s;
    UCHK(get_slot(ctx, slot, &s));
    if (!d_state || !d_stack) return UGSM_ERR_BAD_ARG;
    if (!s->have_pyr) return UGSM_ERR_STATE;
    HIPCHK(ctx, hipSetDevice(ctx->cfg.device));
    UCHK(enqueue_fovea_fine(ctx, *s, slot, d_state, off_x, off_y, d_stack, nullptr, nullptr));
    return mark_done(ctx, *s);
}

int ugsm_submit_foveated(ugsm_ctx *ctx, int slot, const uint8_t *d_rgbL, const uint8_t *d_rgbR, int W, int H, int stride,
                         int off_x, int off_y, float *d_stack, float *d_pyrL, float *d_pyrR)
{
    Slot *s;
    UCHK(get_slot(ctx, slot, &s));
    if (!d_stack) return UGSM_ERR_BAD_ARG;
    HIPCHK(ctx, hipSetDevice(ctx->cfg.device));
    const int F = ctx->cfg.fovea_levels;
    if (F < 2) return UGSM_ERR_BAD_ARG;
    FoveaWin win;
    UCHK(fovea_windows(ctx, W, H, 1, &off_x, &off_y, win));
    UCHK(enqueue_pyramids(ctx, *s, slot, d_rgbL, d_rgbR, W, H, stride, F - 1, &win));
    // level F-1's state is parked in the (otherwise idle) A buffer's tail? No: use a dedicated spot
    // at the end of d_stack's level F-1 block is not 3-plane contiguous, so stage through hout.
    const size_t fn3 = 3 * (size_t)s->w[F - 1] * s->h[F - 1];
    UCHK(grow(ctx, s->hout, s->hout_cap, std::max(fn3, s->hout_cap)));
    UCHK(enqueue_fovea_coarse(ctx, *s, slot, s->hout));
    UCHK(enqueue_fovea_fine(ctx, *s, slot, s->hout, off_x, off_y, d_stack, d_pyrL, d_pyrR));
    return mark_done(ctx, *s);
}

// ---- B pairs per call ---------------------------------------------------------------------------------------------------------
// Contexts whose options need a host round trip per iteration, a second match, or the one-kernel-per-stage path: pair by pair.
static bool batch_runs_pair_by_pair(const ugsm_ctx *ctx)
{
    return ctx->cfg.kernel_path == 1 || ctx->cfg.early_exit_threshold > 0.0f || ctx->cfg.lr_check_threshold > 0.0f;
}

int ugsm_submit_full_batch(ugsm_ctx *ctx, int slot, int n, const uint8_t *const *d_rgbL, const uint8_t *const *d_rgbR, int W, int H, int stride,
                           float *const *d_out)
{
    Slot *s;
    UCHK(get_slot(ctx, slot, &s));
    if (n < 1 || n > UGSM_MAX_BATCH || !d_rgbL || !d_rgbR || !d_out) return UGSM_ERR_BAD_ARG;
    for (int b = 0; b < n; b++)
        if (!d_rgbL[b] || !d_rgbR[b] || !d_out[b]) return UGSM_ERR_BAD_ARG;
    HIPCHK(ctx, hipSetDevice(ctx->cfg.device));
    if (n == 1 || batch_runs_pair_by_pair(ctx)) {
        for (int b = 0; b < n; b++) {
            UCHK(enqueue_pyramids(ctx, *s, slot, d_rgbL[b], d_rgbR[b], W, H, stride, 0));
            UCHK(enqueue_full_lr(ctx, *s, slot, d_out[b]));
        }
        return mark_done(ctx, *s);
    }
    UCHK(enqueue_pyramids(ctx, *s, slot, d_rgbL, d_rgbR, n, W, H, stride));
    s->lr_ran = false;
    UCHK(enqueue_full(ctx, *s, slot, d_out));
    return mark_done(ctx, *s);
}

int ugsm_submit_foveated_batch(ugsm_ctx *ctx, int slot, int n, const uint8_t *const *d_rgbL, const uint8_t *const *d_rgbR, int W, int H, int stride,
                               const int *off_x, const int *off_y, float *const *d_stack, float *const *d_pyrL, float *const *d_pyrR)
{
    Slot *s;
    UCHK(get_slot(ctx, slot, &s));
    if (n < 1 || n > UGSM_MAX_BATCH || !d_rgbL || !d_rgbR || !d_stack) return UGSM_ERR_BAD_ARG;
    for (int b = 0; b < n; b++)
        if (!d_rgbL[b] || !d_rgbR[b] || !d_stack[b]) return UGSM_ERR_BAD_ARG;
    const int F = ctx->cfg.fovea_levels;
    if (F < 2) return UGSM_ERR_BAD_ARG;
    HIPCHK(ctx, hipSetDevice(ctx->cfg.device));
    int zeros[UGSM_MAX_BATCH] = {0};
    const int *ox = off_x ? off_x : zeros, *oy = off_y ? off_y : zeros;
    int fw, fh;
    UCHK(ugsm_fovea_dims(W, H, ctx->cfg.levels, F, &fw, &fh));
    const size_t fn3 = (3 * (size_t)fw * fh + 63) & ~(size_t)63;  // level F-1's state of every pair, staged in the slot's hout
    UCHK(grow(ctx, s->hout, s->hout_cap, std::max(fn3 * n, s->hout_cap)));
    FoveaWin win;
    UCHK(fovea_windows(ctx, W, H, n, ox, oy, win));
    if (n == 1 || batch_runs_pair_by_pair(ctx)) {
        for (int b = 0; b < n; b++) {
            FoveaWin w1;
            UCHK(fovea_windows(ctx, W, H, 1, ox + b, oy + b, w1));
            UCHK(enqueue_pyramids(ctx, *s, slot, d_rgbL[b], d_rgbR[b], W, H, stride, F - 1, &w1));
            UCHK(enqueue_fovea_coarse(ctx, *s, slot, s->hout));
            UCHK(enqueue_fovea_fine(ctx, *s, slot, s->hout, ox[b], oy[b], d_stack[b], d_pyrL ? d_pyrL[b] : nullptr, d_pyrR ? d_pyrR[b] : nullptr));
        }
        return mark_done(ctx, *s);
    }
    float *state[UGSM_MAX_BATCH];
    for (int b = 0; b < n; b++) state[b] = s->hout + b * fn3;
    UCHK(enqueue_pyramids(ctx, *s, slot, d_rgbL, d_rgbR, n, W, H, stride, -1, &win));
    UCHK(enqueue_fovea_coarse(ctx, *s, slot, state));
    UCHK(enqueue_fovea_fine(ctx, *s, slot, state, ox, oy, d_stack, d_pyrL, d_pyrR));
    return mark_done(ctx, *s);
}

int ugsm_wait(ugsm_ctx *ctx, int slot)
{
    Slot *s;
    UCHK(get_slot(ctx, slot, &s, false));
    // (a slot that holds a step of the fovea shard: the shard layer watches the step's deadline and reads the status word the exchange carried)
    const int peer = ctx->hooks.shard_wait ? ctx->hooks.shard_wait(ctx, slot, 1) : UGSM_OK;
    if (s->done_recorded) HIPCHK(ctx, hipEventSynchronize(s->ev_done));  // (shared stream: this slot's pair, not the ones queued behind it)
    else HIPCHK(ctx, hipStreamSynchronize(s->st));
    if (s->forked && s->st2) HIPCHK(ctx, hipStreamSynchronize(s->st2));  // (a submit that failed after its fork: Slot::forked)
    s->forked = false;
    s->busy = false;
    harvest(ctx, *s);
    return peer;
}

int ugsm_poll(ugsm_ctx *ctx, int slot)
{
    Slot *s;
    UCHK(get_slot(ctx, slot, &s, false));
    const hipError_t e = s->done_recorded ? hipEventQuery(s->ev_done) : hipStreamQuery(s->st);
    if (e == hipErrorNotReady) return UGSM_PENDING;
    HIPCHK(ctx, e);
    if (s->forked && s->st2) {  // (a submit that failed after its fork: Slot::forked)
        const hipError_t e2 = hipStreamQuery(s->st2);
        if (e2 == hipErrorNotReady) return UGSM_PENDING;
        HIPCHK(ctx, e2);
    }
    s->forked = false;
    const int peer = ctx->hooks.shard_wait ? ctx->hooks.shard_wait(ctx, slot, 0) : UGSM_OK;  // (the slot is idle: the hook only reads the step's status word)
    s->busy = false;
    harvest(ctx, *s);
    return peer;
}

int ugsm_wait_all(ugsm_ctx *ctx)
{
    if (!ctx) return UGSM_ERR_BAD_ARG;
    // every slot is waited for whatever the ones before it answer (ugsm_shard_finalize frees the slots' exchange buffers behind this call); the
    // first failure is the one reported
    int first = UGSM_OK;
    std::string why;
    for (int i = 0; i < (int)ctx->slots.size(); i++) {
        const int st = ugsm_wait(ctx, i);
        if (st != UGSM_OK && first == UGSM_OK) {
            first = st;
            why = ctx->err;
        }
    }
    if (first != UGSM_OK) ctx->err = why;
    return first;
}

// A blocking entry point that failed part-way: whatever it did enqueue -- uploads that read the caller's images, on the slot's stream or on
// the side stream -- has drained before the caller hears of the failure.  "When the call returns the buffers are the caller's again"
// holds for a failed call too (the reference exit()s instead, MatchGPULib.cpp passim).
static int drained(ugsm_ctx *ctx, int slot, int st)
{
    if (st == UGSM_OK || !ctx || slot < 0 || slot >= (int)ctx->slots.size()) return st;
    const std::string why = ctx->err;
    (void)ugsm_wait(ctx, slot);
    ctx->err = why;
    return st;
}

// The service call on a slot: upload, pyramids, match, results into caller memory.  `sync` = the reference's call (returns when the
// planes are in place; pageable or page-locked memory); otherwise everything is only enqueued and every buffer must be page-locked.
static int match_full_on_slot(ugsm_ctx *ctx, int slot, const uint8_t *rgbL, const uint8_t *rgbR, int W, int H, int stride, float *dispH,
                              float *dispV, float *dispC, bool sync)
{
    Slot *s;
    UCHK(get_slot(ctx, slot, &s));
    if (!dispH || !dispV || !dispC) return UGSM_ERR_BAD_ARG;
    HIPCHK(ctx, hipSetDevice(ctx->cfg.device));
    if (!sync && !(is_pinned(rgbL) && is_pinned(rgbR) && is_pinned(dispH) && is_pinned(dispV) && is_pinned(dispC))) {
        ctx->err = "ugsm_submit_full_host: every host buffer must be page-locked (ugsm_host_alloc, hipHostMalloc or hipHostRegister)";
        return UGSM_ERR_BAD_ARG;
    }
    UCHK(stage_in(ctx, *s, rgbL, rgbR, W, H, stride));
    const size_t n = (size_t)W * H;
    UCHK(grow(ctx, s->hout, s->hout_cap, 3 * n));
    UCHK(enqueue_pyramids(ctx, *s, slot, s->rgbL, s->rgbR, W, H, stride, 0));
    UCHK(enqueue_full_lr(ctx, *s, slot, s->hout));
    float *const dst[3] = {dispH, dispV, dispC};
    if (sync) prefault_planes(ctx, dst, n);  // the GPU is busy for the next ~10 ms: touch the caller's result pages meanwhile
    UCHK(copy_out_planes(ctx, *s, s->hout, n, dst));
    if (sync) s->forked = false;  // (enqueued whole, as mark_done)
    return sync ? ugsm_wait(ctx, slot) : mark_done(ctx, *s);
}

int ugsm_match_full(ugsm_ctx *ctx, const uint8_t *rgbL, const uint8_t *rgbR, int W, int H, int stride, float *dispH,
                    float *dispV, float *dispC)
{
    return drained(ctx, 0, match_full_on_slot(ctx, 0, rgbL, rgbR, W, H, stride, dispH, dispV, dispC, true));
}

int ugsm_submit_full_host(ugsm_ctx *ctx, int slot, const uint8_t *rgbL, const uint8_t *rgbR, int W, int H, int stride, float *dispH,
                          float *dispV, float *dispC)
{
    return match_full_on_slot(ctx, slot, rgbL, rgbR, W, H, stride, dispH, dispV, dispC, false);
}

static int match_foveated_on_slot(ugsm_ctx *ctx, int slot, const uint8_t *rgbL, const uint8_t *rgbR, int W, int H, int stride, int off_x, int off_y,
                                  float *stackH, float *stackV, float *stackC, float *pyrL, float *pyrR, bool sync)
{
    Slot *s;
    UCHK(get_slot(ctx, slot, &s));
    if (!stackH || !stackV || !stackC) return UGSM_ERR_BAD_ARG;
    const int F = ctx->cfg.fovea_levels;
    if (F < 2) return UGSM_ERR_BAD_ARG;
    HIPCHK(ctx, hipSetDevice(ctx->cfg.device));
    if (!sync) {
        bool pinned = is_pinned(rgbL) && is_pinned(rgbR) && is_pinned(stackH) && is_pinned(stackV) && is_pinned(stackC);
        if (pyrL) pinned = pinned && is_pinned(pyrL);
        if (pyrR) pinned = pinned && is_pinned(pyrR);
        if (!pinned) {
            ctx->err = "ugsm_submit_foveated_host: every host buffer must be page-locked (ugsm_host_alloc, hipHostMalloc or hipHostRegister)";
            return UGSM_ERR_BAD_ARG;
        }
    }
    int fw, fh;
    UCHK(ugsm_fovea_dims(W, H, ctx->cfg.levels, F, &fw, &fh));
    UCHK(stage_in(ctx, *s, rgbL, rgbR, W, H, stride));
    const size_t fn = (size_t)fw * fh, stackn = (size_t)F * fn;
    // layout of hout: [state 3*fn][stack 3*stackn][pyrL 3*stackn][pyrR 3*stackn]
    const size_t need = 3 * fn + 3 * stackn + (pyrL ? 3 * stackn : 0) + (pyrR ? 3 * stackn : 0);
    UCHK(grow(ctx, s->hout, s->hout_cap, need));
    float *d_state = s->hout, *d_stack = d_state + 3 * fn;
    float *d_pl = pyrL ? d_stack + 3 * stackn : nullptr;
    float *d_pr = pyrR ? d_stack + 3 * stackn + (pyrL ? 3 * stackn : 0) : nullptr;
    FoveaWin win;
    UCHK(fovea_windows(ctx, W, H, 1, &off_x, &off_y, win));
    UCHK(enqueue_pyramids(ctx, *s, slot, s->rgbL, s->rgbR, W, H, stride, F - 1, &win));
    UCHK(enqueue_fovea_coarse(ctx, *s, slot, d_state));
    UCHK(enqueue_fovea_fine(ctx, *s, slot, d_state, off_x, off_y, d_stack, d_pl, d_pr));
    HIPCHK(ctx, hipMemcpyAsync(stackH, d_stack, stackn * sizeof(float), hipMemcpyDeviceToHost, s->st));
    HIPCHK(ctx, hipMemcpyAsync(stackV, d_stack + stackn, stackn * sizeof(float), hipMemcpyDeviceToHost, s->st));
    HIPCHK(ctx, hipMemcpyAsync(stackC, d_stack + 2 * stackn, stackn * sizeof(float), hipMemcpyDeviceToHost, s->st));
    if (pyrL) HIPCHK(ctx, hipMemcpyAsync(pyrL, d_pl, 3 * stackn * sizeof(float), hipMemcpyDeviceToHost, s->st));
    if (pyrR) HIPCHK(ctx, hipMemcpyAsync(pyrR, d_pr, 3 * stackn * sizeof(float), hipMemcpyDeviceToHost, s->st));
    if (sync) s->forked = false;  // (enqueued whole, as mark_done)
    return sync ? ugsm_wait(ctx, slot) : mark_done(ctx, *s);
}

int ugsm_match_foveated(ugsm_ctx *ctx, const uint8_t *rgbL, const uint8_t *rgbR, int W, int H, int stride, int off_x,
                        int off_y, float *stackH, float *stackV, float *stackC, float *pyrL, float *pyrR)
{
    return drained(ctx, 0, match_foveated_on_slot(ctx, 0, rgbL, rgbR, W, H, stride, off_x, off_y, stackH, stackV, stackC, pyrL, pyrR, true));
}

int ugsm_submit_foveated_host(ugsm_ctx *ctx, int slot, const uint8_t *rgbL, const uint8_t *rgbR, int W, int H, int stride, int off_x,
                              int off_y, float *stackH, float *stackV, float *stackC, float *pyrL, float *pyrR)
{
    return match_foveated_on_slot(ctx, slot, rgbL, rgbR, W, H, stride, off_x, off_y, stackH, stackV, stackC, pyrL, pyrR, false);
}

// ---- batches from page-locked host memory (round 4): the uploads of all pairs, ONE batched match, the downloads of all pairs, enqueued on
// the slot's stream.  The images are staged in the slot (pair b at rgbL / rgbR + b * image bytes), the results in hout.
static bool all_pinned(const void *const *p, int n)
{
    for (int b = 0; b < n; b++)
        if (!p[b] || !is_pinned(p[b])) return false;
    return true;
}
static int stage_in_batch(ugsm_ctx *ctx, Slot &s, int n, const uint8_t *const *rgbL, const uint8_t *const *rgbR, int W, int H, int stride,
                          const uint8_t **dL, const uint8_t **dR)
{
    if (W < 1 || H < 1) return UGSM_ERR_BAD_ARG;
    if (stride < 3 * W) return UGSM_ERR_SIZE_MISMATCH;
    const size_t bytes = ((size_t)stride * H + 255) & ~(size_t)255;
    if (bytes * n > s.rgb_cap) {
        size_t c = s.rgb_cap;
        UCHK(grow(ctx, s.rgbL, c, bytes * n));
        UCHK(grow(ctx, s.rgbR, s.rgb_cap, bytes * n));
    }
    for (int b = 0; b < n; b++) {
        dL[b] = s.rgbL + b * bytes;
        dR[b] = s.rgbR + b * bytes;
        HIPCHK(ctx, hipMemcpyAsync(s.rgbL + b * bytes, rgbL[b], (size_t)stride * H, hipMemcpyHostToDevice, s.st));
        HIPCHK(ctx, hipMemcpyAsync(s.rgbR + b * bytes, rgbR[b], (size_t)stride * H, hipMemcpyHostToDevice, s.st));
    }
    return UGSM_OK;
}

int ugsm_submit_full_batch_host(ugsm_ctx *ctx, int slot, int n, const uint8_t *const *rgbL, const uint8_t *const *rgbR, int W, int H, int stride,
                                float *const *dispH, float *const *dispV, float *const *dispC)
{
    Slot *s;
    UCHK(get_slot(ctx, slot, &s));
    if (n < 1 || n > UGSM_MAX_BATCH || !rgbL || !rgbR || !dispH || !dispV || !dispC) return UGSM_ERR_BAD_ARG;
    HIPCHK(ctx, hipSetDevice(ctx->cfg.device));
    if (!all_pinned((const void *const *)rgbL, n) || !all_pinned((const void *const *)rgbR, n) || !all_pinned((const void *const *)dispH, n) ||
        !all_pinned((const void *const *)dispV, n) || !all_pinned((const void *const *)dispC, n)) {
        ctx->err = "ugsm_submit_full_batch_host: every host buffer must be page-locked (ugsm_host_alloc, hipHostMalloc or hipHostRegister)";
        return UGSM_ERR_BAD_ARG;
    }
    if (n == 1 || batch_runs_pair_by_pair(ctx)) {
        for (int b = 0; b < n; b++) UCHK(match_full_on_slot(ctx, slot, rgbL[b], rgbR[b], W, H, stride, dispH[b], dispV[b], dispC[b], false));
        return UGSM_OK;
    }
    const uint8_t *dL[UGSM_MAX_BATCH], *dR[UGSM_MAX_BATCH];
    UCHK(stage_in_batch(ctx, *s, n, rgbL, rgbR, W, H, stride, dL, dR));
    const size_t px = (size_t)W * H, per = (3 * px + 63) & ~(size_t)63;
    UCHK(grow(ctx, s->hout, s->hout_cap, std::max(per * n, s->hout_cap)));
    float *out[UGSM_MAX_BATCH];
    for (int b = 0; b < n; b++) out[b] = s->hout + b * per;
    UCHK(enqueue_pyramids(ctx, *s, slot, dL, dR, n, W, H, stride));
    s->lr_ran = false;
    UCHK(enqueue_full(ctx, *s, slot, out));
    for (int b = 0; b < n; b++) {
        float *const dst[3] = {dispH[b], dispV[b], dispC[b]};
        UCHK(copy_out_planes(ctx, *s, out[b], px, dst));  // (page-locked destinations: three asynchronous copies)
    }
    return mark_done(ctx, *s);
}

int ugsm_submit_foveated_batch_host(ugsm_ctx *ctx, int slot, int n, const uint8_t *const *rgbL, const uint8_t *const *rgbR, int W, int H, int stride,
                                    const int *off_x, const int *off_y, float *const *stackH, float *const *stackV, float *const *stackC)
{
    Slot *s;
    UCHK(get_slot(ctx, slot, &s));
    if (n < 1 || n > UGSM_MAX_BATCH || !rgbL || !rgbR || !stackH || !stackV || !stackC) return UGSM_ERR_BAD_ARG;
    const int F = ctx->cfg.fovea_levels;
    if (F < 2) return UGSM_ERR_BAD_ARG;
    HIPCHK(ctx, hipSetDevice(ctx->cfg.device));
    if (!all_pinned((const void *const *)rgbL, n) || !all_pinned((const void *const *)rgbR, n) || !all_pinned((const void *const *)stackH, n) ||
        !all_pinned((const void *const *)stackV, n) || !all_pinned((const void *const *)stackC, n)) {
        ctx->err = "ugsm_submit_foveated_batch_host: every host buffer must be page-locked (ugsm_host_alloc, hipHostMalloc or hipHostRegister)";
        return UGSM_ERR_BAD_ARG;
    }
    int zeros[UGSM_MAX_BATCH] = {0};
    const int *ox = off_x ? off_x : zeros, *oy = off_y ? off_y : zeros;
    if (n == 1 || batch_runs_pair_by_pair(ctx)) {
        for (int b = 0; b < n; b++)
            UCHK(match_foveated_on_slot(ctx, slot, rgbL[b], rgbR[b], W, H, stride, ox[b], oy[b], stackH[b], stackV[b], stackC[b], nullptr, nullptr, false));
        return UGSM_OK;
    }
    int fw, fh;
    UCHK(ugsm_fovea_dims(W, H, ctx->cfg.levels, F, &fw, &fh));
    const uint8_t *dL[UGSM_MAX_BATCH], *dR[UGSM_MAX_BATCH];
    UCHK(stage_in_batch(ctx, *s, n, rgbL, rgbR, W, H, stride, dL, dR));
    const size_t fn = (size_t)fw * fh, stackn = (size_t)F * fn;
    const size_t st_per = (3 * fn + 63) & ~(size_t)63, sk_per = (3 * stackn + 63) & ~(size_t)63;  // hout: [states n x st_per][stacks n x sk_per]
    UCHK(grow(ctx, s->hout, s->hout_cap, std::max(n * (st_per + sk_per), s->hout_cap)));
    float *state[UGSM_MAX_BATCH], *stack[UGSM_MAX_BATCH];
    for (int b = 0; b < n; b++) {
        state[b] = s->hout + b * st_per;
        stack[b] = s->hout + n * st_per + b * sk_per;
    }
    FoveaWin win;
    UCHK(fovea_windows(ctx, W, H, n, ox, oy, win));
    UCHK(enqueue_pyramids(ctx, *s, slot, dL, dR, n, W, H, stride, -1, &win));
    UCHK(enqueue_fovea_coarse(ctx, *s, slot, state));
    UCHK(enqueue_fovea_fine(ctx, *s, slot, state, ox, oy, stack, nullptr, nullptr));
    for (int b = 0; b < n; b++) {
        HIPCHK(ctx, hipMemcpyAsync(stackH[b], stack[b], stackn * sizeof(float), hipMemcpyDeviceToHost, s->st));
        HIPCHK(ctx, hipMemcpyAsync(stackV[b], stack[b] + stackn, stackn * sizeof(float), hipMemcpyDeviceToHost, s->st));
        HIPCHK(ctx, hipMemcpyAsync(stackC[b], stack[b] + 2 * stackn, stackn * sizeof(float), hipMemcpyDeviceToHost, s->st));
    }
    return mark_done(ctx, *s);
}

// match(L, R, fov == 1), MatchGPULib.cpp:354-360: foveated matching, then hierarchicalDisparity on the stacks
static int match_foveated_full_on_slot0(ugsm_ctx *ctx, const uint8_t *rgbL, const uint8_t *rgbR, int W, int H, int stride, int off_x, int off_y,
                                        float *outH, float *outV, float *outC)
{
    Slot *s;
    UCHK(get_slot(ctx, 0, &s));
    if (!outH || !outV || !outC) return UGSM_ERR_BAD_ARG;
    const int F = ctx->cfg.fovea_levels;
    if (F < 2) return UGSM_ERR_BAD_ARG;
    HIPCHK(ctx, hipSetDevice(ctx->cfg.device));
    int fw, fh;
    UCHK(ugsm_fovea_dims(W, H, ctx->cfg.levels, F, &fw, &fh));
    UCHK(stage_in(ctx, *s, rgbL, rgbR, W, H, stride));
    const size_t fn = (size_t)fw * fh, stackn = (size_t)F * fn, n = (size_t)W * H;
    UCHK(grow(ctx, s->hout, s->hout_cap, 3 * fn + 3 * stackn + 3 * n));  // [state][stack][full field]
    float *d_state = s->hout, *d_stack = d_state + 3 * fn, *d_full = d_stack + 3 * stackn;
    FoveaWin win;
    UCHK(fovea_windows(ctx, W, H, 1, &off_x, &off_y, win));
    UCHK(enqueue_pyramids(ctx, *s, 0, s->rgbL, s->rgbR, W, H, stride, F - 1, &win));
    UCHK(enqueue_fovea_coarse(ctx, *s, 0, d_state));
    UCHK(enqueue_fovea_fine(ctx, *s, 0, d_state, off_x, off_y, d_stack, nullptr, nullptr));
    UCHK(ugsm_reconstruct_full(ctx, 0, d_stack, d_stack + stackn, d_stack + 2 * stackn, W, H, off_x, off_y, d_full));
    float *const dst[3] = {outH, outV, outC};
    prefault_planes(ctx, dst, n);
    UCHK(copy_out_planes(ctx, *s, d_full, n, dst));
    s->forked = false;  // (enqueued whole, as mark_done)
    return ugsm_wait(ctx, 0);
}
int ugsm_match_foveated_full(ugsm_ctx *ctx, const uint8_t *rgbL, const uint8_t *rgbR, int W, int H, int stride, int off_x, int off_y,
                             float *outH, float *outV, float *outC)
{
    return drained(ctx, 0, match_foveated_full_on_slot0(ctx, rgbL, rgbR, W, H, stride, off_x, off_y, outH, outV, outC));
}

// ---- stage-level ------------------------------------------------------------------------

int ugsm_stage_pyramid(ugsm_ctx *ctx, const uint8_t *d_rgb, int W, int H, int stride, int level, float *d_out3)
{
    Slot *s;
    UCHK(get_slot(ctx, 0, &s));
    if (!d_rgb || !d_out3 || level < 0 || level >= ctx->cfg.levels) return UGSM_ERR_BAD_ARG;
    if (stride < 3 * W) return UGSM_ERR_SIZE_MISMATCH;
    HIPCHK(ctx, hipSetDevice(ctx->cfg.device));
    UCHK(prepare_slot(ctx, *s, W, H));
    UCHK(build_pyramids(ctx, *s, 0, &d_rgb, stride, s->pyrL));
    HIPCHK(ctx, hipMemcpyAsync(d_out3, s->pyrL + s->off[level], sizeof(float) * 3 * (size_t)s->w[level] * s->h[level],
                               hipMemcpyDeviceToDevice, s->st));
    return ugsm_wait(ctx, 0);
}

int ugsm_stage_iterate(ugsm_ctx *ctx, const float *d_L3, const float *d_R3, float *d_d3, int W, int H, int mi, int S,
                       int is_top, int m_from, int m_to, float *d_dbg8)
{
    Slot *s;
    UCHK(get_slot(ctx, 0, &s));
    if (!d_L3 || !d_R3 || !d_d3 || W < 1 || H < 1 || mi < 1 || m_from < 1 || m_to > mi || S < 0) return UGSM_ERR_BAD_ARG;
    if ((long long)W * H > kMaxPixels) return UGSM_ERR_BAD_ARG;
    if (d_dbg8 && ctx->cfg.kernel_path != 1) return UGSM_ERR_BAD_ARG;
    HIPCHK(ctx, hipSetDevice(ctx->cfg.device));
    // level buffers sized for this image; the pyramid buffers are not needed here
    const size_t lvl = 3 * (size_t)W * H;
    UCHK(ensure_level_bufs(ctx, *s, lvl));
    float *cur = s->d0, *other = s->d1;
    HIPCHK(ctx, hipMemcpyAsync(cur, d_d3, lvl * sizeof(float), hipMemcpyDeviceToDevice, s->st));
    const size_t n = (size_t)W * H;
    // the planes arrive ready-made: check their range here (the pyramid kernels do it for the matcher proper).  The slot's range
    // word then describes THESE planes, no longer the slot's pyramids: a fovea phase submitted afterwards must rebuild them first
    s->have_pyr = false;
    s->have_coarse = false;
    s->range_known = ctx->cfg.kernel_path != 1;
    if (s->range_known) {
        HIPCHK(ctx, hipMemsetAsync(s->range_bad, 0, sizeof(unsigned), s->st));
        launch_range_scan(s->st, d_L3, 3 * n, s->range_bad);
        launch_range_scan(s->st, d_R3, 3 * n, s->range_bad);
    }
    s->nb = 1;
    const Img3 Lv{d_L3, W, n}, Rv{d_R3, W, n};
    UCHK(run_level(ctx, *s, 0, &Lv, &Rv, W, H, mi, S, is_top != 0, m_from, m_to, cur, other, d_dbg8));
    HIPCHK(ctx, hipMemcpyAsync(d_d3, cur, lvl * sizeof(float), hipMemcpyDeviceToDevice, s->st));
    return ugsm_wait(ctx, 0);
}

int ugsm_stage_seed(ugsm_ctx *ctx, const float *d_src3, int W, int H, float *d_dst3, int W2, int H2, int Wup, int Hup,
                    int crop_x, int crop_y)
{
    Slot *s;
    UCHK(get_slot(ctx, 0, &s));
    if (!d_src3 || !d_dst3 || W < 1 || H < 1 || W2 < 1 || H2 < 1) return UGSM_ERR_BAD_ARG;
    (void)Wup;
    (void)Hup;
    HIPCHK(ctx, hipSetDevice(ctx->cfg.device));
    launch_seed(s->st, d_src3, W, H, d_dst3, W2, H2, crop_x, crop_y);
    HIPCHK(ctx, hipGetLastError());
    return ugsm_wait(ctx, 0);
}

int ugsm_stage_smooth(ugsm_ctx *ctx, float *d_d3, int W, int H, int passes, int do_box)
{
    Slot *s;
    UCHK(get_slot(ctx, 0, &s));
    if (!d_d3 || W < 1 || H < 1 || passes < 0 || (long long)W * H > kMaxPixels) return UGSM_ERR_BAD_ARG;
    HIPCHK(ctx, hipSetDevice(ctx->cfg.device));
    const size_t lvl = 3 * (size_t)W * H;
    UCHK(ensure_level_bufs(ctx, *s, lvl));
    float *a = s->d0, *b = s->d1;
    HIPCHK(ctx, hipMemcpyAsync(a, d_d3, lvl * sizeof(float), hipMemcpyDeviceToDevice, s->st));
    s->nb = 1;
    UCHK(enqueue_smooth(ctx, *s, 0, a, b, W, H, passes, do_box != 0));
    HIPCHK(ctx, hipGetLastError());
    HIPCHK(ctx, hipMemcpyAsync(d_d3, a, lvl * sizeof(float), hipMemcpyDeviceToDevice, s->st));
    return ugsm_wait(ctx, 0);
}

int ugsm_triangulate(ugsm_ctx *ctx, int slot, const float *d_dispx, const float *d_dispy, int W, int H, const double *P1, const double *P2,
                     float *d_xyz)
{
    Slot *s;
    UCHK(get_slot(ctx, slot, &s));
    if (!d_dispx || !d_dispy || !P1 || !P2 || !d_xyz || W < 1 || H < 1) return UGSM_ERR_BAD_ARG;
    HIPCHK(ctx, hipSetDevice(ctx->cfg.device));
    {
        Timer t(ctx, s, slot, KC_MISC, (double)W * H);
        launch_triangulate(s->st, d_dispx, d_dispy, W, H, P1, P2, d_xyz);
    }
    HIPCHK(ctx, hipGetLastError());
    return UGSM_OK;
}

// CdynamicCalibration::left_marginOf_in / upper_marginOf_in / mapXcoord (getPointCloud.cpp:387-484)
int ugsm_fovea_mapping(int W, int H, int src_level, int dest_level, int *left_margin, int *upper_margin, float *scale)
{
    if (!left_margin || !upper_margin || !scale || W < 1 || H < 1) return UGSM_ERR_BAD_ARG;
    int scaled = 6 - src_level;  // :435 (the reference hard-codes its 7 fovea levels here)
    if (src_level < dest_level) scaled = src_level + dest_level;
    if (scaled < 0 || scaled >= 15 || dest_level < 0 || dest_level >= 15 || src_level < 0) return UGSM_ERR_BAD_ARG;
    int w[16], h[16];
    w[0] = W;
    h[0] = H;
    for (int i = 0; i < 14; i++) {  // :441-443
        w[i + 1] = (int)(w[i] / kScale);
        h[i + 1] = (int)(h[i] / kScale);
    }
    *left_margin = w[dest_level] / 2 - w[scaled] / 2;
    *upper_margin = h[dest_level] / 2 - h[scaled] / 2;
    const float root = (src_level < dest_level) ? (float)0.70710678118654752440 : (float)1.41421356237309504880;
    *scale = powf(root, (float)std::abs(src_level - dest_level));  // pow(float, float), :397
    return UGSM_OK;
}

int ugsm_triangulate_fovea(ugsm_ctx *ctx, int slot, const float *d_stackx, const float *d_stacky, int fovW, int fovH, int src_level,
                           int left_margin, int upper_margin, float scale, const double *P1, const double *P2, float *d_xyz)
{
    Slot *s;
    UCHK(get_slot(ctx, slot, &s));
    if (!d_stackx || !d_stacky || !P1 || !P2 || !d_xyz || fovW < 1 || fovH < 1 || src_level < 0) return UGSM_ERR_BAD_ARG;
    HIPCHK(ctx, hipSetDevice(ctx->cfg.device));
    {
        Timer t(ctx, s, slot, KC_MISC, (double)fovW * fovH);
        launch_triangulate_fovea(s->st, d_stackx, d_stacky, fovW, fovH, src_level, left_margin, upper_margin, scale, P1, P2, d_xyz);
    }
    HIPCHK(ctx, hipGetLastError());
    return UGSM_OK;
}

// hierarchicalDisparity, MatchGPULib.cpp:2589-2701
int ugsm_reconstruct_full(ugsm_ctx *ctx, int slot, const float *d_stackH, const float *d_stackV, const float *d_stackC, int W, int H,
                          int off_x, int off_y, float *d_out3)
{
    Slot *s;
    UCHK(get_slot(ctx, slot, &s));
    if (!d_stackH || !d_stackV || !d_stackC || !d_out3) return UGSM_ERR_BAD_ARG;
    const int levels = ctx->cfg.levels, F = ctx->cfg.fovea_levels;
    int w[UGSM_MAX_LEVELS], h[UGSM_MAX_LEVELS];
    UCHK(level_dims(W, H, levels, w, h));
    if (F < 1 || F > levels) return UGSM_ERR_BAD_ARG;
    HIPCHK(ctx, hipSetDevice(ctx->cfg.device));
    FoveaGeom g;
    if (F >= 2) fovea_geometry(w, h, F, off_x, off_y, g);
    else { g.fw = w[0]; g.fh = h[0]; }
    const size_t fl = (size_t)g.fw * g.fh;
    if (F == 1) {  // the stack is the full frame already
        const float *src[3] = {d_stackH, d_stackV, d_stackC};
        for (int c = 0; c < 3; c++) HIPCHK(ctx, hipMemcpyAsync(d_out3 + c * fl, src[c], fl * sizeof(float), hipMemcpyDeviceToDevice, s->st));
        return UGSM_OK;
    }
    UCHK(ensure_level_bufs(ctx, *s, 3 * (size_t)w[1] * h[1]));
    // level F-1 (whole frame) as a 3-plane field
    float *cur = s->d0, *other = s->d1;
    {
        const float *src[3] = {d_stackH, d_stackV, d_stackC};
        for (int c = 0; c < 3; c++)
            HIPCHK(ctx, hipMemcpyAsync(cur + c * fl, src[c] + (size_t)(F - 1) * fl, fl * sizeof(float), hipMemcpyDeviceToDevice, s->st));
    }
    for (int level = F - 1; level > 0; level--) {
        float *dst = (level == 1) ? d_out3 : other;
        Timer t(ctx, s, slot, KC_MISC, (double)w[level - 1] * h[level - 1]);
        launch_upsample_paste(s->st, cur, w[level], h[level], dst, w[level - 1], h[level - 1], d_stackH + (size_t)(level - 1) * fl,
                              d_stackV + (size_t)(level - 1) * fl, d_stackC + (size_t)(level - 1) * fl, g.fw, g.fh, g.ox[level - 1], g.oy[level - 1]);
        std::swap(cur, other);
    }
    HIPCHK(ctx, hipGetLastError());
    return UGSM_OK;
}

#ifdef UGSM_DEV_LIB
int ugsm_stage_poly_probe(ugsm_ctx *ctx, const float *d_c, const float *d_l, const float *d_r, const float *d_thr, float *d_delta,
                          float *d_corr, float *d_third, int n)
{
    Slot *s;
    UCHK(get_slot(ctx, 0, &s));
    if (!d_c || !d_l || !d_r || !d_thr || !d_delta || !d_corr || !d_third || n < 1) return UGSM_ERR_BAD_ARG;
    HIPCHK(ctx, hipSetDevice(ctx->cfg.device));
    launch_poly_probe(s->st, d_c, d_l, d_r, d_thr, d_delta, d_corr, d_third, n);
    HIPCHK(ctx, hipGetLastError());
    return ugsm_wait(ctx, 0);
}
#endif

int ugsm_stage_weighted_difference(ugsm_ctx *ctx, const float *d_new3, const float *d_old3, int W, int H, float *out2)
{
    Slot *s;
    UCHK(get_slot(ctx, 0, &s));
    if (!d_new3 || !d_old3 || !out2 || W < 1 || H < 1 || H > 65535 || (long long)W * H > kMaxPixels) return UGSM_ERR_BAD_ARG;
    HIPCHK(ctx, hipSetDevice(ctx->cfg.device));
    return weighted_difference(ctx, *s, d_new3, d_old3, W, H, out2);
}

int ugsm_stage_lr_check(ugsm_ctx *ctx, float *d_left3, const float *d_right3, int W, int H, float tau, long long *marked)
{
    Slot *s;
    UCHK(get_slot(ctx, 0, &s));
    if (!d_left3 || !d_right3 || W < 1 || H < 1 || H > 65535 || (long long)W * H > kMaxPixels || !(tau >= 0.0f)) return UGSM_ERR_BAD_ARG;
    HIPCHK(ctx, hipSetDevice(ctx->cfg.device));
    UCHK(grow(ctx, s->lr, s->lr_cap, std::max<size_t>(s->lr_cap, 8)));
    if (!s->lr_host) HIPCHK(ctx, hipHostMalloc((void **)&s->lr_host, sizeof(unsigned long long), hipHostMallocDefault));
    unsigned long long *cnt = reinterpret_cast<unsigned long long *>(s->lr);
    HIPCHK(ctx, hipStreamSynchronize(s->st));  // (the counter shares the slot's LR buffer)
    HIPCHK(ctx, hipMemsetAsync(cnt, 0, sizeof *cnt, s->st));
    launch_lr_check(s->st, d_left3, d_right3, W, H, tau, cnt);
    HIPCHK(ctx, hipGetLastError());
    HIPCHK(ctx, hipMemcpyAsync(s->lr_host, cnt, sizeof *cnt, hipMemcpyDeviceToHost, s->st));
    UCHK(ugsm_wait(ctx, 0));
    if (marked) *marked = (long long)*s->lr_host;
    return UGSM_OK;
}

long long ugsm_last_lr_marked(ugsm_ctx *ctx, int slot)
{
    Slot *s;
    if (get_slot(ctx, slot, &s, false) != UGSM_OK) return -1;
    return (s->lr_ran && s->lr_host) ? (long long)*s->lr_host : -1;
}

int ugsm_slot_stream(ugsm_ctx *ctx, int slot, void **hip_stream)
{
    Slot *s;
    UCHK(get_slot(ctx, slot, &s));
    if (!hip_stream) return UGSM_ERR_BAD_ARG;
    *hip_stream = (void *)s->st;
    return UGSM_OK;
}

int ugsm_last_iterations(ugsm_ctx *ctx, int slot, int *per_level)
{
    Slot *s;
    UCHK(get_slot(ctx, slot, &s, false));  // (reads host state only: the slot does not become busy)
    if (!per_level) return UGSM_ERR_BAD_ARG;
    for (int i = 0; i < UGSM_MAX_LEVELS; i++) per_level[i] = s->iters_run[i];
    return UGSM_OK;
}

#ifdef UGSM_DEV_LIB
int ugsm_stage_div_probe(ugsm_ctx *ctx, const float *d_n, const float *d_d, float *d_q, int n)
{
    Slot *s;
    UCHK(get_slot(ctx, 0, &s));
    if (!d_n || !d_d || !d_q || n < 1) return UGSM_ERR_BAD_ARG;
    HIPCHK(ctx, hipSetDevice(ctx->cfg.device));
    launch_div_probe(s->st, d_n, d_d, d_q, n);
    HIPCHK(ctx, hipGetLastError());
    return ugsm_wait(ctx, 0);
}
#endif

#ifdef UGSM_DEV_LIB
int ugsm_stage_div3_probe(ugsm_ctx *ctx, const float *d_a0, const float *d_a1, const float *d_a2, const float *d_s, float *d_q0, float *d_q1,
                          float *d_q2, int n)
{
    Slot *s;
    UCHK(get_slot(ctx, 0, &s));
    if (!d_a0 || !d_a1 || !d_a2 || !d_s || !d_q0 || !d_q1 || !d_q2 || n < 1) return UGSM_ERR_BAD_ARG;
    HIPCHK(ctx, hipSetDevice(ctx->cfg.device));
    launch_div3_probe(s->st, d_a0, d_a1, d_a2, d_s, d_q0, d_q1, d_q2, n);
    HIPCHK(ctx, hipGetLastError());
    return ugsm_wait(ctx, 0);
}
#endif

// ---- instrumentation / memory helpers ---------------------------------------------------

int ugsm_get_kernel_stats(ugsm_ctx *ctx, ugsm_kernel_stat *out, int cap)
{
    if (!ctx) return 0;
    int n = 0;
    for (int k = 0; k < KC_COUNT; k++)
        for (int l = 0; l <= UGSM_MAX_LEVELS; l++) {
            const StatCell &c = ctx->cells[k][l];
            if (c.launches == 0) continue;
            if (out && n < cap) {
                ugsm_kernel_stat &o = out[n];
                memset(&o, 0, sizeof o);
                snprintf(o.name, sizeof o.name, "%s", kClassName[ctx->cfg.kernel_path][k]);
                o.level = l == kNoLevel ? -1 : l;
                o.launches = c.launches;
                o.total_ms = c.total_ms;
                o.pixel_launches = c.pixel_launches;
            }
            n++;
        }
    return n;
}

int ugsm_reset_kernel_stats(ugsm_ctx *ctx)
{
    if (!ctx) return UGSM_ERR_BAD_ARG;
    for (int k = 0; k < KC_COUNT; k++)
        for (int l = 0; l <= UGSM_MAX_LEVELS; l++) ctx->cells[k][l] = StatCell();
    return UGSM_OK;
}

int ugsm_set_profile_events(ugsm_ctx *ctx, int mode)
{
    if (!ctx || mode < 0 || mode > 2) return UGSM_ERR_BAD_ARG;
    ctx->cfg.profile_events = mode;
    return UGSM_OK;
}

long long ugsm_context_device_bytes(const ugsm_ctx *ctx) { return ctx ? ctx->dev_bytes : -1; }

int ugsm_dev_alloc(ugsm_ctx *ctx, void **d_ptr, long long bytes)
{
    if (!ctx || !d_ptr || bytes < 0) return UGSM_ERR_BAD_ARG;
    HIPCHK(ctx, hipSetDevice(ctx->cfg.device));
    hipError_t e = hipMalloc(d_ptr, (size_t)std::max<long long>(bytes, 1));
    if (e != hipSuccess) {
        ctx->err = std::string("hipMalloc failed: ") + hipGetErrorString(e);
        return UGSM_ERR_NOMEM;
    }
    return UGSM_OK;
}
int ugsm_dev_free(ugsm_ctx *ctx, void *d_ptr)
{
    if (!ctx) return UGSM_ERR_BAD_ARG;
    if (d_ptr) HIPCHK(ctx, hipFree(d_ptr));
    return UGSM_OK;
}
// Page-locked host memory for callers that can place their images / result planes in it: the copies of
// ugsm_match_* then run as plain DMA instead of being staged through the runtime's bounce buffers.
int ugsm_host_alloc(ugsm_ctx *ctx, void **h_ptr, long long bytes)
{
    if (!ctx || !h_ptr || bytes < 0) return UGSM_ERR_BAD_ARG;
    HIPCHK(ctx, hipSetDevice(ctx->cfg.device));
    hipError_t e = hipHostMalloc(h_ptr, (size_t)std::max<long long>(bytes, 1), hipHostMallocDefault);
    if (e != hipSuccess) {
        ctx->err = std::string("hipHostMalloc failed: ") + hipGetErrorString(e);
        return UGSM_ERR_NOMEM;
    }
    return UGSM_OK;
}
int ugsm_host_free(ugsm_ctx *ctx, void *h_ptr)
{
    if (!ctx) return UGSM_ERR_BAD_ARG;
    if (h_ptr) HIPCHK(ctx, hipHostFree(h_ptr));
    return UGSM_OK;
}
int ugsm_copy_to_device(ugsm_ctx *ctx, void *d_dst, const void *h_src, long long bytes)
{
    if (!ctx || !d_dst || !h_src || bytes < 0) return UGSM_ERR_BAD_ARG;
    HIPCHK(ctx, hipMemcpy(d_dst, h_src, (size_t)bytes, hipMemcpyHostToDevice));
    return UGSM_OK;
}
int ugsm_copy_to_host(ugsm_ctx *ctx, void *h_dst, const void *d_src, long long bytes)
{
    if (!ctx || !h_dst || !d_src || bytes < 0) return UGSM_ERR_BAD_ARG;
    HIPCHK(ctx, hipMemcpy(h_dst, d_src, (size_t)bytes, hipMemcpyDeviceToHost));
    return UGSM_OK;
}

}  // extern "C"
