// ugsm_shard.cpp -- the fovea shard's one exchange step inside the library: ncclBroadcast of the coarse state on the slot's own stream.
//
// Round 4 left the exchange to a Python caller (torch.distributed on torch's stream, ordered against the slot's stream by two events and
// a wrapper object); a C++ host -- the node, BASELINE.json: "the ROS host stays C++ and calls HIP through a thin C-ABI layer" -- could not
// run the shard at all.  Here the collective is one more operation in the slot's stream order:
//     pyramids -> [src rank: coarse levels top .. F-1] -> ncclBroadcast(state, 3 x fovH x fovW floats + a status word) -> fine levels F-2 .. 0 of this rank's window
// No event, no second stream, nothing waits on the host.
//
// FAILURE PROTOCOL (round 6): a collective is a promise to the other ranks, so every rank reaches the broadcast whatever
// happened to it before -- a rank whose pyramids or coarse phase were refused (out of memory, a HIP error) still broadcasts / receives, and
// the state it sends carries a STATUS WORD behind the 3 x fovH x fovW floats: 0, or the source's failure status.  Every rank copies the word
// to the host on the same stream; ugsm_wait / ugsm_poll on the slot then answer UGSM_ERR_PEER where the source failed.  A rank that does
// not reach the collective at all (a crashed process) is what ugsm_shard_set_timeout is for: a step that has not finished by its deadline
// aborts the communicator (ncclCommAbort: the kernel stuck in the collective exits) and the wait answers UGSM_ERR_PEER; the shard is then
// dead until ugsm_shard_finalize + ugsm_shard_init.  The reference has one centred fovea on one GPU (MatchGPULib.cpp:1173-1176,
// seeded from level F-1, :1230-1240,1283-1293); the window offset and the exchange are this build's.
//
// RCCL is loaded with dlopen on first use: librccl.so.1 is 570 MB and only a sharding host needs it.  A copy the process has already loaded
// (PyTorch's bundled one) is reused, so that there is one RCCL per process.
#include "ugsm_internal.hpp"

#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <time.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <new>
#include <string>
#include <vector>

using namespace ugsm;

namespace {

struct Rccl {
    void *handle = nullptr;
    std::string origin;
    ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommInitAll)(ncclComm_t *, int, const int *) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*CommAbort)(ncclComm_t) = nullptr;
    ncclResult_t (*CommCount)(const ncclComm_t, int *) = nullptr;
    ncclResult_t (*Broadcast)(const void *, void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*AllReduce)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Send)(const void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Recv)(void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
};

void load_rccl(Rccl &r)
{
    // a copy that is already in the process first (RTLD_NOLOAD), then the system's
    const char *names[] = {"librccl.so.1", "librccl.so"};
    for (const char *n : names) {
        if ((r.handle = dlopen(n, RTLD_NOW | RTLD_NOLOAD))) {
            r.origin = std::string(n) + " (already loaded)";
            break;
        }
    }
    // (development: another library with RCCL's entry points -- tests/fake_rccl.c; like every UGSM_* variable, read under UGSM_DEV=1 only)
    if (!r.handle && dev_env() && getenv("UGSM_RCCL_PATH")) {
        if ((r.handle = dlopen(getenv("UGSM_RCCL_PATH"), RTLD_NOW | RTLD_LOCAL))) r.origin = getenv("UGSM_RCCL_PATH");
    }
    for (const char *n : names) {
        if (r.handle) break;
        if ((r.handle = dlopen(n, RTLD_NOW | RTLD_LOCAL))) r.origin = n;
    }
    if (!r.handle) {
        if ((r.handle = dlopen("/opt/rocm/lib/librccl.so.1", RTLD_NOW | RTLD_LOCAL))) r.origin = "/opt/rocm/lib/librccl.so.1";
    }
    if (!r.handle) return;
    bool ok = true;
    auto sym = [&](const char *name) {
        void *p = dlsym(r.handle, name);
        ok = ok && p != nullptr;
        return p;
    };
    r.GetUniqueId = reinterpret_cast<decltype(r.GetUniqueId)>(sym("ncclGetUniqueId"));
    r.CommInitRank = reinterpret_cast<decltype(r.CommInitRank)>(sym("ncclCommInitRank"));
    r.CommInitAll = reinterpret_cast<decltype(r.CommInitAll)>(sym("ncclCommInitAll"));
    r.CommDestroy = reinterpret_cast<decltype(r.CommDestroy)>(sym("ncclCommDestroy"));
    r.CommAbort = reinterpret_cast<decltype(r.CommAbort)>(sym("ncclCommAbort"));
    r.CommCount = reinterpret_cast<decltype(r.CommCount)>(sym("ncclCommCount"));
    r.Broadcast = reinterpret_cast<decltype(r.Broadcast)>(sym("ncclBroadcast"));
    r.AllReduce = reinterpret_cast<decltype(r.AllReduce)>(sym("ncclAllReduce"));
    r.Send = reinterpret_cast<decltype(r.Send)>(sym("ncclSend"));
    r.Recv = reinterpret_cast<decltype(r.Recv)>(sym("ncclRecv"));
    r.GroupStart = reinterpret_cast<decltype(r.GroupStart)>(sym("ncclGroupStart"));
    r.GroupEnd = reinterpret_cast<decltype(r.GroupEnd)>(sym("ncclGroupEnd"));
    r.GetErrorString = reinterpret_cast<decltype(r.GetErrorString)>(sym("ncclGetErrorString"));
    if (!ok) {
        r.handle = nullptr;  // (left loaded: a library that lacks these symbols is not RCCL)
        r.origin.clear();
    }
}

Rccl *rccl()
{
    static Rccl r;
    static std::once_flag once;
    std::call_once(once, []() noexcept {
        try {
            load_rccl(r);
        } catch (...) {  // (out of host memory for the origin strings: no RCCL then, not an exception across the C-ABI)
            r.handle = nullptr;
        }
    });
    return r.handle ? &r : nullptr;
}

constexpr size_t kStatusWords = 4;  // floats behind the state: [0] = the source's status (bit pattern of an int), the rest padding

struct Shard {
    ncclComm_t comm = nullptr;
    int rank = 0, world = 1;
    bool dead = false;                // the communicator was aborted (a step's deadline passed): finalize and init again
    long long timeout_ms = 0;         // deadline of a step, from its submission (0 = none: ugsm_wait waits as long as it takes)
    std::vector<float *> state;       // per slot: level F-1's (dx, dy, conf) + kStatusWords -- what is broadcast
    std::vector<size_t> state_cap;    // floats (without the status words)
    float *count_buf = nullptr;       // two floats for ugsm_shard_count_ranks
    int *host_word = nullptr;         // page-locked, one int per slot: the status word of the slot's last step, copied on the slot's stream
    std::vector<char> pending;        // per slot: a step whose status word has not been looked at yet
    std::vector<int> src;             // ... and its source rank
    std::vector<long long> deadline;  // ... and its deadline (CLOCK_MONOTONIC ns; 0 = none)
};

long long mono_ns()
{
    timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (long long)ts.tv_sec * 1000000000LL + ts.tv_nsec;
}

int nccl_fail(ugsm_ctx *ctx, const char *what, ncclResult_t e)
{
    char b[384];
    Rccl *r = rccl();
    snprintf(b, sizeof b, "%s failed: %s", what, (r && r->GetErrorString) ? r->GetErrorString(e) : "RCCL error");
    return ctx_fail(ctx, UGSM_ERR_DEVICE, b);
}
#define NCHK(ctx, call)                                            \
    do {                                                           \
        ncclResult_t e__ = (call);                                 \
        if (e__ != ncclSuccess) return nccl_fail(ctx, #call, e__); \
    } while (0)

void shard_free(ugsm_ctx *ctx, void *p)
{
    Shard *s = static_cast<Shard *>(p);
    Rccl *r = rccl();
    (void)hipSetDevice(ctx_config(ctx).device);
    if (s->comm && r) (void)r->CommDestroy(s->comm);
    for (float *b : s->state)
        if (b) (void)ugsm_dev_free(ctx, b);
    if (s->count_buf) (void)ugsm_dev_free(ctx, s->count_buf);
    if (s->host_word) (void)ugsm_host_free(ctx, s->host_word);
    delete s;
}

Shard *shard_of(ugsm_ctx *ctx) { return static_cast<Shard *>(ctx_hooks(ctx).shard); }

int shard_wait(ugsm_ctx *ctx, int slot, int block);

// The context joins the communicator (on failure nothing is attached and `comm` stays the caller's to destroy).
int attach(ugsm_ctx *ctx, ncclComm_t comm, int rank, int world)
{
    return no_throw(ctx, "ugsm_shard_init: out of host memory", [&]() -> int {
        const size_t slots = (size_t)ctx_config(ctx).slots;
        Shard *s = new Shard();
        s->rank = rank;
        s->world = world;
        s->state.assign(slots, nullptr);
        s->state_cap.assign(slots, 0);
        s->pending.assign(slots, 0);
        s->src.assign(slots, 0);
        s->deadline.assign(slots, 0);
        void *p = nullptr;
        const int st = ugsm_host_alloc(ctx, &p, (long long)(slots * sizeof(int)));
        if (st != UGSM_OK) {
            delete s;
            return st;
        }
        s->host_word = static_cast<int *>(p);
        memset(s->host_word, 0, slots * sizeof(int));
        s->comm = comm;
        CtxHooks &h = ctx_hooks(ctx);
        h.shard = s;
        h.shard_free = shard_free;
        h.shard_wait = shard_wait;
        return UGSM_OK;
    });
}

void detach(ugsm_ctx *ctx)
{
    CtxHooks &h = ctx_hooks(ctx);
    if (!h.shard) return;
    shard_free(ctx, h.shard);
    h.shard = nullptr;
    h.shard_free = nullptr;
    h.shard_wait = nullptr;
}

int need_shard(ugsm_ctx *ctx, Shard **out, Rccl **lib)
{
    if (!ctx) return UGSM_ERR_BAD_ARG;
    *out = shard_of(ctx);
    *lib = rccl();
    if (!*out || !*lib) return ctx_fail(ctx, UGSM_ERR_STATE, "the context is not part of a shard (ugsm_shard_init first)");
    if ((*out)->dead)
        return ctx_fail(ctx, UGSM_ERR_STATE, "the shard's communicator was aborted when a step missed its deadline: ugsm_shard_finalize, then ugsm_shard_init again on every rank");
    if (hipSetDevice(ctx_config(ctx).device) != hipSuccess) return ctx_fail(ctx, UGSM_ERR_DEVICE, "hipSetDevice failed");
    return UGSM_OK;
}

int slot_stream(ugsm_ctx *ctx, int slot, hipStream_t *st)
{
    *st = static_cast<hipStream_t>(ctx_slot_stream(ctx, slot));
    return *st ? UGSM_OK : ctx_fail(ctx, UGSM_ERR_BAD_ARG, "slot out of range");
}

// ugsm_wait / ugsm_poll on a slot (CtxHooks::shard_wait).  block: wait for the slot's step, at most until its deadline -- a step that is
// still running then has a rank that never reached the exchange: the communicator is aborted (the kernel stuck in the collective exits; the
// runtime's own wait behind this hook then returns) and the shard is dead.  Then, the slot being idle, the status word of the step.
int shard_wait(ugsm_ctx *ctx, int slot, int block)
{
    Shard *s = shard_of(ctx);
    if (!s || slot < 0 || slot >= (int)s->pending.size() || !s->pending[(size_t)slot]) return UGSM_OK;
    hipStream_t stream;
    if (slot_stream(ctx, slot, &stream) != UGSM_OK) return UGSM_OK;
    if (block && s->deadline[(size_t)slot] > 0 && !s->dead) {
        while (hipStreamQuery(stream) == hipErrorNotReady) {
            if (mono_ns() > s->deadline[(size_t)slot]) {
                Rccl *r = rccl();
                if (r && s->comm) (void)r->CommAbort(s->comm);
                s->comm = nullptr;
                s->dead = true;
                std::fill(s->pending.begin(), s->pending.end(), 0);
                char b[256];
                snprintf(b, sizeof b, "the shard step on slot %d did not finish within %lld ms: a rank never reached the exchange; the communicator was aborted", slot,
                         s->timeout_ms);
                return ctx_fail(ctx, UGSM_ERR_PEER, b);
            }
            timespec ts = {0, 50000};
            nanosleep(&ts, nullptr);
        }
        (void)hipGetLastError();
    }
    if (block && hipStreamSynchronize(stream) != hipSuccess) return UGSM_OK;  // (the runtime's own wait reports the HIP error)
    s->pending[(size_t)slot] = 0;
    const int word = s->host_word[slot];
    if (word == UGSM_OK) return UGSM_OK;
    char b[256];
    snprintf(b, sizeof b, "rank %d, the source of the shard step on slot %d, failed its pyramids or coarse phase with status %d (%s): this rank's stack is not valid", s->src[(size_t)slot],
             slot, word, ugsm_status_string(word));
    return ctx_fail(ctx, UGSM_ERR_PEER, b);
}

}  // namespace

extern "C" {

int ugsm_shard_unique_id(void *id128)
{
    static_assert(sizeof(ncclUniqueId) == UGSM_SHARD_ID_BYTES, "UGSM_SHARD_ID_BYTES is not sizeof(ncclUniqueId)");
    if (!id128) return UGSM_ERR_BAD_ARG;
    Rccl *r = rccl();
    if (!r) return UGSM_ERR_NO_DEVICE;
    ncclUniqueId id;
    if (r->GetUniqueId(&id) != ncclSuccess) return UGSM_ERR_DEVICE;
    memcpy(id128, &id, sizeof id);
    return UGSM_OK;
}

int ugsm_shard_init(ugsm_ctx *ctx, const void *id128, int rank, int world)
{
    if (!ctx || !id128 || world < 1 || rank < 0 || rank >= world) return UGSM_ERR_BAD_ARG;
    if (shard_of(ctx)) return ctx_fail(ctx, UGSM_ERR_STATE, "ugsm_shard_init: the context already belongs to a communicator");
    Rccl *r = rccl();
    if (!r) return ctx_fail(ctx, UGSM_ERR_NO_DEVICE, "ugsm_shard_init: librccl.so.1 not found (set UGSM_RCCL_PATH or the loader path)");
    if (hipSetDevice(ctx_config(ctx).device) != hipSuccess) return ctx_fail(ctx, UGSM_ERR_DEVICE, "hipSetDevice failed");
    ncclUniqueId id;
    memcpy(&id, id128, sizeof id);
    ncclComm_t comm = nullptr;
    NCHK(ctx, r->CommInitRank(&comm, world, id, rank));
    const int st = attach(ctx, comm, rank, world);
    if (st != UGSM_OK) (void)r->CommDestroy(comm);
    return st;
}

int ugsm_shard_init_all(ugsm_ctx *const *ctxs, int n)
{
    if (!ctxs || n < 1 || n > 64) return UGSM_ERR_BAD_ARG;
    for (int i = 0; i < n; i++) {
        if (!ctxs[i]) return UGSM_ERR_BAD_ARG;
        if (shard_of(ctxs[i])) return ctx_fail(ctxs[i], UGSM_ERR_STATE, "ugsm_shard_init_all: a context already belongs to a communicator");
        for (int j = 0; j < i; j++)
            if (ctx_config(ctxs[j]).device == ctx_config(ctxs[i]).device)
                return ctx_fail(ctxs[i], UGSM_ERR_BAD_ARG, "ugsm_shard_init_all: two contexts on one device");
    }
    Rccl *r = rccl();
    if (!r) return ctx_fail(ctxs[0], UGSM_ERR_NO_DEVICE, "ugsm_shard_init_all: librccl.so.1 not found (set UGSM_RCCL_PATH or the loader path)");
    int dev[64];
    ncclComm_t comm[64] = {};
    for (int i = 0; i < n; i++) dev[i] = ctx_config(ctxs[i]).device;
    NCHK(ctxs[0], r->CommInitAll(comm, n, dev));
    for (int i = 0; i < n; i++) {
        const int st = attach(ctxs[i], comm[i], i, n);
        if (st != UGSM_OK) {
            // all or nothing: the contexts attached so far leave again (their ranks go with them), the other ranks are destroyed
            for (int j = 0; j < i; j++) detach(ctxs[j]);
            for (int j = i; j < n; j++) {
                (void)hipSetDevice(dev[j]);
                (void)r->CommDestroy(comm[j]);
            }
            return st;
        }
    }
    return UGSM_OK;
}

int ugsm_shard_rank(const ugsm_ctx *ctx, int *rank, int *world)
{
    if (!ctx) return UGSM_ERR_BAD_ARG;
    const Shard *s = shard_of(const_cast<ugsm_ctx *>(ctx));
    if (!s) return UGSM_ERR_STATE;
    if (rank) *rank = s->rank;
    if (world) *world = s->world;
    return UGSM_OK;
}

int ugsm_shard_count_ranks(ugsm_ctx *ctx, int *ranks)
{
    Shard *s;
    Rccl *r;
    const int st = need_shard(ctx, &s, &r);
    if (st != UGSM_OK) return st;
    if (!ranks) return UGSM_ERR_BAD_ARG;
    if (!s->count_buf) {
        void *p = nullptr;
        const int a = ugsm_dev_alloc(ctx, &p, 2 * sizeof(float));
        if (a != UGSM_OK) return a;
        s->count_buf = static_cast<float *>(p);
    }
    hipStream_t stream;
    const int g = slot_stream(ctx, 0, &stream);
    if (g != UGSM_OK) return g;
    const float one = 1.0f;
    float sum = 0.0f;
    if (hipMemcpyAsync(s->count_buf, &one, sizeof one, hipMemcpyHostToDevice, stream) != hipSuccess) return ctx_fail(ctx, UGSM_ERR_DEVICE, "hipMemcpyAsync failed");
    NCHK(ctx, r->AllReduce(s->count_buf, s->count_buf + 1, 1, ncclFloat, ncclSum, s->comm, stream));
    if (hipMemcpyAsync(&sum, s->count_buf + 1, sizeof sum, hipMemcpyDeviceToHost, stream) != hipSuccess || hipStreamSynchronize(stream) != hipSuccess)
        return ctx_fail(ctx, UGSM_ERR_DEVICE, "ugsm_shard_count_ranks: copy or synchronise failed");
    *ranks = (int)(sum + 0.5f);
    return UGSM_OK;
}

int ugsm_shard_set_timeout(ugsm_ctx *ctx, long long milliseconds)
{
    if (!ctx || milliseconds < 0) return UGSM_ERR_BAD_ARG;
    Shard *s = shard_of(ctx);
    if (!s) return ctx_fail(ctx, UGSM_ERR_STATE, "the context is not part of a shard (ugsm_shard_init first)");
    s->timeout_ms = milliseconds;
    return UGSM_OK;
}

int ugsm_submit_fovea_shard(ugsm_ctx *ctx, int slot, const uint8_t *d_rgbL, const uint8_t *d_rgbR, int W, int H, int stride, int off_x, int off_y,
                            float *d_stack, int src_rank)
{
    Shard *s;
    Rccl *r;
    int st = need_shard(ctx, &s, &r);
    if (st != UGSM_OK) return st;
    // ---- what every rank refuses alike, before anything is enqueued anywhere (the ranks make the same calls with the same geometry) -------
    const ugsm_config &cfg = ctx_config(ctx);
    if (slot < 0 || slot >= cfg.slots || !d_rgbL || !d_rgbR || !d_stack || src_rank < 0 || src_rank >= s->world)
        return ctx_fail(ctx, UGSM_ERR_BAD_ARG, "ugsm_submit_fovea_shard: bad slot, rank or buffer");
    if (cfg.fovea_levels < 2) return ctx_fail(ctx, UGSM_ERR_BAD_ARG, "ugsm_submit_fovea_shard: the context has no fovea levels");
    int fw = 0, fh = 0;
    if ((st = ugsm_fovea_dims(W, H, cfg.levels, cfg.fovea_levels, &fw, &fh)) != UGSM_OK) return st;
    if (stride < 3 * W) return ctx_fail(ctx, UGSM_ERR_SIZE_MISMATCH, "ugsm_submit_fovea_shard: stride < 3 * W");
    if (ctx_hooks(ctx).queue_busy) return ctx_fail(ctx, UGSM_ERR_STATE, "ugsm_submit_fovea_shard: pairs enqueued with ugsm_enqueue_* are outstanding");
    hipStream_t stream;
    if ((st = slot_stream(ctx, slot, &stream)) != UGSM_OK) return st;
    // ---- the one buffer this rank cannot meet the exchange without: allocated before anything is enqueued.  A rank that cannot have even
    // this (3 MB) leaves its peers waiting: their ugsm_shard_set_timeout deadline is what ends that. -----------------------------------------
    const size_t n = 3 * (size_t)fw * fh;
    if (s->state_cap[(size_t)slot] < n) {
        // (a reallocation: the slot's earlier work may still read the old buffer)
        if ((st = ugsm_wait(ctx, slot)) != UGSM_OK && st != UGSM_ERR_PEER) return st;
        if (s->state[(size_t)slot]) (void)ugsm_dev_free(ctx, s->state[(size_t)slot]);
        s->state[(size_t)slot] = nullptr;
        s->state_cap[(size_t)slot] = 0;
        void *p = nullptr;
        if ((st = ugsm_dev_alloc(ctx, &p, (long long)((n + kStatusWords) * sizeof(float)))) != UGSM_OK) return st;
        s->state[(size_t)slot] = static_cast<float *>(p);
        s->state_cap[(size_t)slot] = n;
    }
    float *state = s->state[(size_t)slot];
    // ---- from here on this rank REACHES THE BROADCAST whatever happens to it.  Everything goes onto the slot's stream, in this order; the
    // state buffer belongs to the slot: a later step's broadcast into it is ordered, by the stream, after the fine phase of the step before.
    int mine = ugsm_submit_pyramids(ctx, slot, d_rgbL, d_rgbR, W, H, stride);
    if (mine == UGSM_OK && s->rank == src_rank) mine = ugsm_submit_fovea_coarse(ctx, slot, state);
    char why[512] = "";  // (the message of this rank's own failure, kept past the calls below; no allocation on this path)
    if (mine != UGSM_OK) snprintf(why, sizeof why, "%s", ugsm_last_error(ctx));
    if (s->rank == src_rank) {  // the status word the state carries: what the source made of its part
        // (a stream that refuses even this is broken and the broadcast below will say so too; the promise is kept regardless: no return here)
        if (hipMemsetD32Async(reinterpret_cast<hipDeviceptr_t>(state + n), mine, 1, stream) != hipSuccess && mine == UGSM_OK) {
            mine = UGSM_ERR_DEVICE;
            snprintf(why, sizeof why, "ugsm_submit_fovea_shard: the status word could not be written");
        }
    }
    NCHK(ctx, r->Broadcast(state, state, n + kStatusWords, ncclFloat, src_rank, s->comm, stream));
    if (mine != UGSM_OK) {  // (this rank knows: its submit says so; the peers read the word)
        s->pending[(size_t)slot] = 0;
        ctx_fail(ctx, mine, why);
        return mine;
    }
    if (hipMemcpyAsync(&s->host_word[slot], state + n, sizeof(int), hipMemcpyDeviceToHost, stream) != hipSuccess)
        return ctx_fail(ctx, UGSM_ERR_DEVICE, "ugsm_submit_fovea_shard: copy of the status word");
    s->pending[(size_t)slot] = 1;
    s->src[(size_t)slot] = src_rank;
    s->deadline[(size_t)slot] = s->timeout_ms > 0 ? mono_ns() + s->timeout_ms * 1000000LL : 0;
    // (with a failed source the fine phase below works on a state nobody computed: wasted work on the failure path only, never reported
    // as a result -- ugsm_wait answers UGSM_ERR_PEER)
    return ugsm_submit_fovea_fine(ctx, slot, state, off_x, off_y, d_stack);
}

int ugsm_shard_gather(ugsm_ctx *ctx, int slot, const float *d_stack, long long stack_floats, float *d_all, int dst_rank)
{
    Shard *s;
    Rccl *r;
    int st = need_shard(ctx, &s, &r);
    if (st != UGSM_OK) return st;
    if (slot < 0 || slot >= ctx_config(ctx).slots || !d_stack || stack_floats < 1 || dst_rank < 0 || dst_rank >= s->world ||
        (s->rank == dst_rank && !d_all))
        return ctx_fail(ctx, UGSM_ERR_BAD_ARG, "ugsm_shard_gather: bad slot, rank or buffer");
    hipStream_t stream;
    if ((st = slot_stream(ctx, slot, &stream)) != UGSM_OK) return st;
    const size_t n = (size_t)stack_floats;
    if (s->rank != dst_rank) {
        NCHK(ctx, r->Send(d_stack, n, ncclFloat, dst_rank, s->comm, stream));
        return UGSM_OK;
    }
    if (hipMemcpyAsync(d_all + (size_t)dst_rank * n, d_stack, n * sizeof(float), hipMemcpyDeviceToDevice, stream) != hipSuccess)
        return ctx_fail(ctx, UGSM_ERR_DEVICE, "ugsm_shard_gather: device copy failed");
    if (s->world > 1) {
        NCHK(ctx, r->GroupStart());
        ncclResult_t bad = ncclSuccess;  // (the group is closed whatever a receive answers)
        for (int p = 0; p < s->world && bad == ncclSuccess; p++)
            if (p != dst_rank) bad = r->Recv(d_all + (size_t)p * n, n, ncclFloat, p, s->comm, stream);
        const ncclResult_t end = r->GroupEnd();
        if (bad != ncclSuccess) return nccl_fail(ctx, "ncclRecv", bad);
        NCHK(ctx, end);
    }
    return UGSM_OK;
}

int ugsm_shard_finalize(ugsm_ctx *ctx)
{
    if (!ctx) return UGSM_ERR_BAD_ARG;
    CtxHooks &h = ctx_hooks(ctx);
    if (!h.shard) return UGSM_OK;
    (void)ugsm_wait_all(ctx);
    detach(ctx);
    return UGSM_OK;
}

}  // extern "C"
