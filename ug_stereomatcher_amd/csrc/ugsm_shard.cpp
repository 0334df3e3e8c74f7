// ugsm_shard.cpp -- the fovea shard's one exchange step inside the library: ncclBroadcast of the coarse state on the slot's own stream.
//
// Round 4 left the exchange to a Python caller (torch.distributed on torch's stream, ordered against the slot's stream by two events and
// a wrapper object); a C++ host -- the node, BASELINE.json: "the ROS host stays C++ and calls HIP through a thin C-ABI layer" -- could not
// run the shard at all.  Here the collective is one more operation in the slot's stream order:
//     pyramids -> [src rank: coarse levels top .. F-1] -> ncclBroadcast(state, 3 x fovH x fovW floats) -> fine levels F-2 .. 0 of this rank's window
// No event, no second stream, nothing waits on the host.  The reference has one centred fovea on one GPU (MatchGPULib.cpp:1173-1176,
// seeded from level F-1, :1230-1240,1283-1293); the window offset and the exchange are this build's.
//
// RCCL is loaded with dlopen on first use: librccl.so.1 is 570 MB and only a sharding host needs it.  A copy the process has already loaded
// (PyTorch's bundled one) is reused, so that there is one RCCL per process.
#include "ugsm_internal.hpp"

#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

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
    ncclResult_t (*CommCount)(const ncclComm_t, int *) = nullptr;
    ncclResult_t (*Broadcast)(const void *, void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*AllReduce)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Send)(const void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Recv)(void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
};

Rccl *rccl()
{
    static Rccl r;
    static std::once_flag once;
    std::call_once(once, [] {
        // a copy that is already in the process first (RTLD_NOLOAD), then the system's
        const char *names[] = {"librccl.so.1", "librccl.so"};
        for (const char *n : names) {
            if ((r.handle = dlopen(n, RTLD_NOW | RTLD_NOLOAD))) {
                r.origin = std::string(n) + " (already loaded)";
                break;
            }
        }
        if (!r.handle && getenv("UGSM_RCCL_PATH")) {
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
    });
    return r.handle ? &r : nullptr;
}

struct Shard {
    ncclComm_t comm = nullptr;
    int rank = 0, world = 1;
    std::vector<float *> state;       // per slot: level F-1's (dx, dy, conf) -- what is broadcast
    std::vector<size_t> state_cap;    // floats
    float *count_buf = nullptr;       // two floats for ugsm_shard_count_ranks
};

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
    delete s;
}

Shard *shard_of(ugsm_ctx *ctx) { return static_cast<Shard *>(ctx_hooks(ctx).shard); }

int attach(ugsm_ctx *ctx, ncclComm_t comm, int rank, int world)
{
    Shard *s = new (std::nothrow) Shard();
    if (!s) return ctx_fail(ctx, UGSM_ERR_NOMEM, "ugsm_shard_init: out of host memory");
    s->comm = comm;
    s->rank = rank;
    s->world = world;
    s->state.assign((size_t)ctx_config(ctx).slots, nullptr);
    s->state_cap.assign((size_t)ctx_config(ctx).slots, 0);
    CtxHooks &h = ctx_hooks(ctx);
    h.shard = s;
    h.shard_free = shard_free;
    return UGSM_OK;
}

int need_shard(ugsm_ctx *ctx, Shard **out, Rccl **lib)
{
    if (!ctx) return UGSM_ERR_BAD_ARG;
    *out = shard_of(ctx);
    *lib = rccl();
    if (!*out || !*lib) return ctx_fail(ctx, UGSM_ERR_STATE, "the context is not part of a shard (ugsm_shard_init first)");
    if (hipSetDevice(ctx_config(ctx).device) != hipSuccess) return ctx_fail(ctx, UGSM_ERR_DEVICE, "hipSetDevice failed");
    return UGSM_OK;
}

int slot_stream(ugsm_ctx *ctx, int slot, hipStream_t *st)
{
    void *p = nullptr;
    const int r = ugsm_slot_stream(ctx, slot, &p);
    *st = static_cast<hipStream_t>(p);
    return r;
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
            for (int j = i; j < n; j++) (void)r->CommDestroy(comm[j]);
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

int ugsm_submit_fovea_shard(ugsm_ctx *ctx, int slot, const uint8_t *d_rgbL, const uint8_t *d_rgbR, int W, int H, int stride, int off_x, int off_y,
                            float *d_stack, int src_rank)
{
    Shard *s;
    Rccl *r;
    int st = need_shard(ctx, &s, &r);
    if (st != UGSM_OK) return st;
    const ugsm_config &cfg = ctx_config(ctx);
    if (slot < 0 || slot >= cfg.slots || !d_stack || src_rank < 0 || src_rank >= s->world) return ctx_fail(ctx, UGSM_ERR_BAD_ARG, "ugsm_submit_fovea_shard: bad slot, rank or buffer");
    int fw = 0, fh = 0;
    if ((st = ugsm_fovea_dims(W, H, cfg.levels, cfg.fovea_levels, &fw, &fh)) != UGSM_OK) return st;
    if (cfg.fovea_levels < 2) return ctx_fail(ctx, UGSM_ERR_BAD_ARG, "ugsm_submit_fovea_shard: the context has no fovea levels");
    const size_t n = 3 * (size_t)fw * fh;
    if (s->state_cap[(size_t)slot] < n) {
        // (a reallocation: the slot's earlier work may still read the old buffer)
        if ((st = ugsm_wait(ctx, slot)) != UGSM_OK) return st;
        if (s->state[(size_t)slot]) (void)ugsm_dev_free(ctx, s->state[(size_t)slot]);
        s->state[(size_t)slot] = nullptr;
        s->state_cap[(size_t)slot] = 0;
        void *p = nullptr;
        if ((st = ugsm_dev_alloc(ctx, &p, (long long)(n * sizeof(float)))) != UGSM_OK) return st;
        s->state[(size_t)slot] = static_cast<float *>(p);
        s->state_cap[(size_t)slot] = n;
    }
    float *state = s->state[(size_t)slot];
    hipStream_t stream;
    if ((st = slot_stream(ctx, slot, &stream)) != UGSM_OK) return st;
    // Everything below goes onto the slot's stream, in this order.  The state buffer belongs to the slot: a later step's broadcast into
    // it is ordered, by the stream, after the fine phase of the step before.
    if ((st = ugsm_submit_pyramids(ctx, slot, d_rgbL, d_rgbR, W, H, stride)) != UGSM_OK) return st;
    if (s->rank == src_rank && (st = ugsm_submit_fovea_coarse(ctx, slot, state)) != UGSM_OK) return st;
    NCHK(ctx, r->Broadcast(state, state, n, ncclFloat, src_rank, s->comm, stream));
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
    shard_free(ctx, h.shard);
    h.shard = nullptr;
    h.shard_free = nullptr;
    return UGSM_OK;
}

}  // extern "C"
