// fake_runtime.cpp -- TEST INFRASTRUCTURE: a stand-in for csrc/ugsm_runtime.cpp underneath csrc/ugsm_queue.cpp, so that the queue of
// include/ugsm.h (call formation, the stagger, ordering, back-pressure, the `more` hint of the kernel policy, the failure contract) can be
// driven on a machine without a GPU.  The queue is written against the public slot-level entry points and csrc/ugsm_internal.hpp only;
// this file implements exactly those -- as a recorder: a "submit" notes the call, a slot finishes after a set number of ugsm_poll queries,
// chosen calls fail with a chosen status -- and replaces operator new inside this shared object so that a test can make the Nth host
// allocation fail (what the queue promises then: every accepted pair is still reported exactly once; include/ugsm.h, "Return value of
// every ugsm_enqueue_*").  Built by tests/test_queue_host.py with g++ (no HIP call anywhere in the queue); never shipped, never loaded
// by the product.
#include "../ug_stereomatcher_amd/csrc/ugsm_internal.hpp"

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>
#include <vector>

using namespace ugsm;

// ---- allocation faults: the Nth allocation from now throws (one shot) -------------------------------------------------------------------
static long long g_countdown = -1, g_allocs = 0;
static void *counted_alloc(size_t n, bool nothrow)
{
    g_allocs++;
    if (g_countdown >= 0 && g_countdown-- == 0) {
        if (nothrow) return nullptr;
        throw std::bad_alloc();
    }
    void *p = malloc(n ? n : 1);
    if (!p && !nothrow) throw std::bad_alloc();
    return p;
}
void *operator new(size_t n) { return counted_alloc(n, false); }
void *operator new[](size_t n) { return counted_alloc(n, false); }
void *operator new(size_t n, const std::nothrow_t &) noexcept { return counted_alloc(n, true); }
void *operator new[](size_t n, const std::nothrow_t &) noexcept { return counted_alloc(n, true); }
void operator delete(void *p) noexcept { free(p); }
void operator delete[](void *p) noexcept { free(p); }
void operator delete(void *p, size_t) noexcept { free(p); }
void operator delete[](void *p, size_t) noexcept { free(p); }

struct FakeCall {
    int slot, n, mode, mem, status, more, drained;
    const void *L[UGSM_MAX_BATCH];
};
struct FakeSlot {
    bool busy = false;
    int polls_left = 0;
    long long call = -1;  // index into ugsm_ctx::calls of what the slot holds
};
struct ugsm_ctx {
    ugsm_config cfg;
    CtxHooks hooks;
    char err[256] = "";
    std::vector<FakeSlot> slots;
    std::vector<FakeCall> calls;
    std::vector<int> fail;  // status the k-th submit returns (0 = none)
    int poll_delay = 0;
    int violations = 0;  // submits on a slot that had not been seen finished; calls outside the queue's own (queue_calling unset)
};

namespace ugsm {
CtxHooks &ctx_hooks(ugsm_ctx *ctx) { return ctx->hooks; }
const ugsm_config &ctx_config(const ugsm_ctx *ctx) { return ctx->cfg; }
void *ctx_slot_stream(ugsm_ctx *, int) { return nullptr; }
int ctx_fail(ugsm_ctx *ctx, int status, const char *what)
{
    snprintf(ctx->err, sizeof ctx->err, "%s", what);
    return status;
}
void ctx_host_copy(ugsm_ctx *, void *dst, const void *src, size_t bytes) { memcpy(dst, src, bytes); }
bool host_pinned(const void *) { return true; }
bool dev_env() { return false; }
}  // namespace ugsm

static int submit(ugsm_ctx *ctx, int slot, int n, int mode, int mem, const uint8_t *const *L)
{
    if (slot < 0 || slot >= (int)ctx->slots.size() || n < 1 || n > UGSM_MAX_BATCH) return UGSM_ERR_BAD_ARG;
    FakeSlot &s = ctx->slots[(size_t)slot];
    if (s.busy || !ctx->hooks.queue_calling) ctx->violations++;
    FakeCall c{};
    c.slot = slot;
    c.n = n;
    c.mode = mode;
    c.mem = mem;
    c.more = ctx->hooks.queue_more ? 1 : 0;
    const size_t k = ctx->calls.size();
    c.status = k < ctx->fail.size() ? ctx->fail[k] : UGSM_OK;
    for (int b = 0; b < n; b++) c.L[b] = L[b];
    ctx->calls.push_back(c);  // (may throw: the fake's own bookkeeping counts as the runtime running out of memory -- before anything "ran")
    s.busy = true;  // a failed submit too: it may have put work on the stream before it failed
    s.polls_left = ctx->poll_delay;
    s.call = (long long)k;
    if (c.status != UGSM_OK) return ctx_fail(ctx, c.status, "fake: this submit was told to fail");
    return UGSM_OK;
}

extern "C" {

void ugsm_default_config(ugsm_config *cfg)
{
    memset(cfg, 0, sizeof *cfg);
    cfg->levels = 14;
    cfg->fovea_levels = 7;
    cfg->slots = 1;
}
int ugsm_level_dims(int W, int H, int levels, int *w, int *h)
{
    if (W < 16 || H < 16 || levels < 1 || levels > UGSM_MAX_LEVELS) return UGSM_ERR_BAD_ARG;
    for (int i = 0; i < levels; i++) {
        w[i] = W >> (i / 2) ? W >> (i / 2) : 1;
        h[i] = H >> (i / 2) ? H >> (i / 2) : 1;
    }
    return UGSM_OK;
}
int ugsm_fovea_dims(int W, int H, int, int, int *fw, int *fh)
{
    *fw = W / 8 > 0 ? W / 8 : 1;
    *fh = H / 8 > 0 ? H / 8 : 1;
    return UGSM_OK;
}
const char *ugsm_last_error(const ugsm_ctx *ctx) { return ctx ? ctx->err : "null context"; }
int ugsm_host_alloc(ugsm_ctx *ctx, void **p, long long bytes)
{
    *p = ::operator new((size_t)bytes, std::nothrow);
    return *p ? UGSM_OK : ctx_fail(ctx, UGSM_ERR_NOMEM, "fake: host allocation failed");
}
int ugsm_host_free(ugsm_ctx *, void *p)
{
    ::operator delete(p);
    return UGSM_OK;
}

int ugsm_submit_full(ugsm_ctx *ctx, int slot, const uint8_t *L, const uint8_t *, int, int, int, float *) { return submit(ctx, slot, 1, 0, 0, &L); }
int ugsm_submit_full_batch(ugsm_ctx *ctx, int slot, int n, const uint8_t *const *L, const uint8_t *const *, int, int, int, float *const *)
{
    return submit(ctx, slot, n, 0, 0, L);
}
int ugsm_submit_foveated(ugsm_ctx *ctx, int slot, const uint8_t *L, const uint8_t *, int, int, int, int, int, float *, float *, float *)
{
    return submit(ctx, slot, 1, 1, 0, &L);
}
int ugsm_submit_foveated_batch(ugsm_ctx *ctx, int slot, int n, const uint8_t *const *L, const uint8_t *const *, int, int, int, const int *, const int *,
                               float *const *, float *const *, float *const *)
{
    return submit(ctx, slot, n, 1, 0, L);
}
int ugsm_submit_full_host(ugsm_ctx *ctx, int slot, const uint8_t *L, const uint8_t *, int, int, int, float *, float *, float *) { return submit(ctx, slot, 1, 0, 1, &L); }
int ugsm_submit_full_batch_host(ugsm_ctx *ctx, int slot, int n, const uint8_t *const *L, const uint8_t *const *, int, int, int, float *const *, float *const *,
                                float *const *)
{
    return submit(ctx, slot, n, 0, 1, L);
}
int ugsm_submit_foveated_host(ugsm_ctx *ctx, int slot, const uint8_t *L, const uint8_t *, int, int, int, int, int, float *, float *, float *, float *, float *)
{
    return submit(ctx, slot, 1, 1, 1, &L);
}
int ugsm_submit_foveated_batch_host(ugsm_ctx *ctx, int slot, int n, const uint8_t *const *L, const uint8_t *const *, int, int, int, const int *, const int *,
                                    float *const *, float *const *, float *const *)
{
    return submit(ctx, slot, n, 1, 1, L);
}

int ugsm_poll(ugsm_ctx *ctx, int slot)
{
    if (slot < 0 || slot >= (int)ctx->slots.size()) return UGSM_ERR_BAD_ARG;
    FakeSlot &s = ctx->slots[(size_t)slot];
    if (!s.busy) return UGSM_OK;
    if (s.polls_left > 0) {
        s.polls_left--;
        return UGSM_PENDING;
    }
    s.busy = false;
    if (s.call >= 0) ctx->calls[(size_t)s.call].drained = 1;
    return UGSM_OK;
}
int ugsm_wait(ugsm_ctx *ctx, int slot)
{
    if (slot < 0 || slot >= (int)ctx->slots.size()) return UGSM_ERR_BAD_ARG;
    FakeSlot &s = ctx->slots[(size_t)slot];
    s.busy = false;
    s.polls_left = 0;
    if (s.call >= 0) ctx->calls[(size_t)s.call].drained = 1;
    return UGSM_OK;
}

// ---- the test's handle on the fake ---------------------------------------------------------------------------------------------------
#pragma GCC visibility push(default)
ugsm_ctx *ugsm_fake_create(int slots, int batch, int levels, int fovea_levels)
{
    ugsm_ctx *c = new (std::nothrow) ugsm_ctx();
    if (!c) return nullptr;
    ugsm_default_config(&c->cfg);
    c->cfg.slots = slots;
    c->cfg.batch = batch;
    c->cfg.levels = levels;
    c->cfg.fovea_levels = fovea_levels;
    c->slots.resize((size_t)slots);
    c->calls.reserve(4096);  // (so that the recorder itself rarely allocates inside a scenario)
    return c;
}
void ugsm_fake_destroy(ugsm_ctx *c)
{
    if (!c) return;
    g_countdown = -1;
    if (c->hooks.queue && c->hooks.queue_free) c->hooks.queue_free(c, c->hooks.queue);
    delete c;
}
void ugsm_fake_poll_delay(ugsm_ctx *c, int polls) { c->poll_delay = polls; }
int ugsm_fake_fail_call(ugsm_ctx *c, long long index, int status)
{
    if (index < 0 || index > 1 << 20) return -1;
    if (c->fail.size() <= (size_t)index) c->fail.resize((size_t)index + 1, 0);
    c->fail[(size_t)index] = status;
    return 0;
}
// call k: slot, pairs, mode (0 full / 1 foveated), memory (0 device / 1 host), the status the submit returned, the `more` hint it was sent
// with, whether its slot has been seen finished since, and the left-image pointers of its pairs
long long ugsm_fake_calls(const ugsm_ctx *c) { return (long long)c->calls.size(); }
int ugsm_fake_call(const ugsm_ctx *c, long long k, int *out7, const void **L)
{
    if (k < 0 || k >= (long long)c->calls.size()) return -1;
    const FakeCall &f = c->calls[(size_t)k];
    const int v[7] = {f.slot, f.n, f.mode, f.mem, f.status, f.more, f.drained};
    memcpy(out7, v, sizeof v);
    for (int b = 0; b < f.n; b++) L[b] = f.L[b];
    return 0;
}
int ugsm_fake_violations(const ugsm_ctx *c) { return c->violations; }
void ugsm_fake_fail_alloc_after(long long n) { g_countdown = n; }
long long ugsm_fake_allocs(void) { return g_allocs; }
#pragma GCC visibility pop

}  // extern "C"
