// ugsm_internal.hpp -- what the layers above the slot API (ugsm_queue.cpp: the queue; ugsm_shard.cpp: RCCL) need from the runtime
// besides include/ugsm.h itself.  Both are written against the public slot-level entry points (ugsm_submit_*, ugsm_wait, ugsm_poll,
// ugsm_slot_stream): they are what a host would otherwise have to write, moved behind the C-ABI.
#pragma once

#include "../../include/ugsm.h"

#include <stddef.h>

#include <new>

namespace ugsm {

// State the layers hang on a context; the runtime owns the storage and calls the `free` hooks from ugsm_destroy (before the slots go).
struct CtxHooks {
    void *queue = nullptr;
    void (*queue_free)(ugsm_ctx *, void *) = nullptr;
    void *shard = nullptr;
    void (*shard_free)(ugsm_ctx *, void *) = nullptr;
    bool queue_busy = false;   // pairs are outstanding in the queue: the slots belong to it (slot-level entry points answer UGSM_ERR_STATE)
    bool queue_calling = false;  // ... except while the queue itself is calling them
    bool queue_more = false;     // (while queue_calling) other calls follow the one being sent: it shares the chip (call_alone)
    // The shard's part in ugsm_wait / ugsm_poll (block = 0) on a slot that holds a shard step: called BEFORE the runtime waits for the slot.
    // UGSM_OK: go on (wait / query the slot as usual); UGSM_PENDING: (poll) not finished; anything else is returned to the caller once the
    // slot has drained -- a peer rank failed its part of the step, or the step's deadline passed and the communicator was aborted.
    int (*shard_wait)(ugsm_ctx *, int slot, int block) = nullptr;
};
CtxHooks &ctx_hooks(ugsm_ctx *ctx);
const ugsm_config &ctx_config(const ugsm_ctx *ctx);
// the HIP stream of a slot (a hipStream_t), nullptr for a slot the context does not have; unlike ugsm_slot_stream it leaves the slot's
// bookkeeping alone (the public call takes it that the host is about to enqueue work of its own there: the slot becomes busy)
void *ctx_slot_stream(ugsm_ctx *ctx, int slot);
// sets ugsm_last_error and returns `status`
int ctx_fail(ugsm_ctx *ctx, int status, const char *what);
// memcpy by the context's host team (large copies; falls back to the calling thread)
void ctx_host_copy(ugsm_ctx *ctx, void *dst, const void *src, size_t bytes);
// hipPointerGetAttributes says page-locked host memory
bool host_pinned(const void *p);
// UGSM_DEV=1 is set: the process asked for the development switches (UGSM_* environment variables); nothing reads one without it
bool dev_env();
// the stagger of the queue's first round after idle and the size of every later call (ugsm.h, "the queue")
int queue_target(int batch, int slots, long long calls_since_idle);

// The body of a C entry point that uses growing containers: nothing is thrown across the C-ABI.
template <class F>
int no_throw(ugsm_ctx *ctx, const char *entry, F &&body) noexcept
{
    try {
        return body();
    } catch (const std::bad_alloc &) {
        return ctx ? ctx_fail(ctx, UGSM_ERR_NOMEM, entry) : UGSM_ERR_NOMEM;
    } catch (...) {
        return ctx ? ctx_fail(ctx, UGSM_ERR_STATE, entry) : UGSM_ERR_STATE;
    }
}

}  // namespace ugsm
