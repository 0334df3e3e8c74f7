// ugsm_queue.cpp -- the queue of include/ugsm.h: ugsm_enqueue_* / ugsm_flush / ugsm_next_done.
//
// The library owns the slots: it forms calls from the backlog ("batch what has piled up", staggered first round), deals them onto the
// slots in rotation and reports completions in enqueue order.  This is the host logic that produced the throughput figures of rounds 3
// and 4 from inside bench.py (plan_calls / run / submit), moved behind the C-ABI so that the C++ node's topic path
// (/root/reference/src/gpu_matcher/UG_GPU_matcher.cpp:126-185,414-494 -- one blocking match() per callback of a single-threaded
// ros::spin, :749-752) gets the same pipeline from three calls.  Written against the public slot-level entry points only.
#include "ugsm_internal.hpp"

#include <time.h>  // (no HIP header: the queue makes no HIP call -- tests/test_queue_host.py builds this file with g++ against a fake runtime)

#include <algorithm>
#include <cstring>
#include <deque>
#include <new>
#include <string>
#include <vector>

using namespace ugsm;

namespace {

enum Mode { M_FULL = 0, M_FOVEA = 1 };
enum Mem { MEM_DEVICE = 0, MEM_PINNED = 1, MEM_MANAGED = 2 };

// page-locked staging of one managed pair: both images in, the result planes out
struct Managed {
    uint8_t *in = nullptr;
    size_t in_cap = 0;
    float *out = nullptr;
    size_t out_cap = 0;  // floats
    bool busy = false;
};

struct Item {
    int mode, mem;
    const uint8_t *L, *R;
    int W, H, stride, off_x, off_y;
    float *out[5];  // device: out[0] = d_out / d_stack, out[3], out[4] = d_pyrL / d_pyrR; host: H, V, C planes (+ pyramid stacks)
    uint64_t tag;
    unsigned long long seq;
    int managed;  // index into Queue::pool, or -1
    bool pyr() const { return out[3] != nullptr || out[4] != nullptr; }
    // pairs of one call: one mode, one kind of memory, one geometry; host calls with pyramid stacks go out alone (the batched host
    // entry point has none)
    bool same_kind(const Item &o) const
    {
        return mode == o.mode && mem == o.mem && W == o.W && H == o.H && stride == o.stride && !(mem != MEM_DEVICE && (pyr() || o.pyr()));
    }
};

struct Call {
    int slot = 0, status = UGSM_OK;
    long long index = 0;
    std::vector<Item> items;
    size_t reported = 0;  // pairs already moved to the done list (retire_front can be resumed after an allocation failure)
};

struct Done {
    ugsm_completion c;
    int managed;  // the managed buffer the completion lends, or -1
};

struct Queue {
    std::deque<Item> waiting;
    std::deque<Call> flight;             // in dispatch order; retired from the front
    std::deque<Done> done;               // retired, not yet fetched
    std::vector<Managed> pool;
    std::vector<int> lent;               // managed buffers the last ugsm_next_done lent to the host
    unsigned long long seq = 0, flush_upto = 0;
    bool round_restarts = false;  // a flush has closed the burst: the next pair enqueued opens a new staggered round
    long long calls = 0, calls_since_idle = 0;
    int next_slot = 0;
    std::vector<char> slot_busy;
    int outstanding() const
    {
        size_t n = waiting.size() + done.size();
        for (const Call &c : flight) n += c.items.size() - c.reported;
        return (int)n;
    }
};

long long now_ns()
{
    timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (long long)ts.tv_sec * 1000000000LL + ts.tv_nsec;
}

int batch_of(const ugsm_config &cfg) { return std::min(std::max(cfg.batch, 1), UGSM_MAX_BATCH); }

void queue_free(ugsm_ctx *ctx, void *p)
{
    Queue *q = static_cast<Queue *>(p);
    for (Managed &m : q->pool) {
        if (m.in) (void)ugsm_host_free(ctx, m.in);
        if (m.out) (void)ugsm_host_free(ctx, m.out);
    }
    delete q;
}

Queue *queue_of(ugsm_ctx *ctx)
{
    CtxHooks &h = ctx_hooks(ctx);
    if (!h.queue) {
        Queue *q = new (std::nothrow) Queue();
        if (!q) return nullptr;
        q->slot_busy.assign((size_t)ctx_config(ctx).slots, 0);
        h.queue = q;
        h.queue_free = queue_free;
    }
    return static_cast<Queue *>(h.queue);
}

void update_busy(ugsm_ctx *ctx, Queue *q) { ctx_hooks(ctx).queue_busy = !(q->waiting.empty() && q->flight.empty() && q->done.empty()); }

// the oldest call in flight is complete: its pairs move to the done list
void retire_front(ugsm_ctx *ctx, Queue *q)
{
    Call &c = q->flight.front();
    const long long t = now_ns();
    for (; c.reported < c.items.size(); c.reported++) {  // (push_back may throw: what has been reported is not reported again)
        const Item &it = c.items[c.reported];
        ugsm_completion d{};
        d.tag = it.tag;
        d.status = c.status;
        d.slot = c.slot;
        d.call_pairs = (int)c.items.size();
        d.call_index = c.index;
        d.done_ns = t;
        if (it.mem == MEM_MANAGED)
            for (int k = 0; k < 5; k++) d.result[k] = it.out[k];
        q->done.push_back(Done{d, it.managed});
    }
    q->slot_busy[(size_t)c.slot] = 0;
    q->flight.pop_front();
}

// Retires every call at the front of the flight list that has finished (never blocks).  A call whose submit FAILED is drained like any
// other before its pairs are reported: a submit can fail after work has gone onto the slot's stream (out of memory for the staging of a
// later phase, a launch error mid-call), and a completion tells the host that its input and result buffers -- or the managed staging
// buffer the library lends next -- are free.  The submit's status is the one reported.
int reap(ugsm_ctx *ctx, Queue *q)
{
    while (!q->flight.empty()) {
        Call &c = q->flight.front();
        const int st = ugsm_poll(ctx, c.slot);
        if (st == UGSM_PENDING) break;
        if (st != UGSM_OK && c.status == UGSM_OK) c.status = st;
        retire_front(ctx, q);
    }
    return UGSM_OK;
}

// blocks until the oldest call in flight has finished (or, its submit having failed, until what it did enqueue has drained), and retires it
void wait_front(ugsm_ctx *ctx, Queue *q)
{
    Call &c = q->flight.front();
    const int st = ugsm_wait(ctx, c.slot);
    if (st != UGSM_OK && c.status == UGSM_OK) c.status = st;
    retire_front(ctx, q);
}

// A slot without a call in flight, in rotation order from next_slot; -1 if every slot is busy.  A slot whose call has finished but is
// not at the front of the flight list (calls may finish out of order) stays busy until the calls before it have been retired: results
// are reported in enqueue order, and the slot's buffers belong to its call until then.
int free_slot(const Queue *q)
{
    const int n = (int)q->slot_busy.size();
    for (int k = 0; k < n; k++) {
        const int s = (q->next_slot + k) % n;
        if (!q->slot_busy[(size_t)s]) return s;
    }
    return -1;
}

// One library call for the first n waiting pairs on `slot`.  more: other calls follow this one closely -- pairs wait behind it, or it filled
// up without a flush (a host that submits faster than the chip matches): the runtime then takes the call to share the chip whatever the
// other slots are doing at this instant (call_alone, ugsm_runtime.cpp).
void dispatch(ugsm_ctx *ctx, Queue *q, int n, int slot, bool more)
{
    // the call's place in the flight list first: everything that can run out of host memory happens before anything reaches the GPU,
    // and a call that has reached the GPU is always on the list
    q->flight.emplace_back();
    Call &c = q->flight.back();
    try {
        c.items.assign(q->waiting.begin(), q->waiting.begin() + n);
    } catch (...) {
        q->flight.pop_back();
        throw;  // (to the entry point's no_throw; the pairs still wait)
    }
    c.slot = slot;
    c.index = q->calls++;
    q->waiting.erase(q->waiting.begin(), q->waiting.begin() + n);
    const Item &f = c.items[0];
    const uint8_t *L[UGSM_MAX_BATCH], *R[UGSM_MAX_BATCH];
    float *o0[UGSM_MAX_BATCH], *o1[UGSM_MAX_BATCH], *o2[UGSM_MAX_BATCH], *pl[UGSM_MAX_BATCH], *pr[UGSM_MAX_BATCH];
    int ox[UGSM_MAX_BATCH], oy[UGSM_MAX_BATCH];
    bool any_pyr = false;
    for (int b = 0; b < n; b++) {
        const Item &it = c.items[(size_t)b];
        L[b] = it.L;
        R[b] = it.R;
        o0[b] = it.out[0];
        o1[b] = it.out[1];
        o2[b] = it.out[2];
        pl[b] = it.out[3];
        pr[b] = it.out[4];
        ox[b] = it.off_x;
        oy[b] = it.off_y;
        any_pyr = any_pyr || it.pyr();
    }
    CtxHooks &h = ctx_hooks(ctx);
    h.queue_calling = true;
    h.queue_more = more;
    int st;
    if (f.mem == MEM_DEVICE) {
        if (f.mode == M_FULL)
            st = n == 1 ? ugsm_submit_full(ctx, slot, L[0], R[0], f.W, f.H, f.stride, o0[0]) : ugsm_submit_full_batch(ctx, slot, n, L, R, f.W, f.H, f.stride, o0);
        else
            st = n == 1 ? ugsm_submit_foveated(ctx, slot, L[0], R[0], f.W, f.H, f.stride, ox[0], oy[0], o0[0], pl[0], pr[0])
                        : ugsm_submit_foveated_batch(ctx, slot, n, L, R, f.W, f.H, f.stride, ox, oy, o0, any_pyr ? pl : nullptr, any_pyr ? pr : nullptr);
    } else {
        if (f.mode == M_FULL)
            st = n == 1 ? ugsm_submit_full_host(ctx, slot, L[0], R[0], f.W, f.H, f.stride, o0[0], o1[0], o2[0])
                        : ugsm_submit_full_batch_host(ctx, slot, n, L, R, f.W, f.H, f.stride, o0, o1, o2);
        else
            st = n == 1 ? ugsm_submit_foveated_host(ctx, slot, L[0], R[0], f.W, f.H, f.stride, ox[0], oy[0], o0[0], o1[0], o2[0], pl[0], pr[0])
                        : ugsm_submit_foveated_batch_host(ctx, slot, n, L, R, f.W, f.H, f.stride, ox, oy, o0, o1, o2);
    }
    h.queue_calling = false;
    h.queue_more = false;
    c.status = st;  // (a call that failed to enqueue: its pairs are reported with this status once the slot has drained, reap / wait_front)
    q->slot_busy[(size_t)slot] = 1;
    q->next_slot = (slot + 1) % (int)q->slot_busy.size();
    q->calls_since_idle++;
}

// Forms and sends calls from the backlog.  may_block: a call that is full may wait for the slot of the oldest call in flight
// (ugsm_enqueue_*: back-pressure); otherwise only free slots are used.  How a call went is reported with its pairs (ugsm_completion.status),
// never through the entry point that happened to send it: that call may hold other pairs than the one just enqueued.
void pump(ugsm_ctx *ctx, Queue *q, bool may_block)
{
    const ugsm_config &cfg = ctx_config(ctx);
    while (!q->waiting.empty()) {
        reap(ctx, q);
        if (q->round_restarts && q->waiting.front().seq > q->flush_upto) {
            // the pairs before the last flush have all gone out: what arrived after it is a new burst, staggered again (a host's flush is
            // the one event that says "the pipe is about to drain" without depending on how fast the GPU happens to be)
            q->round_restarts = false;
            q->calls_since_idle = 0;
        }
        const int target = queue_target(batch_of(cfg), cfg.slots, q->calls_since_idle);
        const Item &f = q->waiting.front();
        const int cap = (f.mem != MEM_DEVICE && f.pyr()) ? 1 : target;
        int n = 0;
        bool kind_ends = false;  // a pair of another kind waits behind the group: the group goes out as it is
        for (const Item &it : q->waiting) {
            if (n >= cap) break;
            if (!it.same_kind(f)) {
                kind_ends = true;
                break;
            }
            n++;
        }
        if (n < 1) n = 1;  // (a host pair with pyramid stacks is not the same kind as itself: alone)
        const bool full = n >= cap || kind_ends;
        if (!full) {
            // only the pairs a flush has covered go out in a call that is not full
            int m = 0;
            while (m < n && q->waiting[(size_t)m].seq <= q->flush_upto) m++;
            if (m == 0) break;
            n = m;
        }
        int slot = free_slot(q);
        if (slot < 0) {
            if (!(full && may_block) || q->flight.empty()) break;
            wait_front(ctx, q);
            reap(ctx, q);
            slot = free_slot(q);
            if (slot < 0) break;
        }
        const bool by_itself = full && !kind_ends && cap > 1 && q->waiting[(size_t)n - 1].seq > q->flush_upto;
        dispatch(ctx, q, n, slot, (int)q->waiting.size() > n || by_itself);
    }
    update_busy(ctx, q);
}

int check_geometry(ugsm_ctx *ctx, int W, int H, int stride, bool fovea)
{
    const ugsm_config &cfg = ctx_config(ctx);
    int w[UGSM_MAX_LEVELS], h[UGSM_MAX_LEVELS];
    const int st = ugsm_level_dims(W, H, cfg.levels, w, h);
    if (st != UGSM_OK) return ctx_fail(ctx, st, "ugsm_enqueue_*: bad image size for the context's pyramid");
    if (stride < 3 * W) return ctx_fail(ctx, UGSM_ERR_SIZE_MISMATCH, "ugsm_enqueue_*: stride < 3 * W");
    if (fovea && cfg.fovea_levels < 2) return ctx_fail(ctx, UGSM_ERR_BAD_ARG, "ugsm_enqueue_foveated*: the context has no fovea levels");
    return UGSM_OK;
}

// (slots + 1) x batch pairs may be outstanding -- waiting, in flight or finished and not yet fetched.  In-flight and waiting pairs alone
// never reach that number (a full call goes out, or waits for its slot, inside the enqueue that completes it), so a full queue means
// completions the host has not fetched: refuse instead of letting it recycle a result buffer it has not read.
int room(ugsm_ctx *ctx, const Queue *q)
{
    const ugsm_config &cfg = ctx_config(ctx);
    if (q->outstanding() < (cfg.slots + 1) * batch_of(cfg)) return UGSM_OK;
    return ctx_fail(ctx, UGSM_ERR_STATE, "ugsm_enqueue_*: (slots + 1) x batch pairs are outstanding and completions wait to be fetched: call ugsm_next_done first");
}

int enqueue(ugsm_ctx *ctx, Item it)
{
    return no_throw(ctx, "ugsm_enqueue_*: out of host memory", [&]() -> int {
        Queue *q = queue_of(ctx);
        if (!q) return ctx_fail(ctx, UGSM_ERR_NOMEM, "ugsm_enqueue_*: out of host memory");
        const int r = room(ctx, q);
        if (r != UGSM_OK) return r;
        it.seq = q->seq + 1;
        q->waiting.push_back(it);
        q->seq = it.seq;
        ctx_hooks(ctx).queue_busy = true;
        // the pair is accepted; whatever happens to its call comes out of ugsm_next_done -- and a pump that ran out of host memory leaves
        // the pairs waiting for the next entry point's pump: that is not a rejection either
        (void)no_throw(ctx, "ugsm_enqueue_*: out of host memory", [&]() -> int {
            pump(ctx, q, true);
            return UGSM_OK;
        });
        update_busy(ctx, q);
        return UGSM_OK;
    });
}

// a managed buffer with room for `in_bytes` of images and `out_floats` of results
int managed_get(ugsm_ctx *ctx, Queue *q, size_t in_bytes, size_t out_floats)
{
    int idx = -1;
    for (size_t k = 0; k < q->pool.size(); k++)
        if (!q->pool[k].busy && (idx < 0 || (q->pool[k].in_cap >= in_bytes && q->pool[k].out_cap >= out_floats))) idx = (int)k;
    if (idx < 0) {
        q->pool.emplace_back();
        idx = (int)q->pool.size() - 1;
    }
    Managed &m = q->pool[(size_t)idx];
    if (m.in_cap < in_bytes) {
        if (m.in) (void)ugsm_host_free(ctx, m.in);
        m.in = nullptr;
        m.in_cap = 0;
        void *p = nullptr;
        const int st = ugsm_host_alloc(ctx, &p, (long long)in_bytes);
        if (st != UGSM_OK) return -1;
        m.in = static_cast<uint8_t *>(p);
        m.in_cap = in_bytes;
    }
    if (m.out_cap < out_floats) {
        if (m.out) (void)ugsm_host_free(ctx, m.out);
        m.out = nullptr;
        m.out_cap = 0;
        void *p = nullptr;
        const int st = ugsm_host_alloc(ctx, &p, (long long)(out_floats * sizeof(float)));
        if (st != UGSM_OK) return -1;
        m.out = static_cast<float *>(p);
        m.out_cap = out_floats;
    }
    m.busy = true;
    return idx;
}

int enqueue_managed(ugsm_ctx *ctx, int mode, const uint8_t *rgbL, const uint8_t *rgbR, int W, int H, int stride, int off_x, int off_y, int want_pyr,
                    uint64_t tag)
{
    if (!ctx) return UGSM_ERR_BAD_ARG;
    if (!rgbL || !rgbR) return ctx_fail(ctx, UGSM_ERR_BAD_ARG, "ugsm_enqueue_*_managed: null image");
    const int g = check_geometry(ctx, W, H, stride, mode == M_FOVEA);
    if (g != UGSM_OK) return g;
    Queue *q = queue_of(ctx);
    if (!q) return ctx_fail(ctx, UGSM_ERR_NOMEM, "ugsm_enqueue_*: out of host memory");
    const ugsm_config &cfg = ctx_config(ctx);
    const int r = room(ctx, q);
    if (r != UGSM_OK) return r;
    const size_t row = 3 * (size_t)W, img = (row * (size_t)H + 255) & ~(size_t)255;
    size_t plane, out_floats;
    if (mode == M_FULL) {
        plane = (size_t)W * H;
        out_floats = 3 * plane;
    } else {
        int fw = 0, fh = 0;
        (void)ugsm_fovea_dims(W, H, cfg.levels, cfg.fovea_levels, &fw, &fh);
        plane = (size_t)cfg.fovea_levels * fw * fh;
        out_floats = 3 * plane + (want_pyr ? 6 * plane : 0);
    }
    int idx = -1;
    const int grown = no_throw(ctx, "ugsm_enqueue_*_managed: out of host memory", [&]() -> int {
        idx = managed_get(ctx, q, 2 * img, out_floats);
        return UGSM_OK;
    });
    if (grown != UGSM_OK || idx < 0) return UGSM_ERR_NOMEM;  // (ugsm_host_alloc has set the message)
    Managed &m = q->pool[(size_t)idx];
    // the images, compacted to rows of 3 W bytes, into the staging buffer: after this the caller's memory is not touched again
    for (int side = 0; side < 2; side++) {
        const uint8_t *src = side == 0 ? rgbL : rgbR;
        uint8_t *dst = m.in + side * img;
        if ((size_t)stride == row) {
            ctx_host_copy(ctx, dst, src, row * (size_t)H);
        } else {
            for (int y = 0; y < H; y++) memcpy(dst + (size_t)y * row, src + (size_t)y * stride, row);
        }
    }
    Item it{};
    it.mode = mode;
    it.mem = MEM_MANAGED;
    it.L = m.in;
    it.R = m.in + img;
    it.W = W;
    it.H = H;
    it.stride = (int)row;
    it.off_x = off_x;
    it.off_y = off_y;
    it.out[0] = m.out;
    it.out[1] = m.out + plane;
    it.out[2] = m.out + 2 * plane;
    it.out[3] = (mode == M_FOVEA && want_pyr) ? m.out + 3 * plane : nullptr;
    it.out[4] = (mode == M_FOVEA && want_pyr) ? m.out + 6 * plane : nullptr;
    it.tag = tag;
    it.managed = idx;
    const int st = enqueue(ctx, it);
    if (st != UGSM_OK) q->pool[(size_t)idx].busy = false;  // (rejected: the staging buffer is free again)
    return st;
}

int next_done(ugsm_ctx *ctx, ugsm_completion *out, int block)
{
    Queue *q = queue_of(ctx);
    if (!q) return ctx_fail(ctx, UGSM_ERR_NOMEM, "ugsm_next_done: out of host memory");
    for (int idx : q->lent) q->pool[(size_t)idx].busy = false;  // what the previous call lent comes back
    q->lent.clear();
    if (block && q->flush_upto != q->seq) {
        q->flush_upto = q->seq;
        q->round_restarts = true;
    }
    for (;;) {
        reap(ctx, q);
        pump(ctx, q, false);  // (a slot may just have come free for pairs that wait)
        if (!q->done.empty()) break;
        if (q->flight.empty() && q->waiting.empty()) {
            update_busy(ctx, q);
            return UGSM_EMPTY;
        }
        if (!block) {
            update_busy(ctx, q);
            return UGSM_PENDING;
        }
        if (q->flight.empty()) {
            // pairs wait and no call is in flight, yet pump sent nothing: cannot happen (every slot is free); do not spin
            update_busy(ctx, q);
            return ctx_fail(ctx, UGSM_ERR_STATE, "ugsm_next_done: the queue cannot make progress");
        }
        wait_front(ctx, q);
    }
    const int m = q->done.front().managed;
    if (m >= 0) q->lent.push_back(m);  // (before the completion leaves the list: a push_back that throws loses nothing)
    *out = q->done.front().c;
    q->done.pop_front();
    // (pairs that waited for room may go out now; the slot-level entry points open up again once nothing is outstanding)
    pump(ctx, q, false);
    return UGSM_OK;
}

}  // namespace

namespace ugsm {
// call c (0-based) since the queue was idle: the first `slots` calls are staggered in size, ceil(batch (c + 2) / (slots + 1)) -- 4, 5, 7,
// 8 for batch 8 on four slots -- so that the slots do not march through the levels in phase from a drained pipe (same box, 20 pairs of
// 16 MP, batch 4: 172 against 169 pairs/s for equal calls, 162 for a descending start; profiles/r04_ab_plan.txt); every later call: batch
int queue_target(int batch, int slots, long long c)
{
    if (c >= slots) return batch;
    const long long t = ((long long)batch * (c + 2) + slots) / (slots + 1);
    return (int)std::max(1LL, std::min((long long)batch, t));
}
}  // namespace ugsm

extern "C" {

int ugsm_queue_plan(const ugsm_config *cfg_in, int n_pairs, int *sizes, int cap)
{
    ugsm_config cfg;
    if (cfg_in) cfg = *cfg_in;
    else ugsm_default_config(&cfg);
    if (n_pairs < 0 || cap < 0 || (cap > 0 && !sizes) || cfg.slots < 1) return -1;
    const int B = batch_of(cfg);
    int calls = 0, left = n_pairs;
    while (left > 0) {
        const int nb = std::min(left, queue_target(B, cfg.slots, calls));
        if (calls < cap) sizes[calls] = nb;
        calls++;
        left -= nb;
    }
    return calls;
}

int ugsm_enqueue_full(ugsm_ctx *ctx, const uint8_t *d_rgbL, const uint8_t *d_rgbR, int W, int H, int stride, float *d_out, uint64_t tag)
{
    if (!ctx) return UGSM_ERR_BAD_ARG;
    if (!d_rgbL || !d_rgbR || !d_out) return ctx_fail(ctx, UGSM_ERR_BAD_ARG, "ugsm_enqueue_full: null buffer");
    const int g = check_geometry(ctx, W, H, stride, false);
    if (g != UGSM_OK) return g;
    Item it{};
    it.mode = M_FULL;
    it.mem = MEM_DEVICE;
    it.L = d_rgbL;
    it.R = d_rgbR;
    it.W = W;
    it.H = H;
    it.stride = stride;
    it.out[0] = d_out;
    it.tag = tag;
    it.managed = -1;
    return enqueue(ctx, it);
}

int ugsm_enqueue_foveated(ugsm_ctx *ctx, const uint8_t *d_rgbL, const uint8_t *d_rgbR, int W, int H, int stride, int off_x, int off_y, float *d_stack,
                          float *d_pyrL, float *d_pyrR, uint64_t tag)
{
    if (!ctx) return UGSM_ERR_BAD_ARG;
    if (!d_rgbL || !d_rgbR || !d_stack) return ctx_fail(ctx, UGSM_ERR_BAD_ARG, "ugsm_enqueue_foveated: null buffer");
    const int g = check_geometry(ctx, W, H, stride, true);
    if (g != UGSM_OK) return g;
    Item it{};
    it.mode = M_FOVEA;
    it.mem = MEM_DEVICE;
    it.L = d_rgbL;
    it.R = d_rgbR;
    it.W = W;
    it.H = H;
    it.stride = stride;
    it.off_x = off_x;
    it.off_y = off_y;
    it.out[0] = d_stack;
    it.out[3] = d_pyrL;
    it.out[4] = d_pyrR;
    it.tag = tag;
    it.managed = -1;
    return enqueue(ctx, it);
}

int ugsm_enqueue_full_host(ugsm_ctx *ctx, const uint8_t *rgbL, const uint8_t *rgbR, int W, int H, int stride, float *dispH, float *dispV, float *dispC,
                           uint64_t tag)
{
    if (!ctx) return UGSM_ERR_BAD_ARG;
    if (!rgbL || !rgbR || !dispH || !dispV || !dispC) return ctx_fail(ctx, UGSM_ERR_BAD_ARG, "ugsm_enqueue_full_host: null buffer");
    const int g = check_geometry(ctx, W, H, stride, false);
    if (g != UGSM_OK) return g;
    if (!(host_pinned(rgbL) && host_pinned(rgbR) && host_pinned(dispH) && host_pinned(dispV) && host_pinned(dispC)))
        return ctx_fail(ctx, UGSM_ERR_BAD_ARG, "ugsm_enqueue_full_host: every host buffer must be page-locked (ugsm_host_alloc, hipHostMalloc or hipHostRegister); "
                                               "ugsm_enqueue_full_managed takes any memory");
    Item it{};
    it.mode = M_FULL;
    it.mem = MEM_PINNED;
    it.L = rgbL;
    it.R = rgbR;
    it.W = W;
    it.H = H;
    it.stride = stride;
    it.out[0] = dispH;
    it.out[1] = dispV;
    it.out[2] = dispC;
    it.tag = tag;
    it.managed = -1;
    return enqueue(ctx, it);
}

int ugsm_enqueue_foveated_host(ugsm_ctx *ctx, const uint8_t *rgbL, const uint8_t *rgbR, int W, int H, int stride, int off_x, int off_y, float *stackH,
                               float *stackV, float *stackC, float *pyrL, float *pyrR, uint64_t tag)
{
    if (!ctx) return UGSM_ERR_BAD_ARG;
    if (!rgbL || !rgbR || !stackH || !stackV || !stackC) return ctx_fail(ctx, UGSM_ERR_BAD_ARG, "ugsm_enqueue_foveated_host: null buffer");
    const int g = check_geometry(ctx, W, H, stride, true);
    if (g != UGSM_OK) return g;
    bool pinned = host_pinned(rgbL) && host_pinned(rgbR) && host_pinned(stackH) && host_pinned(stackV) && host_pinned(stackC);
    if (pyrL) pinned = pinned && host_pinned(pyrL);
    if (pyrR) pinned = pinned && host_pinned(pyrR);
    if (!pinned)
        return ctx_fail(ctx, UGSM_ERR_BAD_ARG, "ugsm_enqueue_foveated_host: every host buffer must be page-locked (ugsm_host_alloc, hipHostMalloc or hipHostRegister); "
                                               "ugsm_enqueue_foveated_managed takes any memory");
    Item it{};
    it.mode = M_FOVEA;
    it.mem = MEM_PINNED;
    it.L = rgbL;
    it.R = rgbR;
    it.W = W;
    it.H = H;
    it.stride = stride;
    it.off_x = off_x;
    it.off_y = off_y;
    it.out[0] = stackH;
    it.out[1] = stackV;
    it.out[2] = stackC;
    it.out[3] = pyrL;
    it.out[4] = pyrR;
    it.tag = tag;
    it.managed = -1;
    return enqueue(ctx, it);
}

int ugsm_enqueue_full_managed(ugsm_ctx *ctx, const uint8_t *rgbL, const uint8_t *rgbR, int W, int H, int stride, uint64_t tag)
{
    return enqueue_managed(ctx, M_FULL, rgbL, rgbR, W, H, stride, 0, 0, 0, tag);
}

int ugsm_enqueue_foveated_managed(ugsm_ctx *ctx, const uint8_t *rgbL, const uint8_t *rgbR, int W, int H, int stride, int off_x, int off_y, int want_pyramids,
                                  uint64_t tag)
{
    return enqueue_managed(ctx, M_FOVEA, rgbL, rgbR, W, H, stride, off_x, off_y, want_pyramids, tag);
}

int ugsm_flush(ugsm_ctx *ctx)
{
    if (!ctx) return UGSM_ERR_BAD_ARG;
    return no_throw(ctx, "ugsm_flush: out of host memory", [&]() -> int {
        Queue *q = queue_of(ctx);
        if (!q) return ctx_fail(ctx, UGSM_ERR_NOMEM, "ugsm_flush: out of host memory");
        q->flush_upto = q->seq;
        q->round_restarts = true;
        pump(ctx, q, false);
        return UGSM_OK;
    });
}

int ugsm_next_done(ugsm_ctx *ctx, ugsm_completion *out, int block)
{
    if (!ctx || !out) return UGSM_ERR_BAD_ARG;
    return no_throw(ctx, "ugsm_next_done: out of host memory", [&]() -> int { return next_done(ctx, out, block); });
}

int ugsm_queue_depth(ugsm_ctx *ctx, int *waiting, int *in_flight, int *unreported)
{
    if (!ctx) return UGSM_ERR_BAD_ARG;
    Queue *q = queue_of(ctx);
    if (!q) return ctx_fail(ctx, UGSM_ERR_NOMEM, "ugsm_queue_depth: out of host memory");
    int fl = 0;
    for (const Call &c : q->flight) fl += (int)c.items.size();
    if (waiting) *waiting = (int)q->waiting.size();
    if (in_flight) *in_flight = fl;
    if (unreported) *unreported = (int)q->done.size();
    return UGSM_OK;
}

}  // extern "C"
