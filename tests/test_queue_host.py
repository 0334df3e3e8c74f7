"""The queue of include/ugsm.h (csrc/ugsm_queue.cpp) on a machine without a GPU: the real queue source compiled against a recording
stand-in for the runtime (tests/fake_runtime.cpp -- test infrastructure) and driven through the C-ABI with ctypes.

The queue is host logic written against the public slot-level entry points; what it promises is stated in the header:
  * UGSM_OK from ugsm_enqueue_* = the pair is accepted and is reported by ugsm_next_done EXACTLY ONCE, in enqueue order, with the status
    of the library call it went out in; anything else = rejected, never reported;
  * a failed call's pairs are reported only after the slot has drained;
  * calls hold pairs of one kind, at most `batch` of them, the first `slots` calls of a burst staggered (ugsm_queue_plan);
  * no slot is given a second call before its first has been seen finished;
  * the `more` hint (CtxHooks::queue_more) that makes the runtime take a call to share the chip: set when pairs wait behind the call or
    the call filled up by itself, clear for a flushed call with nothing behind it;
  * nothing is thrown across the C-ABI: with the Nth host allocation failing, entry points answer UGSM_ERR_NOMEM and the first rule
    still holds.
The reference has no counterpart (one blocking match() per callback, UG_GPU_matcher.cpp:126-185, 749-752)."""
import ctypes as C
import os
import random
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OK, BAD_ARG, NOMEM, STATE, PENDING, EMPTY = 0, 1, 6, 7, 8, 9


class Completion(C.Structure):
    _fields_ = [("tag", C.c_uint64), ("status", C.c_int), ("slot", C.c_int), ("call_pairs", C.c_int), ("reserved", C.c_int),
                ("call_index", C.c_longlong), ("done_ns", C.c_longlong), ("result", C.POINTER(C.c_float) * 5)]


@pytest.fixture(scope="module")
def fq(tmp_path_factory):
    out = str(tmp_path_factory.mktemp("fakeq") / "libugsm_queue_fake.so")
    cmd = ["g++", "-std=c++17", "-O1", "-g", "-fPIC", "-shared", "-fvisibility=hidden",
           os.path.join(ROOT, "tests", "fake_runtime.cpp"), os.path.join(ROOT, "ug_stereomatcher_amd", "csrc", "ugsm_queue.cpp"),
           "-Wl,--version-script=" + os.path.join(ROOT, "ug_stereomatcher_amd", "csrc", "ugsm.map"), "-o", out]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]
    lib = C.CDLL(out)
    vp, i, u64, ll = C.c_void_p, C.c_int, C.c_uint64, C.c_longlong
    lib.ugsm_fake_create.restype = vp
    lib.ugsm_fake_create.argtypes = [i, i, i, i]
    lib.ugsm_fake_destroy.argtypes = [vp]
    lib.ugsm_fake_destroy.restype = None
    lib.ugsm_fake_poll_delay.argtypes = [vp, i]
    lib.ugsm_fake_poll_delay.restype = None
    lib.ugsm_fake_fail_call.argtypes = [vp, ll, i]
    lib.ugsm_fake_calls.argtypes = [vp]
    lib.ugsm_fake_calls.restype = ll
    lib.ugsm_fake_call.argtypes = [vp, ll, C.POINTER(i * 7), C.POINTER(vp * 16)]
    lib.ugsm_fake_violations.argtypes = [vp]
    lib.ugsm_fake_fail_alloc_after.argtypes = [ll]
    lib.ugsm_fake_fail_alloc_after.restype = None
    lib.ugsm_fake_allocs.restype = ll
    lib.ugsm_enqueue_full.argtypes = [vp, vp, vp, i, i, i, vp, u64]
    lib.ugsm_enqueue_foveated.argtypes = [vp, vp, vp, i, i, i, i, i, vp, vp, vp, u64]
    lib.ugsm_enqueue_full_host.argtypes = [vp, vp, vp, i, i, i, vp, vp, vp, u64]
    lib.ugsm_enqueue_full_managed.argtypes = [vp, vp, vp, i, i, i, u64]
    lib.ugsm_flush.argtypes = [vp]
    lib.ugsm_next_done.argtypes = [vp, C.POINTER(Completion), i]
    lib.ugsm_queue_depth.argtypes = [vp, C.POINTER(i), C.POINTER(i), C.POINTER(i)]
    lib.ugsm_queue_plan.argtypes = [vp, i, C.POINTER(i), i]
    lib.ugsm_last_error.argtypes = [vp]
    lib.ugsm_last_error.restype = C.c_char_p
    return lib


BASE = 0x7F0000000000   # made-up "device" addresses: the queue hands them on and never looks behind them


def ptr_of(tag):
    return BASE + 4096 * tag


class Host:
    """A host program on the queue; keeps the books the assertions need."""

    def __init__(self, lib, slots, batch, poll_delay=0):
        self.lib, self.slots, self.batch = lib, slots, max(1, batch)
        self.ctx = lib.ugsm_fake_create(slots, batch, 8, 4)
        assert self.ctx
        lib.ugsm_fake_poll_delay(self.ctx, poll_delay)
        self.accepted, self.kind_of, self.reported, self.rejected = [], {}, [], []
        self.next_tag = 0
        self.img = np.zeros((2, 16, 3 * 32), np.uint8)   # a 32 x 16 rgb8 pair for the managed entry point (it copies the images)

    def close(self):
        self.lib.ugsm_fake_destroy(self.ctx)
        self.ctx = None

    def outstanding(self):
        return len(self.accepted) - len(self.reported)

    def enqueue(self, kind, size=(64, 32)):
        tag = self.next_tag
        self.next_tag += 1
        W, H = size
        p = ptr_of(tag)
        if kind == "full":
            st = self.lib.ugsm_enqueue_full(self.ctx, p, p + 1, W, H, 3 * W, p + 2, tag)
        elif kind == "fovea":
            st = self.lib.ugsm_enqueue_foveated(self.ctx, p, p + 1, W, H, 3 * W, 0, 0, p + 2, None, None, tag)
        elif kind == "host":
            st = self.lib.ugsm_enqueue_full_host(self.ctx, p, p + 1, W, H, 3 * W, p + 2, p + 3, p + 4, tag)
        else:
            st = self.lib.ugsm_enqueue_full_managed(self.ctx, self.img[0].ctypes.data, self.img[1].ctypes.data, 32, 16, 96, tag)
        if st == OK:
            self.accepted.append(tag)
            self.kind_of[tag] = (kind, size if kind != "managed" else (32, 16))
        else:
            self.rejected.append((tag, st))
        return st

    def call(self, k):
        v, L = (C.c_int * 7)(), (C.c_void_p * 16)()
        assert self.lib.ugsm_fake_call(self.ctx, k, C.byref(v), C.byref(L)) == 0, k
        return dict(slot=v[0], n=v[1], mode=v[2], mem=v[3], status=v[4], more=v[5], drained=v[6], L=[L[b] for b in range(v[1])])

    def calls(self):
        return [self.call(k) for k in range(self.lib.ugsm_fake_calls(self.ctx))]

    def next_done(self, block):
        c = Completion()
        st = self.lib.ugsm_next_done(self.ctx, C.byref(c), block)
        if st == OK:
            rec = self.call(c.call_index)
            # reported only once its slot has been seen finished -- a failed call too ("a completion always means the buffers are free")
            assert rec["drained"] == 1, (c.tag, rec)
            assert c.status == rec["status"] and c.call_pairs == rec["n"] and c.slot == rec["slot"], (c.tag, c.status, rec)
            self.reported.append(c.tag)
            if self.kind_of[c.tag][0] == "managed":
                assert c.result[0] and c.result[1] and c.result[2]          # the planes the library lends
            else:
                assert not c.result[0]
        return st

    def drain(self, tolerate=False):
        spins = 0
        while True:
            st = self.next_done(1)
            if st == EMPTY:
                return
            if st != OK:
                assert tolerate and st == NOMEM, (st, self.lib.ugsm_last_error(self.ctx))
                spins += 1
                assert spins < 50, "ugsm_next_done keeps failing"

    def check(self, faults=False):
        """The invariants, after a drain."""
        assert self.reported == self.accepted, (self.reported, self.accepted)            # exactly once, in enqueue order
        assert self.lib.ugsm_fake_violations(self.ctx) == 0                               # no slot reused before it was seen finished
        sent = []
        for rec in self.calls():
            assert 1 <= rec["n"] <= self.batch, rec
            tags = [(p - BASE) // 4096 if p and p >= BASE else None for p in rec["L"]]
            kinds = set()
            for t, p in zip(tags, rec["L"]):
                if t is None or t not in self.kind_of or self.kind_of[t][0] == "managed":
                    kinds.add("managed")            # (staged by the library: not the host's pointer)
                else:
                    kinds.add(self.kind_of[t])
                    sent.append(t)
            assert len(kinds) == 1, (rec, kinds)                                          # one kind per call
            assert rec["mode"] == (1 if next(iter(kinds))[0] == "fovea" else 0) if "managed" not in kinds else True
        plain = [t for t in self.accepted if self.kind_of[t][0] != "managed"]
        if not faults:
            assert sent == plain, (sent, plain)                                           # every pair went out once, in order
        else:
            assert sorted(set(sent)) == sorted(sent) and set(plain) <= set(sent)
        w, f, u = C.c_int(), C.c_int(), C.c_int()
        assert self.lib.ugsm_queue_depth(self.ctx, C.byref(w), C.byref(f), C.byref(u)) == OK and (w.value, f.value, u.value) == (0, 0, 0)


def plan(lib, slots, batch, n):
    """ugsm_queue_plan: what the header says a burst of n pairs becomes."""
    class Config(C.Structure):   # ugsm_config (include/ugsm.h; the layout is pinned against the header by tests/test_abi_host.py through _lib.Config)
        _fields_ = [(n, C.c_float if n in ("early_exit_threshold", "lr_check_threshold") else C.c_int) for n in (
            "device", "levels", "fovea_levels", "slots", "kernel_path", "profile_events", "march_min_pixels", "march_np", "march_rows", "march_smooth",
            "early_exit_threshold", "small_max_pixels", "lr_check_threshold", "streams", "batch", "stream_priority")]
    cfg = Config()
    cfg.slots, cfg.batch, cfg.levels, cfg.fovea_levels = slots, batch, 8, 4
    sizes = (C.c_int * 64)()
    k = lib.ugsm_queue_plan(C.byref(cfg), n, sizes, 64)
    return [sizes[i] for i in range(k)]


def test_a_burst_is_staggered_and_hinted(fq):
    """Four slots, calls of up to eight, 32 pairs enqueued back to back from idle while nothing finishes: calls of 4, 5, 7, 8, then the
    enqueue that fills the fifth call waits for the oldest call's slot (back-pressure) and sends 8.  Every call of the burst carries the
    `more` hint (it filled up by itself: a host that submits faster than the chip matches)."""
    h = Host(fq, 4, 8, poll_delay=10 ** 6)
    try:
        for _ in range(32):
            assert h.enqueue("full") == OK
        calls = h.calls()
        assert [c["n"] for c in calls] == [4, 5, 7, 8, 8] == plan(fq, 4, 8, 32), calls
        assert [c["slot"] for c in calls] == [0, 1, 2, 3, 0]
        assert all(c["more"] == 1 for c in calls)
        assert h.next_done(0) == OK and h.reported == [0]          # (the back-pressure wait retired call 0: its pairs are ready)
        assert fq.ugsm_flush(h.ctx) == OK
        h.drain()
        h.check()
    finally:
        h.close()


def test_a_flushed_call_with_nothing_behind_it_is_taken_to_be_alone(fq):
    """The hint the kernel policy reads (call_alone, csrc/ugsm_runtime.cpp): clear for one pair enqueued and flushed (the node's topic path when
    frames arrive slower than they are matched), for a partial call sent by a blocking ugsm_next_done, and for the LAST call of a burst;
    set for a call that leaves pairs waiting (here: a pair of another size behind it)."""
    h = Host(fq, 4, 8)
    try:
        assert h.enqueue("full") == OK and fq.ugsm_flush(h.ctx) == OK
        assert [(c["n"], c["more"]) for c in h.calls()] == [(1, 0)]
        h.drain()
        assert h.enqueue("full") == OK and h.enqueue("full") == OK
        assert fq.ugsm_fake_calls(h.ctx) == 1                      # two pairs of a call of four (the stagger restarts after a flush): they wait
        assert h.next_done(1) == OK                                 # ... until a blocking fetch sends them
        assert [(c["n"], c["more"]) for c in h.calls()][1:] == [(2, 0)]
        h.drain()
        assert h.enqueue("full") == OK and h.enqueue("full") == OK and h.enqueue("full", (128, 64)) == OK
        got = [(c["n"], c["more"]) for c in h.calls()][2:]
        assert got == [(2, 1)], got                                 # the pair of another size closes the group: it goes out, hinted
        assert fq.ugsm_flush(h.ctx) == OK
        assert [(c["n"], c["more"]) for c in h.calls()][3:] == [(1, 0)]
        h.drain()
        h.check()
    finally:
        h.close()


def test_the_queue_refuses_instead_of_recycling_unfetched_results(fq):
    """(slots + 1) x batch pairs outstanding with completions unfetched: UGSM_ERR_STATE, nothing enqueued, and room again after a fetch."""
    h = Host(fq, 2, 2)
    try:
        sts = [h.enqueue("full") for _ in range(8)]
        assert sts[:6] == [OK] * 6 and set(sts[6:]) == {STATE}, sts
        assert h.next_done(1) == OK
        assert h.enqueue("full") == OK
        h.drain()
        h.check()
        assert [t for t, _ in h.rejected] == [6, 7]
    finally:
        h.close()


@pytest.mark.parametrize("seed", range(24))
def test_every_accepted_pair_is_reported_exactly_once_in_order(fq, seed):
    """Random host programs (four kinds of pairs, two sizes, flushes, blocking and non-blocking fetches) on random contexts, with slots
    that take a random number of queries to finish and a random tenth of the library calls failing: the books balance."""
    rnd = random.Random(1000 + seed)
    slots, batch = rnd.choice([1, 2, 4]), rnd.choice([1, 2, 3, 8])
    h = Host(fq, slots, batch, poll_delay=rnd.choice([0, 1, 3, 7]))
    try:
        for k in range(400):
            if rnd.random() < 0.1:
                fq.ugsm_fake_fail_call(h.ctx, k, rnd.choice([5, 6]))
        for _ in range(rnd.randrange(50, 300)):
            r = rnd.random()
            if r < 0.6:
                st = h.enqueue(rnd.choice(["full", "full", "fovea", "host", "managed"]), rnd.choice([(64, 32), (64, 32), (128, 64)]))
                assert st == OK or (st == STATE and h.outstanding() >= (slots + 1) * h.batch), (st, h.outstanding())
            elif r < 0.7:
                assert fq.ugsm_flush(h.ctx) == OK
            elif r < 0.95:
                assert h.next_done(0) in (OK, PENDING, EMPTY)
            else:
                assert h.next_done(1) in (OK, EMPTY)
        h.drain()
        h.check()
    finally:
        h.close()


def scenario(h):
    """40 pairs of four kinds and two sizes, fetched as they finish (the books never fill up: no UGSM_ERR_STATE), then a flush."""
    kinds = [("full", (64, 32))] * 7 + [("managed", None)] * 5 + [("fovea", (64, 32))] * 6 + [("host", (128, 64))] * 3 + [("full", (128, 64))] * 4
    for kind, size in kinds + kinds[:15]:
        h.enqueue(kind, size or (32, 16))
        for _ in range(2):
            if h.next_done(0) not in (OK,):
                break
    h.lib.ugsm_flush(h.ctx)


def test_host_allocation_failures_lose_no_pair(fq):
    """Nothing is thrown across the C-ABI and nothing is lost: the same host program with the 1st, 2nd, 3rd ... host allocation inside the
    library failing.  An entry point may then answer UGSM_ERR_NOMEM (an enqueue: the pair is rejected; a fetch: ask again); every pair whose
    enqueue answered UGSM_OK is still reported exactly once, in order, and no slot is reused early."""
    h = Host(fq, 2, 3, poll_delay=1)
    a0 = fq.ugsm_fake_allocs()
    scenario(h)
    h.drain()
    h.check()
    n_allocs = fq.ugsm_fake_allocs() - a0
    h.close()
    assert n_allocs > 10, n_allocs
    refused = fetch_failed = 0
    for k in range(n_allocs + 2):
        h = Host(fq, 2, 3, poll_delay=1)
        try:
            fq.ugsm_fake_fail_alloc_after(k)
            scenario(h)
            before = len(h.reported)
            h.drain(tolerate=True)
            fq.ugsm_fake_fail_alloc_after(-1)
            h.check(faults=True)
            assert all(st in (NOMEM, STATE) for _, st in h.rejected), h.rejected
            refused += any(st == NOMEM for _, st in h.rejected)
            fetch_failed += len(h.accepted) == 40 and not h.rejected and before < 40
        finally:
            h.close()
    assert refused > 3, (refused, fetch_failed)     # (the faults do land: some in an enqueue, which then refuses its pair)
