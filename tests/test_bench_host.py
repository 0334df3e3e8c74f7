"""Host logic without a GPU: how the LIBRARY's queue deals a burst of pairs into calls (ugsm_queue_plan, include/ugsm.h -- the rule
bench.py's plan_calls had in rounds 3-4, now behind the C-ABI) and which interval of completions bench.py counts as steady state
(steady_window: analysis of the calls the library formed, no planning)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def _plan(n, slots, B):
    import __graft_entry__ as ge
    ge.build_library()
    from ug_stereomatcher_amd import _lib
    return _lib.queue_plan(n, slots=slots, batch=B)


def test_bench_holds_no_call_planning():
    src = open(os.path.join(ROOT, "bench.py")).read()
    assert "def plan_calls" not in src
    timed_loop = src.split("    def run(n):")[1].split("    def call_sizes_of(")[0]      # the body of the timed region
    assert "ctx.enqueue_full(" in timed_loop and "ctx.flush()" in timed_loop and "fetch(" in timed_loop
    assert "submit_full" not in timed_loop and "% slots" not in timed_loop.split('if mode == "fovea-shard":')[0] + timed_loop.split("return\n", 1)[1]


def test_queue_plan_batches_the_backlog_and_staggers_the_start():
    for n in (0, 1, 5, 20, 96, 384, 385, 1000):
        for slots in (1, 2, 4):
            for B in (1, 2, 4, 8, 16):
                sizes = _plan(n, slots, B)
                assert sum(sizes) == n and all(1 <= v <= B for v in sizes), (n, slots, B, sizes)
    assert _plan(20, 4, 1) == [1] * 20                      # single-pair calls: rounds 1-3
    assert _plan(20, 4, 4) == [2, 3, 4, 4, 4, 3]            # staggered start, full-size calls, the rest in one
    assert _plan(20, 4, 8) == [4, 5, 7, 4]                  # the driver's protocol with bench.py's default batch
    s = _plan(384, 4, 4)
    assert s[:6] == [2, 3, 4, 4, 4, 4] and set(s[4:-1]) == {4} and s[-1] == 3   # batches while a backlog exists
    assert _plan(10, 4, 8) == [4, 5, 1] and _plan(5, 1, 8) == [5]
    # bad arguments are an answer, not a crash
    import ctypes as C
    from ug_stereomatcher_amd import _lib
    assert _lib.load().ugsm_queue_plan(None, -1, None, 0) == -1
    assert _lib.load().ugsm_queue_plan(None, 3, None, 0) == 3        # (count only; default config: one slot, batch 1)
    buf = (C.c_int * 2)()
    assert _lib.load().ugsm_queue_plan(None, 3, buf, 2) == 3 and list(buf) == [1, 1]


def test_steady_window():
    assert bench.steady_window(_plan(20, 4, 4), 4) is None           # 20 steps have no middle
    assert bench.steady_window(_plan(20, 4, 1), 4) == (4, 15)        # single pairs: completions slots+1 .. steps-slots
    s = _plan(384, 4, 4)
    lo, hi = bench.steady_window(s, 4)
    first = next(j for j in range(4, len(s)) if all(v == 4 for v in s[j:j + 5]))
    assert lo == sum(s[:first + 1]) - 1 and lo < hi < 384
    # every call between the two ends, and the `slots` calls behind the last one, is full-size
    j_hi = next(j for j in range(len(s)) if sum(s[:j + 1]) - 1 == hi)
    assert all(v == 4 for v in s[first:j_hi + 5]) and s[j_hi + 5] < 4
    assert bench.steady_window([], 4) is None and bench.steady_window([4] * 7, 4) is None


def test_the_further_region_of_a_short_run_always_has_a_middle():
    # bench.py: a region without a steady window (the driver's --steps 20) is followed by one of max(steps, 48 B, 12 slots B) steps
    for slots in (1, 2, 4):
        for B in (1, 2, 4, 8, 16):
            for steps in (1, 5, 20):
                n = max(steps, 48 * B, 12 * slots * B)
                win = bench.steady_window(_plan(n, slots, B), slots)
                assert win is not None and win[1] - win[0] >= 8 * B, (slots, B, steps, win)


def test_the_line_compares_like_with_like_and_carries_the_other_configs():
    """VERDICT r05 #2, pinned on the source (the line itself needs a GPU; tests/test_gpu_queue.py::test_bench_line_keys runs it): `vs_baseline` is the
    blocking host-memory call over the reference's figure -- the reference's own bracket -- never the device-resident throughput; both ratios
    and their brackets are named under baseline_comparison; fovea16mp and 1080p ride along under other_workloads."""
    src = open(os.path.join(ROOT, "bench.py")).read()
    assert '"vs_baseline": None,' in src and 'result["vs_baseline"] = (same / ref) if same else None' in src
    assert 'same = result.get("pcie_inclusive", {}).get("pageable_pairs_per_s")' in src
    assert "value / REFERENCE_PAIRS_PER_S" not in src
    for key in ('"baseline_comparison"', '"same_bracket"', '"throughput"', '"vs_baseline_is": "same_bracket"', '"other_workloads"', '"blocking_call_ms"',
                '"vs_reference_same_bracket"', '"on_the_timed_context"'):
        assert key in src, key
    assert 'for name in ("fovea16mp", "1080p"):' in src
    assert bench.REFERENCE_PAIRS_PER_S == {"full16mp": 0.1, "fovea16mp": 1.0 / 3.0} and bench.DEFAULT_BATCH["1080p"] == 16
