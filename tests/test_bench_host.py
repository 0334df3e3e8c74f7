"""bench.py's host logic without a GPU: how a region's pairs are dealt into calls (plan_calls) and which interval of completions counts as
steady state (steady_window)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def test_plan_calls_batches_the_backlog_and_staggers_the_start():
    for n in (1, 5, 20, 96, 384, 385, 1000):
        for slots in (1, 2, 4):
            for B in (1, 2, 4, 8, 16):
                for taper in (0.0, 0.5, 1.0):
                    sizes = bench.plan_calls(n, slots, B, taper)
                    assert sum(sizes) == n and all(1 <= v <= B for v in sizes), (n, slots, B, taper, sizes)
    assert bench.plan_calls(20, 4, 1) == [1] * 20                      # single-pair calls: rounds 1-3
    assert bench.plan_calls(20, 4, 4) == [2, 3, 4, 4, 4, 3]            # the driver's protocol: staggered start, full-size calls, the rest in one
    s = bench.plan_calls(384, 4, 4)
    assert s[:6] == [2, 3, 4, 4, 4, 4] and set(s[4:-1]) == {4} and s[-1] == 3   # batches while a backlog exists
    assert bench.plan_calls(20, 4, 4, taper=1.0) == [2, 3, 3, 3, 2, 1, 1, 1, 1, 1, 1, 1]   # (the tapered end of the round's first half, kept for A/Bs)
    t = bench.plan_calls(384, 4, 4, taper=1.0)[-20:]
    assert t == sorted(t, reverse=True) and t[-1] == 1
    assert bench.plan_calls(10, 4, 8, head=[8, 8, 8, 8]) == [8, 2] and bench.plan_calls(10, 4, 8) == [4, 5, 1]


def test_steady_window():
    assert bench.steady_window(bench.plan_calls(20, 4, 4), 4) is None           # 20 steps have no middle
    assert bench.steady_window(bench.plan_calls(20, 4, 1), 4) == (4, 15)        # single pairs: completions slots+1 .. steps-slots
    for taper in (0.0, 1.0):
        s = bench.plan_calls(384, 4, 4, taper)
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
                win = bench.steady_window(bench.plan_calls(n, slots, B), slots)
                assert win is not None and win[1] - win[0] >= 8 * B, (slots, B, steps, win)
