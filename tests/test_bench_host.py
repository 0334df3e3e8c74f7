"""bench.py's host logic without a GPU: how a region's pairs are dealt into calls (plan_calls) and which interval of completions counts as
steady state (steady_window)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def test_plan_calls_batches_the_backlog_staggers_the_start_and_tapers_the_end():
    for n in (1, 5, 20, 96, 384, 385, 1000):
        for slots in (1, 2, 4):
            for B in (1, 2, 4, 8, 16):
                sizes = bench.plan_calls(n, slots, B)
                assert sum(sizes) == n and all(1 <= v <= B for v in sizes), (n, slots, B, sizes)
    assert bench.plan_calls(20, 4, 1) == [1] * 20                                # single-pair calls: rounds 1-3
    assert bench.plan_calls(20, 4, 4) == [2, 3, 3, 3, 2, 1, 1, 1, 1, 1, 1, 1]    # the driver's protocol: staggered start, then by what is left
    s = bench.plan_calls(384, 4, 4)
    assert s[:6] == [2, 3, 4, 4, 4, 4] and s[-4:] == [1, 1, 1, 1]
    body = s[4:-20]
    assert set(body) == {4}                                                      # batches while a backlog exists
    tail = s[-20:]
    assert tail == sorted(tail, reverse=True)                                    # ... tapering monotonically to single pairs
    assert bench.plan_calls(10, 4, 8, head=[8, 8, 8, 8]) == [2, 2, 1, 1, 1, 1, 1, 1]   # the backlog rule still caps an overridden start


def test_steady_window():
    assert bench.steady_window(bench.plan_calls(20, 4, 4), 4) is None           # 20 steps have no middle
    assert bench.steady_window(bench.plan_calls(20, 4, 1), 4) == (4, 15)        # single pairs: completions slots+1 .. steps-slots
    s = bench.plan_calls(384, 4, 4)
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
