"""The C oracle against the independent numpy restatement (tests/golden/restate_np.py), bit for bit, on the committed
fixtures.  Neither is the reference (which cannot be built or run here and holds no fixtures): two restatements written by
different routes agreeing is what stands in for a pin -- see the header of restate_np.py."""
import os
import sys

import numpy as np

from conftest import GOLDEN, assert_bit_equal, load_golden

sys.path.insert(0, GOLDEN)
import restate_np as rn  # noqa: E402


def test_constants_and_schedules(orc):
    assert_bit_equal(rn.gauss_taps(), orc.gauss_taps(), "gauss taps")
    assert rn.level_dims(4928, 3264, 14) == tuple(list(v) for v in orc.level_dims(4928, 3264, 14)) or \
        [list(v) for v in rn.level_dims(4928, 3264, 14)] == [list(v) for v in orc.level_dims(4928, 3264, 14)]
    for mi in (2, 4, 6, 8, 10, 12, 22):
        assert_bit_equal(np.array(rn.thresholds(mi), np.float32), orc.threshold_schedule(mi), f"thresholds mi={mi}")
    for i in range(14):
        assert rn.iterations(i) == orc.iterations_for_level(i) and rn.smooth_passes(i) == orc.smooth_passes_for_level(i)


def test_stage_fixture_96x72(orc):
    g = load_golden("stage_96x72.npz")
    pl, pr = rn.planes(g["L"]), rn.planes(g["R"])
    assert_bit_equal(pl, orc.rgb_to_planes(g["L"]), "planes")
    pyr = rn.pyramid(pl, 4)
    for i in (1, 2, 3):
        assert_bit_equal(pyr[i], g[f"pyr{i}"], f"pyramid level {i}")
    # level index 1: mi = 4 iterations, S = 10 passes in the schedule; the fixture ran mi=4, S=5 -> use level 2's S by hand
    d1 = _iterate(rn, pl, pr, g["d0"], mi=4, S=5, is_top=False, m_from=1, m_to=1)
    assert_bit_equal(d1, g["d1"], "after iteration 1")
    d3 = _iterate(rn, pl, pr, g["d0"], mi=4, S=5, is_top=False, m_from=1, m_to=3)
    assert_bit_equal(d3, g["d3"], "after iteration 3")
    assert_bit_equal(rn.smooth_pass(g["d0"]), g["smooth1"], "smooth")
    assert_bit_equal(np.stack([rn.blur(p, rn.BOX, "clamp") for p in g["d0"]]), g["box"], "box")
    assert_bit_equal(rn.seed(g["d0"], g["seed"].shape[2], g["seed"].shape[1]), g["seed"], "seed")


def _iterate(rn, L, R, d, mi, S, is_top, m_from, m_to):
    """restate_np.iterate_level with an explicit (mi, S) instead of the level's schedule (the stage fixture uses its own)"""
    it, sp = rn.iterations, rn.smooth_passes
    rn.iterations, rn.smooth_passes = (lambda i: mi), (lambda i: S)
    try:
        return rn.iterate_level(L, R, d, 0, is_top, m_from, m_to)
    finally:
        rn.iterations, rn.smooth_passes = it, sp


def test_top_level_first_iteration_has_no_blend(orc):
    g = load_golden("stage_96x72.npz")
    pl, pr = rn.planes(g["L"]), rn.planes(g["R"])
    # the fixture: zero seed, mi = 22, S = 10, is_top, iterations 1-2 (tests/golden/make_golden.py)
    assert_bit_equal(_iterate(rn, pl, pr, np.zeros_like(g["d0"]), 22, 10, True, 1, 2), g["dtop"], "top level, iterations 1-2")


def test_full_64x48(orc):
    g = load_golden("full_64x48_l5.npz")
    out = rn.match_full(g["L"], g["R"], int(g["levels"]))
    assert_bit_equal(out, g["out"], "numpy restatement vs fixture")
    assert_bit_equal(out, orc.match_full(g["L"], g["R"], int(g["levels"])), "numpy restatement vs C oracle")


def test_foveated_320x240(orc):
    g = load_golden("fovea_320x240_l9_f4.npz")
    st = rn.match_foveated(g["L"], g["R"], int(g["levels"]), int(g["F"]))
    assert_bit_equal(st, g["stack"], "foveated stack")
