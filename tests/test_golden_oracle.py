"""The CPU oracle against the committed golden fixtures (tests/golden/, made by make_golden.py)."""
import numpy as np

from conftest import assert_bit_equal, load_golden


def test_full_small(orc):
    for name in ("full_64x48_l5.npz", "full_160x120_l8.npz"):
        g = load_golden(name)
        out = orc.match_full(g["L"], g["R"], int(g["levels"]))
        assert_bit_equal(out, g["out"], name)


def test_full_is_thread_count_independent(orc):
    g = load_golden("full_64x48_l5.npz")
    n0 = orc.num_threads()
    try:
        orc.set_num_threads(1)
        a = orc.match_full(g["L"], g["R"], 5)
        orc.set_num_threads(3)
        b = orc.match_full(g["L"], g["R"], 5)
    finally:
        orc.set_num_threads(n0)
    assert_bit_equal(a, b, "1 vs 3 threads")
    assert_bit_equal(a, g["out"], "vs golden")


def test_stage_fixture(orc):
    g = load_golden("stage_96x72.npz")
    pl, pr = orc.rgb_to_planes(g["L"]), orc.rgb_to_planes(g["R"])
    pyr = orc.pyramid(pl, 4)
    for i in (1, 2, 3):
        assert_bit_equal(pyr[i], g[f"pyr{i}"], f"pyramid level {i}")
    d1, dbg = orc.iterate_level(pl, pr, g["d0"], 4, 5, False, 1, 1, want_dbg=True)
    assert_bit_equal(dbg, g["dbg"], "Q/dx'/dy'/kappa")
    assert_bit_equal(d1, g["d1"], "after iteration 1")
    d3, _ = orc.iterate_level(pl, pr, g["d0"], 4, 5, False, 1, 3)
    assert_bit_equal(d3, g["d3"], "after iteration 3")
    # iterating 1..3 in one call == three single calls (the schedule depends on m only)
    d = g["d0"]
    for m in (1, 2, 3):
        d, _ = orc.iterate_level(pl, pr, d, 4, 5, False, m, m)
    assert_bit_equal(d, g["d3"], "stepwise")
    assert_bit_equal(orc.smooth_pass(g["d0"]), g["smooth1"], "smooth")
    assert_bit_equal(orc.box3(g["d0"]), g["box"], "box")
    assert_bit_equal(orc.seed(g["d0"], g["seed"].shape[2], g["seed"].shape[1]), g["seed"], "seed")


def test_fovea_fixture(orc):
    g = load_golden("fovea_320x240_l9_f4.npz")
    levels, F = int(g["levels"]), int(g["F"])
    st, pl, pr = orc.match_foveated(g["L"], g["R"], levels, F, 0, 0, want_pyr=True)
    assert_bit_equal(st, g["stack"], "stack")
    assert_bit_equal(pl, g["pyrL"], "pyrL")
    assert_bit_equal(pr, g["pyrR"], "pyrR")
    ox, oy = (int(v) for v in g["off"])
    st2, _, _ = orc.match_foveated(g["L"], g["R"], levels, F, ox, oy)
    assert_bit_equal(st2, g["stack_off"], "off-centre stack")
    # coarsest fovea level (whole level F-1) does not depend on the window
    assert_bit_equal(st2[:, F - 1], st[:, F - 1], "level F-1 is window independent")
    assert not np.array_equal(st2[:, 0], st[:, 0])


def test_fovea_geometry_reference_case(orc):
    # MatchGPULib.cpp:1143-1146,1173-1176,1612-1615 at 16 MP: fovea 615x407, centred
    fw, fh, ox, oy, cx, cy = orc.fovea_geometry(4928, 3264, 14, 7, 0, 0)
    w, h = orc.level_dims(4928, 3264, 14)
    assert (fw, fh) == (615, 407)
    assert ox == [w[i] // 2 - fw // 2 for i in range(6)] and oy == [h[i] // 2 - fh // 2 for i in range(6)]
    assert cx == [w[5] // 2 - fw // 2] * 6 and cy == [h[5] // 2 - fh // 2] * 6
    # off-centre windows stay inside every level and inside the upsampled parent
    for off in [(900, -500), (-2400, 1600), (5000, 5000)]:
        fw, fh, ox, oy, cx, cy = orc.fovea_geometry(4928, 3264, 14, 7, *off)
        for i in range(6):
            assert 0 <= ox[i] <= w[i] - fw and 0 <= oy[i] <= h[i] - fh
            assert 0 <= cx[i] <= w[5] - fw and 0 <= cy[i] <= h[5] - fh
