"""The latency forms of K-cost and K-smooth that run the coarse pyramid levels (csrc/ugsm_kernels_small.hip) against the CPU
oracle and against the LDS-tiled kernels they replace there, bit for bit.

Default contexts use them below ~0.15 Mpx; `small_max_pixels=-1` switches them off (the LDS-tiled kernels everywhere), a large
`small_max_pixels` forces them on, and UGSM_SMALL_RH pins the K-smooth tile height (18 x 4, 18 x 10 or 18 x 18 pixels).
"""
import numpy as np
import pytest

from conftest import assert_bit_equal
from test_gpu_march import iterate, planes, run_smooth, smooth_ref

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def lib():
    import __graft_entry__ as ge
    ge.build_library()
    from ug_stereomatcher_amd import _lib
    return _lib


def stats_names(c):
    return {s["name"] for s in c.kernel_stats() if s["launches"] > 0}


def test_small_cost_sizes_around_the_tile(lib, orc):
    """Two iterations (cost + five passes + box) on sizes one below, at and one above multiples of the 16 x 12 cost tile and the
    18-column smoothing tile, down to images thinner than a halo; the LDS-tiled path must agree as well."""
    rng = np.random.Generator(np.random.PCG64(501))
    cases = [(16, 12), (17, 13), (15, 11), (32, 24), (33, 25), (31, 23), (54, 36), (100, 70), (48, 13), (1, 40), (40, 1), (5, 3), (3, 5), (200, 150)]
    with lib.Context(levels=1, profile_events=2) as cs, lib.Context(levels=1, small_max_pixels=-1) as ct:
        for (W, H) in cases:
            pl, pr = planes(orc, W, H, 5000 + W)
            d0 = np.stack([rng.normal(0, 6, (H, W)), rng.normal(0, 3, (H, W)), 0.2 + 0.8 * rng.random((H, W))]).astype(np.float32)
            exp, _ = orc.iterate_level(pl, pr, d0, 6, 5, False, 1, 2)
            assert_bit_equal(iterate(cs, pl, pr, d0, 6, 5, False, 1, 2), exp, f"small {W}x{H}")
            assert_bit_equal(iterate(ct, pl, pr, d0, 6, 5, False, 1, 2), exp, f"tiled {W}x{H}")
        assert {"k_cost_small", "k_smooth_small"} <= stats_names(cs), stats_names(cs)


def test_small_cost_large_and_wild_disparities_zero_patches_top_level(lib, orc):
    rng = np.random.Generator(np.random.PCG64(502))
    W, H = 150, 77
    pl, pr = planes(orc, W, H, 5100)
    pl[:, 10:22, 12:30] = 0  # 0/0 -> NaN correlation -> (0, 0.4) branch
    pr[:, 40:60, 100:140] = 0
    d0 = np.stack([rng.normal(0, 60, (H, W)), rng.normal(0, 30, (H, W)), 0.2 + 0.8 * rng.random((H, W))]).astype(np.float32)
    dw = np.stack([rng.normal(0, 5, (H, W)), rng.normal(0, 5, (H, W)), 0.2 + 0.8 * rng.random((H, W))]).astype(np.float32)
    dw[0, 3, 5], dw[1, 7, 9], dw[0, 10, 10], dw[1, 11, 11] = np.nan, np.nan, np.inf, -np.inf
    dw[0, 20, 30], dw[1, 21, 31], dw[0, 22, 32] = 3e38, -3e38, 1e-42
    with lib.Context(levels=1) as c:
        for is_top in (False, True):
            exp, _ = orc.iterate_level(pl, pr, d0, 4, 5, is_top, 1, 3)
            assert_bit_equal(iterate(c, pl, pr, d0, 4, 5, is_top, 1, 3), exp, f"top={is_top}")
        with np.errstate(all="ignore"):
            exp, _ = orc.iterate_level(pl, pr, dw, 4, 0, False, 1, 1)
        assert_bit_equal(iterate(c, pl, pr, dw, 4, 0, False, 1, 1), exp, "wild disparities")


def test_small_cost_values_outside_the_usual_range(lib, orc):
    """Tiny, huge and negative plane values (the marching kernel's range guard does not apply here: these kernels always divide
    with the compiler's full sequence)."""
    rng = np.random.Generator(np.random.PCG64(503))
    W, H = 90, 50
    pl, pr = planes(orc, W, H, 5200)
    pl[:, 5:15, 5:25] *= np.float32(1e-20)
    pr[:, 20:30, 40:70] *= np.float32(1e15)
    pl[0, 30:35, 10:20] *= np.float32(-1.0)
    d0 = np.stack([rng.normal(0, 2, (H, W)), rng.normal(0, 2, (H, W)), 0.2 + 0.8 * rng.random((H, W))]).astype(np.float32)
    with np.errstate(all="ignore"):
        exp, _ = orc.iterate_level(pl, pr, d0, 4, 5, False, 1, 2)
    with lib.Context(levels=1) as c:
        assert_bit_equal(iterate(c, pl, pr, d0, 4, 5, False, 1, 2), exp, "out-of-range planes")


@pytest.mark.parametrize("rh", [18, 24, 32])
def test_small_smooth_every_pass_count_and_tile_height(lib, orc, monkeypatch, rh):
    """0-5 passes with and without the box, and 7 / 10 as two launches, for each K-smooth tile height, on sizes around the tile
    (18 columns x rh-14 rows) and degenerate shapes."""
    monkeypatch.setenv("UGSM_SMALL_RH", str(rh))
    rng = np.random.Generator(np.random.PCG64(504 + rh))
    t = rh - 14
    cases = [(18, t), (19, t + 1), (17, max(t - 1, 1)), (36, 2 * t), (37, 2 * t + 1), (54, 36), (1, 40), (40, 1), (5, 3), (3, 5), (130, 75)]
    with lib.Context(levels=1, profile_events=2) as c:
        for (W, H) in cases:
            d = np.stack([rng.normal(0, 3, (H, W)), rng.normal(0, 3, (H, W)), 0.1 + 0.9 * rng.random((H, W))]).astype(np.float32)
            for passes, box in [(0, 1), (1, 0), (1, 1), (2, 0), (3, 1), (4, 0), (5, 0), (5, 1), (7, 1), (10, 1)]:
                assert_bit_equal(run_smooth(c, d, passes, box), smooth_ref(orc, d, passes, box), f"rh={rh} {W}x{H} passes={passes} box={box}")
        assert "k_smooth_small" in stats_names(c) and "k_smooth_fused" not in stats_names(c), stats_names(c)


@pytest.mark.parametrize("rh", [18, 32])
def test_small_smooth_degenerate_confidence(lib, orc, monkeypatch, rh):
    """Confidence fields that push sumCorr out of the shared-reciprocal range (zero patches: 0/0 -> NaN spreading one pixel per
    pass; negative, 1e-30, 1e30 weights), also on the frame and across tile seams."""
    monkeypatch.setenv("UGSM_SMALL_RH", str(rh))
    rng = np.random.Generator(np.random.PCG64(505))
    W, H = 300, 150
    d = np.stack([rng.normal(0, 3, (H, W)), rng.normal(0, 3, (H, W)), 0.1 + 0.9 * rng.random((H, W))]).astype(np.float32)
    d[2, 40:60, 50:90] = 0.0
    d[2, 100:104, 100:140] = -0.25
    d[2, 120:124, 20:60] = 1e-30
    d[2, 130:134, 20:60] = 1e30
    d[0, 140:144, 20:60] = 0.0
    d[2, H - 30:H - 10, W - 80:W - 40] = 0.0
    d[2, 0:3, 200:230] = 0.0
    d[2, 70:90, 0:4] = 0.0
    d[2, 60:64, W - 3:W] = 0.0
    d[2, H - 2:H, 200:240] = 0.0
    d[1, 10:12, 110:120] = np.nan
    with lib.Context(levels=1) as c:
        for passes, box in [(5, 1), (10, 1), (5, 0)]:
            assert_bit_equal(run_smooth(c, d, passes, box), smooth_ref(orc, d, passes, box), f"degenerate rh={rh} passes={passes} box={box}")


def test_small_kernels_forced_on_at_a_mid_level_size(lib, orc):
    """small_max_pixels above the default: interior tiles in bulk (520 x 300, 1 000 cost tiles), against the oracle."""
    rng = np.random.Generator(np.random.PCG64(506))
    W, H = 520, 300
    pl, pr = planes(orc, W, H, 5300)
    d0 = np.stack([rng.normal(0, 6, (H, W)), rng.normal(0, 3, (H, W)), 0.2 + 0.8 * rng.random((H, W))]).astype(np.float32)
    exp, _ = orc.iterate_level(pl, pr, d0, 6, 5, False, 1, 2)
    with lib.Context(levels=1, small_max_pixels=10**9, march_min_pixels=-1, profile_events=2) as c:
        assert_bit_equal(iterate(c, pl, pr, d0, 6, 5, False, 1, 2), exp, "forced small 520x300")
        assert {"k_cost_small", "k_smooth_small"} <= stats_names(c)


def test_whole_matcher_with_and_without_the_small_kernels(lib):
    """Full and foveated mode, 14 levels at 1080p: the result planes do not depend on which kernels ran the coarse levels."""
    from ug_stereomatcher_amd import synth
    W, H = 1920, 1080
    L, R, _, _ = synth.make_pair(W, H, 5400)
    res = []
    for smp in (0, -1):
        with lib.Context(levels=14, fovea_levels=7, small_max_pixels=smp, profile_events=2) as c:
            full = np.asarray(c_match_full(c, L, R))
            fov = np.asarray(c_match_fov(c, L, R))
            res.append((full, fov, stats_names(c)))
    assert "k_cost_small" in res[0][2] and "k_cost_small" not in res[1][2]
    assert_bit_equal(res[0][0], res[1][0], "full mode")
    assert_bit_equal(res[0][1], res[1][1], "foveated mode")


def c_match_full(c, L, R):
    import ctypes as C
    H, W, _ = L.shape
    out = np.empty((3, H, W), np.float32)
    c.check(c.lib.ugsm_match_full(c.handle, L.ctypes.data_as(C.c_void_p), R.ctypes.data_as(C.c_void_p), W, H, W * 3,
                                  out[0].ctypes.data_as(C.c_void_p), out[1].ctypes.data_as(C.c_void_p), out[2].ctypes.data_as(C.c_void_p)))
    return out


def c_match_fov(c, L, R):
    import ctypes as C
    H, W, _ = L.shape
    fw, fh = C.c_int(), C.c_int()
    c.check(c.lib.ugsm_fovea_dims(W, H, 14, 7, C.byref(fw), C.byref(fh)))
    out = np.empty((3, 7, fh.value, fw.value), np.float32)
    c.check(c.lib.ugsm_match_foveated(c.handle, L.ctypes.data_as(C.c_void_p), R.ctypes.data_as(C.c_void_p), W, H, W * 3, 0, 0,
                                      out[0].ctypes.data_as(C.c_void_p), out[1].ctypes.data_as(C.c_void_p), out[2].ctypes.data_as(C.c_void_p), None, None))
    return out


def test_random_sizes_production_path_against_the_one_kernel_per_stage_path(lib):
    """Which kernel runs a level depends on the level's size and on the slot count (marching from 0.2 / 0.4 Mpx, latency kernels up to
    0.15 Mpx, seeding fused where the next level marches, strip heights from a model): random image sizes, pyramid depths, slot counts
    and fovea offsets, production path against kernel_path 1, full and foveated mode (tools/stress_pipeline.py runs hundreds)."""
    import ctypes as C
    from ug_stereomatcher_amd import synth
    rng = np.random.Generator(np.random.PCG64(507))
    for case in range(10):
        W, H = (int(rng.integers(900, 1800)), int(rng.integers(600, 1100))) if case % 5 == 4 else (int(rng.integers(48, 800)), int(rng.integers(40, 600)))
        max_levels, w, h = 1, W, H
        while max_levels < 14 and int(w / 1.41421356) >= 8 and int(h / 1.41421356) >= 8:
            w, h, max_levels = int(w / 1.41421356), int(h / 1.41421356), max_levels + 1
        levels = int(rng.integers(2, max_levels + 1))
        F = int(rng.integers(2, levels + 1))
        off = (int(rng.integers(-W // 8, W // 8 + 1)), int(rng.integers(-H // 8, H // 8 + 1)))
        slots = int(rng.choice([1, 2, 4]))
        L, R, _, _ = synth.make_pair(W, H, 5500 + case)
        out = []
        for path, sl in ((0, slots), (1, 1)):
            with lib.Context(levels=levels, fovea_levels=F, slots=sl, kernel_path=path) as c:
                full = np.empty((3, H, W), np.float32)
                c.check(c.lib.ugsm_match_full(c.handle, L.ctypes.data, R.ctypes.data, W, H, W * 3, full[0].ctypes.data, full[1].ctypes.data, full[2].ctypes.data))
                fw, fh = C.c_int(), C.c_int()
                c.check(c.lib.ugsm_fovea_dims(W, H, levels, F, C.byref(fw), C.byref(fh)))
                st = np.empty((3, F, fh.value, fw.value), np.float32)
                c.check(c.lib.ugsm_match_foveated(c.handle, L.ctypes.data, R.ctypes.data, W, H, W * 3, off[0], off[1], st[0].ctypes.data,
                                                  st[1].ctypes.data, st[2].ctypes.data, None, None))
                out.append((full, st))
        assert_bit_equal(out[0][0], out[1][0], f"full {W}x{H} levels={levels} slots={slots}")
        assert_bit_equal(out[0][1], out[1][1], f"foveated {W}x{H} levels={levels} F={F} off={off} slots={slots}")
