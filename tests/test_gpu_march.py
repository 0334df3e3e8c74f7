"""K-cost as the marching kernel (csrc/ugsm_kernels_march.hip) against the CPU oracle, bit for bit.

The marching kernel runs the large levels in production; here it is forced on for every size
(march_min_pixels=1) so that strip seams, frame edges, both pixels-per-lane forms and several strip
heights are crossed on images the oracle finishes in seconds.
"""
import numpy as np
import pytest

from conftest import assert_bit_equal

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def lib():
    import __graft_entry__ as ge
    ge.build_library()
    from ug_stereomatcher_amd import _lib
    return _lib


def iterate(c, L3, R3, d3, mi, S, is_top, m_from, m_to):
    _, H, W = L3.shape
    pL, pR, pd = c.to_device(L3), c.to_device(R3), c.to_device(d3)
    try:
        c.check(c.lib.ugsm_stage_iterate(c.handle, pL, pR, pd, W, H, mi, S, int(is_top), m_from, m_to, None))
        return c.to_host(pd, (3, H, W))
    finally:
        for p in (pL, pR, pd):
            c.free(p)


def planes(orc, W, H, seed):
    from ug_stereomatcher_amd import synth
    L, R, _, _ = synth.make_pair(max(W, 16), max(H, 16), seed)
    pl = np.ascontiguousarray(orc.rgb_to_planes(L)[:, :H, :W])
    pr = np.ascontiguousarray(orc.rgb_to_planes(R)[:, :H, :W])
    return pl, pr


@pytest.mark.parametrize("np_lane", [1, 2])
def test_march_one_iteration_sizes_and_strip_heights(lib, orc, np_lane):
    """One cost iteration (no smoothing: S = 0 passes still runs the box, so compare after the full stage) on sizes
    around the strip widths (58 / 122 columns), with strips shorter and taller than the image."""
    rng = np.random.Generator(np.random.PCG64(77))
    cases = [(300, 90, 0), (123, 40, 16), (122, 33, 7), (59, 70, 33), (58, 21, 5), (257, 64, 16), (31, 9, 4), (640, 48, 16)]
    for (W, H, rows) in cases:
        pl, pr = planes(orc, W, H, 4000 + W)
        d0 = np.stack([rng.normal(0, 6, (H, W)), rng.normal(0, 3, (H, W)), 0.2 + 0.8 * rng.random((H, W))]).astype(np.float32)
        exp, _ = orc.iterate_level(pl, pr, d0, 6, 5, False, 1, 2)
        with lib.Context(levels=1, march_min_pixels=1, march_np=np_lane, march_rows=rows) as c:
            got = iterate(c, pl, pr, d0, 6, 5, False, 1, 2)
        assert_bit_equal(got, exp, f"{W}x{H} rows={rows} np={np_lane}")


@pytest.mark.parametrize("np_lane", [1, 2])
def test_march_large_disparities_top_level_and_zero_patches(lib, orc, np_lane):
    rng = np.random.Generator(np.random.PCG64(78))
    W, H = 200, 77
    pl, pr = planes(orc, W, H, 4100)
    pl[:, 10:22, 12:30] = 0  # 0/0 -> NaN correlation -> (0, 0.4) branch
    pr[:, 40:60, 100:150] = 0
    d0 = np.stack([rng.normal(0, 60, (H, W)), rng.normal(0, 30, (H, W)), 0.2 + 0.8 * rng.random((H, W))]).astype(np.float32)
    with lib.Context(levels=1, march_min_pixels=1, march_np=np_lane, march_rows=16) as c:
        for is_top in (False, True):
            exp, _ = orc.iterate_level(pl, pr, d0, 4, 5, is_top, 1, 3)
            got = iterate(c, pl, pr, d0, 4, 5, is_top, 1, 3)
            assert np.isfinite(exp).all()
            assert_bit_equal(got, exp, f"top={is_top} np={np_lane}")


def test_march_equals_tiled_end_to_end(lib, orc, monkeypatch):
    """Whole matcher, every level through the marching kernel, against the oracle."""
    from ug_stereomatcher_amd import MatchGPULib, synth
    L, R, _, _ = synth.make_pair(320, 240, synth.BASE_SEED + 7)
    exp = orc.match_full(L, R, 8)
    monkeypatch.setenv("UGSM_MARCH_MIN_PIXELS", "1")
    for np_lane in (1, 2):
        monkeypatch.setenv("UGSM_MARCH_NP", str(np_lane))
        m = MatchGPULib(levels=8)
        got = m.match(L, R, 0)
        m.close()
        assert_bit_equal(got, exp, f"320x240 full, np={np_lane}")
