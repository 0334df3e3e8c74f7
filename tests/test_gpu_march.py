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


@pytest.mark.parametrize("np_lane", [1])  # (the two-pixels-per-lane development form is not in libugsm.so: tools/kbench.hip)
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


@pytest.mark.parametrize("np_lane", [1])  # (the two-pixels-per-lane development form is not in libugsm.so: tools/kbench.hip)
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


@pytest.mark.parametrize("np_lane", [1])  # (the two-pixels-per-lane development form is not in libugsm.so: tools/kbench.hip)
def test_march_wild_disparities(lib, orc, np_lane):
    """NaN, +-Inf, huge and denormal disparities go through the branch-free texture index exactly like tex_index
    (NaN -> 0, clamp otherwise)."""
    rng = np.random.Generator(np.random.PCG64(79))
    W, H = 150, 40
    pl, pr = planes(orc, W, H, 4200)
    d0 = np.stack([rng.normal(0, 5, (H, W)), rng.normal(0, 5, (H, W)), 0.2 + 0.8 * rng.random((H, W))]).astype(np.float32)
    wild = np.array([np.nan, np.inf, -np.inf, 3e38, -3e38, 1e10, -1e10, 2147483648.0, -2147483904.0, 1e-45, -1e-45, -0.5, -0.49999997],
                    np.float32)
    idx = rng.integers(0, H * W, 400)
    d0[0].ravel()[idx[:200]] = wild[rng.integers(0, len(wild), 200)]
    d0[1].ravel()[idx[200:]] = wild[rng.integers(0, len(wild), 200)]
    exp, _ = orc.iterate_level(pl, pr, d0, 4, 0, False, 1, 1)
    with lib.Context(levels=1, march_min_pixels=1, march_np=np_lane, march_rows=16) as c:
        got = iterate(c, pl, pr, d0, 4, 0, False, 1, 1)
    assert_bit_equal(got, exp, f"wild np={np_lane}")


def test_march_equals_tiled_end_to_end(lib, orc, monkeypatch):
    """Whole matcher, every level through the marching kernel, against the oracle."""
    from ug_stereomatcher_amd import MatchGPULib, synth
    L, R, _, _ = synth.make_pair(320, 240, synth.BASE_SEED + 7)
    exp = orc.match_full(L, R, 8)
    monkeypatch.setenv("UGSM_MARCH_MIN_PIXELS", "1")
    for np_lane in (1,):
        m = MatchGPULib(levels=8)
        got = m.match(L, R, 0)
        m.close()
        assert_bit_equal(got, exp, f"320x240 full, np={np_lane}")


def test_division_in_range_is_ieee(lib):
    """K-cost's range-guarded division (csrc/ugsm_exact.hpp: the compiler's division sequence without v_div_scale /
    v_div_fixup) equals the IEEE binary32 quotient for every operand pair it can meet: 0 or [2^-62, 2^37]."""
    rng = np.random.Generator(np.random.PCG64(41))
    n = 1 << 22

    def rand_pos(m, emin, emax):
        mant = rng.integers(0, 1 << 23, m, dtype=np.uint32)
        ex = rng.integers(emin + 127, emax + 127, m, dtype=np.uint32)
        return ((ex << 23) | mant).view(np.float32)

    num = rand_pos(n, -62, 37)
    den = rand_pos(n, -62, 37)
    q = n // 8
    # pipeline-like: N^2 <= A*B, both around 1e3..1e9
    den[:q] = (rng.uniform(1.0, 255.0, q) ** 4).astype(np.float32)
    num[:q] = (den[:q] * rng.uniform(0.0, 1.001, q)).astype(np.float32)
    # hard cases of a reciprocal-based quotient: all-ones mantissas, denominators just below / at powers of two, exact
    # quotients, equal operands, the ends of the range
    e = rng.integers(-62, 37, 8192)
    den[q:q + 8192] = np.nextafter(np.float32(2.0) ** e.astype(np.float32), np.float32(0))
    num[q:q + 4096] = np.nextafter(np.float32(2.0) ** rng.integers(-62, 37, 4096).astype(np.float32), np.float32(0))
    num[q + 4096:q + 8192] = den[q + 4096:q + 8192]
    den[2 * q:2 * q + 4096] = np.float32(2.0) ** rng.integers(-62, 37, 4096).astype(np.float32)
    num[3 * q:3 * q + 4096] = (den[3 * q:3 * q + 4096] * np.float32(3.0))
    num[4 * q:4 * q + 2048] = np.float32(2.0 ** -62)
    den[4 * q:4 * q + 1024] = np.nextafter(np.float32(2.0 ** 37), np.float32(0))
    num[4 * q + 2048:4 * q + 4096] = np.nextafter(np.float32(2.0 ** 37), np.float32(0))
    den[4 * q + 2048:4 * q + 3072] = np.float32(2.0 ** -62)
    # zeros: 0/d = +0, 0/0 = NaN (the all-zero 5x5 patch)
    num[5 * q:5 * q + 4096] = 0.0
    den[5 * q + 2048:5 * q + 4096] = 0.0
    with np.errstate(all="ignore"):
        exp = (num / den).astype(np.float32)
    with lib.Context(levels=1, dev=True) as c:   # the probe entry points live in libugsm_dev.so (include/ugsm_dev.h)
        pn, pd = c.to_device(num), c.to_device(den)
        pq = c.alloc(4 * n)
        try:
            c.check(c.lib.ugsm_stage_div_probe(c.handle, pn, pd, pq, n))
            got = c.to_host(pq, (n,))
        finally:
            for p in (pn, pd, pq):
                c.free(p)
    assert np.isnan(exp[5 * q + 2048:5 * q + 4096]).all() and (exp[5 * q:5 * q + 2048] == 0).all()
    assert_bit_equal(got, exp, "range-guarded division")


@pytest.mark.parametrize("np_lane", [1])  # (the two-pixels-per-lane development form is not in libugsm.so: tools/kbench.hip)
def test_march_values_outside_the_division_range_take_the_full_division(lib, orc, np_lane):
    """A plane value outside range_ok (tiny, huge, negative) must switch the pair to the compiler's division; results
    stay bit-exact either way.  The in-range case of the same images exercises the guarded division."""
    rng = np.random.Generator(np.random.PCG64(80))
    W, H = 140, 36
    pl, pr = planes(orc, W, H, 4300)
    d0 = np.stack([rng.normal(0, 3, (H, W)), rng.normal(0, 2, (H, W)), 0.2 + 0.8 * rng.random((H, W))]).astype(np.float32)
    for tag, edit in [("in range", None), ("tiny", 1e-7), ("huge", 3000.0), ("denormal", 1e-41)]:
        L2, R2 = pl.copy(), pr.copy()
        if edit is not None:
            L2[1, 10:14, 20:60] = edit
            R2[2, 20:22, 70:90] = edit
        exp, _ = orc.iterate_level(L2, R2, d0, 4, 5, False, 1, 2)
        with lib.Context(levels=1, march_min_pixels=1, march_np=np_lane, march_rows=16) as c:
            got = iterate(c, L2, R2, d0, 4, 5, False, 1, 2)
        assert_bit_equal(got, exp, f"{tag} np={np_lane}")


def test_march_on_fovea_views(lib, orc, monkeypatch):
    """Foveated mode hands the cost kernel (pointer, pitch) views into the pyramid levels (pitch != width); production never
    runs the marching kernel there (the fovea is below its size threshold), forced on here, against the fixture."""
    from conftest import load_golden
    from ug_stereomatcher_amd import MatchGPULib
    g = load_golden("fovea_320x240_l9_f4.npz")
    monkeypatch.setenv("UGSM_MARCH_MIN_PIXELS", "1")
    for np_lane in (1,):
        m = MatchGPULib(3, ["node", "x", str(int(g["F"]))], levels=int(g["levels"]))
        st = m.matchStack(g["L"], g["R"])
        m.close()
        assert_bit_equal(np.ascontiguousarray(np.asarray(st).transpose(1, 0, 2, 3)), g["stack"], f"fovea stack np={np_lane}")


# ---- K-smooth helpers (shared with test_gpu_small.py) ---------------------------------------------------

def smooth_ref(orc, d, passes, box):
    exp = d
    with np.errstate(all="ignore"):
        for _ in range(passes):
            exp = orc.smooth_pass(exp)
        if box:
            exp = orc.box3(exp)
    return exp


def run_smooth(c, d, passes, box):
    _, H, W = d.shape
    p = c.to_device(d)
    try:
        c.check(c.lib.ugsm_stage_smooth(c.handle, p, W, H, passes, box))
        return c.to_host(p, d.shape)
    finally:
        c.free(p)


def test_march_exact_invariances_at_full_level_size(lib):
    """Size-independent exactness properties of one cost + smoothing iteration, checked at a level size the marching kernel runs
    in production (2.1 Mpx; no oracle needed): (1) scaling both images by a power of two scales every product by an exact factor and
    leaves every quotient, hence (dx, dy, conf), bit-identical; (2) exchanging colour channels 0 and 1 only swaps the operands of
    the first (commutative) channel sum ((q0 + q1) + q2) / 3."""
    from ug_stereomatcher_amd import synth
    W, H = 1920, 1080
    L, R, dx, dy = synth.make_pair(W, H, 6100)
    pl = np.ascontiguousarray(L.transpose(2, 0, 1)).astype(np.float32)
    pr = np.ascontiguousarray(R.transpose(2, 0, 1)).astype(np.float32)
    rng = np.random.Generator(np.random.PCG64(93))
    d0 = np.stack([dx + rng.normal(0, 0.3, dx.shape), dy + rng.normal(0, 0.3, dy.shape), 0.3 + 0.6 * rng.random(dx.shape)]).astype(np.float32)
    with lib.Context(levels=1) as c:  # default threshold (0.2-0.4 Mpx) -> marching cost kernel
        base = iterate(c, pl, pr, d0, 6, 5, False, 1, 2)
        for k in (2.0, 0.5):
            got = iterate(c, pl * np.float32(k), pr * np.float32(k), d0, 6, 5, False, 1, 2)
            assert_bit_equal(got, base, f"images scaled by {k}")
        sw = iterate(c, np.ascontiguousarray(pl[[1, 0, 2]]), np.ascontiguousarray(pr[[1, 0, 2]]), d0, 6, 5, False, 1, 2)
        assert_bit_equal(sw, base, "channels 0 and 1 exchanged")
    with lib.Context(levels=1, march_min_pixels=-1) as c:  # and the LDS-tiled kernel agrees with all of it
        assert_bit_equal(iterate(c, pl, pr, d0, 6, 5, False, 1, 2), base, "tiled kernel")


def test_seeding_fused_into_the_first_cost_launch(lib, monkeypatch):
    """A level whose K-cost is the marching kernel is seeded inside its first launch (launch_cost_march_seeded): same planes as with
    the separate k_seed launch, in full and foveated mode (fovea crop offsets, off-centre fovea), and the k_seed launches are gone."""
    import ctypes as C
    from ug_stereomatcher_amd import synth
    W, H = 700, 500
    L, R, _, _ = synth.make_pair(W, H, 6200)
    monkeypatch.setenv("UGSM_MARCH_MIN_PIXELS", "1")
    res = {}
    for fuse in ("1", "0"):
        monkeypatch.setenv("UGSM_FUSE_SEED", fuse)
        with lib.Context(levels=9, fovea_levels=4, profile_events=2) as c:
            full = np.empty((3, H, W), np.float32)
            c.check(c.lib.ugsm_match_full(c.handle, L.ctypes.data, R.ctypes.data, W, H, W * 3, full[0].ctypes.data, full[1].ctypes.data, full[2].ctypes.data))
            seeds_full = sum(s["launches"] for s in c.kernel_stats() if s["name"] == "k_seed")
            fw, fh = C.c_int(), C.c_int()
            c.check(c.lib.ugsm_fovea_dims(W, H, 9, 4, C.byref(fw), C.byref(fh)))
            st = np.empty((3, 4, fh.value, fw.value), np.float32)
            c.check(c.lib.ugsm_match_foveated(c.handle, L.ctypes.data, R.ctypes.data, W, H, W * 3, 37, -21, st[0].ctypes.data, st[1].ctypes.data,
                                              st[2].ctypes.data, None, None))
            res[fuse] = (full, st, seeds_full)
    assert res["1"][2] == 0 and res["0"][2] == 8, (res["1"][2], res["0"][2])
    assert_bit_equal(res["1"][0], res["0"][0], "full mode, fused seeding")
    assert_bit_equal(res["1"][1], res["0"][1], "foveated mode, fused seeding")
