"""Known-answer tests that pin the CPU oracle to the reference source.

The reference has no tests or golden data; these are the analytic cases derivable from its
source plus the one executable fragment (convolutionSeparable_gold.cpp -> blur_gold.npz).
"""
import ctypes as C

import numpy as np
import pytest

from conftest import assert_bit_equal, load_golden


def test_gauss_taps_bit_patterns(orc):
    # MatchGPULib.cpp:761-774; SURVEY.md A.1: LE hex 250eb93d ea9f773e 04d9ab3e
    g = orc.gauss_taps()
    assert g.tobytes().hex() == "250eb93dea9f773e04d9ab3eea9f773e250eb93d"
    assert np.array_equal(orc.box_taps(), np.array([0, 0.3333, 0.3333, 0.3333, 0], np.float32))


def test_level_dims_16mp(orc):
    # SURVEY.md Appendix B
    w, h = orc.level_dims(4928, 3264, 14)
    assert w == [4928, 3484, 2463, 1741, 1231, 870, 615, 434, 306, 216, 152, 107, 75, 53]
    assert h == [3264, 2307, 1631, 1153, 815, 576, 407, 287, 202, 142, 100, 70, 49, 34]
    w, h = orc.level_dims(1920, 1080, 14)
    assert w == [1920, 1357, 959, 678, 479, 338, 239, 168, 118, 83, 58, 41, 28, 19]
    assert h == [1080, 763, 539, 381, 269, 190, 134, 94, 66, 46, 32, 22, 15, 10]
    with pytest.raises(ValueError):
        orc.level_dims(64, 48, 14)


def test_iteration_and_smoothing_schedule(orc):
    assert [orc.iterations_for_level(i) for i in range(14)] == [2, 4, 6, 8, 10, 12] + [22] * 8
    assert [orc.smooth_passes_for_level(i) for i in range(14)] == [10, 10] + [5] * 12


def test_threshold_schedule_table(orc):
    # SURVEY.md A.5 step 5 known-answer table (MatchGPULib.cpp:2299-2306)
    f = np.float32
    table = {
        2: [1, 1],
        4: [1, 1, .1, .1],
        6: [1, 1, .55, .55, .1, .1],
        8: [1, 1, .7, .7, .4, .4, .1, .1],
        10: [1, 1, .775, .775, .55, .55, .325, .325, .1, .1],
        12: [1, 1, .82, .82, .64, .64, .46, .46, .28, .28, .1, .1],
        22: [1] * 10 + [.55, .55, .46, .46, .37, .37, .28, .28, .19, .19, .1, .1],
    }
    for mi, exp in table.items():
        got = orc.threshold_schedule(mi)
        assert np.allclose(got, np.array(exp, f), rtol=0, atol=1e-6), (mi, got)
    # exact: the value is f32(double expression)
    assert orc.threshold_schedule(6)[2] == f((3 - 1 - 1) * ((1 - 0.1) / (3 - 1.0)) + 0.1)


def test_poly_known_answers(orc):
    # (l,c,r) = (0.5,0.9,0.7) -> b1=.1, c1=-.3, delta=1/6, c*=.908333, corr=.9725
    d, k = orc.poly(0.9, 0.5, 0.7, 1.0)
    assert abs(d - 1 / 6) < 1e-6
    assert abs(k - 0.9725) < 1e-6
    # c1 >= 0 -> (0, 0.4)
    assert orc.poly(0.5, 0.5, 0.5, 1.0) == (np.float32(0), np.float32(0.4))
    assert orc.poly(0.2, 0.5, 0.7, 1.0) == (np.float32(0), np.float32(0.4))
    # NaN anywhere -> c1 < 0 is false -> (0, 0.4)   (MatchLib.cu:812,834-836)
    for args in [(np.nan, .5, .7), (.9, np.nan, .7), (.9, .5, np.nan)]:
        assert orc.poly(*args, 1.0) == (np.float32(0), np.float32(0.4))
    # clamp to +-thr
    d, _ = orc.poly(0.9, 0.1, 0.8, 0.1)
    assert d == np.float32(0.1)
    d, _ = orc.poly(0.9, 0.8, 0.1, 0.1)
    assert d == np.float32(-0.1)
    # c* > 1 -> corr = 1 and delta rescaled by (1-c)/(c*-c)
    c, l, r = np.float32(0.99), np.float32(0.2), np.float32(0.9)
    d, k = orc.poly(c, l, r, 1.0)
    b1 = (r - l) / np.float32(2)
    c1 = r - (c + b1)
    dh = np.float32((np.float64(-b1) * 0.5) / np.float64(c1))
    cstar = (c1 * dh + b1) * dh + c
    assert cstar > 1 and k == np.float32(1.0)
    assert d == np.float32(np.float64(dh) * ((1.0 - np.float64(c)) / np.float64(cstar - c)))


def test_blur_matches_reference_gold(orc):
    """Zero-padded row/column conv == the reference's convolutionRowCPU/ColumnCPU outputs
    (fixture generated from oracle/_ref/libgold.so, i.e. from the reference's own code)."""
    g = load_golden("blur_gold.npz")
    taps = g["taps"]
    assert_bit_equal(taps, orc.gauss_taps(), "taps")
    for name in "abcd":
        src = g[f"{name}_src"]
        row = orc.conv_rows_zero(src, taps)
        assert_bit_equal(row, g[f"{name}_row"], f"row {name}")
        assert_bit_equal(orc.conv_cols_zero(row, taps), g[f"{name}_col"], f"col {name}")
    # all-ones row -> [0.66782004, 0.90964097, 1, ...]
    r = orc.conv_rows_zero(np.ones((1, 8), np.float32), taps)
    assert_bit_equal(r, g["ones_row"], "ones")
    assert r[0, 0] == np.float32(0.66782004) and r[0, 1] == np.float32(0.90964097)


def test_blur_matches_live_reference_when_present(orc):
    gold = orc.gold_lib()
    if gold is None:
        pytest.skip("oracle/_ref/libgold.so not built (no /root/reference here)")
    f32p = C.POINTER(C.c_float)
    rng = np.random.Generator(np.random.PCG64(11))
    taps = orc.gauss_taps()
    for H, W in [(17, 23), (2, 2), (40, 131)]:
        src = rng.random((H, W), dtype=np.float32) * 1000
        row = np.empty_like(src)
        col = np.empty_like(src)
        gold.convolutionRowCPU(row.ctypes.data_as(f32p), src.ctypes.data_as(f32p), taps.ctypes.data_as(f32p), W, H, 2)
        gold.convolutionColumnCPU(col.ctypes.data_as(f32p), row.ctypes.data_as(f32p), taps.ctypes.data_as(f32p), W, H, 2)
        assert_bit_equal(orc.conv(src, taps, "zero"), col, f"{W}x{H}")


def test_clamp_conv_constant_field(orc):
    v = np.full((9, 11), 7.0, np.float32)
    out = orc.conv(v, orc.gauss_taps(), "clamp")
    assert np.all(out == out[0, 0]) and abs(out[0, 0] - 7.0) < 1e-5
    # box gain 0.9999 per pass -> 0.9998.. per iteration
    d = np.full((3, 9, 11), 1.0, np.float32)
    b = orc.box3(d)
    assert abs(b[0, 4, 5] - 0.9999 ** 2) < 1e-6 and np.all(b == b[0, 0, 0])


def test_smoothing_leaves_row0_col0(orc):
    rng = np.random.Generator(np.random.PCG64(3))
    d = rng.random((3, 13, 17), dtype=np.float32)
    s = orc.smooth_pass(d)
    assert np.array_equal(s[:, 0, :], d[:, 0, :]) and np.array_equal(s[:, :, 0], d[:, :, 0])
    # interior value: weighted cross mean in the reference's order
    y, x = 5, 7
    w = d[2]
    acc = np.float32(0)
    den = np.float32(0)
    for (yy, xx) in [(y, x), (y, x - 1), (y, x + 1), (y - 1, x), (y + 1, x)]:
        acc = np.float32(d[0, yy, xx] * w[yy, xx]) + acc
        den = den + w[yy, xx]
    assert s[0, y, x] == np.float32(acc / den)
    # right/bottom edges clamp
    y, x = 12, 16
    acc = den = np.float32(0)
    for (yy, xx) in [(y, x), (y, x - 1), (y, x), (y - 1, x), (y, x)]:
        acc = np.float32(d[1, yy, xx] * w[yy, xx]) + acc
        den = den + w[yy, xx]
    assert s[1, y, x] == np.float32(acc / den)


def test_seed_index_map_and_scale(orc):
    # MatchLib.cu:381-394: dst = f32(1.41421356 * src[floor((i+.5f)*0.70710677f)])
    W, H, W2, H2 = 10, 7, 14, 9
    src = np.arange(3 * H * W, dtype=np.float32).reshape(3, H, W)
    dst = orc.seed(src, W2, H2)
    sf = np.float32(1 / 1.41421356)
    for (iy, ix) in [(0, 0), (3, 5), (8, 13), (8, 0)]:
        sx = min(int(np.floor(np.float32(np.float32(ix) + np.float32(0.5)) * sf)), W - 1)
        sy = min(int(np.floor(np.float32(np.float32(iy) + np.float32(0.5)) * sf)), H - 1)
        for c in range(3):
            assert dst[c, iy, ix] == np.float32(1.41421356 * np.float64(src[c, sy, sx]))


def test_pyramid_structure(orc):
    rng = np.random.Generator(np.random.PCG64(5))
    p0 = (rng.random((3, 40, 60), dtype=np.float32) * 255).astype(np.float32)
    pyr = orc.pyramid(p0, 4)
    taps = orc.gauss_taps()
    b0 = orc.conv(np.ascontiguousarray(p0[1]), taps, "zero")
    # level 2 = blur(level 0)[2y+1, 2x+1]
    assert_bit_equal(pyr[2][1], b0[1::2, 1::2][: pyr[2].shape[1], : pyr[2].shape[2]], "level2")
    # level 1 = blur(level 0)[floor((i+.5f)*sqrt2f)]
    sf = np.float32(1.41421356)
    ys = np.floor((np.arange(pyr[1].shape[1], dtype=np.float32) + np.float32(0.5)) * sf).astype(int)
    xs = np.floor((np.arange(pyr[1].shape[2], dtype=np.float32) + np.float32(0.5)) * sf).astype(int)
    assert_bit_equal(pyr[1][1], b0[ys][:, xs], "level1")
    b1 = orc.conv(np.ascontiguousarray(pyr[1][0]), taps, "zero")
    assert_bit_equal(pyr[3][0], b1[1::2, 1::2][: pyr[3].shape[1], : pyr[3].shape[2]], "level3")


def test_identical_images_zero_seed_do_not_move(orc):
    """L == R and a zero seed: the centre correlation is 1 everywhere it is well defined, so the
    interior parabola step is ~0: exactly -b1/(2 c1) with b1 = (Q_r - Q_l)/2 tiny."""
    from ug_stereomatcher_amd import synth
    L, _, _, _ = synth.make_pair(64, 48, 99)
    pl = orc.rgb_to_planes(L)
    d0 = np.zeros((3, 48, 64), np.float32)
    d1, dbg = orc.iterate_level(pl, pl, d0, mi=4, S=5, is_top=True, m_from=1, m_to=1, want_dbg=True)
    inner = (slice(4, -4), slice(4, -4))
    assert np.all(dbg[4][inner] > 0.999)  # Q centre ~ 1
    assert np.all(np.abs(dbg[5][inner]) < 0.05) and np.all(np.abs(dbg[6][inner]) < 0.05)
    assert np.median(np.abs(dbg[5][inner])) < 1e-3


def test_constant_shift_is_recovered(orc):
    """R(x) = L(x-k): interior dx ~ k.  Statistical only: the nearest-neighbour warp makes the
    iteration hover around the rounding boundary of floor(x+.5+dx), so each level settles within
    about half a pixel of the truth (the algorithm's own noise floor)."""
    from ug_stereomatcher_amd import synth
    L, _, _, _ = synth.make_pair(200, 150, 42)
    k = 3
    R = np.roll(L, k, axis=1)
    out = orc.match_full(L, R, levels=8)
    inner = out[0][30:-30, 30:-30]
    assert abs(np.median(inner) - k) < 0.5
    assert abs(np.median(out[1][30:-30, 30:-30])) < 0.5
    assert not np.isnan(out).any()


def test_triangulation_rectified_rig(orc):
    """SURVEY 8f row f-1 (getPointCloud.cpp:886-949).  For a rectified rig P1 = K[I|0], P2 = K[I|t],
    t = (-B, 0, 0), the closed form must give Z = -f*B/dx, X = (x-cx)*Z/f, Y = (y-cy)*Z/f."""
    f, cx, cy, B = 1000.0, 320.0, 240.0, 0.1
    P1 = np.array([[f, 0, cx, 0], [0, f, cy, 0], [0, 0, 1, 0]])
    P2 = np.array([[f, 0, cx, -f * B], [0, f, cy, 0], [0, 0, 1, 0]])
    H, W = 48, 64
    Z = 2.0 + 0.5 * np.random.default_rng(1).random((H, W))
    dx = (-f * B / Z).astype(np.float32)
    xyz = orc.triangulate(dx, np.zeros((H, W), np.float32), P1, P2)
    xs, ys = np.meshgrid(np.arange(W), np.arange(H))
    assert np.abs(xyz[2] - Z).max() < 1e-3
    assert np.abs(xyz[0] - (xs - cx) * Z / f).max() < 1e-3
    assert np.abs(xyz[1] - (ys - cy) * Z / f).max() < 1e-3


def test_fovea_mapping_known_answers(orc):
    """getPointCloud.cpp:431-484 at 16 MP: level 0 of the fovea stack is the 615x407 centre crop of the full
    frame (Appendix B), level k covers width[6-k] x height[6-k] full-resolution pixels; scale = sqrt2^k."""
    l, u, sc = orc.fovea_mapping(4928, 3264, 0)
    assert (l, u) == (4928 // 2 - 615 // 2, 3264 // 2 - 407 // 2) == (2157, 1429)
    assert sc == np.float32(1.0)
    w, h = orc.level_dims(4928, 3264, 14)
    for k in range(1, 7):
        l, u, sc = orc.fovea_mapping(4928, 3264, k)
        assert (l, u) == (w[0] // 2 - w[6 - k] // 2, h[0] // 2 - h[6 - k] // 2)
        assert abs(float(sc) - 2.0 ** (k / 2)) < 2e-6 * 2.0 ** (k / 2)
    l, u, sc = orc.fovea_mapping(4928, 3264, 6)
    assert (l, u) == (0, 0)     # the coarsest fovea level is the whole frame


def test_fovea_triangulation_reduces_to_full_res_formula(orc):
    """With scale 1 and zero margins the foveated branch differs from the full-resolution one only by the
    int truncation of the right-image coordinate (mapXcoord takes an int)."""
    rng = np.random.Generator(np.random.PCG64(12))
    P1 = np.array([[700.0, 0, 320, 0], [0, 700, 240, 0], [0, 0, 1, 0]])
    P2 = np.array([[700.0, 0, 320, -84], [0, 700, 240, 0], [0, 0, 1, 0]])
    F, fh, fw = 3, 20, 31
    sx = rng.normal(-12, 4, (F, fh, fw)).astype(np.float32)
    sy = rng.normal(0, 1, (F, fh, fw)).astype(np.float32)
    lev = 1
    got = orc.triangulate_fovea(sx, sy, lev, 0, 0, 1.0, P1, P2)
    xx, yy = np.meshgrid(np.arange(fw, dtype=np.float32), np.arange(fh, dtype=np.float32))
    tx = np.trunc(xx + sx[lev]).astype(np.float32) - xx
    ty = np.trunc(yy + sy[lev]).astype(np.float32) - yy
    exp = orc.triangulate(tx.astype(np.float32), ty.astype(np.float32), P1, P2)
    np.testing.assert_array_equal(got, exp)
    # rectified rig, no vertical disparity: Z = f*B/d with B = 84/700
    got0 = orc.triangulate_fovea(sx, np.zeros_like(sy), lev, 0, 0, 1.0, P1, P2)
    ok = tx < -0.5
    np.testing.assert_allclose(got0[2][ok], (700.0 * 0.12 / -tx)[ok], rtol=1e-2)   # binary32 closed form


def test_reconstruct_full_known_answers(orc):
    """hierarchicalDisparity: constant coarse level c with zero finer foveae -> outside every window the value is
    c * f32(sqrt2)^k after k upsamplings, inside the level-0 window it is the level-0 fovea."""
    W, H, levels, F = 400, 300, 9, 4
    fw, fh, ox, oy, _, _ = orc.fovea_geometry(W, H, levels, F)
    stack = np.zeros((3, F, fh, fw), np.float32)
    stack[:, F - 1] = 1.0
    for k in range(F - 1):
        stack[:, k] = 10.0 + k
    out = orc.reconstruct_full(stack, W, H, levels)
    s = np.float32(1.41421356)
    corner = np.float32(1.0)
    for _ in range(F - 1):
        corner = np.float32(s * corner)
    assert out[0, 0, 0] == corner and out[2, H - 1, W - 1] == corner
    assert (out[:, oy[0]:oy[0] + fh, ox[0]:ox[0] + fw] == 10.0).all()
    # ring of level-1 fovea (value 11) upsampled once, just outside the level-0 window
    assert out[1, oy[0] - 1, ox[0] + fw // 2] == np.float32(s * np.float32(11.0))


# ---- hand-derived cases for the per-pixel kernels: expected values are computed here, operation by
# operation in numpy float32 scalars, from the reference's formulas -- not by calling any restatement ------------------

def _seq5(vals, taps):
    """sum = 0; sum += v[k] * t[k] for k = 0..4, every product and partial sum rounded to float32"""
    f = np.float32
    s = f(0)
    for v, t in zip(vals, taps):
        s = f(s + f(f(v) * f(t)))
    return s


def test_move_correlation_known_answers(orc):
    """MoveCorrelation, MatchLib.cu:681-687: clamp01((N*N) / (A * B)) with N = zero-padded, A and B = clamp-addressed 5x5
    Gaussian sums (rows first, rounded, then columns).  Constant images make every term a product of tap sums."""
    f = np.float32
    g = orc.gauss_taps()
    a, b = f(10), f(20)
    H, W = 9, 12
    L = np.full((3, H, W), a, f)
    R = np.full((3, H, W), b, f)
    d0 = np.zeros((3, H, W), f)
    d0[2] = 0.5
    _, dbg = orc.iterate_level(L, R, d0, 4, 5, False, 1, 1, want_dbg=True)

    def two_pass(v, n_taps_row, n_taps_col):
        # a constant image: the row pass sees `v` under the first n taps present, zero elsewhere (same for columns)
        row = _seq5([v if k in n_taps_row else 0 for k in range(5)], g)
        return _seq5([row if k in n_taps_col else 0 for k in range(5)], g)
    full = range(5)
    # interior pixel: every tap present in all three sums
    N, A, B = two_pass(f(a * b), full, full), two_pass(f(a * a), full, full), two_pass(f(b * b), full, full)
    q = f(f(N * N) / f(A * B))
    q = f(1) if q > 1 else q
    Q = f(f(f(q + q) + q) / f(3))  # the three channels are identical: ((q0 + q1) + q2) / 3.0f
    for s in range(5):
        assert dbg[s, 4, 6] == Q, (s, dbg[s, 4, 6], Q)
    # corner (0,0), shift (0,0): the zero-padded N has only taps 2..4 (pixels 0..2) in both directions, the clamp-addressed
    # A and B still see all five taps (edge replicated)
    Nc = two_pass(f(a * b), range(2, 5), range(2, 5))
    qc = f(f(Nc * Nc) / f(A * B))
    Qc = f(f(f(qc + qc) + qc) / f(3))
    assert dbg[4, 0, 0] == Qc and Qc < Q


def test_true_confidence_blends_old_075_new_025(orc):
    """TrueConfidence (MatchLib.cu:1003-1007) through its binding (calculateTrueConfidence, MatchLib.cu:1016-1038, called at
    MatchGPULib.cpp:2243 with dispy = the new correlation product, a_Src = the previous confidence): 0.75 * OLD + 0.25 * NEW in
    double.  Identical constant images give l = c = r, hence c1 = 0 -> (delta, rho) = (0, 0.4) for x and y, NEW = 0.4 * 0.4."""
    f = np.float32
    L = np.full((3, 8, 10), 7, f)
    d0 = np.zeros((3, 8, 10), f)
    d0[2] = 0.2
    _, dbg = orc.iterate_level(L, L.copy(), d0, 4, 5, False, 1, 1, want_dbg=True)
    new = f(f(0.4) * f(0.4))
    exp = f(0.75 * float(f(0.2)) + 0.25 * float(new))
    swapped = f(0.75 * float(new) + 0.25 * float(f(0.2)))
    assert exp != swapped
    assert (dbg[7] == exp).all() and (dbg[5] == 0).all() and (dbg[6] == 0).all()
    # the coarsest level's first iteration has no old confidence to blend (MatchGPULib.cpp:2223)
    _, dbg = orc.iterate_level(L, L.copy(), d0, 22, 5, True, 1, 1, want_dbg=True)
    assert (dbg[7] == new).all()


def test_smooth_kernel_weights_and_order(orc):
    """smoothKernel, MatchLib.cu:1108-1139: sumDisp = v*w + sumDisp over centre, west, east, north, south, sumCorr likewise,
    east / south clamped at the last column / row, result sumDisp / sumCorr; every plane weighted by the confidence plane."""
    f = np.float32
    v = np.array([[1.5, -2.25, 4.0], [0.5, 3.0, -1.0], [2.0, 8.0, 0.125]], f)
    w = np.array([[0.9, 0.1, 0.4], [0.3, 0.7, 0.2], [0.6, 0.05, 0.8]], f)
    out = orc.smooth_pass(np.stack([v, v * f(2), w]))

    def at(y, x):
        yy, xx = min(y + 1, 2), min(x + 1, 2)
        nb = [(y, x), (y, x - 1), (y, xx), (y - 1, x), (yy, x)]
        sd, sc, sk = f(0), f(0), f(0)
        for (j, i) in nb:
            sd = f(f(v[j, i] * w[j, i]) + sd)
            sk = f(f(w[j, i] * w[j, i]) + sk)
            sc = f(sc + w[j, i])
        return f(sd / sc), f(sk / sc)
    for (y, x) in [(1, 1), (2, 2), (1, 2), (2, 1)]:
        ev, ek = at(y, x)
        assert out[0, y, x] == ev and out[2, y, x] == ek, (y, x)
    assert (out[:, 0, :] == np.stack([v, v * f(2), w])[:, 0, :]).all() and (out[:, :, 0] == np.stack([v, v * f(2), w])[:, :, 0]).all()


def test_foveated_seed_crop_known_answers(orc):
    """foveatedsubsampleDisp, MatchGPULib.cpp:1595-1655: the fovea-sized field is upsampled to the parent level's size
    (Wup x Hup) with subsampleDispKernel -- dst = SCALE * src[floor((i + .5f) * (float)(1/SCALE))] -- and the fovea window at
    origin (l, u) is cut out (:1612-1615, :1642-1644)."""
    f = np.float32
    fw, fh, Wup, Hup = 11, 7, 15, 10
    src = (np.arange(3 * fh * fw, dtype=f).reshape(3, fh, fw) * f(0.37) - f(5)).astype(f)
    l, u = Wup // 2 - fw // 2, Hup // 2 - fh // 2
    out = orc.seed_fovea(src, Wup, Hup, l, u)
    sf = f(1.0 / 1.41421356)
    for (y, x) in [(0, 0), (3, 5), (fh - 1, fw - 1), (2, 9)]:
        sy = min(int(np.floor(f(f(u + y) + f(0.5)) * sf)), fh - 1)
        sx = min(int(np.floor(f(f(l + x) + f(0.5)) * sf)), fw - 1)
        for c in range(3):
            assert out[c, y, x] == f(1.41421356 * float(src[c, sy, sx])), (c, y, x)


def test_weighted_difference_is_the_weighted_mean_absolute_change(orc):
    """Row f-4, weightedDifference (MatchGPULib.cpp:1336-1437): sum(|D - OldD| * conf) / sum(conf) per disparity plane."""
    f = np.float32
    new = np.zeros((3, 4, 5), f)
    old = np.zeros((3, 4, 5), f)
    new[2] = 0.5
    new[0, 1, 2], old[0, 1, 2] = 3.0, 1.0   # |2| * 0.5
    new[1, 3, 4], old[1, 3, 4] = -1.0, 0.5  # |-1.5| * 0.5
    dh, dv = orc.weighted_difference(new, old)
    assert dh == f(1.0 / 10.0) and dv == f(0.75 / 10.0)
    new[2, 0, 0] = 2.0  # weights are the NEW field's confidence
    dh2, _ = orc.weighted_difference(new, old)
    assert dh2 == f(1.0 / 11.5)


def test_lr_check_known_answers(orc):
    """The LR-consistency check has NO reference counterpart: these cases pin the build's own definition
    (DESIGN.md section 8).  A left pixel (x, y) with (dx, dy) looks at the right-to-left field at the pixel the matcher would fetch,
    floor(x + .5 + dx), floor(y + .5 + dy) clamped to the frame, and keeps its confidence only if that field points back within tau
    in x AND in y; NaN sums count as inconsistent; dx, dy are never changed."""
    f = np.float32
    H, W = 8, 12
    left = np.zeros((3, H, W), f)
    right = np.zeros((3, H, W), f)
    left[2] = 0.5
    # (x=2, y=3): dx = +3.4 -> fetches right (5, 3) [floor(2.5 + 3.4) = 5]; the right field there says -3.0: |0.4| <= 0.5 -> kept
    left[0, 3, 2], right[0, 3, 5] = 3.4, -3.0
    # (x=7, y=1): dx = +1.0, dy = +2.0 -> right (8, 3) says (-1.0, -1.2): x fine, |0.8| > 0.5 in y -> confidence 0
    left[0, 1, 7], left[1, 1, 7] = 1.0, 2.0
    right[0, 3, 8], right[1, 3, 8] = -1.0, -1.2
    # (x=11, y=7): dx = +50 -> clamped to the last column, right (11, 7) says 0 -> |50| > tau -> confidence 0
    left[0, 7, 11] = 50.0
    # (x=0, y=0): NaN disparity -> fetch index 0 (tex_index maps NaN to 0), the sum is NaN -> confidence 0
    left[0, 0, 0] = np.nan
    with np.errstate(all="ignore"):
        out, n = orc.lr_check(left, right, 0.5)
    marked = {(y, x) for y in range(H) for x in range(W) if out[2, y, x] == 0}
    # besides the three above, the right field's own nonzero entries make the LEFT zero-disparity pixels at (3, 5) and (3, 8) inconsistent
    assert marked == {(1, 7), (7, 11), (0, 0), (3, 5), (3, 8)} and n == 5
    assert out[2, 3, 2] == f(0.5)
    same = (out[:2].view(np.uint32) == left[:2].view(np.uint32))
    assert same.all()
    # tau = 0 keeps exact round trips only; a large tau keeps everything but the NaN
    assert orc.lr_check(left, right, 0.0)[1] == 6 and orc.lr_check(left, right, 100.0)[1] == 1
