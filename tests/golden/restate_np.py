#!/usr/bin/env python3
"""A SECOND, independent restatement of the hot path in vectorised numpy -- test infrastructure, build container only.

Purpose ("reduce common-mode risk"): oracle/ugsm_oracle.c and the HIP kernels were written from the same
reading of the reference; the reference holds no fixtures, so a shared misreading would be invisible.  This file restates
the algorithm again by a different route -- whole-array numpy operations in float32 with explicit float64 promotions,
written from /root/reference/src/gpu_matcher/{MatchLib.cu, MatchGPULib.cpp} (line numbers cited per function) and
SURVEY.md Appendix A, without looking at the C oracle's loops -- and tests/test_oracle_np.py checks the C oracle against
it BIT FOR BIT on the committed fixtures.  Two restatements agreeing does not pin the reference's binary (nothing can,
here), but a slip of the pen in either one now shows.

numpy never contracts a*b+c, every ufunc on float32 arrays rounds to float32, and float32 / float32 is the correctly
rounded quotient: the same float contract as oracle/ (DESIGN.md section 3).
"""
from __future__ import annotations

import numpy as np

F32 = np.float32
F64 = np.float64
SCALE = 1.41421356  # MatchLib_common.h:15 (a double literal)


# ---- constants, geometry, schedules ---------------------------------------------------------------------------------

def gauss_taps():
    """MatchGPULib.cpp:761-774: five float literals, then divided by their float sum accumulated left to right."""
    k = np.array([0.0816475, 0.218507, 0.303281, 0.218507, 0.0816475], F32)
    s = F32(0)
    for v in k:
        s = F32(s + v)
    return (k / s).astype(F32)


BOX = np.array([0.0, 0.3333, 0.3333, 0.3333, 0.0], F32)  # MatchGPULib.cpp:344-348


def level_dims(W, H, levels):
    """MatchGPULib.cpp:1224-1228: w[i+1] = (int)(w[i] / SCALE) in double."""
    w, h = [W], [H]
    for _ in range(levels - 1):
        w.append(int(w[-1] / SCALE))
        h.append(int(h[-1] / SCALE))
    return w, h


def iterations(i):  # MatchGPULib.cpp:1741 with level = 13 - i
    return 22 if i > 5 else 2 * (i + 1)


def smooth_passes(i):  # MatchGPULib.cpp:2257-2261
    return 10 if i < 2 else 5


def thresholds(mi):
    """MatchGPULib.cpp:1673 + 2299-2306: the clamp used by iteration m (1-based) is the value left by iteration m-1."""
    out, thr = [], F32(1.0)
    for m in range(1, mi + 1):
        out.append(thr)
        if m % 2 == 0:
            h = mi // 2 - m // 2
            with np.errstate(all="ignore"):  # mi = 2: 0.9 / 0 -- the value is computed after its last use
                thr = F32(F64(h - 1) * (F64(1 - 0.1) / F64(mi // 2 - 1.0)) + 0.1) if h < 7 else F32(1.0)
    return out


# ---- texture fetch and convolutions -----------------------------------------------------------------------------------

def tex_idx(coord, n):
    """Default texture reference (MatchLib.cu:56-60): unnormalised, point sampled, clamped -> clamp(floor(coord)).
    NaN -> 0 by the build's definition."""
    f = np.floor(coord.astype(F32))
    f = np.where(np.isnan(f), F32(0), f)
    return np.clip(f, 0, n - 1).astype(np.int64)


def conv1d(img, taps, axis, mode):
    """sum = 0; for k = -2..2: sum += src[pos + k] * taps[2 - k]  (MatchLib.cu:127-134 zero padded smem version,
    :1484-1487 / :1616-1619 clamp-addressed texture versions).  Each product and each partial sum rounds to float32."""
    a = np.moveaxis(img.astype(F32), axis, -1)
    n = a.shape[-1]
    if mode == "zero":
        p = np.concatenate([np.zeros(a.shape[:-1] + (2,), F32), a, np.zeros(a.shape[:-1] + (2,), F32)], axis=-1)
    else:
        p = np.concatenate([a[..., :1], a[..., :1], a, a[..., -1:], a[..., -1:]], axis=-1)
    s = np.zeros_like(a)
    for k in range(-2, 3):
        s = (s + p[..., 2 + k:2 + k + n] * taps[2 - k]).astype(F32)
    return np.moveaxis(s, -1, axis)


def blur(img, taps, mode):
    """rows, stored to float, then columns (MatchGPULib.cpp:912-920 / 1866-1896 / 2361-2412)."""
    return conv1d(conv1d(img, taps, -1, mode), taps, -2, mode)


# ---- pyramid (MatchGPULib.cpp:1033-1125; subsampleKernel MatchLib.cu:311-339) ---------------------------------------------

def planes(rgb):
    """MatchGPULib.cpp:332-338"""
    return np.ascontiguousarray(rgb.transpose(2, 0, 1)).astype(F32)


def subsample(src, W2, H2, sf):
    sy = tex_idx((np.arange(H2, dtype=F32) + F32(0.5)) * F32(sf), src.shape[-2])
    sx = tex_idx((np.arange(W2, dtype=F32) + F32(0.5)) * F32(sf), src.shape[-1])
    return src[..., sy[:, None], sx[None, :]]


def pyramid(p0, levels):
    g = gauss_taps()
    w, h = level_dims(p0.shape[2], p0.shape[1], levels)
    lv = [p0] + [None] * (levels - 1)
    for i in range(levels):
        if i == 0 and levels > 1:
            lv[1] = subsample(blur(lv[0], g, "zero"), w[1], h[1], F32(SCALE))  # :1082-1087, sf = (float)SCALE
        if i + 2 < levels:
            lv[i + 2] = subsample(blur(lv[i], g, "zero"), w[i + 2], h[i + 2], F32(2.0))  # :1088-1096
    return lv


# ---- one level (matchlevel, MatchGPULib.cpp:1662-2489) -----------------------------------------------------------------

def shift_clamped(img, sx, sy):
    """img[clamp(y + sy), clamp(x + sx)] (CompareMove / MoveCorrelation fetch at x+thresholdx, y+thresholdy)"""
    H, W = img.shape
    yy = np.clip(np.arange(H) + sy, 0, H - 1)
    xx = np.clip(np.arange(W) + sx, 0, W - 1)
    return img[yy[:, None], xx[None, :]]


def clamp01(v):
    """if (v > 1) v = 1; if (v < 0) v = 0;  -- a NaN fails both tests and stays"""
    v = np.where(v > 1, F32(1), v)
    return np.where(v < 0, F32(0), v).astype(F32)


def poly(c, l, r, thr):
    """PolyDisparity, MatchLib.cu:805-836; the literals 0.5, 0.0, 1.0, 1e-10, 0.3, 0.7 are doubles."""
    b1 = ((r - l) / F32(2)).astype(F32)
    c1 = (r - (c + b1)).astype(F32)
    neg = c1 < 0
    with np.errstate(all="ignore"):
        dh = ((-b1).astype(F64) * 0.5 / c1.astype(F64)).astype(F32)
        dh = np.minimum(F64(thr), np.maximum(dh.astype(F64), 0.0 - F64(thr))).astype(F32)
        cstar = (((c1 * dh).astype(F32) + b1).astype(F32) * dh).astype(F32) + c
        cstar = cstar.astype(F32)
        over = cstar.astype(F64) > 1.0
        d = (cstar - c).astype(F32)
        resc = (dh.astype(F64) * ((1.0 - c.astype(F64)) / d.astype(F64))).astype(F32)
        dh_over = np.where(d.astype(F64) > 1e-10, resc, dh)
        corr_in = (0.3 * cstar.astype(F64) + 0.7).astype(F32)
    delta = np.where(neg, np.where(over, dh_over, dh), F32(0)).astype(F32)
    corr = np.where(neg, np.where(over, F32(1), corr_in), F32(0.4)).astype(F32)
    return delta, corr


def smooth_pass(d):
    """smoothKernel, MatchLib.cu:1092-1145: pixels with ix > 0 and iy > 0 only; neighbours x-1, clamp(x+1), y-1, clamp(y+1);
    sumDisp = v*w + sumDisp in the order centre, west, east, north, south; all three planes weighted by the PRE-pass
    confidence (MatchGPULib.cpp:2264-2289)."""
    w = d[2]
    H, W = w.shape

    def nb(a, sx, sy):
        return shift_clamped(a, sx, sy)
    ws = [w, nb(w, -1, 0), nb(w, 1, 0), nb(w, 0, -1), nb(w, 0, 1)]
    sc = np.zeros_like(w)
    for x in ws:
        sc = (sc + x).astype(F32)
    out = []
    with np.errstate(all="ignore"):
        for v in d:
            vs = [v, nb(v, -1, 0), nb(v, 1, 0), nb(v, 0, -1), nb(v, 0, 1)]
            sd = np.zeros_like(v)
            for a, b in zip(vs, ws):
                sd = ((a * b).astype(F32) + sd).astype(F32)
            q = (sd / sc).astype(F32)
            q[0, :] = v[0, :]
            q[:, 0] = v[:, 0]
            out.append(q)
    return np.stack(out)


def iterate_level(L, R, d, i, is_top, m_from=1, m_to=None):
    g = gauss_taps()
    mi, S = iterations(i), smooth_passes(i)
    thr = thresholds(mi)
    _, H, W = L.shape
    xs = np.arange(W, dtype=F32) + F32(0.5)
    ys = np.arange(H, dtype=F32) + F32(0.5)
    moves = [(-1, 0), (1, 0), (0, -1), (0, 1), (0, 0)]  # MatchGPULib.cpp:1677
    d = d.astype(F32).copy()
    for m in range(m_from, (m_to or mi) + 1):
        dx, dy, cf = d
        sx = tex_idx(xs[None, :] + dx, W)  # warpAbyB, MatchLib.cu:510-515
        sy = tex_idx(ys[:, None] + dy, H)
        Q = [None] * 5
        for k in range(3):
            Rw = R[k][sy, sx]
            A = blur((L[k] * L[k]).astype(F32), g, "clamp")
            B = blur((Rw * Rw).astype(F32), g, "clamp")
            for s, (mx, my) in enumerate(moves):
                P = (L[k] * shift_clamped(Rw, mx, my)).astype(F32)  # CompareMove, MatchLib.cu:622-624
                N = blur(P, g, "zero")
                with np.errstate(all="ignore"):
                    q = clamp01(((N * N).astype(F32) / (A * shift_clamped(B, mx, my)).astype(F32)).astype(F32))  # MatchLib.cu:681-687
                if k == 0:
                    Q[s] = q
                elif k == 1:
                    Q[s] = (q + Q[s]).astype(F32)  # Disparity kernel: a + b (MatchLib.cu:951-953)
                else:
                    Q[s] = ((Q[s] + q).astype(F32) / F32(3.0)).astype(F32)  # floatrescale (a + b) / m, m = 3.0f
        ddx, cx = poly(Q[4], Q[0], Q[1], thr[m - 1])
        ddy, cy = poly(Q[4], Q[2], Q[3], thr[m - 1])
        kap = (cy * cx).astype(F32)  # compCorrelation
        if not (is_top and m == 1):
            kap = clamp01((0.75 * cf.astype(F64) + 0.25 * kap.astype(F64)).astype(F32))  # TrueConfidence, old = texSrc
        nd = np.stack([(dx + ddx).astype(F32), (dy + ddy).astype(F32), kap])
        for _ in range(S):
            nd = smooth_pass(nd)
        d = np.stack([blur(p, BOX, "clamp") for p in nd])
    return d


def seed(src, W2, H2):
    """subsampleDispKernel, MatchLib.cu:372-401: dst = SCALE * tex(src, x*sf, y*sf), sf = (float)(1/SCALE); the product is
    taken in double (SCALE is a double macro) and stored to float."""
    sf = F32(1.0 / SCALE)
    return (SCALE * subsample(src, W2, H2, sf).astype(F64)).astype(F32)


# ---- drivers (matching, MatchGPULib.cpp:1196-1318) ------------------------------------------------------------------------

def match_full(rgbL, rgbR, levels):
    pl, pr = pyramid(planes(rgbL), levels), pyramid(planes(rgbR), levels)
    cur = np.zeros_like(pl[levels - 1])
    for i in range(levels - 1, -1, -1):
        cur = iterate_level(pl[i], pr[i], cur, i, i == levels - 1)
        if i > 0:
            cur = seed(cur, pl[i - 1].shape[2], pl[i - 1].shape[1])
    return cur


def match_foveated(rgbL, rgbR, levels, Fv):
    """matchStack: levels >= Fv-1 whole, levels < Fv-1 cropped to the centred fovea of level Fv-1's size
    (CreateFoveatedPyramid :1143-1176); seeds for the cropped levels are upsampled to level Fv-2's size, then centre-cropped
    (foveatedsubsampleDisp :1595-1655).  Returns the stack [3][Fv][fovH][fovW] as the node packs it."""
    pl, pr = pyramid(planes(rgbL), levels), pyramid(planes(rgbR), levels)
    w, h = level_dims(rgbL.shape[1], rgbL.shape[0], levels)
    fw, fh = w[Fv - 1], h[Fv - 1]

    def crop(a, W_, H_):
        l, u = W_ // 2 - fw // 2, H_ // 2 - fh // 2
        return np.ascontiguousarray(a[:, u:u + fh, l:l + fw])
    for i in range(Fv - 1):
        pl[i], pr[i] = crop(pl[i], w[i], h[i]), crop(pr[i], w[i], h[i])
    stack = np.zeros((3, Fv, fh, fw), F32)
    cur = np.zeros_like(pl[levels - 1])
    for i in range(levels - 1, -1, -1):
        cur = iterate_level(pl[i], pr[i], cur, i, i == levels - 1)
        if i < Fv:
            stack[:, i] = cur
        if i > 0:
            if i >= Fv:
                cur = seed(cur, w[i - 1], h[i - 1])
            else:
                cur = crop(seed(cur, w[Fv - 2], h[Fv - 2]), w[Fv - 2], h[Fv - 2])
    return stack
