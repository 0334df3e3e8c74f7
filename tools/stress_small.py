#!/usr/bin/env python3
"""Randomised A/B on the GPU of (1) the coarse levels' latency kernels (k_cost_small, k_smooth_small at every tile height) against the
LDS-tiled kernels, through ugsm_stage_iterate / ugsm_stage_smooth, and (2) the seeding fused into the first marching K-cost launch
against the separate k_seed launch, through whole full-mode and foveated matches with every level marching.  The tiled path is pinned
to the oracle by tests/.  Development tool:  python tools/stress_small.py [cases [seed]]"""
import ctypes as C
import os
os.environ["UGSM_DEV"] = "1"  # the kernel-choice overrides used below are development switches
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ug_stereomatcher_amd import _lib, synth  # noqa: E402

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 200
rng = np.random.Generator(np.random.PCG64(int(sys.argv[2]) if len(sys.argv) > 2 else 20260411))  # (second argument: another seed)


def bits_equal(a, b):
    return ((a.view(np.uint32) == b.view(np.uint32)) | (np.isnan(a) & np.isnan(b))).all()


def iterate(c, L3, R3, d3, mi, S, top, m0, m1):
    _, H, W = L3.shape
    pL, pR, pd = c.to_device(L3), c.to_device(R3), c.to_device(d3)
    try:
        c.check(c.lib.ugsm_stage_iterate(c.handle, pL, pR, pd, W, H, mi, S, int(top), m0, m1, None))
        return c.to_host(pd, (3, H, W))
    finally:
        for p in (pL, pR, pd):
            c.free(p)


def smooth(c, d3, passes, box):
    _, H, W = d3.shape
    p = c.to_device(d3)
    try:
        c.check(c.lib.ugsm_stage_smooth(c.handle, p, W, H, passes, box))
        return c.to_host(p, d3.shape)
    finally:
        c.free(p)


def match_both(c, L, R, levels, F, off):
    H, W, _ = L.shape
    full = np.empty((3, H, W), np.float32)
    c.check(c.lib.ugsm_match_full(c.handle, L.ctypes.data, R.ctypes.data, W, H, W * 3, full[0].ctypes.data, full[1].ctypes.data, full[2].ctypes.data))
    fw, fh = C.c_int(), C.c_int()
    c.check(c.lib.ugsm_fovea_dims(W, H, levels, F, C.byref(fw), C.byref(fh)))
    st = np.empty((3, F, fh.value, fw.value), np.float32)
    c.check(c.lib.ugsm_match_foveated(c.handle, L.ctypes.data, R.ctypes.data, W, H, W * 3, off[0], off[1], st[0].ctypes.data, st[1].ctypes.data,
                                      st[2].ctypes.data, None, None))
    return full, st


tiled = _lib.Context(levels=1, march_min_pixels=-1, small_max_pixels=-1)
bad = 0
for case in range(n_cases):
    if case % 8 == 7:  # ---- fused against separate seeding, whole matcher, every level marching
        W, H = int(rng.integers(120, 900)), int(rng.integers(90, 600))
        levels = int(rng.integers(3, 9))
        F = int(rng.integers(2, levels + 1))
        off = (int(rng.integers(-40, 41)), int(rng.integers(-30, 31)))
        L, R, _, _ = synth.make_pair(W, H, 7000 + case)
        os.environ["UGSM_MARCH_MIN_PIXELS"] = "1"
        out = []
        for fuse in ("1", "0"):
            os.environ["UGSM_FUSE_SEED"] = fuse
            with _lib.Context(levels=levels, fovea_levels=F, slots=int(rng.integers(1, 3))) as c:
                out.append(match_both(c, L, R, levels, F, off))
        del os.environ["UGSM_MARCH_MIN_PIXELS"], os.environ["UGSM_FUSE_SEED"]
        ok = bits_equal(out[0][0], out[1][0]) and bits_equal(out[0][1], out[1][1])
        what = f"seeding {W}x{H} levels={levels} F={F} off={off}"
    else:  # ---- latency kernels against tiled kernels, one stage
        W, H = int(rng.integers(1, 420)), int(rng.integers(1, 320))
        if case % 5 == 0:
            W, H = int(rng.choice([15, 16, 17, 18, 19, 31, 32, 33, 35, 36, 37, 54])), int(rng.choice([3, 4, 5, 10, 11, 12, 13, 18, 19, 24, 36]))
        rh = int(rng.choice([18, 24, 32]))
        L3 = rng.integers(1, 256, (3, H, W)).astype(np.float32)
        R3 = np.roll(L3, int(rng.integers(-3, 4)), axis=2) + rng.integers(0, 3, (3, H, W)).astype(np.float32)
        kind = case % 6
        if kind == 1 and H > 3 and W > 3:  # zero patches -> 0/0
            y, x = int(rng.integers(0, H - 2)), int(rng.integers(0, W - 2))
            L3[:, y:y + 12, x:x + 20] = 0
            R3[:, max(y - 3, 0):y + 6, x:x + 30] = 0
        if kind == 2:
            L3[rng.integers(0, 3), rng.integers(0, H), :] = rng.choice([1e-6, 2000.0, 1e-40, -3.0])
        d3 = np.stack([rng.normal(0, 8, (H, W)), rng.normal(0, 4, (H, W)), 0.05 + rng.random((H, W))]).astype(np.float32)
        if kind == 3:  # wild disparities
            idx = rng.integers(0, H * W, 20)
            d3[0].ravel()[idx] = rng.choice([np.nan, np.inf, -np.inf, 1e30, -1e30, 3e9], 20)
        if kind == 4:  # degenerate confidences
            idx = rng.integers(0, H * W, 30)
            d3[2].ravel()[idx] = rng.choice([0.0, -0.25, 1e-30, 1e30], 30)
        mi = int(rng.choice([4, 6, 22]))
        m0 = int(rng.integers(1, mi))
        m1 = min(mi, m0 + int(rng.integers(0, 2)))
        S, top = int(rng.choice([0, 3, 5, 7, 10])), bool(rng.integers(0, 2))
        passes, box = int(rng.integers(0, 11)), int(rng.integers(0, 2))
        if passes == 0:
            box = 1
        os.environ["UGSM_SMALL_RH"] = str(rh)
        with _lib.Context(levels=1, march_min_pixels=-1, small_max_pixels=10**9, slots=int(rng.integers(1, 3))) as c:
            with np.errstate(all="ignore"):
                a = iterate(c, L3, R3, d3, mi, S, top, m0, m1)
                sa = smooth(c, d3, passes, box)
        del os.environ["UGSM_SMALL_RH"]
        b = iterate(tiled, L3, R3, d3, mi, S, top, m0, m1)
        sb = smooth(tiled, d3, passes, box)
        ok = bits_equal(a, b) and bits_equal(sa, sb)
        what = f"{W}x{H} rh={rh} kind={kind} mi={mi} m={m0}..{m1} S={S} top={top} passes={passes} box={box} iterate={bits_equal(a, b)} smooth={bits_equal(sa, sb)}"
    if not ok:
        bad += 1
        print(f"MISMATCH case {case}: {what}", flush=True)
    if case % 25 == 24:
        print(f"{case + 1} cases, {bad} mismatches", flush=True)
tiled.close()
print("done:", n_cases, "cases,", bad, "mismatches")
sys.exit(1 if bad else 0)
