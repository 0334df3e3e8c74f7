#!/usr/bin/env python3
"""Randomised A/B of the marching kernels against the LDS-tiled ones on the GPU (both through ugsm_stage_iterate /
ugsm_stage_smooth; the tiled path is itself pinned to the oracle by tests/): random sizes, strip heights, pixels per lane,
disparity fields with outliers, zero patches, out-of-range plane values.  Development tool:  python tools/stress_march.py [cases [seed]]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ug_stereomatcher_amd import _lib  # noqa: E402

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 200
rng = np.random.Generator(np.random.PCG64(int(sys.argv[2]) if len(sys.argv) > 2 else 20260410))  # (second argument: another seed)


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


ref = _lib.Context(levels=1, march_min_pixels=-1)
bad = 0
for case in range(n_cases):
    W, H = int(rng.integers(6, 700)), int(rng.integers(5, 300))
    if case % 10 == 0:
        W = int(rng.choice([57, 58, 59, 63, 64, 65, 116, 121, 122, 123, 128, 174, 180]))
    npl, rows = int(rng.integers(1, 3)), int(rng.choice([0, 3, 8, 16, 23, 64]))
    L3 = rng.integers(1, 256, (3, H, W)).astype(np.float32)
    R3 = np.roll(L3, int(rng.integers(-3, 4)), axis=2) + rng.integers(0, 3, (3, H, W)).astype(np.float32)
    kind = case % 5
    if kind == 1:  # zero patches -> 0/0
        y, x = int(rng.integers(0, H - 2)), int(rng.integers(0, W - 2))
        L3[:, y:y + 12, x:x + 20] = 0
        R3[:, max(y - 3, 0):y + 6, x:x + 30] = 0
    if kind == 2:  # values outside the guarded-division range: the pair must take the full division
        L3[rng.integers(0, 3), rng.integers(0, H), :] = rng.choice([1e-6, 2000.0, 1e-40])
    d3 = np.stack([rng.normal(0, 8, (H, W)), rng.normal(0, 4, (H, W)), 0.05 + rng.random((H, W))]).astype(np.float32)
    if kind == 3:  # wild disparities
        idx = rng.integers(0, H * W, 50)
        d3[0].ravel()[idx] = rng.choice([np.nan, np.inf, -np.inf, 1e30, -1e30, 3e9], 50)
    mi = int(rng.choice([4, 6, 22]))
    m0 = int(rng.integers(1, mi))
    m1 = min(mi, m0 + int(rng.integers(0, 2)))
    S, top = int(rng.choice([5, 10])), bool(rng.integers(0, 2))
    with _lib.Context(levels=1, march_min_pixels=1, march_np=npl, march_rows=rows) as c:
        a = iterate(c, L3, R3, d3, mi, S, top, m0, m1)
        sa = smooth(c, d3, 5, case % 2)
    b = iterate(ref, L3, R3, d3, mi, S, top, m0, m1)
    sb = smooth(ref, d3, 5, case % 2)
    ok = bits_equal(a, b) and bits_equal(sa, sb)
    if not ok:
        bad += 1
        print(f"MISMATCH case {case}: {W}x{H} np={npl} rows={rows} kind={kind} mi={mi} m={m0}..{m1} S={S} top={top} "
              f"iterate={bits_equal(a, b)} smooth={bits_equal(sa, sb)}", flush=True)
    if case % 25 == 24:
        print(f"{case + 1} cases, {bad} mismatches", flush=True)
ref.close()
print("done:", n_cases, "cases,", bad, "mismatches")
sys.exit(1 if bad else 0)
