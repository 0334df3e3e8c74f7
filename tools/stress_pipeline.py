#!/usr/bin/env python3
"""Randomised whole-matcher A/B on the GPU: kernel_path 0 (production: marching / tiled / latency kernels chosen per level and per
slot count, fused seeding, strip-height model) against kernel_path 1 (one kernel per reference stage, itself pinned to the oracle by
tests/), full and foveated mode, random image sizes, pyramid depths, slot counts and fovea offsets.  Development tool:
python tools/stress_pipeline.py [cases [seed]]"""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ug_stereomatcher_amd import _lib, synth  # noqa: E402

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = np.random.Generator(np.random.PCG64(int(sys.argv[2]) if len(sys.argv) > 2 else 20260412))  # (second argument: another seed)


def bits_equal(a, b):
    return ((a.view(np.uint32) == b.view(np.uint32)) | (np.isnan(a) & np.isnan(b))).all()


def run(c, L, R, levels, F, off, slot):
    H, W, _ = L.shape
    lib = c.lib
    dL, dR = c.to_device(L), c.to_device(R)
    full = np.empty((3, H, W), np.float32)
    dO = c.alloc(full.nbytes)
    c.check(lib.ugsm_submit_full(c.handle, slot, dL, dR, W, H, W * 3, dO))
    c.check(lib.ugsm_wait(c.handle, slot))
    full = c.to_host(dO, (3, H, W))
    fw, fh = C.c_int(), C.c_int()
    c.check(lib.ugsm_fovea_dims(W, H, levels, F, C.byref(fw), C.byref(fh)))
    dS = c.alloc(3 * F * fh.value * fw.value * 4)
    c.check(lib.ugsm_submit_foveated(c.handle, slot, dL, dR, W, H, W * 3, off[0], off[1], dS, None, None))
    c.check(lib.ugsm_wait(c.handle, slot))
    st = c.to_host(dS, (3, F, fh.value, fw.value))
    for p in (dL, dR, dO, dS):
        c.free(p)
    return full, st


bad = 0
for case in range(n_cases):
    big = case % 6 == 5
    W = int(rng.integers(900, 2600)) if big else int(rng.integers(48, 900))
    H = int(rng.integers(600, 1700)) if big else int(rng.integers(40, 700))
    max_levels = 1
    w, h = W, H
    while max_levels < 14 and int(w / 1.41421356) >= 8 and int(h / 1.41421356) >= 8:
        w, h = int(w / 1.41421356), int(h / 1.41421356)
        max_levels += 1
    levels = int(rng.integers(2, max_levels + 1))
    F = int(rng.integers(2, levels + 1))
    off = (int(rng.integers(-W // 8, W // 8 + 1)), int(rng.integers(-H // 8, H // 8 + 1)))
    slots = int(rng.choice([1, 2, 4]))
    slot = int(rng.integers(0, slots))
    L, R, _, _ = synth.make_pair(W, H, 9000 + case)
    try:
        with _lib.Context(levels=levels, fovea_levels=F, slots=slots, kernel_path=0) as c0:
            a = run(c0, L, R, levels, F, off, slot)
        with _lib.Context(levels=levels, fovea_levels=F, slots=1, kernel_path=1) as c1:
            b = run(c1, L, R, levels, F, off, 0)
    except _lib.UgsmError as e:
        print(f"case {case}: {W}x{H} levels={levels} F={F}: {e}", flush=True)
        continue
    ok = bits_equal(a[0], b[0]) and bits_equal(a[1], b[1])
    if not ok:
        bad += 1
        print(f"MISMATCH case {case}: {W}x{H} levels={levels} F={F} off={off} slots={slots} full={bits_equal(a[0], b[0])} fovea={bits_equal(a[1], b[1])}", flush=True)
    if case % 10 == 9:
        print(f"{case + 1} cases, {bad} mismatches", flush=True)
print("done:", n_cases, "cases,", bad, "mismatches")
sys.exit(1 if bad else 0)
