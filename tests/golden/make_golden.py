#!/usr/bin/env python3
"""Generates the committed golden fixtures under tests/golden/.

Run from the repo root in the build container:  python tests/golden/make_golden.py

What pins what:
  * blur_gold.npz  -- expected outputs come from the REFERENCE's own
    convolutionRowCPU / convolutionColumnCPU (oracle/_ref/libgold.so, compiled by
    oracle/Makefile from /root/reference/src/gpu_matcher/convolutionSeparable_gold.cpp where
    it lies).  This is the only executable piece of the reference in this image.
  * every other file -- inputs from ug_stereomatcher_amd.synth, expected outputs from the
    CPU restatement (oracle/).  The reference has no tests or golden data for these
    ("parity unpinned"): the fixtures freeze the restatement so that an accidental change
    of either the oracle or the HIP kernels shows up.
Only data (inputs / expected outputs) is stored; no reference source text.
"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import oracle as orc  # noqa: E402
from ug_stereomatcher_amd import synth  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))
f32p = C.POINTER(C.c_float)


def fp(a):
    return a.ctypes.data_as(f32p)


def blur_gold():
    gold = orc.gold_lib()
    if gold is None:
        print("oracle/_ref/libgold.so missing: keeping existing blur_gold.npz")
        return
    rng = np.random.Generator(np.random.PCG64(7))
    taps = orc.gauss_taps()
    cases = {}
    for name, (H, W) in {"a": (29, 37), "b": (5, 3), "c": (1, 9), "d": (64, 130)}.items():
        src = (rng.random((H, W), dtype=np.float32) * 255).astype(np.float32)
        row = np.empty_like(src)
        col = np.empty_like(src)
        gold.convolutionRowCPU(fp(row), fp(src), fp(taps), W, H, 2)
        gold.convolutionColumnCPU(fp(col), fp(row), fp(taps), W, H, 2)
        cases[f"{name}_src"], cases[f"{name}_row"], cases[f"{name}_col"] = src, row, col
    ones = np.ones((1, 8), np.float32)
    r = np.empty_like(ones)
    gold.convolutionRowCPU(fp(r), fp(ones), fp(taps), 8, 1, 2)
    cases["ones_row"] = r
    cases["taps"] = taps
    np.savez_compressed(os.path.join(OUT, "blur_gold.npz"), **cases)
    print("blur_gold: ones row ->", r[0, :3])


def full(W, H, levels, seed):
    L, R, dx, dy = synth.make_pair(W, H, seed)
    out = orc.match_full(L, R, levels)
    np.savez_compressed(os.path.join(OUT, f"full_{W}x{H}_l{levels}.npz"), L=L, R=R, out=out, levels=levels)
    m = 12
    print(f"full {W}x{H} l{levels}: median |dx-truth| {np.median(np.abs(out[0] - dx)[m:-m, m:-m]):.3f}")


def stage(W, H, seed):
    L, R, dx, dy = synth.make_pair(W, H, seed)
    pl = orc.rgb_to_planes(L)
    pr = orc.rgb_to_planes(R)
    pyr = orc.pyramid(pl, 4)
    rng = np.random.Generator(np.random.PCG64(seed))
    # a smooth, non-trivial seed field: truth + low-amplitude noise, conf in (0.5, 1)
    d0 = np.stack([dx + 0.3 * rng.standard_normal(dx.shape).astype(np.float32),
                   dy + 0.3 * rng.standard_normal(dx.shape).astype(np.float32),
                   0.5 + 0.5 * rng.random(dx.shape, dtype=np.float32)]).astype(np.float32)
    d1, dbg = orc.iterate_level(pl, pr, d0, mi=4, S=5, is_top=False, m_from=1, m_to=1, want_dbg=True)
    d3, _ = orc.iterate_level(pl, pr, d0, mi=4, S=5, is_top=False, m_from=1, m_to=3)
    dtop, _ = orc.iterate_level(pl, pr, np.zeros_like(d0), mi=22, S=10, is_top=True, m_from=1, m_to=2)
    sm = orc.smooth_pass(d0)
    bx = orc.box3(d0)
    w, h = orc.level_dims(W, H, 2)
    sd = orc.seed(d0, int(W * 1.41421356) + 1, int(H * 1.41421356) + 1)
    np.savez_compressed(os.path.join(OUT, f"stage_{W}x{H}.npz"), L=L, R=R, pyr1=pyr[1], pyr2=pyr[2], pyr3=pyr[3],
                        d0=d0, d1=d1, dbg=dbg, d3=d3, dtop=dtop, smooth1=sm, box=bx, seed=sd)


def fovea(W, H, levels, F, seed):
    L, R, _, _ = synth.make_pair(W, H, seed)
    st, pl, pr = orc.match_foveated(L, R, levels, F, 0, 0, want_pyr=True)
    st2, _, _ = orc.match_foveated(L, R, levels, F, 24, -14)
    np.savez_compressed(os.path.join(OUT, f"fovea_{W}x{H}_l{levels}_f{F}.npz"), L=L, R=R, stack=st, pyrL=pl, pyrR=pr,
                        stack_off=st2, off=np.array([24, -14]), levels=levels, F=F)


if __name__ == "__main__":
    blur_gold()
    full(64, 48, 5, synth.BASE_SEED + 100)
    full(160, 120, 8, synth.BASE_SEED + 101)
    stage(96, 72, synth.BASE_SEED + 102)
    fovea(320, 240, 9, 4, synth.BASE_SEED + 103)
    for f in sorted(os.listdir(OUT)):
        if f.endswith(".npz"):
            print(f, os.path.getsize(os.path.join(OUT, f)))
