#!/usr/bin/env python3
"""Randomised A/B of the batch dimension on the GPU: ugsm_submit_full_batch / ugsm_submit_foveated_batch (B pairs marching through the
levels in lockstep, one launch per level for all of them) against the same pairs through the single-pair calls of the same context --
random image sizes, pyramid depths, batch sizes, slot counts, batch thresholds, kernel-choice overrides, distinct images and distinct
fovea offsets inside a batch, pyramid stacks on and off.  (The single-pair calls are pinned to the oracle by tests/.)  Development tool:
python tools/stress_batch.py [cases [seed]]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["UGSM_DEV"] = "1"
from ug_stereomatcher_amd import _lib, synth  # noqa: E402

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
rng = np.random.Generator(np.random.PCG64(int(sys.argv[2]) if len(sys.argv) > 2 else 20261004))  # (second argument: another seed)
KNOBS = [{}, {}, {}, {"UGSM_ALONE": "0"}, {"UGSM_ALONE": "1"}, {"UGSM_MARCH_MIN_PIXELS": "1"}, {"UGSM_MARCH4": "1,2000000000"},
         {"UGSM_FUSE_SEED": "0"}, {"UGSM_BATCH_MAX_PIXELS": "60000"}, {"UGSM_BATCH_MAX_PIXELS": "100000000"}, {"UGSM_PYR_STREAM": "0"},
         {"UGSM_SMALL_MAX_PIXELS": "-1"}]


def bits_equal(a, b):
    return ((a.view(np.uint32) == b.view(np.uint32)) | (np.isnan(a) & np.isnan(b))).all()


bad = 0
for case in range(n_cases):
    big = case % 7 == 6
    W = int(rng.integers(900, 2400)) if big else int(rng.integers(48, 900))
    H = int(rng.integers(600, 1500)) if big else int(rng.integers(40, 700))
    max_levels, w, h = 1, W, H
    while max_levels < 14 and int(w / 1.41421356) >= 8 and int(h / 1.41421356) >= 8:
        w, h, max_levels = int(w / 1.41421356), int(h / 1.41421356), max_levels + 1
    levels = int(rng.integers(2, max_levels + 1))
    F = int(rng.integers(2, levels + 1))
    B = int(rng.choice([2, 3, 4, 5, 8, 16] if not big else [2, 3, 4]))
    slots = int(rng.choice([1, 2, 4]))
    slot = int(rng.integers(0, slots))
    knobs = KNOBS[int(rng.integers(0, len(KNOBS)))]
    offs = [(int(rng.integers(-W // 6, W // 6 + 1)), int(rng.integers(-H // 6, H // 6 + 1))) for _ in range(B)]
    want_pyr = bool(rng.integers(0, 2))
    n_img = min(B, 3)
    imgs = [synth.make_pair(W, H, 12000 + 17 * case + j)[:2] for j in range(n_img)]
    for k, v in knobs.items():
        os.environ[k] = v
    try:
        with _lib.Context(levels=levels, fovea_levels=F, slots=slots, batch=int(rng.choice([0, B]))) as c:
            fw, fh = _lib.fovea_dims(W, H, levels, F)
            dL = [c.to_device(L) for L, _ in imgs]
            dR = [c.to_device(R) for _, R in imgs]
            sel = [b % n_img for b in range(B)]
            nf, ns = 3 * W * H * 4, 3 * F * fh * fw * 4
            dO = [c.alloc(nf) for _ in range(B + n_img)]
            dS = [c.alloc(ns) for _ in range(2 * B)]
            dP = [c.alloc(ns) for _ in range(4 * B)] if want_pyr else None
            # single-pair calls
            for j in range(n_img):
                c.check(c.lib.ugsm_submit_full(c.handle, slot, dL[j], dR[j], W, H, 3 * W, dO[B + j]))
                c.check(c.lib.ugsm_wait(c.handle, slot))
            for b in range(B):
                c.check(c.lib.ugsm_submit_foveated(c.handle, slot, dL[sel[b]], dR[sel[b]], W, H, 3 * W, offs[b][0], offs[b][1], dS[B + b],
                                                   dP[2 * B + b] if want_pyr else None, dP[3 * B + b] if want_pyr else None))
                c.check(c.lib.ugsm_wait(c.handle, slot))
            # the batch calls
            c.submit_full_batch(slot, [dL[k] for k in sel], [dR[k] for k in sel], W, H, 3 * W, dO[:B])
            c.submit_foveated_batch(slot, [dL[k] for k in sel], [dR[k] for k in sel], W, H, 3 * W, offs, dS[:B],
                                    dP[:B] if want_pyr else None, dP[B:2 * B] if want_pyr else None)
            c.check(c.lib.ugsm_wait(c.handle, slot))
            ok = True
            for b in range(B):
                ok = ok and bits_equal(c.to_host(dO[b], (3, H, W)), c.to_host(dO[B + sel[b]], (3, H, W)))
                ok = ok and bits_equal(c.to_host(dS[b], (3, F, fh, fw)), c.to_host(dS[B + b], (3, F, fh, fw)))
                if want_pyr:
                    ok = ok and bits_equal(c.to_host(dP[b], (F, 3, fh, fw)), c.to_host(dP[2 * B + b], (F, 3, fh, fw)))
                    ok = ok and bits_equal(c.to_host(dP[B + b], (F, 3, fh, fw)), c.to_host(dP[3 * B + b], (F, 3, fh, fw)))
            for p in dL + dR + dO + dS + (dP or []):
                c.free(p)
    except _lib.UgsmError as e:
        print(f"case {case}: {W}x{H} levels={levels} F={F} B={B}: {e}", flush=True)
        ok = True
    finally:
        for k in knobs:
            os.environ.pop(k, None)
    if not ok:
        bad += 1
        print(f"MISMATCH case {case}: {W}x{H} levels={levels} F={F} B={B} slots={slots} offs={offs} knobs={knobs} pyr={want_pyr}", flush=True)
    if case % 10 == 9:
        print(f"{case + 1} cases, {bad} mismatches", flush=True)
print("done:", n_cases, "cases,", bad, "mismatches")
sys.exit(1 if bad else 0)
