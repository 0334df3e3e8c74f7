#!/usr/bin/env python3
"""Per (kernel, pyramid level) time of ONE call in flight (HIP events on every launch): where the latency of a pair -- or of a batch
of pairs marching in lockstep -- goes.
usage: python tools/level_breakdown.py [--size W H] [--calls N] [--batch B] [--fovea F] [--slots S]
(--slots only selects the kernel choices: the calls run one at a time on slot 0)"""
import argparse, os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch
from ug_stereomatcher_amd import _lib, synth

ap = argparse.ArgumentParser()
ap.add_argument("--size", type=int, nargs=2, default=(4928, 3264))
ap.add_argument("--calls", type=int, default=3)
ap.add_argument("--batch", type=int, default=1)
ap.add_argument("--fovea", type=int, default=0)
ap.add_argument("--slots", type=int, default=1)
a = ap.parse_args()
W, H = a.size
B, F = a.batch, a.fovea
L, R, _, _ = synth.make_pair(W, H, 11)
dev = torch.device("cuda:0")
dL, dR = torch.from_numpy(L).to(dev), torch.from_numpy(R).to(dev)
fw, fh = _lib.fovea_dims(W, H, 14, F) if F else (W, H)
outs = [torch.empty((3, F, fh, fw) if F else (3, H, W), dtype=torch.float32, device=dev) for _ in range(B)]
with _lib.Context(levels=14, slots=a.slots, fovea_levels=F or 7, profile_events=0, batch=B) as c:
    def call():
        if B > 1:
            if F:
                c.submit_foveated_batch(0, [dL.data_ptr()] * B, [dR.data_ptr()] * B, W, H, W * 3, None, [o.data_ptr() for o in outs])
            else:
                c.submit_full_batch(0, [dL.data_ptr()] * B, [dR.data_ptr()] * B, W, H, W * 3, [o.data_ptr() for o in outs])
        elif F:
            c.check(c.lib.ugsm_submit_foveated(c.handle, 0, dL.data_ptr(), dR.data_ptr(), W, H, W * 3, 0, 0, outs[0].data_ptr(), None, None))
        else:
            c.check(c.lib.ugsm_submit_full(c.handle, 0, dL.data_ptr(), dR.data_ptr(), W, H, W * 3, outs[0].data_ptr()))
        c.check(c.lib.ugsm_wait(c.handle, 0))
    for _ in range(2):
        call()
    t0 = time.perf_counter()
    for _ in range(a.calls):
        call()
    wall0 = (time.perf_counter() - t0) / a.calls * 1e3
    c.reset_kernel_stats(); c.set_profile_events(2)
    t0 = time.perf_counter()
    for _ in range(a.calls):
        call()
    wall = (time.perf_counter() - t0) / a.calls * 1e3
    st = c.kernel_stats()
lv = {}
for s in st:
    lv.setdefault(s["level"], []).append(s)
tot = 0.0
n = a.calls
print(f"{W}x{H}{' foveated' if F else ''}, batch {B}, kernel choices of a {a.slots}-slot context: wall {wall0:.2f} ms per call without events "
      f"({wall0 / B:.3f} ms per pair), {wall:.2f} ms with")
by_kernel = {}
for l in sorted(lv):
    row = sorted(lv[l], key=lambda s: -s["total_ms"])
    ms = sum(s["total_ms"] for s in row) / n
    tot += ms
    for s in row:
        by_kernel[s["name"]] = by_kernel.get(s["name"], 0.0) + s["total_ms"] / n
    print(f"level {l:2d}: {ms:7.3f} ms  " + "  ".join(f"{s['name']}:{s['launches'] // n}x{s['total_ms'] / s['launches'] * 1e3:.1f}us" for s in row))
print(f"kernel total {tot:.3f} ms per call = {tot / B:.3f} ms per pair")
print("by kernel (ms per call): " + "  ".join(f"{k}:{v:.3f}" for k, v in sorted(by_kernel.items(), key=lambda kv: -kv[1])))
