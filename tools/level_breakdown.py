#!/usr/bin/env python3
"""Per (kernel, pyramid level) time of ONE pair in flight (HIP events on every launch): where the single-pair latency goes.
usage: python tools/level_breakdown.py [W H] [pairs]"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch
from ug_stereomatcher_amd import _lib, synth

W, H = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (4928, 3264)
pairs = int(sys.argv[3]) if len(sys.argv) > 3 else 3
L, R, _, _ = synth.make_pair(W, H, 11)
dev = torch.device("cuda:0")
dL, dR = torch.from_numpy(L).to(dev), torch.from_numpy(R).to(dev)
out = torch.empty((3, H, W), dtype=torch.float32, device=dev)
with _lib.Context(levels=14, slots=1, profile_events=0) as c:
    for _ in range(2):
        c.check(c.lib.ugsm_submit_full(c.handle, 0, dL.data_ptr(), dR.data_ptr(), W, H, W * 3, out.data_ptr())); c.check(c.lib.ugsm_wait(c.handle, 0))
    c.reset_kernel_stats(); c.set_profile_events(2)
    t0 = time.perf_counter()
    for _ in range(pairs):
        c.check(c.lib.ugsm_submit_full(c.handle, 0, dL.data_ptr(), dR.data_ptr(), W, H, W * 3, out.data_ptr())); c.check(c.lib.ugsm_wait(c.handle, 0))
    wall = (time.perf_counter() - t0) / pairs * 1e3
    st = c.kernel_stats()
lv = {}
for s in st:
    lv.setdefault(s["level"], []).append(s)
tot = 0.0
print(f"{W}x{H}: wall {wall:.2f} ms/pair with events")
for l in sorted(lv):
    row = sorted(lv[l], key=lambda s: -s["total_ms"])
    ms = sum(s["total_ms"] for s in row) / pairs
    tot += ms
    print(f"level {l:2d}: {ms:7.3f} ms  " + "  ".join(f"{s['name']}:{s['launches'] // pairs}x{s['total_ms'] / s['launches'] * 1e3:.1f}us" for s in row))
print(f"kernel total {tot:.3f} ms/pair")
