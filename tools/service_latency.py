#!/usr/bin/env python3
"""PCIe-inclusive latency of the service path (ugsm_match_full: pageable host buffers in and out),
the number DESIGN.md quotes beside bench.py's HBM-resident throughput.  Development tool."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ug_stereomatcher_amd import MatchGPULib, synth  # noqa: E402

for (W, H) in [(4928, 3264), (1920, 1080)]:
    L, R, _, _ = synth.make_pair(W, H, synth.BASE_SEED + 2)
    m = MatchGPULib()
    m.match(L, R, 0)  # first call allocates the context buffers
    ts = []
    for _ in range(5):
        t0 = time.perf_counter()
        m.match(L, R, 0)
        ts.append(time.perf_counter() - t0)
    print(f"{W}x{H} full mode, host->device->host: median {sorted(ts)[2] * 1e3:.1f} ms, min {min(ts) * 1e3:.1f} ms per pair", flush=True)
    # the same call with the images and the result planes in page-locked host memory (ugsm_host_alloc)
    c = m._ctx
    pl, pr = c.host_array(L.shape, L.dtype), c.host_array(R.shape, R.dtype)
    pl[...] = L
    pr[...] = R
    out = c.host_array((3, H, W))
    ts = []
    for _ in range(5):
        t0 = time.perf_counter()
        c.check(c.lib.ugsm_match_full(c.handle, pl.ctypes.data, pr.ctypes.data, W, H, 3 * W, out[0].ctypes.data, out[1].ctypes.data, out[2].ctypes.data))
        ts.append(time.perf_counter() - t0)
    print(f"{W}x{H} full mode, page-locked host buffers:  median {sorted(ts)[2] * 1e3:.1f} ms, min {min(ts) * 1e3:.1f} ms per pair", flush=True)
    del pl, pr, out
    m.close()
