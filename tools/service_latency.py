#!/usr/bin/env python3
"""PCIe-inclusive latency of the service path (ugsm_match_full: host buffers in and out), the number DESIGN.md quotes beside
bench.py's HBM-resident throughput.  Development tool.

Three caller patterns: result planes that were touched before (a node that keeps its buffers), FRESH pageable result planes for
every call (what the reference node does: it mallocs and frees them per call, UG_GPU_matcher.cpp:414-418,487-489) and page-locked
buffers (ugsm_host_alloc).  For the fresh pattern the caller's own cost of releasing the previous result (munmap of 193 MB at
16 MP) is shown separately: it is outside the library call."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ug_stereomatcher_amd import _lib, synth  # noqa: E402


def med(v):
    return sorted(v)[len(v) // 2] * 1e3


for (W, H) in [(4928, 3264), (1920, 1080)]:
    L, R, _, _ = synth.make_pair(W, H, synth.BASE_SEED + 2)
    c = _lib.Context(levels=14)

    def call(Lh, Rh, out):
        t0 = time.perf_counter()
        c.check(c.lib.ugsm_match_full(c.handle, Lh.ctypes.data, Rh.ctypes.data, W, H, 3 * W, out[0].ctypes.data, out[1].ctypes.data, out[2].ctypes.data))
        return time.perf_counter() - t0
    out = np.empty((3, H, W), np.float32)
    call(L, R, out)  # first call allocates the context buffers
    reused = [call(L, R, out) for _ in range(5)]
    fresh, freed = [], []
    for _ in range(5):
        t0 = time.perf_counter()
        out = None  # the caller releases the previous result ...
        freed.append(time.perf_counter() - t0)
        out = np.empty((3, H, W), np.float32)  # ... and allocates fresh, untouched planes
        fresh.append(call(L, R, out))
    pl, pr, po = c.host_array(L.shape, L.dtype), c.host_array(R.shape, R.dtype), c.host_array((3, H, W))
    pl[...] = L
    pr[...] = R
    pinned = [call(pl, pr, po) for _ in range(5)]
    print(f"{W}x{H} full mode, ugsm_match_full per pair (median of 5): pageable buffers touched before {med(reused):.1f} ms; "
          f"FRESH pageable result planes per call {med(fresh):.1f} ms (+ {med(freed):.1f} ms in the caller to free the previous ones); "
          f"page-locked buffers {med(pinned):.1f} ms", flush=True)
    del pl, pr, po
    c.close()
