#!/usr/bin/env python3
"""Accuracy of the matcher against the synthetic ground truth (not the parity figure: that is bit-exactness against
the oracle).  Development tool; numbers quoted in DESIGN.md."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ug_stereomatcher_amd import MatchGPULib, synth  # noqa: E402
for (W, H) in [(1920, 1080), (4928, 3264)]:
    L, R, dx, dy = synth.make_pair(W, H, synth.BASE_SEED + 2)
    m = MatchGPULib()
    out = m.match(L, R, 0)
    m.close()
    b = 64
    ex, ey = (out[0] - dx)[b:-b, b:-b], (out[1] - dy)[b:-b, b:-b]
    print(f"{W}x{H}: dx median |err| {np.median(np.abs(ex)):.3f} px, RMSE {np.sqrt(np.mean(ex**2)):.3f} px; dy median |err| {np.median(np.abs(ey)):.3f} px, "
          f"RMSE {np.sqrt(np.mean(ey**2)):.3f} px; |dx err| < 0.5 px on {100*np.mean(np.abs(ex)<0.5):.1f} % of the interior; conf mean {out[2].mean():.3f}", flush=True)
