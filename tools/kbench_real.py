#!/usr/bin/env python3
"""Writes the level-0 planes of a synthetic stereo pair (L, R as float planes, a (dx,dy,conf) state near the truth)
for `tools/kbench W H reps 0 <file>`: the kernels' speed depends on the data (random images behave differently)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ug_stereomatcher_amd import synth  # noqa: E402
W, H, out = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3]
L, R, dx, dy = synth.make_pair(W, H, synth.BASE_SEED + 2)
rng = np.random.Generator(np.random.PCG64(1))
with open(out, "wb") as f:
    for img in (L, R):
        f.write(np.ascontiguousarray(np.transpose(img, (2, 0, 1)).astype(np.float32)).tobytes())
    f.write(np.stack([dx + rng.normal(0, 0.2, dx.shape), dy + rng.normal(0, 0.2, dy.shape), 0.3 + 0.7 * rng.random(dx.shape)]).astype(np.float32).tobytes())
