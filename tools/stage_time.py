#!/usr/bin/env python3
"""Times one level-0 iteration pair (sqblur + 2 x (K-cost + K-smooth 5 + K-smooth 5+box)) on a real synthetic pair
through ugsm_stage_iterate -- a product-level A/B probe for compiler-flag experiments.  Development tool."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ug_stereomatcher_amd import _lib, synth  # noqa: E402

W, H = 4928, 3264
L, R, dx, dy = synth.make_pair(W, H, synth.BASE_SEED + 2)
pl = np.ascontiguousarray(np.transpose(L, (2, 0, 1)).astype(np.float32))
pr = np.ascontiguousarray(np.transpose(R, (2, 0, 1)).astype(np.float32))
rng = np.random.Generator(np.random.PCG64(1))
d3 = np.stack([dx + rng.normal(0, 0.2, dx.shape), dy + rng.normal(0, 0.2, dy.shape), 0.3 + 0.7 * rng.random(dx.shape)]).astype(np.float32)
ctx = _lib.Context(levels=3)
a, b, d = ctx.to_device(pl), ctx.to_device(pr), ctx.to_device(d3)
def run():
    ctx.check(ctx.lib.ugsm_stage_iterate(ctx.handle, a, b, d, W, H, 2, 10, 0, 1, 2, None))
run()
ts = []
for _ in range(8):
    t0 = time.perf_counter(); run(); ts.append(time.perf_counter() - t0)
print(f"level-0 x2 iterations (real pair): median {1e3 * sorted(ts)[4]:.3f} ms, min {1e3 * min(ts):.3f} ms")
ctx.close()
