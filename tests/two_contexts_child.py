"""Child process of tests/test_gpu_batch.py::test_two_contexts_in_one_process_with_stream_priority_pools: two contexts of four slots in one
(fresh) process, in different stream-priority pools, each at the rate a context alone reaches; a context at the process default priority
for comparison.  Prints the rates and "TWO_CONTEXTS_OK"; any failure raises."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from ug_stereomatcher_amd import _lib as lib, synth  # noqa: E402

W, H, lv, slots, n = 4928, 3264, 14, 4, 32   # (16 MP: bound by the GPU, not by the submitting thread -- the rates repeat to a per cent)
L, R, _, _ = synth.make_pair(W, H, synth.BASE_SEED + 77)


def rate(c, o, dL, dR):
    def run(k):
        for i in range(k):
            s = i % slots
            c.check(c.lib.ugsm_wait(c.handle, s))
            c.check(c.lib.ugsm_submit_full(c.handle, s, dL, dR, W, H, 3 * W, o[s]))
        c.check(c.lib.ugsm_wait_all(c.handle))
    run(2 * slots)
    best = 0.0
    for _ in range(3):
        t0 = time.perf_counter()
        run(n)
        best = max(best, n / (time.perf_counter() - t0))
    return best


def same(a, b):
    return bool((a.view(np.uint32) == b.view(np.uint32)).all())


with lib.Context(levels=lv, slots=slots) as a:
    dL, dR = a.to_device(L), a.to_device(R)   # (device memory belongs to the process: both contexts use these buffers)
    outs = [a.alloc(3 * W * H * 4) for _ in range(2 * slots)]
    alone = rate(a, outs[:slots], dL, dR)
    ref = a.to_host(outs[0], (3, H, W))
    with lib.Context(levels=lv, slots=slots, stream_priority=3) as b:       # a pool of its own: the least priority
        rb = rate(b, outs[slots:], dL, dR)
        ra = rate(a, outs[:slots], dL, dR)
        assert same(b.to_host(outs[slots], (3, H, W)), ref), "results do not depend on the stream priorities"
    with lib.Context(levels=lv, slots=slots, stream_priority=1) as d:       # the opt-out: the process default priority
        rd = rate(d, outs[slots:], dL, dR)
        assert same(d.to_host(outs[slots], (3, H, W)), ref), "results do not depend on the stream priorities"
    for p in [dL, dR] + outs:
        a.free(p)
print(f"pairs/s at 16 MP, four slots: context alone {alone:.0f}; second context in the least-priority pool {rb:.0f}, the first again {ra:.0f}; "
      f"a context at the process default priority (beside the null stream) {rd:.0f}")
assert rb >= 0.85 * alone and ra >= 0.85 * alone, (alone, rb, ra)
print("TWO_CONTEXTS_OK")
