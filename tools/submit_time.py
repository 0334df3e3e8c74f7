import sys, time, numpy as np
sys.path.insert(0, '/root/repo')
import torch
from ug_stereomatcher_amd import _lib, synth
W, H = 4928, 3264
L, R, *_ = synth.make_pair(W, H, seed=3)
ctx = _lib.Context(levels=14, slots=4, profile_events=0)
dL = torch.from_numpy(L).cuda(); dR = torch.from_numpy(R).cuda()
outs = [torch.empty((3, H, W), dtype=torch.float32, device='cuda') for _ in range(4)]
torch.cuda.synchronize()
lib = ctx.lib
def submit(slot):
    ctx.check(lib.ugsm_submit_full(ctx.handle, slot, dL.data_ptr(), dR.data_ptr(), W, H, 3 * W, outs[slot].data_ptr()))
for s in range(4): submit(s)
for s in range(4): ctx.check(lib.ugsm_wait(ctx.handle, s))
# enqueue cost with an empty queue
t0 = time.perf_counter(); submit(0); t1 = time.perf_counter(); ctx.check(lib.ugsm_wait(ctx.handle, 0)); t2 = time.perf_counter()
print(f"submit returns after {1e3*(t1-t0):.2f} ms; pair done after {1e3*(t2-t0):.2f} ms")
# 4 back to back
t0 = time.perf_counter()
for s in range(4): submit(s)
t1 = time.perf_counter()
for s in range(4): ctx.check(lib.ugsm_wait(ctx.handle, s))
t2 = time.perf_counter()
print(f"4 submits return after {1e3*(t1-t0):.2f} ms; all done after {1e3*(t2-t0):.2f} ms")
ctx.close()
