# round 3: does spacing the four slots' phases evenly help?  (slots that end close together stay close together: every pair takes the same time)
import sys, time, torch
sys.path.insert(0, '.')
from ug_stereomatcher_amd import _lib, synth
W, H = 4928, 3264
dev = torch.device("cuda:0")
pairs = []
for j in range(2):
    L, R, _, _ = synth.make_pair(W, H, synth.BASE_SEED + 2 + 16 * j); pairs.append((torch.from_numpy(L).to(dev), torch.from_numpy(R).to(dev)))
slots = 4
outs = [torch.empty((3, H, W), dtype=torch.float32, device=dev) for _ in range(slots)]
with _lib.Context(levels=14, slots=slots) as c:
    lib, h = c.lib, c.handle
    def run(n, gap_ms, paced_pairs):
        torch.cuda.synchronize()
        t0 = time.perf_counter(); last = -1.0; done = []
        for k in range(n):
            s = k % slots
            c.check(lib.ugsm_wait(h, s))
            if k >= slots: done.append(time.perf_counter() - t0)
            if k < paced_pairs:
                while time.perf_counter() - last < gap_ms * 1e-3: pass
            last = time.perf_counter()
            a, b = pairs[k % 2]
            c.check(lib.ugsm_submit_full(h, s, a.data_ptr(), b.data_ptr(), W, H, 3 * W, outs[s].data_ptr()))
        c.check(lib.ugsm_wait_all(h))
        T = time.perf_counter() - t0
        iv = [1e3 * (done[i] - done[i - 1]) for i in range(max(1, len(done) - 40), len(done))] or [0.0]
        return n / T, min(iv), max(iv)
    run(16, 0, 0)
    for gap, paced in ((0, 0), (5.0, 8), (6.0, 8), (5.5, 400), (4.0, 400), (0, 0), (6.0, 8)):
        r, lo, hi = run(400, gap, paced)
        print(f"min gap {gap} ms on the first {paced} submissions: {r:.2f} pairs/s; last 40 completion intervals {lo:.1f} .. {hi:.1f} ms")
