# round 3: where a 20-pair burst (bench.py --steps 20, the driver's command) loses against the steady state: completion time of every pair
import sys, time, torch
sys.path.insert(0, '.')
from ug_stereomatcher_amd import _lib, synth
W, H = 4928, 3264
dev = torch.device("cuda:0")
pairs = []
for j in range(2):
    L, R, _, _ = synth.make_pair(W, H, synth.BASE_SEED + 2 + 16 * j); pairs.append((torch.from_numpy(L).to(dev), torch.from_numpy(R).to(dev)))
slots = int(sys.argv[1]) if len(sys.argv) > 1 else 4
outs = [torch.empty((3, H, W), dtype=torch.float32, device=dev) for _ in range(slots)]
with _lib.Context(levels=14, slots=slots) as c:
    lib, h = c.lib, c.handle
    def burst(n):
        torch.cuda.synchronize()
        t0 = time.perf_counter(); done = {}; sub = {}
        for k in range(n):
            s = k % slots
            c.check(lib.ugsm_wait(h, s))
            if k >= slots: done[k - slots] = time.perf_counter() - t0
            a, b = pairs[k % 2]
            c.check(lib.ugsm_submit_full(h, s, a.data_ptr(), b.data_ptr(), W, H, 3 * W, outs[s].data_ptr()))
            sub[k] = time.perf_counter() - t0
        for k in range(n - slots, n):
            c.check(lib.ugsm_wait(h, k % slots)); done[k] = time.perf_counter() - t0
        return sub, done
    burst(8)
    for rep in range(2):
        sub, done = burst(20)
        print("total %.1f ms = %.1f pairs/s" % (1e3 * done[19], 20 / done[19]))
        print("submitted at (ms):", " ".join("%.1f" % (1e3 * sub[k]) for k in range(20)))
        print("done at (ms):     ", " ".join("%.1f" % (1e3 * done[k]) for k in range(20)))
        d = [1e3 * done[k] for k in range(20)]
        print("intervals (ms):   ", " ".join("%.1f" % (d[k] - d[k - 1]) for k in range(1, 20)))
