#!/usr/bin/env python3
"""Randomised check of the queue (ugsm_enqueue_* / ugsm_flush / ugsm_next_done) against single blocking calls on the GPU: random slot and
batch counts, image sizes, modes (full / foveated with random window offsets, with and without pyramid stacks), memory kinds (device,
page-locked host, managed), and random points at which the host flushes, fetches without blocking, or drains.  Whatever calls the library
forms, every pair must come back in enqueue order, with its tag, bit-identical to the single call, and the bookkeeping must balance
(at most (slots + 1) x batch outstanding; depth 0 after a drain).  Development tool:  python tools/stress_queue.py [cases [seed]]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ug_stereomatcher_amd import _lib, synth  # noqa: E402

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 100
rng = np.random.Generator(np.random.PCG64(int(sys.argv[2]) if len(sys.argv) > 2 else 20261005))  # (second argument: another seed)


def same(a, b):
    return bool(((a.view(np.uint32) == b.view(np.uint32)) | (np.isnan(a) & np.isnan(b))).all())


bad = 0
for case in range(n_cases):
    slots, batch = int(rng.integers(1, 5)), int(rng.choice([1, 2, 3, 4, 8]))
    levels = int(rng.integers(5, 10))
    F = int(rng.integers(2, min(levels, 6) + 1))
    sizes = [(int(rng.integers(96, 400)), int(rng.integers(80, 300))) for _ in range(int(rng.integers(1, 3)))]
    n = int(rng.integers(1, 3 * (slots + 1) * batch + 2))
    cap = (slots + 1) * batch
    with _lib.Context(levels=levels, fovea_levels=F, slots=slots, batch=batch) as q, _lib.Context(levels=levels, fovea_levels=F) as ref:
        imgs = {}
        for (W, H) in sizes:
            prs = [synth.make_pair(W, H, int(rng.integers(1, 1 << 30)))[:2] for _ in range(2)]
            imgs[(W, H)] = [(L, R, q.to_device(L), q.to_device(R)) for L, R in prs]
        jobs, expect, outs, keep = [], {}, {}, []
        outstanding = 0

        def fetch(block):
            global bad, outstanding
            while True:
                c = q.next_done(block)
                if c is None:
                    return
                tag = int(c.tag)
                kind, shp, kindmem = jobs[tag][0], jobs[tag][1], jobs[tag][2]
                if kindmem == "managed":
                    planes = q.managed_planes(c, shp)
                    got = [np.array(p) for p in planes]
                elif kindmem == "pinned":
                    got = [np.array(a) for a in outs[tag]]
                else:
                    got = [q.to_host(p, s) for p, s in zip(outs[tag], shp)]
                ok = all(same(g.reshape(e.shape), e) for g, e in zip(got, expect[tag])) and tag == fetch.next
                if not ok:
                    bad += 1
                    print(f"case {case}: pair {tag} ({kind}, {kindmem}) differs or out of order (expected tag {fetch.next})")
                fetch.next += 1
                outstanding -= 1
                if kindmem == "device":
                    for p in outs[tag]:
                        q.free(p)
                outs.pop(tag, None)
                if block == "one":
                    return
        fetch.next = 0
        for k in range(n):
            W, H = sizes[int(rng.integers(len(sizes)))]
            L, R, dL, dR = imgs[(W, H)][int(rng.integers(2))]
            mode = "full" if rng.random() < 0.6 else "fovea"
            mem = str(rng.choice(["device", "pinned", "managed"]))
            off = (int(rng.integers(-W, W)), int(rng.integers(-H, H)))
            want_pyr = mode == "fovea" and rng.random() < 0.3
            fw, fh = _lib.fovea_dims(W, H, levels, F)
            # the single blocking call
            if mode == "full":
                e = np.empty((3, H, W), np.float32)
                ref.check(ref.lib.ugsm_match_full(ref.handle, L.ctypes.data, R.ctypes.data, W, H, 3 * W, e[0].ctypes.data, e[1].ctypes.data, e[2].ctypes.data))
                exp = [e[0], e[1], e[2]] if mem != "device" else [e]
                shp = [(H, W)] * 3 if mem != "device" else [(3, H, W)]
            else:
                st = np.empty((3, F, fh, fw), np.float32)
                pl = np.empty((F, 3, fh, fw), np.float32)
                pr = np.empty((F, 3, fh, fw), np.float32)
                ref.check(ref.lib.ugsm_match_foveated(ref.handle, L.ctypes.data, R.ctypes.data, W, H, 3 * W, off[0], off[1], st[0].ctypes.data, st[1].ctypes.data,
                                                      st[2].ctypes.data, pl.ctypes.data if want_pyr else None, pr.ctypes.data if want_pyr else None))
                if mem == "device":
                    exp, shp = [st] + ([pl, pr] if want_pyr else []), [(3, F, fh, fw)] + ([(F, 3, fh, fw)] * 2 if want_pyr else [])
                else:
                    exp, shp = [st[0], st[1], st[2]] + ([pl, pr] if want_pyr else []), [(F, fh, fw)] * 3 + ([(F, 3, fh, fw)] * 2 if want_pyr else [])
            while outstanding >= cap:   # the host's side of the back-pressure rule
                fetch("one")
            jobs.append((mode, shp, mem))
            expect[k] = exp
            if mem == "device":
                bufs = [q.alloc(int(np.prod(s)) * 4) for s in shp]
                outs[k] = bufs
                if mode == "full":
                    q.enqueue_full(dL, dR, W, H, 3 * W, bufs[0], k)
                else:
                    q.enqueue_foveated(dL, dR, W, H, 3 * W, off, bufs[0], k, bufs[1] if want_pyr else None, bufs[2] if want_pyr else None)
            elif mem == "pinned":
                pL_, pR_ = q.host_array(L.shape, L.dtype), q.host_array(R.shape, R.dtype)
                pL_[...] = L
                pR_[...] = R
                keep.append((pL_, pR_))
                arrs = [q.host_array(s) for s in shp]
                outs[k] = arrs
                if mode == "full":
                    q.check(q.lib.ugsm_enqueue_full_host(q.handle, pL_.ctypes.data, pR_.ctypes.data, W, H, 3 * W, arrs[0].ctypes.data, arrs[1].ctypes.data, arrs[2].ctypes.data, k))
                else:
                    q.check(q.lib.ugsm_enqueue_foveated_host(q.handle, pL_.ctypes.data, pR_.ctypes.data, W, H, 3 * W, off[0], off[1], arrs[0].ctypes.data, arrs[1].ctypes.data,
                                                             arrs[2].ctypes.data, arrs[3].ctypes.data if want_pyr else None, arrs[4].ctypes.data if want_pyr else None, k))
            else:
                if mode == "full":
                    q.enqueue_full_managed(L, R, k)
                else:
                    q.enqueue_foveated_managed(L, R, off, want_pyr, k)
            outstanding += 1
            w_, f_, u_ = q.queue_depth()
            if w_ + f_ + u_ != outstanding or outstanding > cap:
                bad += 1
                print(f"case {case}: bookkeeping: depth {(w_, f_, u_)} but {outstanding} outstanding (cap {cap})")
            r = rng.random()
            if r < 0.25:
                q.flush()
            elif r < 0.5:
                fetch(False)
            elif r < 0.6:
                fetch(True)
        fetch(True)
        if fetch.next != n or q.queue_depth() != (0, 0, 0):
            bad += 1
            print(f"case {case}: {fetch.next} of {n} pairs reported, depth {q.queue_depth()}")
        for lst in imgs.values():
            for (_, _, dL, dR) in lst:
                q.free(dL)
                q.free(dR)
    if (case + 1) % 10 == 0:
        print(f"{case + 1} cases, {bad} mismatches", flush=True)
print(f"done: {n_cases} cases, {bad} mismatches")
sys.exit(1 if bad else 0)
