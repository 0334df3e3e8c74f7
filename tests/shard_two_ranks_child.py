"""One rank of tests/test_gpu_dist.py::test_fovea_shard_two_ranks_over_a_fake_transport: the library's shard code with TWO ranks on one GPU.

RCCL refuses two ranks on one device, so the transport is tests/fake_rccl.c (shared memory between the two processes, loaded through
UGSM_RCCL_PATH); everything else is the product: ugsm_shard_init, ugsm_submit_fovea_shard (here also on the rank that RECEIVES the coarse state
and never runs the coarse phase), ugsm_shard_count_ranks, ugsm_shard_gather.  Every rank's stacks must equal ugsm_submit_foveated at that
rank's window, bit for bit, with rank 0 and then rank 1 as the source.  No torch in this process (its bundled RCCL must not be found)."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import numpy as np
    from ug_stereomatcher_amd import _lib, synth
    rank, world, idfile = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3]
    W, H, levels, F, steps = (int(v) for v in sys.argv[4:9])
    offsets = [(0, 0), (W // 5, -H // 7)]
    fw, fh = _lib.fovea_dims(W, H, levels, F)
    slots = 2
    n = 3 * F * fh * fw
    imgs = [synth.make_pair(W, H, synth.BASE_SEED + 60 + j)[:2] for j in range(2)]
    with _lib.Context(levels=levels, fovea_levels=F, slots=slots) as c:
        lib, h = c.lib, c.handle
        dL = [c.to_device(L) for L, _ in imgs]
        dR = [c.to_device(R) for _, R in imgs]
        # what every rank must get: the one-shot foveated match at ITS window (rank 0 also computes rank 1's, for the gather)
        expect = {}
        tmp = c.alloc(n * 4)
        for r in range(world):
            if r != rank and rank != 0:
                continue
            for j in range(2):
                c.check(lib.ugsm_submit_foveated(h, 0, dL[j], dR[j], W, H, 3 * W, offsets[r][0], offsets[r][1], tmp, None, None))
                c.check(lib.ugsm_wait(h, 0))
                expect[(r, j)] = c.to_host(tmp, (3, F, fh, fw))
        c.free(tmp)
        # the id: made by rank 0, handed over through a file
        if rank == 0:
            ident = _lib.shard_unique_id()
            with open(idfile + ".tmp", "wb") as f:
                f.write(ident)
            os.replace(idfile + ".tmp", idfile)
        else:
            t0 = time.time()
            while not os.path.exists(idfile):
                if time.time() - t0 > 120:
                    raise SystemExit("no shard id from rank 0")
                time.sleep(0.05)
            ident = open(idfile, "rb").read()
        c.shard_init(ident, rank, world)
        assert c.shard_count_ranks() == world
        outs = [c.alloc(n * 4) for _ in range(slots)]
        checked = 0
        for src in range(world):            # rank 0 as the source of the coarse state, then rank 1
            results = []
            for k in range(steps):
                s = k % slots
                if k >= slots:
                    c.check(lib.ugsm_wait(h, s))
                    results.append((k - slots, c.to_host(outs[s], (3, F, fh, fw))))
                c.submit_fovea_shard(s, dL[k % 2], dR[k % 2], W, H, 3 * W, offsets[rank], outs[s], src)
            for k in range(max(0, steps - slots), steps):
                c.check(lib.ugsm_wait(h, k % slots))
                results.append((k, c.to_host(outs[k % slots], (3, F, fh, fw))))
            for k, got in results:
                exp = expect[(rank, k % 2)]
                if not (got.view(np.uint32) == exp.view(np.uint32)).all():
                    raise SystemExit(f"rank {rank}, source {src}, step {k}: {int((got.view(np.uint32) != exp.view(np.uint32)).sum())} values differ from ugsm_submit_foveated at {offsets[rank]}")
                checked += 1
        # every rank's last stack to rank 0
        last = (steps - 1) % slots
        d_all = c.alloc(world * n * 4) if rank == 0 else None
        c.shard_gather(last, outs[last], n, d_all, 0)
        c.check(lib.ugsm_wait(h, last))
        if rank == 0:
            allst = c.to_host(d_all, (world, 3, F, fh, fw))
            for r in range(world):
                if not (allst[r].view(np.uint32) == expect[(r, (steps - 1) % 2)].view(np.uint32)).all():
                    raise SystemExit(f"gather: rank {r}'s stack differs")
            c.free(d_all)
        c.shard_finalize()
        for p in dL + dR + outs:
            c.free(p)
    print(f"SHARD2_OK rank={rank} world={world} steps_checked={checked} window={offsets[rank]} fovea={fw}x{fh}", flush=True)


if __name__ == "__main__":
    main()
