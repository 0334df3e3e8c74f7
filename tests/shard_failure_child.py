"""One rank of tests/test_gpu_dist.py::test_a_failing_rank_of_the_shard_does_not_strand_its_peers (include/ugsm.h "when a rank
fails"): two ranks on one GPU over tests/fake_rccl.c.  Rank 0's context is created under UGSM_MEM_LIMIT_MB (development switch): a 16 MP step
does not fit it, a 1280 x 960 step does.  Every rank makes the same sequence of steps:

  A  16 MP, source 0: rank 0's pyramids are refused (UGSM_ERR_NOMEM from ugsm_submit_fovea_shard) -- it still broadcasts; rank 1's submit
     succeeds and its ugsm_wait answers UGSM_ERR_PEER, naming rank 0 and its status, within a bounded time;
  B  1280 x 960, source 0: a good step on both ranks, bit-equal to ugsm_submit_foveated -- the communicator survived;
  C  16 MP, source 1: rank 0 (NOT the source) fails the same way and still receives; rank 1's result is valid and bit-exact;
  D  1280 x 960, source 1: good on both;  ugsm_shard_count_ranks is still 2."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import numpy as np
    from ug_stereomatcher_amd import _lib, synth
    rank, world, idfile = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3]
    levels, F = 14, 7
    big, small = (4928, 3264), (1280, 960)
    offsets = [(0, 0), (300, -200)]
    imgs = {sz: synth.make_pair(sz[0], sz[1], synth.BASE_SEED + 70)[:2] for sz in (big, small)}
    with _lib.Context(levels=levels, fovea_levels=F, slots=2) as c:
        lib, h = c.lib, c.handle
        dev, expect, outs = {}, {}, {}
        for sz in (big, small):
            W, H = sz
            fw, fh = _lib.fovea_dims(W, H, levels, F)
            dev[sz] = (c.to_device(imgs[sz][0]), c.to_device(imgs[sz][1]))
            outs[sz] = c.alloc(3 * F * fh * fw * 4)
            if sz == big and rank == 0:
                continue   # (does not fit rank 0's memory limit: that is the point)
            c.check(lib.ugsm_submit_foveated(h, 0, dev[sz][0], dev[sz][1], W, H, 3 * W, offsets[rank][0], offsets[rank][1], outs[sz], None, None))
            c.check(lib.ugsm_wait(h, 0))
            expect[sz] = c.to_host(outs[sz], (3, F, fh, fw))
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
        c.shard_set_timeout(120000)
        log = []

        def step(sz, src):
            W, H = sz
            fw, fh = _lib.fovea_dims(W, H, levels, F)
            t0 = time.time()
            st = lib.ugsm_submit_fovea_shard(h, 1, dev[sz][0], dev[sz][1], W, H, 3 * W, offsets[rank][0], offsets[rank][1], outs[sz], src)
            msg_submit = lib.ugsm_last_error(h).decode()
            stw = lib.ugsm_wait(h, 1)
            msg_wait = lib.ugsm_last_error(h).decode()
            took = time.time() - t0
            got = c.to_host(outs[sz], (3, F, fh, fw)) if st == _lib.UGSM_OK and stw == _lib.UGSM_OK else None
            log.append((sz, src, st, stw, round(took, 2)))
            return st, stw, got, msg_submit, msg_wait, took

        def same(got, sz, what):
            if not (got.view(np.uint32) == expect[sz].view(np.uint32)).all():
                raise SystemExit(f"rank {rank}: {what}: differs from ugsm_submit_foveated")

        # A: the source fails
        st, stw, got, ms, mw, took = step(big, 0)
        if rank == 0:
            assert st == _lib.UGSM_ERR_NOMEM and "hipMalloc" in ms, (st, ms)
            assert stw == _lib.UGSM_OK, (stw, mw)                      # (its slot drains; it already knows)
        else:
            assert st == _lib.UGSM_OK, (st, ms)
            assert stw == _lib.UGSM_ERR_PEER and "rank 0" in mw and "status 6" in mw and "not valid" in mw, (stw, mw)
        assert took < 60, took
        # B: the communicator survived
        st, stw, got, ms, mw, took = step(small, 0)
        assert (st, stw) == (_lib.UGSM_OK, _lib.UGSM_OK), (st, stw, ms, mw)
        same(got, small, "B")
        # C: a rank that is not the source fails
        st, stw, got, ms, mw, took = step(big, 1)
        if rank == 0:
            assert st == _lib.UGSM_ERR_NOMEM, (st, ms)
        else:
            assert (st, stw) == (_lib.UGSM_OK, _lib.UGSM_OK), (st, stw, ms, mw)
            same(got, big, "C")
        assert took < 60, took
        # D
        st, stw, got, ms, mw, took = step(small, 1)
        assert (st, stw) == (_lib.UGSM_OK, _lib.UGSM_OK), (st, stw, ms, mw)
        same(got, small, "D")
        assert c.shard_count_ranks() == world
        c.shard_finalize()
        for sz in (big, small):
            for p in dev[sz] + (outs[sz],):
                c.free(p)
    print(f"SHARD_FAIL_OK rank={rank} steps={log}", flush=True)


if __name__ == "__main__":
    main()
