"""Child process of tests/test_gpu_dist.py: the fovea shard with its exchange INSIDE the library, on one GPU.

ugsm_shard_init joins a ONE-RANK RCCL communicator (the id is made by ugsm_shard_unique_id and handed round by
ug_stereomatcher_amd.dist.exchange_shard_id over a one-rank torch.distributed "nccl" group -- the control plane bench.py uses); every
step is ONE call, ugsm_submit_fovea_shard: pyramids -> coarse levels -> ncclBroadcast of the state on the slot's own stream -> fine levels
of an off-centre window.  Runs in a process of its own so that a hang inside RCCL cannot take the test session with it.  Checks, bit for bit:

  A  `steps` steps dealt over two slots without any host wait inside a step, against ugsm_submit_foveated at the same offset;
  B  ugsm_shard_count_ranks (ncclAllReduce of ones) = 1, ugsm_shard_rank = (0, 1);
  C  ugsm_shard_gather: the consumer rank's own stack lands in d_all (one rank: the device copy; no send / receive);
  D  call-sequence errors: a shard call on a context that has not joined (UGSM_ERR_STATE), a second ugsm_shard_init (UGSM_ERR_STATE), a
     source rank outside the communicator (UGSM_ERR_BAD_ARG) -- all refused BEFORE any collective is enqueued;
  E  after ugsm_shard_finalize the context works as before (ugsm_submit_foveated);
  F  the deadline (round 6, include/ugsm.h "when a rank fails"): with ugsm_shard_set_timeout(1 ms) a step that is still running at its deadline
     makes ugsm_wait abort the communicator (ncclCommAbort of the REAL library) and answer UGSM_ERR_PEER; every later shard call answers
     UGSM_ERR_STATE; after ugsm_shard_finalize + ugsm_shard_init (a new id) a step is bit-exact again;
  G  ugsm_shard_init_all with one context (ncclCommInitAll), a step, ugsm_shard_finalize.

Prints one line "RCCL_SHARD_OK ..." on success; any failure raises.
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

os.environ.setdefault("RANK", "0")
os.environ.setdefault("LOCAL_RANK", "0")
os.environ.setdefault("WORLD_SIZE", "1")
os.environ["UGSM_FORCE_DIST"] = "1"


def main():
    import torch
    import torch.distributed as dist
    from ug_stereomatcher_amd import _lib, dist as ud, synth

    W, H, levels, F = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
    off = (int(sys.argv[5]), int(sys.argv[6]))
    steps = int(sys.argv[7])
    rank, local_rank, world = ud.init()          # before any other GPU call of this process
    assert dist.is_initialized() and dist.get_backend() == "nccl" and world == 1
    dev = torch.device("cuda", local_rank)
    fw, fh = _lib.fovea_dims(W, H, levels, F)
    slots = 2
    pairs = []
    for j in range(2):
        L, R, _, _ = synth.make_pair(W, H, synth.BASE_SEED + 40 + j)
        pairs.append((torch.from_numpy(L).to(dev), torch.from_numpy(R).to(dev)))
    stride = 3 * W

    def same(a, b, what):
        if not torch.equal(a.view(torch.int32), b.view(torch.int32)):
            bad = int((a.view(torch.int32) != b.view(torch.int32)).sum())
            raise SystemExit(f"{what}: {bad} of {a.numel()} values differ")

    with _lib.Context(device=local_rank, levels=levels, fovea_levels=F, slots=slots) as ctx:
        lib, h = ctx.lib, ctx.handle
        # what every step must reproduce: the one-shot foveated match at the same offset
        expect = []
        for (Lt, Rt) in pairs:
            o = torch.empty((3, F, fh, fw), dtype=torch.float32, device=dev)
            ctx.check(lib.ugsm_submit_foveated(h, 0, Lt.data_ptr(), Rt.data_ptr(), W, H, stride, off[0], off[1], o.data_ptr(), None, None))
            ctx.check(lib.ugsm_wait(h, 0))
            expect.append(o)
        torch.cuda.synchronize()

        # ---- D (first half): not part of a shard yet --------------------------------------------------------------------------
        o = torch.empty((3, F, fh, fw), dtype=torch.float32, device=dev)
        st = lib.ugsm_submit_fovea_shard(h, 0, pairs[0][0].data_ptr(), pairs[0][1].data_ptr(), W, H, stride, off[0], off[1], o.data_ptr(), 0)
        assert st == _lib.UGSM_ERR_STATE, st

        # ---- B: join, count -------------------------------------------------------------------------------------------------------
        counted = ud.shard_init(ctx, rank, world)
        assert counted == 1, counted
        import ctypes as C
        r_, w_ = C.c_int(-1), C.c_int(-1)
        ctx.check(lib.ugsm_shard_rank(h, C.byref(r_), C.byref(w_)))
        assert (r_.value, w_.value) == (0, 1)
        # ---- D (second half) --------------------------------------------------------------------------------------------------------
        ident = (C.c_char * 128)(*_lib.shard_unique_id())
        assert lib.ugsm_shard_init(h, ident, 0, 1) == _lib.UGSM_ERR_STATE
        st = lib.ugsm_submit_fovea_shard(h, 0, pairs[0][0].data_ptr(), pairs[0][1].data_ptr(), W, H, stride, off[0], off[1], o.data_ptr(), 1)
        assert st == _lib.UGSM_ERR_BAD_ARG, st
        st = lib.ugsm_submit_fovea_shard(h, slots, pairs[0][0].data_ptr(), pairs[0][1].data_ptr(), W, H, stride, off[0], off[1], o.data_ptr(), 0)
        assert st == _lib.UGSM_ERR_BAD_ARG, st

        # ---- A: `steps` steps over two slots; the host never waits inside a step -----------------------------------------------
        outs = [torch.full((3, F, fh, fw), float("nan"), dtype=torch.float32, device=dev) for _ in range(slots)]
        results = []
        for k in range(steps):
            s = k % slots
            if k >= slots:
                ctx.check(lib.ugsm_wait(h, s))      # the slot is free again: bench.py's loop does the same
                results.append((k - slots, outs[s].clone()))
                outs[s].fill_(float("nan"))
                torch.cuda.current_stream().synchronize()
            Lt, Rt = pairs[k % 2]
            ctx.submit_fovea_shard(s, Lt.data_ptr(), Rt.data_ptr(), W, H, stride, off, outs[s].data_ptr(), 0)
        for k in range(max(0, steps - slots), steps):
            ctx.check(lib.ugsm_wait(h, k % slots))
            results.append((k, outs[k % slots].clone()))
        torch.cuda.synchronize()
        assert len(results) == steps
        for k, o_ in results:
            same(o_, expect[k % 2], f"A: ugsm_submit_fovea_shard step {k} (slot {k % slots}) vs ugsm_submit_foveated at offset {off}")

        # ---- C: the gather on the consumer rank ------------------------------------------------------------------------------------
        n = 3 * F * fh * fw
        d_all = torch.full((1, n), float("nan"), dtype=torch.float32, device=dev)
        last = (steps - 1) % slots
        ctx.shard_gather(last, outs[last].data_ptr(), n, d_all.data_ptr(), 0)
        ctx.check(lib.ugsm_wait(h, last))
        same(d_all[0].view(3, F, fh, fw), expect[(steps - 1) % 2], "C: ugsm_shard_gather")

        # ---- E: leave the communicator; the context is an ordinary one again ---------------------------------------------------
        ctx.shard_finalize()
        assert lib.ugsm_shard_rank(h, None, None) == _lib.UGSM_ERR_STATE
        o2 = torch.empty((3, F, fh, fw), dtype=torch.float32, device=dev)
        ctx.check(lib.ugsm_submit_foveated(h, 1, pairs[1][0].data_ptr(), pairs[1][1].data_ptr(), W, H, stride, off[0], off[1], o2.data_ptr(), None, None))
        ctx.check(lib.ugsm_wait(h, 1))
        same(o2, expect[1], "E: ugsm_submit_foveated after ugsm_shard_finalize")
        # ---- F: a step that misses its deadline -----------------------------------------------------------------------------------
        ud.shard_init(ctx, rank, world)
        ctx.shard_set_timeout(1)
        t_f = "skipped (the step finished inside its millisecond)"
        for attempt in range(3):           # (a 16 MP step takes ~2 ms: the deadline passes while it runs; smaller frames may need the queue behind them)
            for q in range(3):
                ctx.submit_fovea_shard(0, pairs[0][0].data_ptr(), pairs[0][1].data_ptr(), W, H, stride, off, outs[0].data_ptr(), 0)
            st = lib.ugsm_wait(h, 0)
            if st == _lib.UGSM_ERR_PEER:
                t_f = "deadline fired"
                msg = lib.ugsm_last_error(h).decode()
                assert "did not finish within 1 ms" in msg and "aborted" in msg, msg
                st2 = lib.ugsm_submit_fovea_shard(h, 0, pairs[0][0].data_ptr(), pairs[0][1].data_ptr(), W, H, stride, off[0], off[1], outs[0].data_ptr(), 0)
                assert st2 == _lib.UGSM_ERR_STATE and "aborted" in lib.ugsm_last_error(h).decode(), (st2, lib.ugsm_last_error(h))
                assert lib.ugsm_shard_count_ranks(h, C.byref(r_)) == _lib.UGSM_ERR_STATE
                break
            assert st == _lib.UGSM_OK, st
        ctx.shard_finalize()
        ud.shard_init(ctx, rank, world)          # a fresh communicator: the shard works again
        ctx.submit_fovea_shard(1, pairs[1][0].data_ptr(), pairs[1][1].data_ptr(), W, H, stride, off, outs[1].data_ptr(), 0)
        ctx.check(lib.ugsm_wait(h, 1))
        same(outs[1], expect[1], "F: a step after ugsm_shard_finalize + ugsm_shard_init")
        ctx.shard_finalize()

        # ---- G: one process that owns the GPUs of the shard (here: one) --------------------------------------------------------------
        ctx.shard_init_all()
        assert ctx.shard_count_ranks() == 1
        ctx.submit_fovea_shard(0, pairs[0][0].data_ptr(), pairs[0][1].data_ptr(), W, H, stride, off, outs[0].data_ptr(), 0)
        ctx.check(lib.ugsm_wait(h, 0))
        same(outs[0], expect[0], "G: a step over ugsm_shard_init_all")
        ctx.shard_finalize()
    dist.destroy_process_group()
    print(f"RCCL_SHARD_OK deadline: {t_f}; steps={steps} slots={slots} offset={off} fovea={fw}x{fh} rccl_ranks={counted} exchange=inside-the-library", flush=True)


if __name__ == "__main__":
    main()
