"""Child process of tests/test_gpu_dist.py: the fovea shard's DEVICE-ORDERED exchange on one GPU.

Runs in a process of its own because the RCCL process group must exist before anything else touches the GPU
(ug_stereomatcher_amd.dist.init with UGSM_FORCE_DIST=1: a ONE-RANK group over backend "nccl" == RCCL).  Three checks, each bit for bit:

  A  the real UgsmShardDriver through fovea_shard_step, `steps` steps dealt over two slots, an off-centre window, no host wait inside
     a step (orders_on_device is true over nccl) -- against ugsm_submit_foveated at the same offset, and gather_stacks;
  B  current_after_slot: the torch stream the collective is launched from reads the state only after the slot's coarse phase has
     written it (the source rank's side of the exchange), shown by copying the state on that stream right after the call;
  C  slot_after_current: the slot's fine phase starts only after the work enqueued on the torch stream has delivered the state (a
     receiving rank's side): the state arrives late, behind some milliseconds of other work on that stream, and starts out as NaN.

B and C use a torch copy in the place of the broadcast: with one rank the collective moves nothing, so the ordering it needs would not
show in A alone.  Prints one line "RCCL_SHARD_OK ..." on success; any failure raises.
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
    import numpy as np
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
        # what every step must reproduce: the one-shot foveated match at the same offset, and the coarse state on its own
        expect, expect_state = [], []
        for (Lt, Rt) in pairs:
            o = torch.empty((3, F, fh, fw), dtype=torch.float32, device=dev)
            ctx.check(lib.ugsm_submit_foveated(h, 0, Lt.data_ptr(), Rt.data_ptr(), W, H, stride, off[0], off[1], o.data_ptr(), None, None))
            ctx.check(lib.ugsm_wait(h, 0))
            expect.append(o)
            st = torch.empty((3, fh, fw), dtype=torch.float32, device=dev)
            ctx.check(lib.ugsm_submit_pyramids(h, 0, Lt.data_ptr(), Rt.data_ptr(), W, H, stride))
            ctx.check(lib.ugsm_submit_fovea_coarse(h, 0, st.data_ptr()))
            ctx.check(lib.ugsm_wait(h, 0))
            expect_state.append(st)
        torch.cuda.synchronize()

        # ---- A: the real driver, device-ordered, no host wait inside a step -------------------------------------------------
        drv = ud.UgsmShardDriver(ctx)
        waits = []
        real_wait = drv.wait
        drv.wait = lambda slot: (waits.append(slot), real_wait(slot))[1]
        states = [torch.full((3, fh, fw), float("nan"), dtype=torch.float32, device=dev) for _ in range(slots)]
        outs = [torch.full((3, F, fh, fw), float("nan"), dtype=torch.float32, device=dev) for _ in range(slots)]
        assert drv.orders_on_device(states[0])
        results = []
        for k in range(steps):
            s = k % slots
            if k >= slots:
                real_wait(s)                      # the slot (and its state buffer) is free again: bench.py's submit() does the same
                results.append((k - slots, outs[s].clone()))
                outs[s].fill_(float("nan"))
                states[s].fill_(float("nan"))
                torch.cuda.current_stream().synchronize()
            Lt, Rt = pairs[k % 2]
            n_before = len(waits)
            ud.fovea_shard_step(drv, s, Lt, Rt, W, H, stride, states[s], off, outs[s], rank)
            assert len(waits) == n_before, "a device-ordered step must not block the host"
        for k in range(max(0, steps - slots), steps):
            real_wait(k % slots)
            results.append((k, outs[k % slots].clone()))
        torch.cuda.synchronize()
        assert len(results) == steps
        for k, o in results:
            same(o, expect[k % 2], f"A: fovea_shard_step {k} (slot {k % slots}) vs ugsm_submit_foveated at offset {off}")
        got = ud.gather_stacks(results[-1][1], 0)
        assert got is not None and len(got) == 1
        same(got[0], expect[(steps - 1) % 2], "A: gather_stacks")

        # ---- B: current_after_slot (source side) ---------------------------------------------------------------------------------
        side = torch.cuda.Stream()               # "the stream the collective is launched from"
        for rep in range(3):
            Lt, Rt = pairs[rep % 2]
            st = states[0]
            st.fill_(float("nan"))
            sent = torch.full_like(st, float("nan"))
            torch.cuda.synchronize()
            drv.submit_pyramids(0, Lt, Rt, W, H, stride)
            drv.submit_coarse(0, st)
            with torch.cuda.stream(side):
                drv.current_after_slot(0)
                sent.copy_(st)                   # stands in for the broadcast reading the state
            side.synchronize()
            same(sent, expect_state[rep % 2], f"B: state read on the collective's stream after current_after_slot (rep {rep})")
            real_wait(0)

        # ---- C: slot_after_current (receiving side) ----------------------------------------------------------------------------
        big = torch.randn((4096, 4096), device=dev)
        for rep in range(3):
            Lt, Rt = pairs[rep % 2]
            st, o = states[1], outs[1]
            st.fill_(float("nan"))
            o.fill_(float("nan"))
            torch.cuda.synchronize()
            drv.submit_pyramids(1, Lt, Rt, W, H, stride)
            with torch.cuda.stream(side):
                drv.current_after_slot(1)
                acc = big
                for _ in range(6):               # some milliseconds of other work in front of the state's arrival
                    acc = acc @ big
                    acc = acc / acc.abs().max()
                st.copy_(expect_state[rep % 2])  # stands in for the broadcast delivering the state
                drv.slot_after_current(1)
            drv.submit_fine(1, st, off, o)
            real_wait(1)
            side.synchronize()
            same(o, expect[rep % 2], f"C: fine phase after slot_after_current (rep {rep})")
    dist.destroy_process_group()
    print(f"RCCL_SHARD_OK steps={steps} slots={slots} offset={off} fovea={fw}x{fh} backend=nccl world=1", flush=True)


if __name__ == "__main__":
    main()
