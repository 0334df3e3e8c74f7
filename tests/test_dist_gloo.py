"""N>1 plumbing on CPU: world_size-2 gloo process group exercising what bench.py and the
fovea-shard mode use (pair sharding, max-over-ranks timing, the one broadcast of the coarse state,
optional gather of the stacks).  No GPU, no compute: the kernels are covered by -m gpu tests."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, out_q):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    from ug_stereomatcher_amd import dist as ud
    r, lr, w = ud.init(backend="gloo")
    assert (r, w) == (rank, world)
    # independent pairs: disjoint, complete cover, no collective needed
    mine = ud.shard_pairs(11, rank, world)
    # bench contract: duration = max over ranks; work = sum over ranks
    tmax = ud.max_over_ranks(1.0 + rank)
    total = ud.sum_over_ranks(float(len(mine)))
    # fovea sharding: rank 0 owns the coarse state (3 x fovH x fovW), everyone receives it
    fh, fw = 13, 17
    state = torch.zeros((3, fh, fw), dtype=torch.float32)
    if rank == 0:
        state = torch.arange(3 * fh * fw, dtype=torch.float32).reshape(3, fh, fw).contiguous()
    ud.barrier()
    ud.broadcast_coarse_state(state, 0)
    ok_state = bool(torch.equal(state, torch.arange(3 * fh * fw, dtype=torch.float32).reshape(3, fh, fw)))
    # rgb pair broadcast
    L = torch.full((4, 5, 3), 7 if rank == 0 else 0, dtype=torch.uint8)
    R = torch.full((4, 5, 3), 9 if rank == 0 else 0, dtype=torch.uint8)
    ud.broadcast_pair(L, R, 0)
    # gather of the per-window stacks on rank 0
    stack = torch.full((3, 2, fh, fw), float(rank), dtype=torch.float32)
    got = ud.gather_stacks(stack, 0)
    ok_gather = True
    if rank == 0:
        ok_gather = len(got) == world and all(float(g.mean()) == float(i) for i, g in enumerate(got))
    out_q.put((rank, mine, tmax, total, ok_state, int(L[0, 0, 0]), int(R[0, 0, 0]), ok_gather))
    dist.destroy_process_group()


def test_world_size_2_gloo():
    world = 2
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    pairs = sorted(res[0][1] + res[1][1])
    assert pairs == list(range(11)) and not set(res[0][1]) & set(res[1][1])
    for r in res:
        assert r[2] == 2.0          # max over ranks of (1+rank)
        assert r[3] == 11.0         # pairs processed by the whole job
        assert r[4] and r[5] == 7 and r[6] == 9 and r[7]


def test_single_process_helpers_are_noops():
    sys.path.insert(0, ROOT)
    from ug_stereomatcher_amd import dist as ud
    assert ud.shard_pairs(5, 0, 1) == [0, 1, 2, 3, 4]
    assert ud.max_over_ranks(3.5) == 3.5 and ud.sum_over_ranks(2.0) == 2.0
    s = torch.ones((3, 2, 2))
    assert ud.broadcast_coarse_state(s) is s


def test_fovea_window_offsets():
    from ug_stereomatcher_amd import dist as ud
    assert ud.fovea_window_offsets(1, 4928, 3264, 615, 407) == [(0, 0)]
    offs = ud.fovea_window_offsets(8, 4928, 3264, 615, 407)
    assert len(offs) == 8 and offs[0] == (0, 0) and len(set(offs)) == 8
    assert all(abs(x) <= 4928 // 2 and abs(y) <= 3264 // 2 for x, y in offs)


# ---- the fovea-shard submit sequence, two ranks, two slots ---------------------------------------------

class OracleShardDriver:
    """The four calls UgsmShardDriver makes, answered by the CPU oracle's stage functions (test double: lives in tests/).
    Work is deferred to wait() / the next use of the slot, like the asynchronous library, so that a state buffer shared
    between slots -- the hazard the per-slot buffers remove -- would be caught."""

    def __init__(self, orc, levels, F, on_device=False):
        self.orc, self.levels, self.F = orc, levels, F
        self.slot = {}
        self.on_device = on_device  # answer of orders_on_device(): stands for "CUDA tensors over RCCL" (round 3: no host wait on the source rank)
        self.log = []

    def orders_on_device(self, state):
        return self.on_device

    def current_after_slot(self, slot):
        self.log.append(("current_after_slot", slot))  # (the double's coarse phase has already written the state: nothing to order)

    def slot_after_current(self, slot):
        self.log.append(("slot_after_current", slot))

    def submit_pyramids(self, slot, L, R, W, H, stride):
        self.log.append(("pyramids", slot))
        self._finish(slot)
        o = self.orc
        self.slot[slot] = dict(pl=o.pyramid(o.rgb_to_planes(L.numpy()), self.levels), pr=o.pyramid(o.rgb_to_planes(R.numpy()), self.levels),
                               W=W, H=H, fine=None)

    def submit_coarse(self, slot, state):
        self.log.append(("coarse", slot))
        o, s = self.orc, self.slot[slot]
        top = self.levels - 1
        cur = np.zeros_like(s["pl"][top])
        for i in range(top, self.F - 2, -1):
            mi = o.iterations_for_level(i)
            cur, _ = o.iterate_level(s["pl"][i], s["pr"][i], cur, mi, o.smooth_passes_for_level(i), i == top)
            if i > self.F - 1:
                cur = o.seed(cur, s["pl"][i - 1].shape[2], s["pl"][i - 1].shape[1])
        state.copy_(torch.from_numpy(cur))

    def wait(self, slot):
        self.log.append(("wait", slot))
        self._finish(slot)

    def submit_fine(self, slot, state, off, out):
        self.log.append(("fine", slot))
        self.slot[slot]["fine"] = (state, off, out)  # reads `state` later, like the stream-ordered copy in the library

    def _finish(self, slot):
        s = self.slot.get(slot)
        if not s or not s["fine"]:
            return
        state, off, out = s["fine"]
        s["fine"] = None
        o, F = self.orc, self.F
        fw, fh, ox, oy, cx, cy = o.fovea_geometry(s["W"], s["H"], self.levels, F, off[0], off[1])
        cur = state.numpy().copy()
        out[:, F - 1] = torch.from_numpy(cur)
        wup, hup = s["pl"][F - 2].shape[2], s["pl"][F - 2].shape[1]
        for i in range(F - 2, -1, -1):
            cur = o.seed_fovea(cur, wup, hup, cx[i], cy[i])
            L3 = np.ascontiguousarray(s["pl"][i][:, oy[i]:oy[i] + fh, ox[i]:ox[i] + fw])
            R3 = np.ascontiguousarray(s["pr"][i][:, oy[i]:oy[i] + fh, ox[i]:ox[i] + fw])
            cur, _ = o.iterate_level(L3, R3, cur, o.iterations_for_level(i), o.smooth_passes_for_level(i), False)
            out[:, i] = torch.from_numpy(cur)


def _shard_worker(rank, world, port, out_q):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                      UGSM_ORACLE_THREADS="2")
    from oracle import oracle as orc
    from ug_stereomatcher_amd import dist as ud, synth
    ud.init(backend="gloo")
    W, H, levels, F, slots, steps = 200, 150, 8, 4, 2, 3
    fw, fh, *_ = orc.fovea_geometry(W, H, levels, F)
    pairs = [tuple(torch.from_numpy(a) for a in synth.make_pair(W, H, 900 + j)[:2]) for j in range(2)]
    offsets = [(0, 0), (37, -21)]
    ok, n_got, no_host_wait = True, 0, True
    for on_device in (False, True):
        drv = OracleShardDriver(orc, levels, F, on_device)
        states = [torch.zeros((3, fh, fw)) for _ in range(slots)]
        outs = [torch.zeros((3, F, fh, fw)) for _ in range(slots)]
        got = []
        for k in range(steps):  # bench.py's submit(): slot free? then the step
            s = k % slots
            drv.wait(s)
            if k >= slots:
                got.append((k - slots, outs[s].clone()))
            L, R = pairs[k % 2]
            mark = len(drv.log)
            ud.fovea_shard_step(drv, s, L, R, W, H, 3 * W, states[s], offsets[rank], outs[s], rank)
            step_log = drv.log[mark:]
            if on_device:
                # the step itself never waits on the host, on any rank; the source rank orders the collective after its coarse phase
                # and every rank orders its fine phase after the collective, both on the device
                no_host_wait = no_host_wait and all(c[0] != "wait" for c in step_log)
                want = [("pyramids", s)] + ([("coarse", s)] if rank == 0 else []) + [("current_after_slot", s), ("slot_after_current", s), ("fine", s)]
                no_host_wait = no_host_wait and step_log == want
            elif rank == 0:
                no_host_wait = no_host_wait and ("wait", s) in step_log  # (the host path: the state is complete before gloo sends it)
        for k in range(max(steps - slots, 0), steps):
            drv.wait(k % slots)
            got.append((k, outs[k % slots].clone()))
        for k, st in got:
            L, R = pairs[k % 2]
            exp, _, _ = orc.match_foveated(L.numpy(), R.numpy(), levels, F, offsets[rank][0], offsets[rank][1])
            ok = ok and bool((st.numpy().view(np.uint32) == exp.view(np.uint32)).all())
        n_got += len(got)
        # the optional gather of every rank's stack on one consumer rank
        last = outs[(steps - 1) % slots]
        stacks = ud.gather_stacks(last, dst=0)
        if rank == 0:
            L, R = pairs[(steps - 1) % 2]
            ok = ok and stacks is not None and len(stacks) == world
            for r in range(world):
                exp, _, _ = orc.match_foveated(L.numpy(), R.numpy(), levels, F, offsets[r][0], offsets[r][1])
                ok = ok and bool((stacks[r].numpy().view(np.uint32) == exp.view(np.uint32)).all())
        else:
            ok = ok and stacks is None
    out_q.put((rank, ok and no_host_wait, n_got))
    dist.destroy_process_group()


def test_fovea_shard_sequence_world_size_2_two_slots():
    """coarse on rank 0 -> one broadcast -> fine on every rank, three steps over two slots: rank 0's window (the centred fovea)
    equals the one-shot foveated result bit for bit, and so does rank 1's off-centre window.  Run twice: with the host-side wait a
    gloo broadcast needs, and with the device-side ordering the RCCL path uses (the step then never calls wait: rank 0 keeps
    submitting); the stacks gathered on rank 0 equal each rank's own result."""
    world = 2
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_shard_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=300) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert res == [(0, True, 6), (1, True, 6)]
