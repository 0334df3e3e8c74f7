"""N>1 plumbing on CPU: a world_size-2 gloo process group exercising bench.py's control plane (pair sharding, barrier, max-over-ranks
timing, the hand-over of the RCCL unique id) and a MODEL of the fovea-shard protocol -- pyramids, coarse levels on the source rank, ONE
broadcast of the coarse state, fine levels of every rank's own window -- run by the CPU oracle over gloo: it shows that the state of
level F-1 is all a rank needs from the source, for any window.  The library's implementation of that protocol (ugsm_submit_fovea_shard:
ncclBroadcast on the slot's stream, csrc/ugsm_shard.cpp) needs a GPU and is covered by tests/test_gpu_dist.py."""
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
    ud.barrier()
    # the RCCL unique id of the library's communicator: made on rank 0 only, 128 bytes, the same on every rank afterwards
    made = []

    def make_id():
        made.append(rank)
        return bytes(range(128))
    ident = ud.exchange_shard_id(make_id, rank, 0)
    ok_id = ident == bytes(range(128)) and made == ([0] if rank == 0 else [])
    out_q.put((rank, mine, tmax, total, ok_id))
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
        assert r[4]                 # the shard id reached every rank, made once


def test_single_process_helpers_are_noops():
    sys.path.insert(0, ROOT)
    from ug_stereomatcher_amd import dist as ud
    assert ud.shard_pairs(5, 0, 1) == [0, 1, 2, 3, 4]
    assert ud.max_over_ranks(3.5) == 3.5 and ud.sum_over_ranks(2.0) == 2.0
    assert ud.exchange_shard_id(lambda: b"x" * 128, 0) == b"x" * 128   # (no process group: the id stays where it was made)


def test_fovea_window_offsets():
    from ug_stereomatcher_amd import dist as ud
    assert ud.fovea_window_offsets(1, 4928, 3264, 615, 407) == [(0, 0)]
    offs = ud.fovea_window_offsets(8, 4928, 3264, 615, 407)
    assert len(offs) == 8 and offs[0] == (0, 0) and len(set(offs)) == 8
    assert all(abs(x) <= 4928 // 2 and abs(y) <= 3264 // 2 for x, y in offs)


# ---- a model of the fovea-shard protocol, two ranks, two slots ---------------------------------------------

class OracleShardModel:
    """ugsm_submit_fovea_shard's stream order -- pyramids; coarse levels on the source rank; one broadcast of the state; fine levels of this
    rank's window -- with the CPU oracle's stage functions for the phases and a gloo broadcast for the exchange (test double: lives in
    tests/).  The fine phase is deferred to wait() / the next use of the slot, like the asynchronous library, so that a state buffer shared
    between slots -- the hazard the per-slot state buffers of csrc/ugsm_shard.cpp remove -- would be caught."""

    def __init__(self, orc, levels, F, slots, fw, fh):
        self.orc, self.levels, self.F = orc, levels, F
        self.slot = {}
        self.state = [torch.zeros((3, fh, fw)) for _ in range(slots)]   # per slot, as in the library

    def submit_fovea_shard(self, slot, L, R, W, H, off, out, rank, src=0):
        self.wait(slot)
        o = self.orc
        s = self.slot[slot] = dict(pl=o.pyramid(o.rgb_to_planes(L.numpy()), self.levels), pr=o.pyramid(o.rgb_to_planes(R.numpy()), self.levels),
                                   W=W, H=H, fine=None)
        state = self.state[slot]
        if rank == src:
            top = self.levels - 1
            cur = np.zeros_like(s["pl"][top])
            for i in range(top, self.F - 2, -1):
                mi = o.iterations_for_level(i)
                cur, _ = o.iterate_level(s["pl"][i], s["pr"][i], cur, mi, o.smooth_passes_for_level(i), i == top)
                if i > self.F - 1:
                    cur = o.seed(cur, s["pl"][i - 1].shape[2], s["pl"][i - 1].shape[1])
            state.copy_(torch.from_numpy(cur))
        dist.broadcast(state, src=src)
        s["fine"] = (state, off, out)  # reads `state` later, like the stream-ordered fine phase in the library

    def wait(self, slot):
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
    ok = True
    drv = OracleShardModel(orc, levels, F, slots, fw, fh)
    outs = [torch.zeros((3, F, fh, fw)) for _ in range(slots)]
    got = []
    for k in range(steps):  # bench.py's loop: slot free? then the step
        s = k % slots
        drv.wait(s)
        if k >= slots:
            got.append((k - slots, outs[s].clone()))
        L, R = pairs[k % 2]
        drv.submit_fovea_shard(s, L, R, W, H, offsets[rank], outs[s], rank)
    for k in range(max(steps - slots, 0), steps):
        drv.wait(k % slots)
        got.append((k, outs[k % slots].clone()))
    for k, st in got:
        L, R = pairs[k % 2]
        exp, _, _ = orc.match_foveated(L.numpy(), R.numpy(), levels, F, offsets[rank][0], offsets[rank][1])
        ok = ok and bool((st.numpy().view(np.uint32) == exp.view(np.uint32)).all())
    out_q.put((rank, ok, len(got)))
    dist.destroy_process_group()


def test_fovea_shard_protocol_world_size_2_two_slots():
    """coarse on rank 0 -> one broadcast -> fine on every rank, three steps over two slots: rank 0's window (the centred fovea)
    equals the one-shot foveated result bit for bit, and so does rank 1's off-centre window."""
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
    assert res == [(0, True, 3), (1, True, 3)]
