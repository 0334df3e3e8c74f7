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
