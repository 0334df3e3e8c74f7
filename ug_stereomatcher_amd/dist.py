"""Multi-GPU plumbing: one process per GPU, torch.distributed (backend "nccl" == RCCL on ROCm).

What shards and what does not (DESIGN.md "Multi-GPU"):
  * One stereo pair does NOT shard: levels are strictly sequential and every iteration couples
    the whole field through the smoothing; a spatial split would need a >=9-px halo exchange
    176+ times per pair.
  * Independent pairs shard trivially: rank r takes pairs r, r+N, ... -- no data-path collective
    ("weak" scaling).  This is what `bench.py --gpus N` measures by default.
  * Fovea windows of ONE pair shard with a single exchange step: the coarse full-frame levels
    (top..F-1) are computed once, their (dx, dy, conf) state -- 3 x fovH x fovW float32, 3.0 MB at
    16 MP -- is broadcast over RCCL/xGMI, and each rank runs the fine levels F-2..0 for its own
    window.  Since round 5 the whole step is ONE library call, ugsm_submit_fovea_shard
    (csrc/ugsm_shard.cpp): the ncclBroadcast sits on the slot's own stream between the coarse and
    the fine phase.  The reference has exactly one, centred, fovea (MatchGPULib.cpp:1173-1176);
    with the centre window on every rank the result equals the single-GPU one, which is the
    parity check for this mode.

What is left here is the harness's control plane: process-group set-up, the barrier and the
max-over-ranks of bench.py's contract, the hand-over of the RCCL unique id, the window grid.
It works over gloo on CPU too (tests/test_dist_gloo.py).
"""
from __future__ import annotations

import os

import torch
import torch.distributed as dist


def env_rank_world():
    return int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))


def init(backend: str | None = None):
    """Initialises the default process group from the torchrun environment (no-op for 1 process)."""
    rank, local_rank, world = env_rank_world()
    backend = backend or os.environ.get("UGSM_DIST_BACKEND")  # rehearsals: "gloo" on a box with fewer GPUs than ranks
    force = os.environ.get("UGSM_FORCE_DIST") == "1"          # rehearsal: a 1-rank RCCL group on a 1-GPU box
    if (world > 1 or force) and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        kw = {}
        if backend == "nccl":
            torch.cuda.set_device(local_rank)
            kw["device_id"] = torch.device("cuda", local_rank)
        dist.init_process_group(backend=backend, rank=rank, world_size=world, **kw)
    return rank, local_rank, world


def _coll_device(device=None):
    """Scalars reduced over RCCL live on the GPU; over gloo (CPU rehearsals) on the host."""
    if dist.get_backend() != "nccl":
        return "cpu"
    return device or "cuda"


def backend():
    """The process group's backend ("nccl" = RCCL on ROCm, "gloo"), or None without one."""
    return dist.get_backend() if dist.is_initialized() else None


def shard_pairs(n_pairs: int, rank: int, world: int):
    """Pairs handled by `rank`: r, r+world, ... (independent pairs, no collective)."""
    return list(range(rank, n_pairs, world))


def barrier():
    if dist.is_initialized():
        dist.barrier()


def max_over_ranks(seconds: float, device=None) -> float:
    """bench.py contract: the timed region's duration is the MAX over ranks."""
    if not dist.is_initialized():
        return seconds
    t = torch.tensor([seconds], dtype=torch.float64, device=_coll_device(device))
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def sum_over_ranks(value: float, device=None) -> float:
    if not dist.is_initialized():
        return value
    t = torch.tensor([value], dtype=torch.float64, device=_coll_device(device))
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return float(t.item())


def exchange_shard_id(make_id, rank: int, src: int = 0) -> bytes:
    """The 128 bytes of an RCCL unique id from rank `src` (which makes it: make_id()) to every rank, over torch.distributed's control
    plane -- out of band, once; works over gloo and nccl alike.  A C++ host does the same with whatever it has (a ROS parameter, a
    file, MPI)."""
    box = [make_id() if rank == src else None]
    if dist.is_initialized():
        dist.broadcast_object_list(box, src=src)
    if not isinstance(box[0], (bytes, bytearray)) or len(box[0]) != 128:
        raise RuntimeError("shard id exchange failed")
    return bytes(box[0])


def shard_init(ctx, rank: int, world: int, src: int = 0) -> int:
    """Joins the context to the library's own RCCL communicator (ugsm_shard_init; include/ugsm.h): rank `src` makes the id, everyone joins,
    and the number of ranks RCCL itself counts (an all-reduce of ones on the slot's stream) is returned -- bench.py's `rccl_ranks`.
    From here on the exchange of the fovea shard is ugsm_submit_fovea_shard: ncclBroadcast on the slot's stream inside the library, no
    Python on the data path (rounds 3-4 issued the broadcast from torch's stream and ordered it against the slot with two events)."""
    from . import _lib
    ctx.shard_init(exchange_shard_id(_lib.shard_unique_id, rank, src), rank, max(world, 1))
    return ctx.shard_count_ranks()


def fovea_window_offsets(n_windows: int, W: int, H: int, fovW: int, fovH: int):
    """Window-centre offsets (level-0 pixels from the image centre) for `n_windows` foveae tiling
    the frame on a near-square grid; window 0 is always the reference's centred fovea."""
    if n_windows <= 1:
        return [(0, 0)]
    import math
    cols = int(math.ceil(math.sqrt(n_windows * W / max(H, 1))))
    cols = max(1, min(cols, n_windows))
    rows = int(math.ceil(n_windows / cols))
    offs = [(0, 0)]
    for r in range(rows):
        for c in range(cols):
            ox = int(round((c + 0.5) / cols * W - W / 2))
            oy = int(round((r + 0.5) / rows * H - H / 2))
            if (ox, oy) != (0, 0):
                offs.append((ox, oy))
    return offs[:n_windows]
