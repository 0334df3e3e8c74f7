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
    window (ugsm_submit_fovea_coarse / ugsm_submit_fovea_fine).  The reference has exactly one,
    centred, fovea (MatchGPULib.cpp:1173-1176); with the centre window on every rank the result
    equals the single-GPU one, which is the parity check for this mode.

Everything here works on CPU tensors with the gloo backend too (tests/test_dist_gloo.py).
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


def broadcast_coarse_state(state: torch.Tensor, src: int = 0) -> torch.Tensor:
    """The one exchange step of fovea sharding: level F-1's (dx, dy, conf), shape (3, fovH, fovW)
    float32, from the rank that computed the coarse levels to every rank."""
    assert state.dtype == torch.float32 and state.dim() == 3 and state.shape[0] == 3 and state.is_contiguous()
    if dist.is_initialized():
        dist.broadcast(state, src=src)
    return state


def broadcast_pair(rgbL: torch.Tensor, rgbR: torch.Tensor, src: int = 0):
    """Optional: ship the rgb8 pair (2 x 48.3 MB at 16 MP) from the rank that received it."""
    assert rgbL.dtype == torch.uint8 and rgbR.dtype == torch.uint8
    if dist.is_initialized():
        dist.broadcast(rgbL, src=src)
        dist.broadcast(rgbR, src=src)
    return rgbL, rgbR


def gather_stacks(stack: torch.Tensor, dst: int = 0):
    """Optional: collect every rank's fovea stack (21 MB each at 16 MP) on one consumer rank."""
    if not dist.is_initialized():
        return [stack]
    world = dist.get_world_size()
    if dist.get_backend() == "nccl":
        out = [torch.empty_like(stack) for _ in range(world)]
        dist.all_gather(out, stack)
        return out if dist.get_rank() == dst else None
    out = [torch.empty_like(stack) for _ in range(world)] if dist.get_rank() == dst else None
    dist.gather(stack, out, dst=dst)
    return out


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


# ---- the fovea-shard step (bench.py --workload fovea-shard) ------------------------------------------

class UgsmShardDriver:
    """The C-ABI calls of one fovea-shard step on a Context, addressed by slot.  Tensors are torch CUDA tensors;
    tests drive fovea_shard_step with a double that has the same methods (tests/test_dist_gloo.py)."""

    def __init__(self, ctx):
        self.ctx = ctx
        self._streams = {}

    def submit_pyramids(self, slot, L, R, W, H, stride):
        c = self.ctx
        c.check(c.lib.ugsm_submit_pyramids(c.handle, slot, L.data_ptr(), R.data_ptr(), W, H, stride))

    def submit_coarse(self, slot, state):
        c = self.ctx
        c.check(c.lib.ugsm_submit_fovea_coarse(c.handle, slot, state.data_ptr()))

    def wait(self, slot):
        c = self.ctx
        c.check(c.lib.ugsm_wait(c.handle, slot))

    def submit_fine(self, slot, state, off, out):
        c = self.ctx
        c.check(c.lib.ugsm_submit_fovea_fine(c.handle, slot, state.data_ptr(), off[0], off[1], out.data_ptr()))

    # ---- device-side ordering between a slot's stream and the stream the collective is launched from (round 3) -------------
    def orders_on_device(self, state: torch.Tensor) -> bool:
        """True when the exchange can be ordered against the slot's work on the device: a CUDA tensor sent over RCCL (a gloo
        broadcast works through the host and needs the data complete when it is called)."""
        return bool(state.is_cuda) and dist.is_initialized() and dist.get_backend() == "nccl"

    def _stream(self, slot):
        st = self._streams.get(slot)
        if st is None:
            import ctypes
            p = ctypes.c_void_p()
            c = self.ctx
            c.check(c.lib.ugsm_slot_stream(c.handle, slot, ctypes.byref(p)))
            st = self._streams[slot] = torch.cuda.ExternalStream(p.value)
        return st

    def current_after_slot(self, slot):
        """The current torch stream (the one the collective synchronises with) waits, on the device, for everything enqueued on the slot."""
        ev = torch.cuda.Event()
        ev.record(self._stream(slot))
        torch.cuda.current_stream().wait_event(ev)

    def slot_after_current(self, slot):
        """The slot's stream waits, on the device, for everything enqueued on the current torch stream (the finished collective)."""
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream())
        self._stream(slot).wait_event(ev)


def _collective_done(t: torch.Tensor):
    """Host waits for the collective that was just enqueued on the current stream -- for it alone, not for the device:
    the other slots' kernels keep running.  (gloo: the call has already completed.)"""
    if t.is_cuda:
        ev = torch.cuda.Event()
        ev.record()
        ev.synchronize()


def fovea_shard_step(drv, slot: int, L, R, W: int, H: int, stride: int, state: torch.Tensor, off, out, rank: int, src: int = 0):
    """One pair, one fovea window per rank: every rank builds the pyramids; rank `src` runs the coarse full-frame levels into
    `state`; ONE broadcast of `state` (3 x fovH x fovW floats); every rank runs the fine levels of its window `off` into `out`.
    `state` belongs to `slot`: the caller reuses a slot (and its state buffer) only after drv.wait(slot), i.e. after the fine
    phase that reads the state has finished, so a later step's broadcast can never overwrite a state still in use; nothing
    here synchronises the whole device, so the slots overlap.

    Over RCCL the step never blocks the host (round 3): the collective's stream waits for the slot's coarse
    phase through an event, and the slot's stream waits for the collective the same way, so rank `src` goes on submitting the
    next pair's pyramids and coarse levels while this pair's state is still being computed and sent.  Over gloo (CPU rehearsals)
    the data must be complete when the call is made: there the host waits for the slot first."""
    drv.submit_pyramids(slot, L, R, W, H, stride)
    on_device = bool(getattr(drv, "orders_on_device", lambda t: False)(state))
    if rank == src:
        drv.submit_coarse(slot, state)
        if on_device:
            drv.current_after_slot(slot)  # (device-side: the broadcast reads the state after the coarse phase has written it)
        else:
            drv.wait(slot)  # the state is complete before it is sent
    elif on_device:
        drv.current_after_slot(slot)  # (a receiving rank: whatever still reads this slot's state buffer has been enqueued before)
    broadcast_coarse_state(state, src)
    if on_device:
        drv.slot_after_current(slot)  # the fine phase starts, on the device, when the state has arrived
    else:
        _collective_done(state)
    drv.submit_fine(slot, state, off, out)
