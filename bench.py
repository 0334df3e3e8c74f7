#!/usr/bin/env python3
"""bench.py -- stereo pairs/s of the MI355X matcher on BASELINE.json's headline workload.

    python bench.py --gpus N --steps K --warmup W

One "step" = one synthetic 16 MP (4928x3264) stereo pair through the full-resolution 14-level
pyramid path (BASELINE.json configs[2], the configuration `metric` is quoted on), inputs already
resident in HBM.  The timed region is a loop over ugsm_enqueue_full + ugsm_next_done (include/ugsm.h, "the queue"): WHICH pairs share a
library call, which of the `--slots` slots takes it and when a slot is free again is decided inside libugsm.so -- this file holds no
call planning (rounds 3-4 kept that logic here: plan_calls / run / submit).  For N > 1 the driver launches
one rank per GPU (torch.distributed.run); run without a launcher, `--gpus N` starts the ranks itself.
Pairs are independent, so ranks share nothing on the data path ("weak" scaling, no collective); the
barrier and the max-over-ranks reduction go over RCCL.  --workload fovea-shard: one fovea window per rank, the coarse state broadcast by
the library's own RCCL communicator on the slot's stream (ugsm_submit_fovea_shard); `rccl_ranks` = what an all-reduce of ones counts.

How the line is put together (rank 0 prints ONE JSON line):
  value, ms_per_step  -- the timed region: W warm-up steps, then exactly K steps between two
                  barrier + synchronize brackets, NO event recording.  value_repeats: the same K steps timed twice more.
  steady_state -- the rate between the completions of the first and the last call that have the pipe full of full-size calls
                  behind them, inside the same regions (a region too short to have such calls: from one further region).
  roofline, kernels, event_pass -- a SEPARATE pass after the timed region: `--profile-pairs` CALLS (of `--batch` pairs each, the
                  timed region's full-size calls) submitted one at a time on slot 0; every launch carries two HIP events IN its
                  dispatch on that stream (hipExtLaunchKernelGGL: the kernel's own begin and end, what rocprofv3 --kernel-trace
                  reports for the same launches; rounds 1-3 and the first half of round 4 recorded the events AROUND the launch,
                  which adds the gaps to its neighbours: 6 us per launch), so the durations are uncontended.  roofline = the kernel with the largest total time
                  (the per-iteration cost kernel of the large levels): algorithmic bytes (48 B per
                  pixel-iteration) / duration, over all its launches and for level 0 alone;
                  HBM peak 8 TB/s.  kernels[] carries the same for every kernel (K-smooth: 24 B per pixel).
  valu_roofline -- both hot kernels are bound by VALU issue, not bytes: the modelled VALU time of
                  the level-0 launch (instruction counts of the kernel x per-instruction issue cost measured in
                  actual cycles, profiles/<PROFILE_TAG>_valu_model.json) against its measured duration.
  pcie_inclusive -- the drop-in service call (ugsm_match_full: host buffers in and out), pageable and
                  page-locked; never `value`.  device_copy_GBps: a 1 GiB device-to-device copy.
  cpu_baseline -- the CPU oracle (a port: the reference has no CPU matcher) timed on this host, rank 0 /
                  N=1 only, on a bounded sample (one 1920x1080 pair), 1 thread and all cores, median of 3.
  vs_baseline, baseline_comparison -- the reference publishes one figure for this workload: 10 s per 16 MP pair, `ros::WallTime` around ONE
                  BLOCKING library call, host images in, host planes out (README.md:15; UG_GPU_matcher.cpp:422-426).  `vs_baseline` is the SAME
                  BRACKET on this build -- ugsm_match_full from pageable host memory, one call at a time (pcie_inclusive) -- over that
                  figure; null when the run skipped that leg.  baseline_comparison names both brackets and also carries the device-resident
                  throughput ratio, which is NOT like for like.
  other_workloads -- BASELINE configs[3] (the 16 MP foveated stack: the reference's "3 s" target) and configs[1] (1920 x 1080), 64 pairs each
                  through the same queue on a context of their own: value, steady state and the blocking host-memory call (rank 0, N = 1).
Other workloads (--workload 1080p | fovea16mp | fovea-shard) as the headline line are parity/scaling cases.
"""
from __future__ import annotations

import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
BYTES_PER_PIXEL_ITER = 48.0  # L 12 + R 12 + (dx,dy,conf) in 12 + out 12
BYTES_PER_PIXEL = {"k_cost": BYTES_PER_PIXEL_ITER, "k_smooth": 24.0, "k_box": 24.0, "k_warp": 36.0, "k_sqblur": 24.0, "k_seed": 24.0}
REFERENCE_PAIRS_PER_S = {"full16mp": 0.1, "fovea16mp": 1.0 / 3.0}  # BASELINE.md section 1 (README.md:15-16): wall clock around one blocking call, copies included
DEFAULT_BATCH = {"full16mp": 8, "fovea16mp": 8, "1080p": 16}
PROFILE_TAG = "r06"

WORKLOADS = {
    "full16mp": dict(W=4928, H=3264, mode="full", desc="16MP (4928x3264) stereo pair, full-res 14-level pyramid"),
    "1080p": dict(W=1920, H=1080, mode="full", desc="1920x1080 stereo pair, full-res 14-level pyramid"),
    "fovea16mp": dict(W=4928, H=3264, mode="fovea", desc="16MP stereo pair, foveated stack, fovea 615x407, 7 fovea levels"),
    "fovea-shard": dict(W=4928, H=3264, mode="fovea-shard",
                        desc="16MP stereo pair, one fovea window per GPU, coarse state broadcast over RCCL"),
}


def whole_pair_algorithmic_bytes(W: int, H: int, levels: int, F: int) -> float:
    """SURVEY.md 8d, derived from the level sizes: matching 48 B per pixel-iteration; pyramid per image = rgb8 read (3 B/px of level 0)
    + level-0 planes written (12 B/px) + for every level >= 1 its parent read once and the level written (12 B/px each; level 1's parent
    is level 0, level i+2's parent is level i); seeding = 12 B x (source + destination pixels) per level transition.  Full mode only
    distinguishes F = 0; the foveated stack crops levels < F-1 to the fovea for matching and seeding (the pyramids are built whole)."""
    from ug_stereomatcher_amd import _lib
    ws, hs = _lib.level_dims(W, H, levels)
    px = [w * h for (w, h) in zip(ws, hs)]
    fpx = px[F - 1] if F >= 2 else None
    mpx = [(fpx if (fpx is not None and i < F - 1) else px[i]) for i in range(levels)]
    match = BYTES_PER_PIXEL_ITER * _lib.pixel_iterations(W, H, levels, F)
    pyr = 3.0 * px[0] + 12.0 * px[0]
    for i in range(1, levels):
        parent = px[0] if i == 1 else px[i - 2]
        pyr += 12.0 * (parent + px[i])
    seed = sum(12.0 * (mpx[i] + mpx[i - 1]) for i in range(1, levels))
    return match + 2.0 * pyr + seed


def kernel_source_sha16() -> str:
    """The hash tools/valu_model.py stamps its model with: sha256 over ug_stereomatcher_amd/csrc/*.hip, *.hpp."""
    import hashlib
    csrc = os.path.join(ROOT, "ug_stereomatcher_amd", "csrc")
    h = hashlib.sha256()
    for f in sorted(os.listdir(csrc)):
        if f.endswith((".hip", ".hpp")):
            h.update(open(os.path.join(csrc, f), "rb").read())
    return h.hexdigest()[:16]


def log(*a):
    print("[bench]", *a, file=sys.stderr, flush=True)


def cpu_model() -> str:
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def usable_cpus() -> int:
    """CPUs this process may actually use: the affinity mask and the cgroup CPU quota, whichever is smaller."""
    n = os.cpu_count() or 1
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except (AttributeError, OSError):
        pass
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    n = min(n, max(1, int(int(txt[0]) / int(txt[1]))))
            else:
                q = int(txt[0])
                per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
                if q > 0:
                    n = min(n, max(1, q // per))
        except (OSError, ValueError, IndexError):
            pass
    return max(1, n)


def cpu_baseline(wl: dict, runs: int = 3):
    """The CPU oracle (kind 'port': the reference has no CPU matcher).  All cores: timed on THE WORKLOAD'S OWN PAIR (16 MP for the
    headline line), median of `runs`, wall-clock bracket around the library call only (UG_GPU_matcher.cpp:422-426).
    One thread: a 1920x1080 pair of the same generator scaled by pixel-iterations (a 16 MP single-thread run takes minutes)."""
    from oracle import oracle as orc
    from ug_stereomatcher_amd import _lib, synth
    orc.build()
    F = 0 if wl["mode"] == "full" else 7
    # "all cores" = what this process may use, capped at the GPU box's CPU share of 16 per GPU (UGSM_CPU_THREADS overrides):
    # an OpenMP team the size of the host's 256 hardware threads inside a 16-CPU share does not finish in minutes
    ncpu = int(os.environ.get("UGSM_CPU_THREADS", min(usable_cpus(), 16)))

    def leg(threads, W, H, seed):
        L, R, _, _ = synth.make_pair(W, H, seed)
        orc.set_num_threads(threads)
        ts = []
        for _ in range(runs):
            t0 = time.perf_counter()
            if F == 0:
                orc.match_full(L, R, 14)
            else:
                orc.match_foveated(L, R, 14, F)
            ts.append(time.perf_counter() - t0)
        med = sorted(ts)[len(ts) // 2]
        pi_sample = _lib.pixel_iterations(W, H, 14, F)
        pi_unit = _lib.pixel_iterations(wl["W"], wl["H"], 14, F)
        return {"threads": threads, "size": f"{W}x{H}", "seconds_median": med, "seconds": ts, "value": (pi_sample / med) / pi_unit,
                "scaled_by_pixel_iterations": (W, H) != (wl["W"], wl["H"])}

    log(f"cpu_baseline: oracle on the {wl['W']}x{wl['H']} pair itself, {ncpu} threads x{runs} ...")
    allc = leg(ncpu, wl["W"], wl["H"], synth.BASE_SEED + 2)
    log(f"cpu_baseline: ... and 1 thread on one 1920x1080 pair x{runs} (scaled) ...")
    one = leg(1, 1920, 1080, synth.BASE_SEED + 2)
    return {"value": allc["value"], "unit": "pairs/s", "cores": ncpu, "kind": "port",
            "sample": f"all cores: the {wl['W']}x{wl['H']} {wl['mode']}-mode pair of the timed workload itself, median of {runs} runs "
                      f"({allc['seconds_median']:.2f} s each); one thread: one 1920x1080 pair scaled by pixel-iterations; wall-clock bracket "
                      "around the library call only, as UG_GPU_matcher.cpp:422-426",
            "host_cpu_count": os.cpu_count(), "usable_cpus": usable_cpus(), "cpu_model": cpu_model(), "all_cores": allc, "one_thread": one}


def steady_window(call_sizes, slots: int):
    """Analysis only -- the calls are formed by the library's queue (ugsm_completion.call_index / call_pairs say which).
    (lo, hi): indices of the LAST PAIR of the first and of the last call that have the pipe full of FULL-size calls behind them (the call
    itself and the `slots` calls after it are of the region's largest size, and at least `slots` calls precede it), or None when the region has
    no such middle.  The rate between those two completions counts no fill, no drain and no staggered or tapered call: such calls hold
    less work in flight, and an interval that touched them would be credited with work done outside it."""
    if not call_sizes:
        return None
    full = max(call_sizes)
    ok = [j for j in range(slots, len(call_sizes) - slots) if all(call_sizes[i] == full for i in range(j, j + slots + 1))]
    if len(ok) < 3:
        return None
    return sum(call_sizes[:ok[0] + 1]) - 1, sum(call_sizes[:ok[-1] + 1]) - 1


def spawn_ranks(args) -> int:
    """`--gpus N` without a launcher: start one rank per GPU the way the driver does, from a process that has not touched
    the GPU, and hand back the child's exit code."""
    port = os.environ.get("MASTER_PORT", "29533")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", port, os.path.abspath(__file__)] + sys.argv[1:]
    log("no launcher environment: starting", " ".join(cmd))
    return subprocess.run(cmd).returncode


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=384, help="pairs of the timed region (a short region under-reports: filling and draining the four slots costs about "
                    "one call's in-flight time whatever its length -- 3 %% of the driver's 20 steps, under 1 %% of 384; `steady_state` is the "
                    "figure without it)")
    ap.add_argument("--warmup", type=int, default=16)
    ap.add_argument("--workload", default="full16mp", choices=sorted(WORKLOADS))
    ap.add_argument("--slots", type=int, default=4, help="pairs in flight per GPU")
    ap.add_argument("--streams", type=int, default=0, help="HIP streams the slots are dealt onto (ugsm_config.streams; 0 = one per slot)")
    ap.add_argument("--batch", type=int, default=0, help="pairs per call at most (ugsm_submit_*_batch: the pairs of a call march through the levels in lockstep, "
                    "one launch per level of <= 9 Mpx for all of them); 1 = the single-pair calls of rounds 1-3; 0 = by the workload: 8 for 16 MP "
                    "full mode and for the foveated stack, 16 for 1080p (tools/ab.py, profiles/r04_ab_batch.txt; 16 MP full mode: 4 in the "
                    "round's first half -- 8 is +0.5 %% over 384 steps, +1.3 %% over 20, profiles/r04_ab_plan.txt)")
    ap.add_argument("--kernel-path", type=int, default=0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--profile-pairs", type=int, default=3, help="calls (of --batch pairs each) of the event pass after the timed region (0 = skip)")
    ap.add_argument("--repeats", type=int, default=2, help="extra timed repetitions of the K steps (value_repeats)")
    ap.add_argument("--no-service", action="store_true", help="skip the PCIe-inclusive service-call leg")
    ap.add_argument("--single-pairs", type=int, default=12, help="pairs of the un-instrumented one-slot leg (single_pair_no_events; 0 = skip)")
    ap.add_argument("--steady-steps", type=int, default=-1, help="steps of the further region `steady_state` is taken from when the timed region is too short to "
                    "have a steady window (-1 = max(steps, 48 x batch, 12 x slots x batch); 0 = no further region: steady_state null)")
    ap.add_argument("--other-steps", type=int, default=64, help="pairs of each leg of `other_workloads` (fovea16mp, 1080p; the default full16mp run on one GPU only; 0 = skip)")
    ap.add_argument("--no-events", action="store_true", help="same as --profile-pairs 0 --no-service --repeats 0 --single-pairs 0 --steady-steps 0 --other-steps 0 (bare throughput line)")
    args = ap.parse_args()
    if args.no_events:
        args.profile_pairs, args.no_service, args.repeats, args.single_pairs, args.steady_steps, args.other_steps = 0, True, 0, 0, 0, 0
    if args.gpus > 1 and "RANK" not in os.environ and "WORLD_SIZE" not in os.environ:
        sys.exit(spawn_ranks(args))
    # stdout carries exactly one line, the JSON result: whatever libraries print on file descriptor 1 on the way
    # (the RCCL version banner at communicator creation, driver notices) is sent to stderr instead
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)

    import numpy as np
    import torch
    import __graft_entry__ as ge
    from ug_stereomatcher_amd import _lib, dist as ud, synth

    rank, local_rank, world = ud.init()
    if world != args.gpus:
        log(f"warning: WORLD_SIZE={world} but --gpus {args.gpus}; reporting n_gpus={max(world, 1)}")
    n_gpus = max(world, 1)
    if rank == 0:
        ge.build_library()
    ud.barrier()
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (there is no CPU fallback); the CPU oracle is only the baseline leg")
    if os.environ.get("UGSM_BENCH_DEVICE") is not None:  # rehearsal of N ranks on a box with fewer GPUs
        local_rank = int(os.environ["UGSM_BENCH_DEVICE"])
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)

    wl = WORKLOADS[args.workload]
    W, H, mode = wl["W"], wl["H"], wl["mode"]
    if args.batch <= 0:
        args.batch = DEFAULT_BATCH.get(args.workload, 1)
    slots = max(1, args.slots)
    F = 7
    ctx = _lib.Context(device=local_rank, levels=14, fovea_levels=F, slots=slots, kernel_path=args.kernel_path, profile_events=0, streams=args.streams,
                       batch=max(1, min(args.batch, _lib.UGSM_MAX_BATCH)))
    fw, fh = _lib.fovea_dims(W, H, 14, F)

    # synthetic inputs: two distinct pairs per rank, resident in HBM before the timed region
    t0 = time.perf_counter()
    pairs, host_pair = [], None
    for j in range(2):
        L, R, _, _ = synth.make_pair(W, H, synth.BASE_SEED + 2 + 16 * j + rank)
        if j == 0:
            host_pair = (L, R)
        pairs.append((torch.from_numpy(L).to(dev), torch.from_numpy(R).to(dev)))
    stride = 3 * W
    if rank == 0:
        log(f"synthetic inputs ready in {time.perf_counter() - t0:.1f} s; workload: {wl['desc']}; slots={slots}")
    B = max(1, min(args.batch, _lib.UGSM_MAX_BATCH)) if mode != "fovea-shard" else 1
    # result buffers: a ring of (slots + 1) x batch, recycled in enqueue order -- the most the queue lets be outstanding (ugsm.h)
    cap = (slots + 1) * B
    shape = (3, H, W) if mode == "full" else (3, F, fh, fw)
    ring = [torch.empty(shape, dtype=torch.float32, device=dev) for _ in range(cap)]
    offsets = ud.fovea_window_offsets(n_gpus, W, H, fw, fh)
    my_off = offsets[rank % len(offsets)]
    # ranks RCCL itself counts (an all-reduce of ones), so that a multi-GPU run shows how many ranks the collective library really saw:
    # fovea-shard -- the LIBRARY's communicator (ugsm_shard_count_ranks; rank 0 makes the id, torch.distributed's control plane hands it
    # round, out of band, once); the other workloads -- torch.distributed's, when its backend is nccl (= RCCL; None over gloo / one process)
    rccl_ranks = None
    if mode == "fovea-shard":
        rccl_ranks = ud.shard_init(ctx, rank, n_gpus)
    elif ud.backend() == "nccl":
        rccl_ranks = int(round(ud.sum_over_ranks(1.0, dev)))
    torch.cuda.synchronize()

    done = []   # completions of the region being timed, in enqueue order (ugsm_completion: tag, call_index, call_pairs, done_ns)

    def fetch(block):
        while True:
            c = ctx.next_done(block)
            if c is None:
                return
            done.append((c.tag, c.call_index, c.call_pairs, c.done_ns))

    def run(n):
        """n pairs through the library: enqueue, fetch what has finished, flush at the end, drain.  (fovea-shard: slot-level calls, every
        rank the same sequence -- collectives match by order -- on the slots in rotation.)"""
        del done[:]
        if mode == "fovea-shard":
            for k in range(n):
                sl = k % slots
                ctx.check(ctx.lib.ugsm_wait(ctx.handle, sl))
                Lt, Rt = pairs[k % 2]
                ctx.submit_fovea_shard(sl, Lt.data_ptr(), Rt.data_ptr(), W, H, stride, my_off, ring[sl].data_ptr(), 0)
            ctx.check(ctx.lib.ugsm_wait_all(ctx.handle))
            return
        for k in range(n):
            Lt, Rt = pairs[k % 2]
            if mode == "full":
                ctx.enqueue_full(Lt.data_ptr(), Rt.data_ptr(), W, H, stride, ring[k % cap].data_ptr(), k)
            else:
                ctx.enqueue_foveated(Lt.data_ptr(), Rt.data_ptr(), W, H, stride, (0, 0), ring[k % cap].data_ptr(), k)
            fetch(False)
        ctx.flush()
        fetch(True)
        assert [d[0] for d in done] == list(range(n))

    def call_sizes_of(dn):
        out, last = [], None
        for (_, ci, cp, _) in dn:
            if ci != last:
                out.append(cp)
                last = ci
        return out

    def steady_state(n):
        """Pairs/s between the completions of the first and the last call of the region just timed that have the pipe full of full-size calls
        behind them (steady_window; SURVEY 8d defines the metric as steady state with the slots full): the region starts from a drained pipe
        and ends by draining it, which costs about one call's in-flight time whatever its length -- 4 % of 96 steps, more of 20 -- and hides
        changes of a few per cent.  None when the region is too short to have a middle.  Completion times are the
        library's (ugsm_completion.done_ns: CLOCK_MONOTONIC when it noticed the call complete)."""
        win = steady_window(call_sizes_of(done), slots)
        if len(done) != n or win is None or done[win[1]][3] <= done[win[0]][3]:
            return None
        return (win[1] - win[0]) / ((done[win[1]][3] - done[win[0]][3]) * 1e-9)

    def timed(n):
        torch.cuda.synchronize()
        ud.barrier()
        t0 = time.perf_counter()
        run(n)
        torch.cuda.synchronize()
        ud.barrier()
        return ud.max_over_ranks(time.perf_counter() - t0, dev)

    run(args.warmup)
    if B > 1:   # ... and enough pairs, untimed, that every slot has run a full-size call: whatever a first batched call sets up is not the workload
        run(2 * slots * B)
    dt = timed(args.steps)
    work = n_gpus if mode != "fovea-shard" else 1
    value = work * args.steps / dt
    timed_calls = call_sizes_of(done)
    steady = [steady_state(args.steps)]
    repeats = []
    for _ in range(max(0, args.repeats)):
        repeats.append(work * args.steps / timed(args.steps))
        steady.append(steady_state(args.steps))
    # A region too short to have a middle (the driver's --steps 20 is all fill and drain) gets its steady-state figure from one more region,
    # long enough to have one, run after the regions `value` and `value_repeats` come from and outside them.
    steady_region, steady_steps = "the timed region of `value`", args.steps
    steady_calls = timed_calls
    if args.steady_steps != 0 and mode != "fovea-shard" and steady_window(timed_calls, slots) is None:   # (the same answer on every rank: the queue's rule is deterministic)
        steady_steps = args.steady_steps if args.steady_steps > 0 else max(args.steps, 48 * B, 12 * slots * B)
        timed(steady_steps)
        steady[0] = steady_state(steady_steps)
        steady_calls = call_sizes_of(done)
        steady_region = (f"a further region of {steady_steps} steps after the timed ones (the timed region of {args.steps} steps has no call with the pipe "
                         "full of full-size calls on both sides); `value` and `value_repeats` do not include it")
    pi = _lib.pixel_iterations(W, H, 14, 0 if mode == "full" else F)

    result = {
        "metric": "stereo pairs/sec at 16MP full-res pyramid" if args.workload == "full16mp" else f"stereo pairs/sec ({args.workload})",
        "value": value,
        "unit": "pairs/s",
        "n_gpus": n_gpus,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": 1e3 * dt / args.steps,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,   # filled below from the blocking host-memory call (the reference's own bracket), when that leg runs
        "dtype": "f32",
        "data": "synthetic",
        "config": {"workload": wl["desc"], "slots_per_gpu": slots, "pairs_per_call_max": B, "pairs_in_flight_per_gpu": slots * B,
                   "streams_per_gpu": args.streams or slots, "kernel_path": args.kernel_path,
                   "pixel_iterations_per_pair": pi, "parallelism": f"replicas x{n_gpus}" if mode != "fovea-shard" else f"fovea windows x{n_gpus}",
                   "host_loop": "ugsm_enqueue_* + ugsm_next_done (the library forms the calls and owns the slots)" if mode != "fovea-shard"
                                else "ugsm_submit_fovea_shard on the slots in rotation (ncclBroadcast on the slot's stream, inside the library)"},
        # the calls the LIBRARY formed from the timed region's pairs (ugsm_completion.call_pairs), e.g. [4, 5, 7, 4] for 20 steps
        "calls_formed_by_the_library": timed_calls if len(timed_calls) <= 16 else {"calls": len(timed_calls), "head": timed_calls[:6], "tail": timed_calls[-6:]},
        # ranks RCCL counts by an all-reduce of ones (fovea-shard: the library's own communicator; else torch.distributed's nccl group; None
        # without one): = n_gpus when RCCL really spans them
        "rccl_ranks": rccl_ranks,
        "value_repeats": repeats,
        # this rank's pairs/s between the completions of pair slots + 1 and pair steps - slots, inside the same timed regions as `value`
        # and `value_repeats` (first entry: the region `value` comes from); x n_gpus for independent replicas
        "steady_state": {"value": (work * steady[0]) if steady[0] else None, "repeats": [(work * v) if v else None for v in steady[1:]],
                         "unit": "pairs/s", "region": steady_region, "steps": steady_steps, "calls": len(steady_calls),
                         "call_sizes_head_tail": [steady_calls[:6], steady_calls[-6:]],
                         "note": "ugsm_completion.done_ns (CLOCK_MONOTONIC when the library noticed a call complete) of every pair, in enqueue order; "
                                 "the rate between the completions of the first and the last call that have the pipe full of full-size calls "
                                 "behind them: no fill, no drain, no staggered calls"},
        "whole_pair_algorithmic_bytes": whole_pair_algorithmic_bytes(W, H, 14, 0 if mode == "full" else F),
        "whole_pair_algorithmic_GBps": whole_pair_algorithmic_bytes(W, H, 14, 0 if mode == "full" else F) * value / n_gpus / 1e9,
    }

    # ---- event pass: uncontended kernel durations of the launches the timed region makes (rank 0) -------------------------------
    # One CALL at a time on slot 0 -- a full-size call of B pairs, as the timed region submits them (round 4; rounds 1-3: single pairs) --
    # with two HIP events in every launch's dispatch (its own begin and end): the kernels and grids are those of the timed region and of
    # `rocprofv3 --kernel-trace --stats` of this command, whose average durations these must agree with.
    if rank == 0 and args.profile_pairs > 0 and mode != "fovea-shard":
        ctx.set_profile_events(2)
        ctx.reset_kernel_stats()
        t0 = time.perf_counter()
        for k in range(args.profile_pairs):   # (an explicit call of B pairs on slot 0: ugsm_submit_*_batch, the full-size call of the timed region)
            sel = [pairs[(k * B + b) % 2] for b in range(B)]
            dL, dR, dO = [p_[0].data_ptr() for p_ in sel], [p_[1].data_ptr() for p_ in sel], [o.data_ptr() for o in ring[:B]]
            if mode == "full":
                ctx.submit_full_batch(0, dL, dR, W, H, stride, dO)
            else:
                ctx.submit_foveated_batch(0, dL, dR, W, H, stride, None, dO)
            ctx.check(ctx.lib.ugsm_wait(ctx.handle, 0))
        t_single = (time.perf_counter() - t0) / (args.profile_pairs * B)
        ctx.set_profile_events(0)
        stats = ctx.kernel_stats()
        n_pairs = args.profile_pairs * B
        by_name = {}
        for s in stats:
            e = by_name.setdefault(s["name"], {"launches": 0, "total_ms": 0.0, "pixel_launches": 0.0, "levels": {}})
            e["launches"] += s["launches"]
            e["total_ms"] += s["total_ms"]
            e["pixel_launches"] += s["pixel_launches"]
            e["levels"][s["level"]] = s
        kernels = []
        for name, e in sorted(by_name.items(), key=lambda kv: -kv[1]["total_ms"]):
            bpp = next((v for k, v in BYTES_PER_PIXEL.items() if name.startswith(k)), None)
            row = {"name": name, "launches_per_pair": e["launches"] / n_pairs, "ms_per_pair": e["total_ms"] / n_pairs,
                   "avg_us": 1e3 * e["total_ms"] / e["launches"]}
            if bpp:
                row["bytes_per_pixel"] = bpp
                row["GBps"] = bpp * e["pixel_launches"] / (e["total_ms"] * 1e-3) / 1e9
                row["frac_of_hbm_peak"] = row["GBps"] / HBM_PEAK_GBS
                l0 = e["levels"].get(0)
                if l0 and l0["launches"]:
                    row["level0"] = {"launches_per_pair": l0["launches"] / n_pairs, "avg_us": 1e3 * l0["total_ms"] / l0["launches"],
                                     "GBps": bpp * l0["pixel_launches"] / (l0["total_ms"] * 1e-3) / 1e9}
                    row["level0"]["frac_of_hbm_peak"] = row["level0"]["GBps"] / HBM_PEAK_GBS
            kernels.append(row)
        result["kernels"] = kernels
        kernel_ms = sum(e["total_ms"] for e in by_name.values()) / n_pairs
        result["event_pass"] = {"ms_per_pair_wall": 1e3 * t_single, "pairs_per_s": 1.0 / t_single, "kernel_ms_per_pair": kernel_ms,
                                "pairs": n_pairs, "pairs_per_call": B,
                                "note": "one call of pairs_per_call pairs in flight; every launch carries two HIP events in its dispatch on its stream (hipExtLaunchKernelGGL: "
                                        "the kernel's own begin and end timestamps)"}
        # the dominant kernel: the cost kernel that carries most of the pair's pixel-iterations (the marching kernel of the large
        # levels; the LDS-tiled one only serves the latency-bound small levels)
        dom = max((k for k in kernels if k["name"].startswith("k_cost")), key=lambda k: by_name[k["name"]]["pixel_launches"], default=None)
        if dom:
            e = by_name[dom["name"]]
            traffic, prof = None, {}
            tpath = os.path.join(ROOT, "profiles", "pmc_traffic.json")
            if os.path.exists(tpath):
                try:
                    prof = json.load(open(tpath))
                    traffic = prof.get(args.workload, {}).get(dom["name"])
                except Exception:
                    prof = {}
            result["roofline"] = {
                "bound": "hbm", "achieved": dom["GBps"], "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": dom["GBps"] / HBM_PEAK_GBS,
                "traffic": traffic, "kernel": dom["name"], "launches_per_pair": dom["launches_per_pair"], "avg_launch_us": dom["avg_us"],
                "algorithmic_bytes_per_launch_avg": BYTES_PER_PIXEL_ITER * e["pixel_launches"] / e["launches"],
                "level0": dom.get("level0"),
                # the same figure level by level (level 0 = the full frame; each next level has half the pixels): the average above
                # weights every launch equally, and the coarser levels' launches last as long as one strip, not as long as their bytes
                "per_level": [{"level": lv, "launches_per_pair": st["launches"] / n_pairs, "avg_us": 1e3 * st["total_ms"] / st["launches"],
                               "frac": BYTES_PER_PIXEL_ITER * st["pixel_launches"] / (st["total_ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS}
                              for lv, st in sorted(e["levels"].items()) if st["launches"]],
                # the same launch-weighted figure over the levels of >= 0.2 Mpx only (levels 0-6 at 16 MP: the launches rounds 1-2 ran through
                # this kernel; from round 3 on the throughput choices also send the 50-200 k-pixel levels through it, 44 launches of
                # ~19 us that move 1.5-3 MB each and pull the all-launch average down while the pairs/s go up)
                "frac_levels_of_200k_pixels_and_more": (
                    (lambda big: (BYTES_PER_PIXEL_ITER * sum(st["pixel_launches"] for st in big) / (sum(st["total_ms"] for st in big) * 1e-3) / 1e9
                                  / HBM_PEAK_GBS) if big else None)(
                        [st for lv, st in e["levels"].items() if st["launches"] and st["pixel_launches"] / st["launches"] >= 200000])),
                "traffic_source": (f"profiles/pmc_traffic.json ({prof.get('_tag', '?')}: PMC passes of tools/profile_round.sh, not measured in this run)"
                                   if traffic is not None else None),
                "note": "algorithmic bytes (48 B per pixel-iteration x pixels of the launch, all its pairs) / HIP-event duration on the launching "
                        "stream, one call at a time after the timed region, the launches of the timed region (uncontended: agrees with rocprofv3 "
                        "--kernel-trace --stats of this command, which serialises launches); the kernel is bound by VALU issue, see valu_roofline"}
            vpath = os.path.join(ROOT, "profiles", f"{PROFILE_TAG}_valu_model.json")
            if os.path.exists(vpath) and dom.get("level0") and args.workload == "full16mp":
                try:
                    vm = json.load(open(vpath))
                    result["valu_roofline"] = valu_roofline(vm, by_name, n_pairs, W * H)
                except Exception as ex:  # a stale model file must not take the line down
                    result["valu_roofline"] = {"error": str(ex)}

    # ---- one pair at a time, un-instrumented, on a ONE-SLOT context: the reference's call pattern (UG_GPU_matcher.cpp:497-694) ------
    if rank == 0 and n_gpus == 1 and mode != "fovea-shard" and args.single_pairs > 0:
        ctx1 = _lib.Context(device=local_rank, levels=14, fovea_levels=F, slots=1, kernel_path=args.kernel_path, profile_events=0)
        out1 = ring[0]

        def one(k):
            Lt, Rt = pairs[k % 2]
            if mode == "full":
                ctx1.check(ctx1.lib.ugsm_submit_full(ctx1.handle, 0, Lt.data_ptr(), Rt.data_ptr(), W, H, stride, out1.data_ptr()))
            else:
                ctx1.check(ctx1.lib.ugsm_submit_foveated(ctx1.handle, 0, Lt.data_ptr(), Rt.data_ptr(), W, H, stride, 0, 0, out1.data_ptr(), None, None))
            ctx1.check(ctx1.lib.ugsm_wait(ctx1.handle, 0))
        for k in range(3):
            one(k)
        ts = []
        for k in range(args.single_pairs):
            t0 = time.perf_counter()
            one(k)
            ts.append(time.perf_counter() - t0)
        ctx1.close()
        med = sorted(ts)[len(ts) // 2]
        # ... and the same on THE TIMED CONTEXT (four slots, batch 8): since round 6 the kernel choices follow what is in flight, not
        # ugsm_config.slots, so a call alone gets the same choices there (it used to lose 11.6 %)
        ctx1, ts4 = ctx, []
        for k in range(3 + args.single_pairs):
            t0 = time.perf_counter()
            one(k)
            ts4.append(time.perf_counter() - t0)
        med4 = sorted(ts4[3:])[len(ts4[3:]) // 2]
        result["single_pair_no_events"] = {"ms_per_pair_median": 1e3 * med, "pairs_per_s": 1.0 / med, "ms_per_pair_mean": 1e3 * sum(ts) / len(ts),
                                           "pairs": len(ts),
                                           "on_the_timed_context": {"slots": slots, "batch": B, "ms_per_pair_median": 1e3 * med4, "pairs_per_s": 1.0 / med4,
                                                                    "vs_one_slot_context": med / med4},
                                           "note": "one-slot context, one pair in flight (submit, then wait), inputs resident in HBM, no event recorded "
                                                   "anywhere; host wall clock per pair; on_the_timed_context: the same calls, one at a time, on the "
                                                   "context the headline was timed on"}

    # ---- device copy rate and the PCIe-inclusive service call (rank 0, N = 1) ------------------------------------------
    if rank == 0 and n_gpus == 1 and not args.no_service:
        torch.cuda.synchronize()
        a = torch.zeros(1 << 28, dtype=torch.float32, device=dev)
        b = torch.empty_like(a)
        for _ in range(3):
            torch.add(a, 1.0, out=b)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(8):
            torch.add(a, 1.0, out=b)  # a streaming elementwise kernel: 1 GiB read + 1 GiB written per pass
        e1.record()
        torch.cuda.synchronize()
        result["device_copy_GBps"] = 8 * 2 * a.numel() * 4 / (e0.elapsed_time(e1) * 1e-3) / 1e9  # read + write
        del a, b
        if mode == "full":
            L, R = host_pair
            # one call at a time = the reference node's pattern: on a ONE-SLOT context, as that node would create it
            ctx_s = _lib.Context(device=local_rank, levels=14, fovea_levels=F, slots=1, kernel_path=args.kernel_path, profile_events=0)

            def call(Lh, Rh, oh):
                ts = []
                for _ in range(4):
                    o = oh if oh is not None else np.empty((3, H, W), np.float32)  # fresh, untouched result planes per call
                    t0 = time.perf_counter()
                    ctx_s.check(ctx_s.lib.ugsm_match_full(ctx_s.handle, Lh.ctypes.data, Rh.ctypes.data, W, H, stride,
                                                          o[0].ctypes.data, o[1].ctypes.data, o[2].ctypes.data))
                    ts.append(time.perf_counter() - t0)
                return sorted(ts[1:])[1]  # median of the last three
            t_page = call(L, R, None)
            pl, pr, po = ctx.host_array(L.shape, L.dtype), ctx.host_array(R.shape, R.dtype), ctx.host_array((3, H, W))
            pl[...] = L
            pr[...] = R
            t_pin = call(pl, pr, po)
            ctx_s.close()
            # several pairs in flight from host memory through the queue ("end-to-end from pinned host memory"): the host enqueues
            # and fetches, the library forms the calls.  Own contexts (batch 1 and batch 2: (slots + 1) x batch page-locked buffer sets of
            # 290 MB each); `managed` = pageable images in, library-owned page-locked planes out (what the node's topic path uses).
            def in_flight(Bh, managed):
                with _lib.Context(device=local_rank, levels=14, fovea_levels=F, slots=slots, kernel_path=args.kernel_path, batch=Bh) as ch:
                    cap_h = (slots + 1) * Bh
                    hb = [] if managed else [(ch.host_array(L.shape, L.dtype), ch.host_array(R.shape, R.dtype), ch.host_array((3, H, W))) for _ in range(cap_h)]
                    for (a_, b_, _) in hb:
                        a_[...] = L
                        b_[...] = R

                    def go(n):
                        t0 = time.perf_counter()
                        for k in range(n):
                            if managed:
                                ch.enqueue_full_managed(L, R, k)
                            else:
                                ch.enqueue_full_host(hb[k % cap_h][0], hb[k % cap_h][1], hb[k % cap_h][2], k)
                            while ch.next_done(False) is not None:
                                pass
                        ch.drain()
                        return (time.perf_counter() - t0) / n
                    go(2 * slots * Bh)
                    return min(go(12 * slots * Bh) for _ in range(2))
            t_piped = in_flight(1, False)
            Bh = 2
            t_piped_b = in_flight(Bh, False)
            t_managed = in_flight(Bh, True)
            result["pcie_inclusive"] = {"pageable_ms_per_pair": 1e3 * t_page, "pageable_pairs_per_s": 1.0 / t_page,
                                        "pinned_ms_per_pair": 1e3 * t_pin, "pinned_pairs_per_s": 1.0 / t_pin,
                                        "pinned_in_flight_pairs_per_s": 1.0 / t_piped, "pinned_in_flight_slots": slots,
                                        "pinned_in_flight_batched_pairs_per_s": 1.0 / t_piped_b, "pinned_in_flight_batch": Bh,
                                        "managed_in_flight_pairs_per_s": 1.0 / t_managed,
                                        "pinned_in_flight_GBps_over_pcie": (2 * H * stride + 12 * W * H) / min(t_piped, t_piped_b) / 1e9,
                                        "note": "ugsm_match_full on a one-slot context, one call at a time: rgb8 pair in (2 x 48 MB at 16 MP), three float planes out "
                                                "(193 MB); median of 3 calls; pageable = fresh result planes for every call, as the reference "
                                                "node allocates them (UG_GPU_matcher.cpp:414-418); the caller's free() is not in the call; "
                                                "pinned_in_flight = ugsm_enqueue_full_host + ugsm_next_done (the queue; page-locked images and planes of the "
                                                "host); managed_in_flight = ugsm_enqueue_full_managed (pageable images copied in by the library, results "
                                                "lent from its page-locked ring), batch 2"}

    # ---- the reference's own bracket: one blocking call, host memory in and out (README.md:15; UG_GPU_matcher.cpp:422-426) -------------
    ref = REFERENCE_PAIRS_PER_S.get(args.workload)
    if rank == 0 and ref:
        same = result.get("pcie_inclusive", {}).get("pageable_pairs_per_s")
        result["vs_baseline"] = (same / ref) if same else None
        result["baseline_comparison"] = {
            "reference_pairs_per_s": ref,
            "reference_bracket": "ros::WallTime around one blocking MatchGPULib::match call, host images in, host planes out, pyramid construction and every "
                                 "copy included (README.md:15: 10 s per 16 MP pair; UG_GPU_matcher.cpp:422-426); card not stated (GTX 750 Ti / 970 / 1080)",
            "same_bracket": {"pairs_per_s": same, "ratio": (same / ref) if same else None,
                             "what": "pcie_inclusive.pageable_pairs_per_s: ugsm_match_full, one call at a time, pageable images in, fresh pageable planes out"},
            "throughput": {"pairs_per_s": value, "ratio": value / ref,
                           "what": "`value`: inputs resident in HBM, 32 pairs in flight through the queue -- NOT the reference's bracket; quoted for completeness"},
            "vs_baseline_is": "same_bracket"}

    # ---- BASELINE configs[3] and configs[1] beside the headline (rank 0, N = 1, the default workload) -------------------------------------
    if rank == 0 and n_gpus == 1 and args.workload == "full16mp" and args.other_steps > 0:
        ctx.close()   # (its 43 GB of slot buffers are not needed any more; closed again below: a no-op)
        result["other_workloads"] = {}
        for name in ("fovea16mp", "1080p"):
            t0 = time.perf_counter()
            try:
                result["other_workloads"][name] = other_workload(name, args, local_rank, dev, host_pair if name == "fovea16mp" else None)
            except Exception as e:  # a side leg must never take the headline down
                result["other_workloads"][name] = {"error": str(e)}
            log(f"other workload {name}: {time.perf_counter() - t0:.1f} s")

    if rank == 0:
        if n_gpus == 1 and not args.no_cpu_baseline:
            try:
                result["cpu_baseline"] = cpu_baseline(wl)
            except Exception as e:  # the baseline leg must never take the measurement down
                result["cpu_baseline"] = {"value": None, "unit": "pairs/s", "cores": os.cpu_count(), "kind": "port", "sample": f"failed: {e}"}
        sys.stdout.flush()
        os.write(json_fd, (json.dumps(result) + "\n").encode())
    ctx.close()
    import torch.distributed as td
    if td.is_initialized():
        td.destroy_process_group()


def other_workload(name: str, args, local_rank: int, dev, host_pair=None) -> dict:
    """One of BASELINE's other single-GPU configurations through the same queue, on a context of its own: `steps` pairs resident in HBM (value,
    steady state), then the reference's bracket -- the blocking call from pageable host memory, one at a time, on the same context."""
    import numpy as np
    import torch
    from ug_stereomatcher_amd import _lib, synth
    wl = WORKLOADS[name]
    W, H, mode, F, slots = wl["W"], wl["H"], wl["mode"], 7, max(1, args.slots)
    B = DEFAULT_BATCH[name]
    n = args.other_steps
    fw, fh = _lib.fovea_dims(W, H, 14, F)
    L, R = host_pair if host_pair is not None else synth.make_pair(W, H, synth.BASE_SEED + 2)[:2]
    dL, dR = torch.from_numpy(L).to(dev), torch.from_numpy(R).to(dev)
    cap = (slots + 1) * B
    shape = (3, H, W) if mode == "full" else (3, F, fh, fw)
    ring = [torch.empty(shape, dtype=torch.float32, device=dev) for _ in range(cap)]
    with _lib.Context(device=local_rank, levels=14, fovea_levels=F, slots=slots, kernel_path=args.kernel_path, batch=B) as c:
        def run(m):
            done = []
            for k in range(m):
                if mode == "full":
                    c.enqueue_full(dL.data_ptr(), dR.data_ptr(), W, H, 3 * W, ring[k % cap].data_ptr(), k)
                else:
                    c.enqueue_foveated(dL.data_ptr(), dR.data_ptr(), W, H, 3 * W, (0, 0), ring[k % cap].data_ptr(), k)
                while True:
                    d = c.next_done(False)
                    if d is None:
                        break
                    done.append((d.call_index, d.call_pairs, d.done_ns))
            for d in c.drain():
                done.append((d.call_index, d.call_pairs, d.done_ns))
            return done
        run(2 * slots * B)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        run(n)
        torch.cuda.synchronize()
        value = n / (time.perf_counter() - t0)
        # steady state from a region long enough to have a middle (steady_window), outside `value`
        m = max(n, 12 * slots * B)
        done = run(m)
        sizes, last = [], None
        for (ci, cp, _) in done:
            if ci != last:
                sizes.append(cp)
                last = ci
        win = steady_window(sizes, slots)
        steady = (win[1] - win[0]) / ((done[win[1]][2] - done[win[0]][2]) * 1e-9) if win and done[win[1]][2] > done[win[0]][2] else None
        # the blocking call from host memory, fresh result planes per call (the node's service call and its one-at-a-time topic path)
        ts = []
        for _ in range(5):
            if mode == "full":
                o = np.empty((3, H, W), np.float32)
                t0 = time.perf_counter()
                c.check(c.lib.ugsm_match_full(c.handle, L.ctypes.data, R.ctypes.data, W, H, 3 * W, o[0].ctypes.data, o[1].ctypes.data, o[2].ctypes.data))
            else:
                o = np.empty((3, F, fh, fw), np.float32)
                t0 = time.perf_counter()
                c.check(c.lib.ugsm_match_foveated(c.handle, L.ctypes.data, R.ctypes.data, W, H, 3 * W, 0, 0, o[0].ctypes.data, o[1].ctypes.data, o[2].ctypes.data,
                                                  None, None))
            ts.append(time.perf_counter() - t0)
        t_call = sorted(ts[2:])[1]
    ref = REFERENCE_PAIRS_PER_S.get(name)
    return {"workload": wl["desc"], "value": value, "unit": "pairs/s", "steps": n, "slots": slots, "pairs_per_call_max": B,
            "steady_state": steady, "steady_steps": m,
            "blocking_call_ms": 1e3 * t_call, "blocking_call_pairs_per_s": 1.0 / t_call,
            "reference_pairs_per_s": ref, "vs_reference_same_bracket": (1.0 / t_call / ref) if ref else None,
            "note": "value: `steps` pairs resident in HBM through ugsm_enqueue_* / ugsm_next_done on a context of its own (no event recorded); "
                    "blocking_call: ugsm_match_full / ugsm_match_foveated from pageable host memory into fresh planes, one call at a time on the same "
                    "context (median of the last three of five) -- the bracket of the reference's README figure"}


def valu_roofline(vm: dict, by_name: dict, n_pairs: int, px0: int) -> dict:
    """Modelled VALU time of the level-0 launches of the hot kernels against their measured durations.
    vm = profiles/rNN_valu_model.json (tools/valu_model.py): per kernel the mean issue cost of a VALU instruction of its hot
    loops (ISA mix x tools/valubench.hip costs in actual cycles) and the in-kernel clock; profiles/pmc_traffic.json holds
    SQ_INSTS_VALU of the level-0 launch (PMC pass).  model = instructions x mean cost / (SIMDs x clock)."""
    built = kernel_source_sha16()
    out = {"source": vm.get("_source"), "clock_GHz": vm["clock_GHz"], "simds": vm["simds"], "kernels": [],
           "model_kernel_source_sha16": vm.get("_kernel_source_sha16"), "built_kernel_source_sha16": built,
           # instruction counts and mixes come from committed profiles: they describe the sources they were collected on
           "stale": vm.get("_kernel_source_sha16") != built}
    try:
        pj = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))
        insts, clocks = pj.get("valu_insts_level0", {}), pj.get("clock_GHz_level0", {})
    except Exception:
        insts, clocks = {}, {}
    for name, m in vm["kernels"].items():
        e = by_name.get(name)
        l0 = e["levels"].get(0) if e else None
        if not l0 or not l0["launches"] or name not in insts:
            continue
        measured_us = 1e3 * l0["total_ms"] / l0["launches"]
        clk = clocks.get(name) if isinstance(clocks.get(name), (int, float)) else vm["clock_GHz"]  # the clock held during that launch
        model_us = insts[name] * m["mean_cycles_per_valu"] / vm["simds"] / (clk * 1e3)
        guide_us = insts[name] * m.get("mean_cycles_per_valu_guide", m["mean_cycles_per_valu"]) / vm["simds"] / (clk * 1e3)
        out["kernels"].append({"name": name, "level0_measured_us": measured_us, "level0_valu_model_us": model_us, "frac": model_us / measured_us,
                               "level0_valu_model_us_guide_costs": guide_us, "frac_guide_costs": guide_us / measured_us,
                               "mean_cycles_per_valu_guide": m.get("mean_cycles_per_valu_guide"),
                               "clock_GHz": clk,
                               "valu_instructions_level0": insts[name], "mean_cycles_per_valu": m["mean_cycles_per_valu"],
                               "valu_lane_instructions_per_pixel": insts[name] * 64.0 / px0})
    out["note"] = ("frac = time the launch's VALU instruction stream needs at the MEASURED per-instruction issue costs (tools/valubench, actual cycles, "
                   "four waves per SIMD: 2.2 / 4.2 / 8.1 for full-rate / half-rate / transcendental instructions) / measured duration; "
                   "frac_guide_costs = the same at the hardware guide's nominal 2 / 4 / 8; near 1 = VALU-issue bound.  Instruction counts and mixes "
                   "come from profiles/ (PMC pass, ISA of the hot loops); `stale` = those were collected on other kernel sources than the ones built")
    return out


if __name__ == "__main__":
    main()
