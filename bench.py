#!/usr/bin/env python3
"""bench.py -- stereo pairs/s of the MI355X matcher on BASELINE.json's headline workload.

    python bench.py --gpus N --steps K --warmup W

One "step" = one synthetic 16 MP (4928x3264) stereo pair through the full-resolution 14-level
pyramid path (BASELINE.json configs[2], the configuration `metric` is quoted on), inputs already
resident in HBM, `--slots` pairs in flight on separate HIP streams.  For N > 1 the driver launches
one rank per GPU (torch.distributed.run); pairs are independent, so ranks share nothing on the data
path ("weak" scaling, no collective); the barrier and the max-over-ranks reduction go over RCCL.

Rank 0 prints ONE JSON line with, besides the contract fields,
  roofline     -- dominant kernel (the per-iteration cost kernel): algorithmic bytes
                  (48 B per pixel-iteration) / HIP-event duration of its launches,
                  measured live on slot 0's stream during the timed region; HBM peak 8 TB/s.
  cpu_baseline -- the CPU oracle (a port: the reference has no CPU matcher) timed on this host,
                  rank 0 / N=1 only, on a bounded sample; reported, not a target.
Other workloads (--workload 1080p | fovea16mp | fovea-shard) are parity/scaling cases, not the
headline line.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
BYTES_PER_PIXEL_ITER = 48.0  # L 12 + R 12 + (dx,dy,conf) in 12 + out 12
REFERENCE_PAIRS_PER_S = {"full16mp": 0.1, "fovea16mp": 1.0 / 3.0}  # BASELINE.md section 1 (README.md:15-16)

WORKLOADS = {
    "full16mp": dict(W=4928, H=3264, mode="full", desc="16MP (4928x3264) stereo pair, full-res 14-level pyramid"),
    "1080p": dict(W=1920, H=1080, mode="full", desc="1920x1080 stereo pair, full-res 14-level pyramid"),
    "fovea16mp": dict(W=4928, H=3264, mode="fovea", desc="16MP stereo pair, foveated stack, fovea 615x407, 7 fovea levels"),
    "fovea-shard": dict(W=4928, H=3264, mode="fovea-shard",
                        desc="16MP stereo pair, one fovea window per GPU, coarse state broadcast over RCCL"),
}


def log(*a):
    print("[bench]", *a, file=sys.stderr, flush=True)


def cpu_baseline(sample: str, threads: int, wl: dict):
    """Times the CPU oracle (kind 'port') on a bounded sample; returns the JSON object."""
    from oracle import oracle as orc
    from ug_stereomatcher_amd import _lib, synth
    orc.build()
    orc.set_num_threads(threads)
    if sample == "full":
        W, H = wl["W"], wl["H"]
    else:
        W, H = 1920, 1080
    L, R, _, _ = synth.make_pair(W, H, synth.BASE_SEED + 2)
    log(f"cpu_baseline: oracle on one {W}x{H} pair, {threads} threads ...")
    t0 = time.perf_counter()
    if wl["mode"] == "full":
        orc.match_full(L, R, 14)
        pi_sample = _lib.pixel_iterations(W, H, 14, 0)
        pi_unit = _lib.pixel_iterations(wl["W"], wl["H"], 14, 0)
    else:
        orc.match_foveated(L, R, 14, 7)
        pi_sample = _lib.pixel_iterations(W, H, 14, 7)
        pi_unit = _lib.pixel_iterations(wl["W"], wl["H"], 14, 7)
    dt = time.perf_counter() - t0
    # scaled to the metric's unit: pixel-iterations/s divided by the pixel-iterations of one workload pair
    value = (pi_sample / dt) / pi_unit
    return {"value": value, "unit": "pairs/s", "cores": threads, "kind": "port",
            "sample": f"one {W}x{H} {wl['mode']}-mode pair ({pi_sample} pixel-iterations) in {dt:.2f} s, "
                      f"scaled by pixel-iterations to the {wl['W']}x{wl['H']} workload ({pi_unit})",
            "seconds": dt}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=24)
    ap.add_argument("--warmup", type=int, default=4)
    ap.add_argument("--workload", default="full16mp", choices=sorted(WORKLOADS))
    ap.add_argument("--slots", type=int, default=4, help="pairs in flight per GPU (HIP streams)")
    ap.add_argument("--kernel-path", type=int, default=0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample", default="full", choices=["1080p", "full"], help="full = one pair of the workload itself (about 3-6 s at 16 MP)")
    ap.add_argument("--cpu-threads", type=int, default=16)
    ap.add_argument("--no-events", action="store_true", help="do not record HIP events on slot 0")
    ap.add_argument("--all-events", action="store_true", help="bracket every kernel class, not only the dominant one (slower)")
    args = ap.parse_args()
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
    if world != args.gpus and world > 1:
        log(f"warning: WORLD_SIZE={world} but --gpus {args.gpus}")
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
    slots = max(1, args.slots)
    F = 7
    ctx = _lib.Context(device=local_rank, levels=14, fovea_levels=F, slots=slots, kernel_path=args.kernel_path,
                       profile_events=0 if args.no_events else (2 if args.all_events else 1))
    fw, fh = _lib.fovea_dims(W, H, 14, F)

    # synthetic inputs: two distinct pairs per rank, resident in HBM before the timed region
    t0 = time.perf_counter()
    pairs = []
    for j in range(2):
        L, R, _, _ = synth.make_pair(W, H, synth.BASE_SEED + 2 + 16 * j + rank)
        pairs.append((torch.from_numpy(L).to(dev), torch.from_numpy(R).to(dev)))
    stride = 3 * W
    if rank == 0:
        log(f"synthetic inputs ready in {time.perf_counter() - t0:.1f} s; workload: {wl['desc']}; slots={slots}")
    if mode == "full":
        outs = [torch.empty((3, H, W), dtype=torch.float32, device=dev) for _ in range(slots)]
    else:
        outs = [torch.empty((3, F, fh, fw), dtype=torch.float32, device=dev) for _ in range(slots)]
    state = torch.empty((3, fh, fw), dtype=torch.float32, device=dev)
    offsets = ud.fovea_window_offsets(n_gpus, W, H, fw, fh)
    my_off = offsets[rank % len(offsets)]
    torch.cuda.synchronize()

    def submit(k):
        s = k % slots
        ctx.check(ctx.lib.ugsm_wait(ctx.handle, s))  # slot free?
        Lt, Rt = pairs[k % 2]
        if mode == "full":
            ctx.check(ctx.lib.ugsm_submit_full(ctx.handle, s, Lt.data_ptr(), Rt.data_ptr(), W, H, stride, outs[s].data_ptr()))
        elif mode == "fovea":
            ctx.check(ctx.lib.ugsm_submit_foveated(ctx.handle, s, Lt.data_ptr(), Rt.data_ptr(), W, H, stride, 0, 0,
                                                   outs[s].data_ptr(), None, None))
        else:  # fovea-shard: pyramids everywhere, coarse on rank 0, one RCCL broadcast, fine per window
            ctx.check(ctx.lib.ugsm_submit_pyramids(ctx.handle, s, Lt.data_ptr(), Rt.data_ptr(), W, H, stride))
            if rank == 0:
                ctx.check(ctx.lib.ugsm_submit_fovea_coarse(ctx.handle, s, state.data_ptr()))
                ctx.check(ctx.lib.ugsm_wait(ctx.handle, s))
            ud.broadcast_coarse_state(state, 0)
            torch.cuda.synchronize()
            ctx.check(ctx.lib.ugsm_submit_fovea_fine(ctx.handle, s, state.data_ptr(), my_off[0], my_off[1], outs[s].data_ptr()))

    def run(n):
        for k in range(n):
            submit(k)
        ctx.check(ctx.lib.ugsm_wait_all(ctx.handle))

    run(args.warmup)
    ctx.reset_kernel_stats()
    torch.cuda.synchronize()
    ud.barrier()
    t0 = time.perf_counter()
    run(args.steps)
    torch.cuda.synchronize()
    ud.barrier()
    dt = time.perf_counter() - t0
    dt = ud.max_over_ranks(dt, dev)

    stats = ctx.kernel_stats()
    value = n_gpus * args.steps / dt if mode != "fovea-shard" else args.steps / dt
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
        "vs_baseline": (value / REFERENCE_PAIRS_PER_S[args.workload]) if args.workload in REFERENCE_PAIRS_PER_S else None,
        "dtype": "f32",
        "data": "synthetic",
        "config": {"workload": wl["desc"], "pairs_in_flight_per_gpu": slots, "kernel_path": args.kernel_path,
                   "pixel_iterations_per_pair": pi, "parallelism": f"replicas x{n_gpus}" if mode != "fovea-shard" else f"fovea windows x{n_gpus}"},
    }
    if rank == 0:
        # roofline of the dominant kernel from slot 0's HIP events (same stream as the launches)
        dom = max((s for s in stats if s["name"].startswith("k_cost")), key=lambda s: s["total_ms"], default=None)
        kernels = []
        for s in stats:
            if s["launches"] == 0:
                continue
            bpp = {"k_cost": BYTES_PER_PIXEL_ITER, "k_smooth": 24.0, "k_box": 24.0, "k_warp": 36.0}.get(
                next((p for p in ("k_cost", "k_smooth", "k_box", "k_warp") if s["name"].startswith(p)), ""), None)
            kernels.append({"name": s["name"], "launches": s["launches"], "total_ms": s["total_ms"],
                            "avg_us": 1e3 * s["total_ms"] / s["launches"],
                            "GBps": (bpp * s["pixel_launches"] / (s["total_ms"] * 1e-3) / 1e9) if bpp and s["total_ms"] > 0 else None})
        if dom and dom["total_ms"] > 0:
            achieved = BYTES_PER_PIXEL_ITER * dom["pixel_launches"] / (dom["total_ms"] * 1e-3) / 1e9
            traffic = None
            tpath = os.path.join(ROOT, "profiles", "pmc_traffic.json")
            valu_busy = None
            if os.path.exists(tpath):
                try:
                    pj = json.load(open(tpath))
                    traffic = pj.get(args.workload, {}).get(dom["name"])
                    valu_busy = pj.get("valu_insts_per_simd_cycle_level0", {}).get(dom["name"])
                except Exception:
                    traffic = None
            result["roofline"] = {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                  "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "kernel": dom["name"],
                                  "launches": dom["launches"], "avg_launch_us": 1e3 * dom["total_ms"] / dom["launches"],
                                  "algorithmic_bytes_per_launch_avg": BYTES_PER_PIXEL_ITER * dom["pixel_launches"] / dom["launches"],
                                  "valu_insts_per_simd_cycle_level0": valu_busy,
                                  "note": "48 B per pixel-iteration x pixels per launch / HIP-event duration, slot 0, timed region; the kernel is "
                                          "bound by VALU issue, not bytes: valu_insts_per_simd_cycle_level0 = SQ_INSTS_VALU / (SQ_BUSY_CU_CYCLES x 4) "
                                          "from profiles/, times ~3.3 cycles per instruction of this mix = busy fraction (DESIGN.md section 6)"}
            # whole-pair figure: all algorithmic bytes of a pair / the pair's share of wall time
            result["whole_pair_algorithmic_GBps"] = (BYTES_PER_PIXEL_ITER * pi * value / n_gpus) / 1e9
        result["kernels"] = kernels
        if n_gpus == 1 and not args.no_cpu_baseline:
            try:
                result["cpu_baseline"] = cpu_baseline(args.cpu_sample, args.cpu_threads, wl)
            except Exception as e:  # the baseline leg must never take the measurement down
                result["cpu_baseline"] = {"value": None, "unit": "pairs/s", "cores": args.cpu_threads, "kind": "port",
                                          "sample": f"failed: {e}"}
        sys.stdout.flush()
        os.write(json_fd, (json.dumps(result) + "\n").encode())
    ctx.close()
    import torch.distributed as td
    if td.is_initialized():
        td.destroy_process_group()


if __name__ == "__main__":
    main()
