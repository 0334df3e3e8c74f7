#!/usr/bin/env python3
"""A/B of development overrides on ONE box: child processes under different UGSM_* settings (UGSM_DEV=1) take turns on the same
16 MP pairs, alternating order; rates per configuration as median / min / max over the processes.  Box-to-box differences
(+-1.5 %) are larger than most of the effects being compared, so the comparisons quoted in DESIGN.md from round 3 on come from
this tool, not from bench.py runs on different boxes.

usage: python tools/ab.py [--slots S] [--pairs P] [--rounds R] [--size W H] "name:VAR=val;VAR2=val" "name2:" ...
"""
import argparse, os, statistics, subprocess, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))

ap = argparse.ArgumentParser()
ap.add_argument("--slots", type=int, default=4)
ap.add_argument("--pairs", type=int, default=48)
ap.add_argument("--rounds", type=int, default=3)
ap.add_argument("--levels", type=int, default=14)
ap.add_argument("--size", type=int, nargs=2, default=(4928, 3264))
ap.add_argument("--streams", type=int, default=0, help="ugsm_config.streams (0 = one per slot)")
ap.add_argument("--fovea", type=int, default=0, help="foveated mode with this many fovea levels (0 = full mode)")
ap.add_argument("--batch", type=int, default=1, help="pairs per ugsm_submit_*_batch call (1 = the single-pair calls); a configuration may override it with BATCH=n and the slots with SLOTS=n")
ap.add_argument("--serial", type=int, default=0, help="1: one call at a time, waited for before the next goes out -- the node's service call (a configuration: SERIAL=1)")
ap.add_argument("--child", action="store_true", help="(internal) measure under the current environment, print the rate")
ap.add_argument("configs", nargs="*")
args = ap.parse_args()
W, H = args.size

if args.child:
    # one context per process: HIP deals a process's streams onto its hardware queues in creation order, and a second context's
    # streams share queues with the first one's (a context measured second in the same process ran 15 % slower whatever its settings)
    # NO_TORCH=1: device buffers through the C-ABI and no torch in the process -- libugsm.so then runs on the HIP runtime it links
    # (/opt/rocm), not on the one PyTorch's wheel bundles and loads first (ug_stereomatcher_amd/_lib.py, _share_torch_hip_runtime)
    no_torch = os.environ.get("NO_TORCH") == "1"
    if no_torch:
        os.environ["UGSM_NO_TORCH_RUNTIME"] = "1"
    else:
        import torch
        dev = torch.device("cuda:0")
    from ug_stereomatcher_amd import _lib, synth
    B = int(os.environ.get("BATCH", args.batch))
    args.slots = int(os.environ.get("SLOTS", args.slots))
    args.streams = int(os.environ.get("STREAMS", args.streams))
    serial = int(os.environ.get("SERIAL", args.serial))
    cap = max(B, int(os.environ.get("CTXBATCH", 0)))   # CTXBATCH=n: the context holds n pairs per slot while the calls carry B
    slot0 = os.environ.get("SLOT0") == "1"             # SLOT0=1 (with SERIAL=1): every call on slot 0
    class Buf:   # a device buffer with torch's data_ptr() face
        def __init__(self, p):
            self.p = p

        def data_ptr(self):
            return self.p
    F = args.fovea
    fw, fh = _lib.fovea_dims(W, H, args.levels, F) if F else (W, H)
    host_pairs = [synth.make_pair(W, H, synth.BASE_SEED + j)[:2] for j in range(max(args.slots, 2))]
    with _lib.Context(levels=args.levels, slots=args.slots, fovea_levels=F, streams=args.streams, batch=cap) as c:
        lib, h = c.lib, c.handle
        if no_torch:
            pairs = [(Buf(c.to_device(L)), Buf(c.to_device(R))) for L, R in host_pairs]
            outs = [[Buf(c.alloc(4 * 3 * (F * fh * fw if F else H * W))) for _ in range(B)] for _ in range(args.slots)]
        else:
            pairs = [(torch.from_numpy(L).to(dev), torch.from_numpy(R).to(dev)) for L, R in host_pairs]
            outs = [[torch.empty((3, F, fh, fw) if F else (3, H, W), dtype=torch.float32, device=dev) for _ in range(B)] for _ in range(args.slots)]
            torch.cuda.synchronize()

        def run(n):
            for i in range((n + B - 1) // B):          # n pairs, B per call
                s = 0 if (serial and slot0) else i % args.slots
                if serial and i > 0:
                    c.check(lib.ugsm_wait(h, 0 if slot0 else (i - 1) % args.slots))   # the call before has finished: this one has the chip to itself
                elif i >= args.slots:
                    c.check(lib.ugsm_wait(h, s))
                if B > 1:
                    sel = [pairs[(i * B + b) % len(pairs)] for b in range(B)]
                    dLs, dRs, dOs = [p[0].data_ptr() for p in sel], [p[1].data_ptr() for p in sel], [o.data_ptr() for o in outs[s]]
                    if F:
                        c.submit_foveated_batch(s, dLs, dRs, W, H, W * 3, None, dOs)
                    else:
                        c.submit_full_batch(s, dLs, dRs, W, H, W * 3, dOs)
                    continue
                dL, dR = pairs[i % len(pairs)]
                if F:
                    c.check(lib.ugsm_submit_foveated(h, s, dL.data_ptr(), dR.data_ptr(), W, H, W * 3, 0, 0, outs[s][0].data_ptr(), None, None))
                else:
                    c.check(lib.ugsm_submit_full(h, s, dL.data_ptr(), dR.data_ptr(), W, H, W * 3, outs[s][0].data_ptr()))
            c.check(lib.ugsm_wait_all(h))

        run(3 * args.slots * B)
        best = []
        for _ in range(3):
            t0 = time.perf_counter()
            run(args.pairs)
            best.append(args.pairs / (time.perf_counter() - t0))
    print("RATE", statistics.median(best))
    sys.exit(0)

cfgs = []
for spec in args.configs:
    name, _, rest = spec.partition(":")
    cfgs.append((name, dict(kv.split("=", 1) for kv in rest.split(";") if kv)))
rates = [[] for _ in cfgs]
for r in range(args.rounds):
    order = list(range(len(cfgs))) if r % 2 == 0 else list(reversed(range(len(cfgs))))
    for i in order:
        env = dict(os.environ, UGSM_DEV="1", **cfgs[i][1])
        out = subprocess.run([sys.executable, os.path.abspath(__file__), "--child", "--slots", str(args.slots), "--pairs", str(args.pairs), "--levels",
                              str(args.levels), "--size", str(W), str(H), "--fovea", str(args.fovea), "--streams", str(args.streams), "--batch", str(args.batch), "--serial", str(args.serial)], env=env, capture_output=True, text=True, timeout=300)
        line = [l for l in out.stdout.splitlines() if l.startswith("RATE")]
        if not line:
            print(out.stdout[-2000:], out.stderr[-2000:])
            sys.exit(1)
        rates[i].append(float(line[0].split()[1]))
        print(f"round {r} {cfgs[i][0]}: {rates[i][-1]:.2f}", flush=True)
base = statistics.median(rates[0])
print(f"{W}x{H}, {args.levels} levels{', fovea levels ' + str(args.fovea) if args.fovea else ''}, slots={args.slots}{' on ' + str(args.streams) + ' streams' if args.streams else ''}, {args.pairs} pairs x 3, {args.rounds} processes each (pairs/s: median  min  max  vs first)")
for (name, env), rs in zip(cfgs, rates):
    m = statistics.median(rs)
    print(f"  {name:28s} {m:8.2f} {min(rs):8.2f} {max(rs):8.2f}  {100.0 * (m / base - 1.0):+6.2f} %   {env}")
