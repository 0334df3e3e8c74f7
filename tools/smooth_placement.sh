#!/bin/bash
# VERDICT r05 #7 (time-boxed): why does K-smooth stretch 3.3 x with four calls in flight when K-cost stretches 1.55 x?  Placement or cost?
#   gpurun --timeout 900 -- 'bash tools/smooth_placement.sh gpurun_out/smooth_placement.txt'
# The timed region of bench.py (96 steps, calls of eight on four slots) under rocprofv3 --kernel-trace --stats, four ways:
#   four     the product: four calls in flight
#   one      one call in flight (--slots 1): the kernels uncontended
#   wg1      four in flight, K-smooth's 112 x 36 tile held to ONE workgroup per CU (UGSM_SMOOTH_LDS_EXTRA: 10 000 more LDS bytes per workgroup)
#   streams2 four slots on two streams (two calls in flight on the device, two queued behind them)
# and the same four without the profiler (384 steps) for the pairs/s.  Per-kernel sums = the profiler's "duration" of a launch, which runs
# from the moment its first workgroup is placed to the end of its last one: a launch that WAITS for room beside another kernel's
# workgroups is long without costing anything.
out=$1
export UGSM_DEV=1
R=$PWD
O=$R/gpurun_out/placement
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
B="--warmup 8 --no-cpu-baseline --no-service --single-pairs 0 --profile-pairs 0 --repeats 0 --other-steps 0 --steady-steps 0"
prof() { name=$1; shift; timeout -k 10 200 rocprofv3 --kernel-trace --stats -d $O/$name -o run --output-format csv -- python3 $R/bench.py --steps 96 $B "$@" > $O/$name.json 2> $O/$name.err; echo "[placement] traced $name ($(date +%T))"; }
bare() { name=$1; shift; timeout -k 10 200 python3 $R/bench.py --steps 384 $B "$@" > $O/${name}_bare.json 2>> $O/$name.err; echo "[placement] timed $name ($(date +%T))"; }
for mode in prof bare; do
  unset UGSM_SMOOTH_LDS_EXTRA
  $mode four
  $mode one --slots 1
  $mode streams2 --streams 2
  export UGSM_SMOOTH_LDS_EXTRA=10000
  $mode wg1
done
unset UGSM_SMOOTH_LDS_EXTRA
cd $R
python3 - $O > $out <<'PY'
import csv, glob, json, sys
O = sys.argv[1]
print("# K-smooth under co-scheduling: placement or cost?  (tools/smooth_placement.sh; rocprofv3 --kernel-trace --stats of bench.py's timed region,")
print("# 96 steps of calls of eight; pairs/s from the same region without the profiler, 384 steps)")
print(f"{'configuration':10s} {'pairs/s':>8s} {'traced':>8s} | " + " | ".join(f"{k:>26s}" for k in ("k_cost_march", "k_smooth_fused<112,36>", "k_cost_march4", "everything else")) + " | sum of all kernel durations")
rows = {}
for name in ("one", "four", "streams2", "wg1"):
    f = glob.glob(f"{O}/{name}/**/*kernel_stats.csv", recursive=True)
    if not f:
        continue
    acc = {"k_cost_march": [0, 0.0], "k_smooth_fused<112": [0, 0.0], "k_cost_march4": [0, 0.0], "other": [0, 0.0]}
    for r in csv.DictReader(open(f[0])):
        n = r["Name"].replace("void ", "").replace("ugsm::", "")
        key = "k_cost_march4" if n.startswith("k_cost_march4") else ("k_cost_march" if n.startswith("k_cost_march") else ("k_smooth_fused<112" if n.startswith("k_smooth_fused<112") else "other"))
        acc[key][0] += int(r["Calls"])
        acc[key][1] += float(r["TotalDurationNs"]) / 1e6
    def val(p):
        try:
            return json.load(open(p))["value"]
        except Exception:
            return float("nan")
    rows[name] = acc
    tot = sum(v[1] for v in acc.values())
    cells = " | ".join(f"{acc[k][1]:8.1f} ms {acc[k][0]:6d} x {1e3 * acc[k][1] / max(acc[k][0], 1):6.1f} us" for k in ("k_cost_march", "k_smooth_fused<112", "k_cost_march4", "other"))
    print(f"{name:10s} {val(f'{O}/{name}_bare.json'):8.1f} {val(f'{O}/{name}.json'):8.1f} | {cells} | {tot:8.1f} ms")
if "one" in rows:
    print("\nstretch against one call in flight (mean duration of a launch):")
    for name in ("four", "streams2", "wg1"):
        if name in rows:
            print(f"  {name:9s} " + "   ".join(f"{k} x {(rows[name][k][1] / max(rows[name][k][0], 1)) / (rows['one'][k][1] / max(rows['one'][k][0], 1)):.2f}" for k in ("k_cost_march", "k_smooth_fused<112", "k_cost_march4", "other")))
PY
cat $out
