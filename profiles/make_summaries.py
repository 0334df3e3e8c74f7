#!/usr/bin/env python3
"""Turns one gpurun_out/prof_rNN directory (bench JSONs, rocprofv3 kernel trace, PMC passes) into the
committed summaries under profiles/.   usage: make_summaries.py gpurun_out/prof_r01d r01"""
import collections
import csv
import glob
import json
import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
base, tag = sys.argv[1], sys.argv[2]
out = os.path.join(ROOT, "profiles")
shutil.copy(glob.glob(base + "/trace/**/*kernel_stats.csv", recursive=True)[0], f"{out}/{tag}_kernel_stats.csv")
for f in ("bench_default.json", "bench_under_rocprof.json"):
    shutil.copy(f"{base}/{f}", f"{out}/{tag}_{f}")
for f in ("kbench_age_16mp.txt", "kbench_graph.txt", "census_16mp.txt", "kbench_16mp.txt", "kbench_smooth_16mp.txt", "kbench_small.txt", "kbench_aux_16mp.txt", "kbench_strips.txt", "level_breakdown.txt", "valubench.txt", "ldsbench.txt", "service_latency.txt", "bench_slots1.json", "bench_1080p.json",
          "bench_fovea16mp.json", "ab_policies.txt", "ab_batch.txt", "bench_steps20.json", "bench_batch1.json", "kbench_smooth_pipe.txt", "rehearsal_2ranks.txt", "rccl_and_contexts.txt",
          "kbench_march4.txt", "queue_probe.txt", "ab_queue.txt", "kbench_march_issue_raw.txt", "ab_alone.txt", "kbench_two_streams.txt", "smooth_placement.txt"):
    if os.path.exists(f"{base}/{f}"):
        shutil.copy(f"{base}/{f}", f"{out}/{tag}_{f}")


def load(d):
    f = glob.glob(f"{base}/{d}/**/*counter_collection.csv", recursive=True)[0]
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("ugsm::", "")
        agg[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return agg


from ug_stereomatcher_amd import _lib  # noqa: E402
pi = _lib.pixel_iterations(4928, 3264, 14, 0)
rd, wr, fe = load("pmc_rdreq"), load("pmc_write"), load("pmc_fetch")
lines = [f"# {tag}: HBM traffic per kernel (PMC, separate passes; `bench.py --steps 8 --warmup 0 --slots 1 --batch 8`: the launches of the timed "
         "region -- levels 1-13 carry eight pairs per launch, level 0 one)\n",
         "Read bytes = 32*RDREQ_32B + 64*(RDREQ - RDREQ_32B - RDREQ_128B) + 128*RDREQ_128B (TCC_EA0_*_sum).  FETCH_SIZE is shown "
         "beside it doubled: on gfx950 it tallies 128-B requests at 64 B, i.e. exactly half (MI355X_MICROARCH.md, HBM section) -- the two "
         "agree.  Write bytes = WRITE_SIZE x 1024.  Algorithmic = 48 B x pixels of the launch (SURVEY.md 8d).\n",
         "| kernel | launches | read MB/launch | 2 x FETCH_SIZE MB/launch | write MB/launch | algorithmic MB/launch | traffic / algorithmic |",
         "|---|---|---|---|---|---|---|"]
traffic, traffic_n = {}, {}
for k in sorted(rd, key=lambda k: -sum(rd[k]["TCC_EA0_RDREQ_sum"])):
    c = rd[k]
    n = len(c["TCC_EA0_RDREQ_sum"])
    tot, r32, r128 = sum(c["TCC_EA0_RDREQ_sum"]), sum(c["TCC_EA0_RDREQ_32B_sum"]), sum(c["TCC_EA0_RDREQ_128B_sum"])
    rb = 32 * r32 + 64 * (tot - r32 - r128) + 128 * r128
    f = sum(fe[k]["FETCH_SIZE"]) * 1024 * 2
    w = sum(wr[k]["WRITE_SIZE"]) * 1024
    a = ratio = ""
    if k.startswith("k_cost") or k.startswith("k_smooth"):
        traffic[k.split("<")[0]] = traffic.get(k.split("<")[0], 0.0) + (rb + w)
        traffic_n[k.split("<")[0]] = traffic_n.get(k.split("<")[0], 0) + n
    lines.append(f"| {k} | {n} | {rb / n / 1e6:.2f} | {f / n / 1e6:.2f} | {w / n / 1e6:.2f} | {a} | {ratio} |")
traffic = {k: v / traffic_n[k] for k, v in traffic.items()}
lines.append("")
lines.append("Mean HBM bytes per launch over all launches of a pair (what `roofline.traffic` in the bench line quotes): "
             + ", ".join(f"{k} {v / 1e6:.2f} MB" for k, v in sorted(traffic.items())))
# round 4: the foveated call's k_pyr_base stores level 0 inside the fovea window only (a WRITE_SIZE pass of `bench.py --workload fovea16mp`)
try:
    fov = load("pmc_write_fovea")
    for k in fov:
        if k.startswith("k_pyr_base") or k.startswith("k_blur_decimate2<16"):
            v = fov[k]["WRITE_SIZE"]
            lines.append("" if k.startswith("k_blur") else "\n## The foveated call (round 4: level 0 stored inside the fovea window only)\n\n"
                         "`rocprofv3 --kernel-trace --pmc WRITE_SIZE -- python3 bench.py --workload fovea16mp --steps 4 --warmup 1 --slots 1 --batch 1 "
                         "--no-cpu-baseline --no-events`:\n")
            lines.append(f"* `{k}`: {len(v)} launches, WRITE_SIZE {sum(v) * 1024 / len(v) / 1e6:.1f} MB per launch"
                         + (" (full mode, table above: 338.1 MB -- levels 1 and 2 are written whole, level 0 only in the window's tiles)" if k.startswith("k_pyr") else ""))
except Exception:
    pass
open(f"{out}/{tag}_hbm_traffic.md", "w").write("\n".join(lines) + "\n")
# VALU issue rate of the two hot kernels at level 0: SQ_INSTS_VALU (pass sq1) per SIMD cycle (SQ_BUSY_CU_CYCLES of pass sq2
# x 4 SIMDs), largest grid of each kernel.  Times the mean issue cost of the mix (tools/valubench: 3.0 cycles for plain
# binary32 operations up to 8.4 for v_rcp_f32; about 3.3 for these kernels) this is the busy fraction of the VALU.
def _largest(agg, prefix, counter):
    f = glob.glob(f"{base}/{agg}/**/*counter_collection.csv", recursive=True)[0]
    rows = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("ugsm::", "")
        if (name.startswith(prefix) if "<" in prefix else name.split("<")[0] == prefix) and r["Counter_Name"] == counter:   # (k_cost_march is not k_cost_march4)
            rows[int(r["Grid_Size"])].append(float(r["Counter_Value"]))
    g = max(rows)
    # (round 3: the marching K-cost launches 3 x 256 workgroups at every level that runs three waves per SIMD, so the largest grid no
    # longer identifies level 0 by itself: of that grid's dispatches keep those within 25 % of the largest value -- the level-0 ones)
    top = max(rows[g])
    keep = [v for v in rows[g] if v >= 0.75 * top]
    return sum(keep) / len(keep)
valu_busy = {}
valu_insts = {}
for kname, prefix in (("k_cost_march", "k_cost_march"), ("k_cost_split", "k_cost_split"), ("k_smooth_fused", "k_smooth_fused<112")):
    try:
        valu_insts[kname] = _largest("pmc_sq1", prefix, "SQ_INSTS_VALU")
        valu_busy[kname] = round(valu_insts[kname] / (4.0 * _largest("pmc_sq2", prefix, "SQ_BUSY_CU_CYCLES")), 4)
    except Exception:  # a pass without those counters
        valu_busy[kname] = None
# the clock the chip held during the level-0 launches: GRBM_GUI_ACTIVE (summed over the 8 XCDs) / 8 / duration of the same dispatch
# (MI355X_MICROARCH.md, DVFS give-back)
clock = {}
try:
    f = glob.glob(f"{base}/pmc_sq2/**/*counter_collection.csv", recursive=True)[0]
    t = glob.glob(f"{base}/pmc_sq2/**/*kernel_trace.csv", recursive=True)[0]
    dur = {r["Dispatch_Id"]: int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in csv.DictReader(open(t))}
    rows = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == "GRBM_GUI_ACTIVE" and r["Dispatch_Id"] in dur:
            name = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("ugsm::", "").split("<")[0]
            rows[name][int(r["Grid_Size"])].append(float(r["Counter_Value"]) / 8.0 / dur[r["Dispatch_Id"]])
    for name, g in rows.items():
        if name in ("k_cost_march", "k_smooth_fused", "k_cost_split"):
            v = g[max(g)]
            clock[name] = round(sum(v) / len(v), 3)
except Exception as ex:  # a pass without the counter
    clock = {"error": str(ex)}
json.dump({"_tag": tag, "full16mp": traffic, "clock_GHz_level0": clock, "valu_insts_per_simd_cycle_level0": valu_busy, "valu_insts_level0": valu_insts,
           "_note": "HBM bytes per launch (mean over the launches of a 16 MP pair) of each hot kernel: exact read bytes from "
                    f"the size-binned TCC_EA0_RDREQ counters + WRITE_SIZE; see profiles/{tag}_hbm_traffic.md.  valu_insts_level0: "
                    "SQ_INSTS_VALU of the largest grid of each kernel (the level-0 launch), mean per dispatch"},
          open(f"{out}/pmc_traffic.json", "w"), indent=1)
an = os.path.join(out, "analyze_trace.py")
bl = ["--bench-line", base + "/bench_under_rocprof.json"] if os.path.exists(base + "/bench_under_rocprof.json") else []
subprocess.check_call([sys.executable, an, base + "/trace", "--out", f"{out}/{tag}_trace_summary.md"] + bl, stdout=subprocess.DEVNULL)
with open(f"{out}/{tag}_pmc_summary.md", "w") as fo:
    fo.write(f"# {tag}: SQ counters (separate PMC passes, single slot, 5 pairs; mean per dispatch)\n")
    for d in ("pmc_sq1", "pmc_sq2", "pmc_sq3"):
        txt = subprocess.check_output([sys.executable, an, base + "/" + d]).decode()
        keep = [l for l in txt.split("\n") if l.startswith("- k_cost") or l.startswith("- k_smooth")]
        fo.write(f"\n## {d}\n" + "\n".join(keep[:14]) + "\n")
vb = f"{out}/{tag}_valubench.txt"
if os.path.exists(vb):
    subprocess.check_call([sys.executable, os.path.join(ROOT, "tools", "valu_model.py"), vb, tag], stdout=subprocess.DEVNULL)
print("\n".join(lines))
