#!/usr/bin/env python3
"""Summarises rocprofv3 CSV output (kernel trace and/or PMC passes) per kernel and grid size.

usage: analyze_trace.py <dir-with-*_kernel_trace.csv / *_counter_collection.csv> [--out summary.md]
The grid size identifies the pyramid level, so per-level timings fall out of the trace.
"""
import collections
import csv
import glob
import os
import sys


def short(name):
    name = name.split("(")[0]
    return name.replace("void ", "").replace("ugsm::", "")


def main():
    d = sys.argv[1]
    out = sys.argv[sys.argv.index("--out") + 1] if "--out" in sys.argv else None
    lines = []
    traces = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)
    disp = {}
    if traces:
        agg = collections.defaultdict(lambda: [0, 0.0])
        per_kernel = collections.defaultdict(lambda: [0, 0.0])
        for t in traces:
            for r in csv.DictReader(open(t)):
                k = (short(r["Kernel_Name"]), int(r["Grid_Size_X"]) // int(r["Workgroup_Size_X"]), int(r["Grid_Size_Y"]))
                us = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
                agg[k][0] += 1
                agg[k][1] += us
                per_kernel[k[0]][0] += 1
                per_kernel[k[0]][1] += us
                disp[r["Dispatch_Id"]] = k
        tot = sum(v[1] for v in per_kernel.values())
        lines.append("## kernel totals\n")
        lines.append("| kernel | calls | total ms | avg us | % |\n|---|---|---|---|---|")
        for k, (c, us) in sorted(per_kernel.items(), key=lambda kv: -kv[1][1]):
            lines.append(f"| {k} | {c} | {us / 1e3:.3f} | {us / c:.1f} | {100 * us / tot:.1f} |")
        lines.append("\n## per kernel and grid (blocks_x x blocks_y)\n")
        lines.append("| kernel | grid | calls | total ms | avg us |\n|---|---|---|---|---|")
        for (k, gx, gy), (c, us) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:60]:
            lines.append(f"| {k} | {gx}x{gy} | {c} | {us / 1e3:.3f} | {us / c:.1f} |")
    pmcs = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
    if pmcs:
        agg = collections.defaultdict(lambda: collections.defaultdict(float))
        cnt = collections.defaultdict(lambda: collections.defaultdict(int))
        for t in pmcs:
            for r in csv.DictReader(open(t)):
                k = (short(r["Kernel_Name"]), int(r["Grid_Size"]) // max(int(r["Workgroup_Size"]), 1))
                agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
                cnt[k][r["Counter_Name"]] += 1
        lines.append("\n## PMC counters, mean per dispatch (kernel, workgroups)\n")
        for k in sorted(agg, key=lambda k: -k[1])[:40]:
            vals = ", ".join(f"{c}={agg[k][c] / cnt[k][c]:.4g}" for c in sorted(agg[k]))
            lines.append(f"- {k[0]} [{k[1]} wg, {max(cnt[k].values())} dispatches]: {vals}")
    text = "\n".join(lines)
    print(text)
    if out:
        open(out, "w").write(text + "\n")


if __name__ == "__main__":
    main()
