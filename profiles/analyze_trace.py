#!/usr/bin/env python3
"""Summarises rocprofv3 CSV output (kernel trace and/or PMC passes) per kernel and grid size.

usage: analyze_trace.py <dir-with-*_kernel_trace.csv / *_counter_collection.csv> [--out summary.md] [--bench-line bench_under_rocprof.json]
The grid size identifies the pyramid level, so per-level timings fall out of the trace.
--bench-line: the JSON line bench.py printed in the traced run; the trace's launches of its `roofline.kernel` are then split into the
event pass's (the last event_pass.pairs x roofline.launches_per_pair of them: one call in flight) and the ones before (the timed regions:
`slots` calls in flight), whose means bracket `roofline.avg_launch_us`.
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
    if traces and "--bench-line" in sys.argv:
        import json
        line = json.load(open(sys.argv[sys.argv.index("--bench-line") + 1]))
        roof, ev = line.get("roofline") or {}, line.get("event_pass") or {}
        if roof.get("kernel") and ev.get("pairs"):
            n_ev = int(round(ev["pairs"] * roof["launches_per_pair"]))
            rows = []
            for t in traces:
                for r in csv.DictReader(open(t)):
                    if short(r["Kernel_Name"]).split("<")[0] == roof["kernel"]:
                        rows.append((int(r["Start_Timestamp"]), (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3))
            rows.sort()
            if len(rows) > n_ev > 0:
                before, last = [u for _, u in rows[:-n_ev]], [u for _, u in rows[-n_ev:]]
                lines.append(f"\n## {roof['kernel']}: the timed regions' launches against the event pass's\n")
                lines.append(f"* all {len(rows)} launches: {sum(u for _, u in rows) / len(rows):.1f} us mean (the AverageNs of the kernel summary)")
                lines.append(f"* the {len(before)} launches of the warm-up and the timed regions ({line['config'].get('slots_per_gpu', '?')} calls in flight: "
                             f"launches of different slots share the chip): {sum(before) / len(before):.1f} us")
                lines.append(f"* the last {n_ev} launches = the event pass ({ev['pairs']} pairs in calls of {ev.get('pairs_per_call', 1)}, one call in flight): "
                             f"{sum(last) / len(last):.1f} us by the trace's timestamps; {roof['avg_launch_us']:.1f} us by the HIP events the same "
                             "launches carry in their dispatch (`roofline.avg_launch_us` of that run's line)")
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
