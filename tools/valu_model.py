#!/usr/bin/env python3
"""VALU-issue model of the hot kernels (development tool; writes profiles/<tag>_valu_model.{json,md}).

Both hot kernels are bound by VALU issue, not by bytes.  This tool makes that claim checkable:
  1. it compiles the kernel sources to gfx950 assembly and finds each hot kernel's innermost loops;
  2. it prices every VALU instruction of those loops with the issue cost measured by tools/valubench.hip in ACTUAL
     shader cycles (cycles per wave-instruction per SIMD with four waves per SIMD, wall time x in-kernel clock);
  3. it reports the instruction mix, the mean cost per VALU instruction and -- for the marching kernels, whose loop
     trip is one image row per wave -- the VALU cycles per wave row step;
  4. with the dynamic VALU instruction counts of the level-0 launches (SQ_INSTS_VALU, PMC pass of tools/profile_round.sh)
     bench.py turns this into a modelled VALU time and compares it with the measured duration (`valu_roofline`).

usage: valu_model.py <valubench.txt> <tag> [pmc_dir]
"""
import collections
import json
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "ug_stereomatcher_amd", "csrc")
FLAGS = ["-O3", "-std=c++17", "-DUGSM_DEV_LIB", "--offload-arch=gfx950", "-ffp-contract=off", "-fno-fast-math", "-fno-slp-vectorize", "-x", "hip", "-S",
         "--cuda-device-only"]

# valubench kernel -> cost class
BENCH_CLASS = {"k_fma": "fma32", "k_fmac": "fma32", "k_mul": "plain32", "k_add": "plain32", "k_mov": "mov",
               "k_add_dpp_wave_shr": "dpp", "k_add_dpp_row_shr": "dpp", "k_mov_dpp": "dpp", "k_rcp": "trans32", "k_exp": "trans32",
               "k_div_scale": "div_scale32", "k_div_fmas": "slow32", "k_div_fixup": "slow32", "k_minimum3": "slow32", "k_min3": "slow32",
               "k_max": "slow32", "k_med3": "slow32", "k_cmp_vcc": "slow32", "k_floor": "slow32", "k_cvt_i32": "slow32", "k_fma64": "f64",
               "k_mul64": "f64", "k_add64": "f64", "k_cvt_f64_f32": "f64", "k_cvt_f32_f64": "f64", "k_add_u32": "int", "k_mad_u32_u24": "int3",
               "k_lshlrev_b32": "int3", "k_add_lshl_u32": "int3", "k_pk_mul": "packed", "k_pk_add": "packed", "k_pk_fma": "packed"}
# the guide's nominal figures (MI355X_MICROARCH.md, "Per-instruction cycle constants": v_fma_f32 2 per SIMD at >= 2 waves, transcendentals 8;
# half-rate classes at twice the full rate) -- the second cost table `valu_roofline` is quoted against
GUIDE_COSTS = {"fma32": 2.0, "plain32": 2.0, "int": 2.0, "mov": 2.0, "dpp": 4.0, "slow32": 4.0, "int3": 4.0, "f64": 4.0, "div_scale32": 4.0,
               "trans32": 8.0, "trans64": 16.0, "packed": 4.0}


def read_costs(path):
    """cycles per wave-instruction per SIMD from tools/valubench (round-3 format: kernel, waves/SIMD, stamped, wall, clock GHz): the WALL
    figure at four waves per SIMD -- the throughput a saturated SIMD sustains (wall time x in-kernel clock / instructions)."""
    acc = collections.defaultdict(list)
    clocks = []
    for line in open(path):
        m = re.match(r"(k_\w+)\s+(\d)\s+([\d.]+)\s+([\d.]+)\s+([\d.]+)\s*$", line)
        if m and m.group(2) == "4" and m.group(1) in BENCH_CLASS:
            acc[BENCH_CLASS[m.group(1)]].append(float(m.group(4)))
            clocks.append(float(m.group(5)))
    costs = {c: sum(v) / len(v) for c, v in acc.items()}
    costs.setdefault("trans64", 2 * costs.get("trans32", 8.2))
    return costs, sum(clocks) / max(len(clocks), 1)


def source_hash():
    """sha256 (first 16 hex digits) over the kernel sources: bench.py reports valu_roofline as stale when the built sources differ"""
    import hashlib
    h = hashlib.sha256()
    for f in sorted(os.listdir(CSRC)):   # (the product's kernel sources: csrc/*.hip, *.hpp -- not csrc/dev/)
        if f.endswith((".hip", ".hpp")):
            h.update(open(os.path.join(CSRC, f), "rb").read())
    return h.hexdigest()[:16]


def classify(op, text):
    if "dpp" in text:
        return "dpp"
    if op.startswith("v_pk_"):
        return "packed"
    if op.endswith("_f64") or "_f64_" in op or op in ("v_ldexp_f64",):
        return "trans64" if op.startswith("v_rcp_f64") or op.startswith("v_rsq_f64") else "f64"
    if op.startswith(("v_rcp_", "v_rsq_", "v_sqrt_", "v_exp_", "v_log_", "v_sin_", "v_cos_")):
        return "trans32"
    if op.startswith("v_div_scale_f32"):
        return "div_scale32"
    if op.startswith(("v_fma_f32", "v_fmac_f32")):
        return "fma32"
    if op.startswith("v_mov_b32"):
        return "mov"
    if op.startswith(("v_add_f32", "v_sub_f32", "v_subrev_f32", "v_mul_f32")):
        return "plain32"
    if op.startswith(("v_mad_u32", "v_mad_i32", "v_add_lshl", "v_lshl_add", "v_add3", "v_med3_i32", "v_med3_u32", "v_lshl_or", "v_and_or", "v_bfe",
                      "v_mul_lo", "v_mul_hi", "v_mad_u64")):
        return "int3"
    if op.startswith(("v_add_u32", "v_sub_u32", "v_subrev_u32", "v_add_co", "v_addc", "v_add_i32", "v_sub_i32")):
        return "int"
    if op.startswith(("v_lshlrev_b32", "v_lshrrev_b32", "v_ashrrev_i32", "v_and_b32", "v_or_b32", "v_xor_b32", "v_mul_u32_u24", "v_mul_i32_i24", "v_not_b32")):
        return "int3"  # (shifts and logic measure at the half rate: k_lshlrev_b32)
    return "slow32"  # compares, selects, min/max/med3, floor, conversions, div_fmas/div_fixup, readlane ...


def loops_of(asm_path, name_re, all_loops=False):
    """{kernel symbol: {loop header label: [instruction text]}} for innermost loops (all_loops: loops of any depth, each block counted
    under the innermost loop that holds it -- for kernels whose hot loop contains small rarely-taken loops).  Basic blocks that hold a binary64
    division (v_rcp_f64: the literal fallbacks of PolyDisparity's quotients, entered only by the lanes that need them and
    skipped by s_cbranch_execz otherwise) are left out: they are not on the hot path."""
    out = {}
    cur = None
    header = None
    block = []

    def flush():
        if cur and header and block and not any("v_rcp_f64" in t for t in block):
            out[cur][header] += block

    for line in open(asm_path):
        m = re.match(r"^(_Z\S+):", line)
        if m:
            flush()
            block = []
            cur = m.group(1) if re.search(name_re, m.group(1)) else None
            if cur:
                out[cur] = collections.OrderedDict()
            header = None
            continue
        if cur is None:
            continue
        if line.startswith(".Lfunc_end"):
            flush()
            block = []
            cur = None
            continue
        m = re.match(r"^(\.LBB\d+_\d+):\s*;?\s*(.*)", line) or re.match(r"^; %bb\.(\d+):\s*;?\s*(.*)", line)
        if m:
            flush()
            block = []
            lab, com = m.group(1), m.group(2)
            if "Inner Loop Header" in com or (all_loops and "Loop Header" in com):
                header = lab
                out[cur].setdefault(header, [])
            else:
                mm = re.search(r"in Loop: Header=(BB\d+_\d+)", com)
                header = ".L" + mm.group(1) if mm and (".L" + mm.group(1)) in out[cur] else None
            continue
        t = line.strip()
        if t and not t.startswith((";", ".")):
            block.append(t)
    flush()
    return out


def price(instrs, costs):
    mix = collections.Counter()
    other = collections.Counter()
    for t in instrs:
        op = t.split()[0]
        if op.startswith("v_"):
            mix[classify(op, t)] += 1
        elif op.startswith("s_"):
            other["salu"] += 1
        elif op.startswith(("global_", "buffer_", "flat_", "scratch_")):
            other["vmem"] += 1
        elif op.startswith("ds_"):
            other["lds"] += 1
    n = sum(mix.values())
    cyc = sum(k * costs.get(c, costs["slow32"]) for c, k in mix.items())
    cyc_g = sum(k * GUIDE_COSTS.get(c, 4.0) for c, k in mix.items())
    return {"valu_instructions": n, "valu_cycles": cyc, "mean_cycles_per_valu": cyc / max(n, 1), "valu_cycles_guide": cyc_g,
            "mean_cycles_per_valu_guide": cyc_g / max(n, 1), "mix": dict(mix), "other": dict(other)}


def main():
    bench_txt, tag = sys.argv[1], sys.argv[2]
    costs, clock = read_costs(bench_txt)
    costs.setdefault("int", costs["plain32"])
    costs.setdefault("int3", costs["slow32"])
    costs.setdefault("mov", costs["plain32"])
    costs.setdefault("packed", 2 * costs["plain32"])
    tmp = "/tmp/ugsm_valu_model"
    os.makedirs(tmp, exist_ok=True)
    report = {"_source": f"tools/valu_model.py on {os.path.basename(bench_txt)}", "_tag": tag, "_kernel_source_sha16": source_hash(),
              "clock_GHz": round(clock, 3), "simds": 1024,
              "cost_cycles_per_wave_instruction": {k: round(v, 2) for k, v in sorted(costs.items())},
              "guide_cycles_per_wave_instruction": GUIDE_COSTS, "kernels": {}}
    md = [f"# {tag}: VALU-issue model of the hot kernels\n",
          f"Issue costs (actual shader cycles per wave-instruction per SIMD, four waves per SIMD; `{os.path.basename(bench_txt)}`; mean in-kernel "
          f"clock {clock:.2f} GHz):\n", "| class | cycles | what is in it |", "|---|---|---|"]
    what = {"plain32": "v_add/sub/mul_f32", "mov": "v_mov_b32", "fma32": "v_fma_f32, v_fmac_f32", "dpp": "any VALU instruction with a DPP operand (wave_shr/shl)",
            "trans32": "v_rcp_f32", "div_scale32": "v_div_scale_f32", "slow32": "v_cmp*, v_cndmask, v_min/max/med3, v_floor, v_cvt_*, v_div_fmas/fixup_f32",
            "f64": "v_fma/mul/add_f64, v_cvt_f64_f32, v_cvt_f32_f64", "trans64": "v_rcp_f64", "int": "v_add_u32, shifts, logic", "int3": "v_mad_u32_u24, v_add_lshl_u32, v_med3_i32",
            "packed": "v_pk_*_f32"}
    for k, v in sorted(costs.items()):
        md.append(f"| {k} | {v:.2f} | {what.get(k, '')} |")
    jobs = [("ugsm_kernels_march.hip", r"k_cost_marchE", "k_cost_march", {"valid_pixels_per_wave_step": 58, "steps_per_trip": 2}),
            ("ugsm_kernels_smooth.hip", r"k_smooth_fusedILi112ELi36ELi512ELb1E", "k_smooth_fused", None),
            ("dev/ugsm_dev_cost_tiled.hip", r"k_cost_splitE", "k_cost_split", None),
            ("ugsm_kernels_march4.hip", r"k_cost_march4E", "k_cost_march4", None)]
    for src, name_re, short, geom in jobs:
        asm = os.path.join(tmp, os.path.basename(src).replace(".hip", ".s"))
        if not os.path.exists(asm) or os.path.getmtime(asm) < os.path.getmtime(os.path.join(CSRC, src)):
            subprocess.check_call(["/opt/rocm/bin/hipcc"] + FLAGS + [os.path.join(CSRC, src), "-o", asm], stderr=subprocess.DEVNULL)
        lp = loops_of(asm, name_re, all_loops=geom is None)
        for sym, loops in lp.items():
            if not loops:
                continue
            priced = {h: price(ins, costs) for h, ins in loops.items()}
            entry = {"symbol": sym}
            if geom:
                # the row loops: four bodies (interior / frame strips x guarded / full division); the production case on in-range
                # data is the interior body with the guarded division = the loop with no v_div_scale_f32 and the fewest selects
                rows = {h: p for h, p in priced.items() if p["valu_instructions"] > 400}
                pick = min(rows, key=lambda h: (rows[h]["mix"].get("div_scale32", 0), rows[h]["valu_instructions"]))
                p = rows[pick]
                entry.update({"loop": pick, "valu_instructions_per_wave_step": p["valu_instructions"] / geom["steps_per_trip"],
                              "valu_cycles_per_wave_step": p["valu_cycles"] / geom["steps_per_trip"],
                              "mean_cycles_per_valu": p["mean_cycles_per_valu"], "mean_cycles_per_valu_guide": p["mean_cycles_per_valu_guide"],
                              "valu_cycles_per_wave_step_guide": p["valu_cycles_guide"] / geom["steps_per_trip"],
                              "mix_per_trip": p["mix"], "other_per_trip": p["other"],
                              "valid_pixels_per_wave_step": geom["valid_pixels_per_wave_step"]})
            else:
                tot = collections.Counter()
                allins = []
                for h, ins in loops.items():
                    allins += ins
                p = price(allins, costs)
                entry.update({"loops": len(loops), "mean_cycles_per_valu": p["mean_cycles_per_valu"],
                              "mean_cycles_per_valu_guide": p["mean_cycles_per_valu_guide"], "mix_static": p["mix"]})
            report["kernels"][short] = entry
            md.append(f"\n## {short}\n")
            md.append("```\n" + json.dumps(entry, indent=1) + "\n```")
    out = os.path.join(ROOT, "profiles")
    json.dump(report, open(os.path.join(out, f"{tag}_valu_model.json"), "w"), indent=1)
    open(os.path.join(out, f"{tag}_valu_model.md"), "w").write("\n".join(md) + "\n")
    print("\n".join(md))


if __name__ == "__main__":
    main()
