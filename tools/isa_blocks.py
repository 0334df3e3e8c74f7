#!/usr/bin/env python3
"""Instruction census of a kernel's basic blocks from hipcc -S output (development tool).

    hipcc ... --cuda-device-only -S file.hip -o file.s;  python tools/isa_blocks.py file.s <kernel-name-substring> [n_blocks]

Per block: instructions, VALU, DPP-operand VALU, s_nop (count and cycles = sum of imm + 1), other SALU, s_waitcnt, memory.  The hot
loop of a kernel is its largest block(s)."""
import re
import sys


def census(path, key, top=8):
    lines = open(path).read().split("\n")
    start = next(i for i, l in enumerate(lines) if key in l and l.rstrip().endswith(":") and not l.startswith("\t") or (key in l and re.match(r"^_Z\S+:\s*;", l)))
    end = next(i for i, l in enumerate(lines) if i > start and l.strip().startswith(".Lfunc_end"))
    blocks, cur = [], ["entry", []]
    blocks.append(cur)
    for l in lines[start + 1:end]:
        m = re.match(r"^(\.LBB\d+_\d+):", l)
        if m:
            cur = [m.group(1), []]
            blocks.append(cur)
            continue
        t = l.strip()
        if t and not t.startswith(";") and not t.startswith("."):
            cur[1].append(t)
    rows = []
    for name, ins in blocks:
        snop = [i for i in ins if i.startswith("s_nop")]
        rows.append(dict(block=name, n=len(ins), valu=sum(i.startswith("v_") for i in ins),
                         dpp=sum(("row_shr" in i or "row_shl" in i or "wave_sh" in i or "quad_perm" in i or "row_bcast" in i) for i in ins),
                         s_nop=len(snop), nop_cycles=sum(int(i.split()[1]) + 1 for i in snop),
                         salu=sum(i.startswith("s_") and not i.startswith("s_nop") and not i.startswith("s_waitcnt") for i in ins),
                         waitcnt=sum(i.startswith("s_waitcnt") for i in ins),
                         mem=sum(i.startswith(("global_", "buffer_", "flat_", "scratch_")) for i in ins),
                         lds=sum(i.startswith("ds_") for i in ins)))
    rows.sort(key=lambda r: -r["n"])
    return rows[:top], sum(r["n"] for r in rows)


if __name__ == "__main__":
    rows, total = census(sys.argv[1], sys.argv[2], int(sys.argv[3]) if len(sys.argv) > 3 else 8)
    print(f"{sys.argv[2]}: {total} instructions in all blocks")
    for r in rows:
        print("  ".join(f"{k}={v}" for k, v in r.items()))
