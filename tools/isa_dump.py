#!/usr/bin/env python3
"""Per-kernel instruction streams of a built library (development tool): which kernels did a source change touch?

    python tools/isa_dump.py ug_stereomatcher_amd/libugsm.so out.json          # one record per device function
    python tools/isa_dump.py --diff before.json after.json                     # added / removed / changed symbols
    python tools/isa_dump.py --table ug_stereomatcher_amd/libugsm.so           # instructions, registers, LDS, scratch, DPP and MFMA counts per kernel

The gfx950 code objects are cut out of the library's .hip_fatbin section (one clang offload bundle per translation unit), disassembled
with llvm-objdump, and every function's instructions -- addresses and encodings stripped, so that moving a kernel to another file or
deleting its neighbours changes nothing -- are hashed.  A refactoring that only moves code must leave every hash as it was; the
register counts come from the code objects' notes (llvm-readelf --notes)."""
import hashlib
import json
import os
import re
import subprocess
import sys
import tempfile

LLVM = "/opt/rocm/lib/llvm/bin"
MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"


def code_objects(lib, tmp):
    raw = os.path.join(tmp, "fatbin")
    subprocess.check_call([os.path.join(LLVM, "llvm-objcopy"), "-O", "binary", "--only-section=.hip_fatbin", lib, raw])
    blob = open(raw, "rb").read()
    starts = [m.start() for m in re.finditer(re.escape(MAGIC), blob)]
    out = []
    for k, s in enumerate(starts):
        e = starts[k + 1] if k + 1 < len(starts) else len(blob)
        b = blob[s:e]
        # bundle header: magic, u64 count, then per entry u64 offset, u64 size, u64 id length, id
        n = int.from_bytes(b[24:32], "little")
        p = 32
        for _ in range(n):
            off, size, idl = (int.from_bytes(b[p + 8 * j:p + 8 * j + 8], "little") for j in range(3))
            ident = b[p + 24:p + 24 + idl].decode()
            p += 24 + idl
            if "gfx950" in ident and size:
                path = os.path.join(tmp, f"co_{k}.o")
                open(path, "wb").write(b[off:off + size])
                out.append(path)
    return out


def functions(co):
    txt = subprocess.run([os.path.join(LLVM, "llvm-objdump"), "-d", "--no-show-raw-insn", co], capture_output=True, text=True, check=True).stdout
    funcs, cur = {}, None
    for line in txt.split("\n"):
        m = re.match(r"^[0-9a-f]+ <(.+)>:$", line)
        if m:
            cur = m.group(1)
            funcs[cur] = []
            continue
        if cur is None or not line.strip():
            continue
        ins = re.sub(r"\s*//.*$", "", line).strip()  # (the trailing comment holds the address)
        ins = re.sub(r"^[0-9a-f]+:\s*", "", ins)
        ins = re.sub(r"<[^>]+>", "<sym>", ins)          # branch targets print as <kernel+0x..>: the relative immediate stays
        if ins:
            funcs[cur].append(ins)
    for ins in funcs.values():  # (the last function of a code object runs into the section's padding: cut after the last s_endpgm)
        last = max((i for i, x in enumerate(ins) if x.startswith("s_endpgm")), default=len(ins) - 1)
        del ins[last + 1:]
    return funcs


def notes(co):
    txt = subprocess.run([os.path.join(LLVM, "llvm-readelf"), "--notes", co], capture_output=True, text=True).stdout
    # a kernel's fields come in alphabetical order at four spaces of indentation, .wavefront_size last
    regs, fields = {}, {}
    keep = ("vgpr_count", "sgpr_count", "group_segment_fixed_size", "private_segment_fixed_size", "vgpr_spill_count", "agpr_count")
    for line in txt.split("\n"):
        m = re.match(r"^(?:    |  - )\.(\w+):\s*(.+)$", line)
        if not m:
            continue
        k, v = m.group(1), m.group(2).strip()
        fields[k] = v
        if k == "wavefront_size":
            if "name" in fields:
                regs[fields["name"]] = {f: int(fields[f]) for f in keep if f in fields}
            fields = {}
    return regs


def dump(lib):
    rec = {}
    with tempfile.TemporaryDirectory() as tmp:
        for co in code_objects(lib, tmp):
            regs = notes(co)
            for name, ins in functions(co).items():
                r = {"n": len(ins), "sha": hashlib.sha256("\n".join(ins).encode()).hexdigest()[:16],
                     "mfma": sum(i.startswith("v_mfma") for i in ins), "dpp": sum("row_sh" in i or "quad_perm" in i or "wave_sh" in i for i in ins)}
                r.update(regs.get(name, {}))
                rec[name] = r
    return rec


def demangle(names):
    out = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout.split("\n")
    return dict(zip(names, out))


if __name__ == "__main__":
    if sys.argv[1] == "--diff":
        a, b = json.load(open(sys.argv[2])), json.load(open(sys.argv[3]))
        dm = demangle(sorted(set(a) | set(b)))
        for s in sorted(set(a) - set(b)):
            print("removed ", dm[s])
        for s in sorted(set(b) - set(a)):
            print("added   ", dm[s])
        ch = [s for s in sorted(set(a) & set(b)) if a[s]["sha"] != b[s]["sha"]]
        for s in ch:
            print(f"changed  {dm[s]}: {a[s]['n']} -> {b[s]['n']} instructions, vgpr {a[s].get('vgpr_count')} -> {b[s].get('vgpr_count')}")
        print(f"{len(set(a) & set(b)) - len(ch)} functions identical, {len(ch)} changed, {len(set(a) - set(b))} removed, {len(set(b) - set(a))} added")
        sys.exit(0)
    if sys.argv[1] == "--table":
        rec = dump(sys.argv[2])
        dm = demangle(sorted(rec))
        print(f"# {os.path.basename(sys.argv[2])}: device functions of the gfx950 code objects (llvm-objdump / llvm-readelf --notes; static LDS only: dynamic LDS is the launcher's)")
        print(f"{'kernel':44s} {'instr':>6s} {'VGPR':>5s} {'SGPR':>5s} {'LDS B':>6s} {'scratch':>7s} {'DPP':>5s} {'v_mfma':>6s}")
        for s_ in sorted(rec, key=lambda k: -rec[k]["n"]):
            r = rec[s_]
            name = dm[s_].split("(")[0].replace("void ", "").replace("ugsm::", "")
            print(f"{name:44s} {r['n']:6d} {r.get('vgpr_count', 0):5d} {r.get('sgpr_count', 0):5d} {r.get('group_segment_fixed_size', 0):6d} "
                  f"{r.get('private_segment_fixed_size', 0):7d} {r['dpp']:5d} {r['mfma']:6d}")
        print(f"{len(rec)} device functions, {sum(r['mfma'] for r in rec.values())} v_mfma")
        sys.exit(0)
    rec = dump(sys.argv[1])
    json.dump(rec, open(sys.argv[2], "w"), indent=0, sort_keys=True)
    print(f"{len(rec)} device functions, {sum(r['mfma'] for r in rec.values())} v_mfma")
