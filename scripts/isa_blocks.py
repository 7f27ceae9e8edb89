#!/usr/bin/env python3
"""Static per-basic-block instruction counts of one kernel's gfx950 ISA (from `hipcc -S`):
usage: python scripts/isa_blocks.py file.s <mangled-name-substring> [min_valu]
Prints each basic block with its VALU / SALU / LDS / VMEM counts and the loop depth the compiler annotated."""
import re, sys
src, pat = sys.argv[1], sys.argv[2]
minv = int(sys.argv[3]) if len(sys.argv) > 3 else 8
lines = open(src).read().split("\n")
start = next(i for i, l in enumerate(lines) if re.match(r"^_Z\w*:", l) and pat in l)
blocks, cur = [], dict(name="entry", depth=0, valu=0, salu=0, lds=0, vmem=0, trans=0, line=start)
for i in range(start + 1, len(lines)):
    l = lines[i]
    m = re.match(r"^(\.LBB\d+_\d+):(.*)", l)
    if m:
        blocks.append(cur)
        d = re.search(r"Depth=(\d+)", m.group(2))
        cur = dict(name=m.group(1), depth=int(d.group(1)) if d else 0, valu=0, salu=0, lds=0, vmem=0, trans=0, line=i)
        continue
    t = l.strip()
    if not t or t.startswith(";") or t.startswith("."):
        continue
    op = t.split()[0]
    if op.startswith("v_"):
        cur["valu"] += 1
        if re.match(r"v_(exp|log|rcp|rsq|sqrt|sin|cos)_", op):
            cur["trans"] += 1
    elif op.startswith("s_"):
        cur["salu"] += 1
    elif op.startswith("ds_"):
        cur["lds"] += 1
    elif op.startswith(("global_", "buffer_", "flat_", "scratch_")):
        cur["vmem"] += 1
    if op == "s_endpgm":
        break
blocks.append(cur)
tot = sum(b["valu"] for b in blocks)
print(f"{len(blocks)} blocks, {tot} VALU static")
for b in blocks:
    if b["valu"] >= minv:
        print(f"{b['name']:12s} line {b['line']-start:5d} depth {b['depth']} valu {b['valu']:4d} (trans {b['trans']:2d}) salu {b['salu']:4d} lds {b['lds']:3d} vmem {b['vmem']:3d}")
