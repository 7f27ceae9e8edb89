#!/usr/bin/env python3
"""Scan the gfx950 ISA of every kernel in a .hip file for serialized global loads: a
`global_load` whose next vector-memory event is an `s_waitcnt vmcnt(0)` (instead of another load)
pays a full memory round trip on its own.  Usage: python scripts/isa_scan.py denet_amd/csrc/classify.hip"""
import re
import subprocess
import sys

src = sys.argv[1]
flt = sys.argv[2] if len(sys.argv) > 2 else ""
asm = subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-S", "--cuda-device-only",
                      src, "-o", "-"], capture_output=True, text=True).stdout
kern = None
stats = {}
for line in asm.split("\n"):
    m = re.match(r"^(_Z\w+):", line)
    if m:
        kern = m.group(1)
        stats[kern] = dict(loads=0, waits0=0, serial=0, last=None, scratch=0, waits=0)
        continue
    if kern is None:
        continue
    t = line.strip()
    op = t.split()[0] if t else ""
    st = stats[kern]
    if op.startswith(("global_load", "buffer_load", "flat_load")):
        st["loads"] += 1
        st["last"] = "load"
    elif op == "s_waitcnt" and "vmcnt" in t:
        st["waits"] += 1
        if "vmcnt(0)" in t:
            st["waits0"] += 1
            if st["last"] == "load":
                st["serial"] += 1
        st["last"] = "wait"
    elif op.startswith("scratch_"):
        st["scratch"] += 1
    elif op == "s_endpgm":
        kern = None
demangle = subprocess.run(["c++filt"], input="\n".join(stats), capture_output=True, text=True).stdout.split("\n")
for name, (k, st) in zip(demangle, stats.items()):
    if flt and flt not in name:
        continue
    print(f"{name[:70]:70s} loads={st['loads']:4d} vmcnt_waits={st['waits']:4d} vmcnt0={st['waits0']:4d} "
          f"load->vmcnt(0) groups={st['serial']:4d} scratch={st['scratch']}")
