#!/usr/bin/env python3
"""Fold a `rocprofv3 --pmc ... --output-format csv` directory into one line per (kernel, counter): the mean counter
value per launch and the number of launches.  usage: counter_fold.py <label> <dir>   (prints; raw CSVs are megabytes)"""
import csv, glob, os, sys
from collections import defaultdict
label, d = sys.argv[1], sys.argv[2]
f = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
if not f:
    print(f"{label}: no counter_collection.csv under {d}")
    sys.exit(0)
acc = defaultdict(list)
for r in csv.DictReader(open(max(f, key=os.path.getmtime))):
    acc[(r["Kernel_Name"].split("(")[0][:56], r["Counter_Name"])].append(float(r["Counter_Value"]))
print(f"== {label}")
for (k, c), v in sorted(acc.items()):
    print(f"{k:56s} {c:24s} launches={len(v):5d} mean={sum(v) / len(v):16.1f}")
