#!/usr/bin/env python3
"""Print a compact per-kernel table from a rocprofv3 --kernel-trace --stats run."""
import csv, glob, sys
d = sys.argv[1]
f = glob.glob(d + "/**/*kernel_stats.csv", recursive=True)[0]
tot = 0.0
rows = []
for r in csv.DictReader(open(f)):
    rows.append((r["Name"].split("(")[0][:58], int(r["Calls"]), float(r["AverageNs"]) / 1e3, float(r["Percentage"])))
for n, c, a, p in rows:
    print(f"{n:58s} calls={c:5d} avg_us={a:9.2f} pct={p:6.2f}")
