#!/usr/bin/env python3
"""Fold rocprofv3 outputs of `bench.py` into the files committed under profiles/.

    python scripts/pmc_summary.py <tag> <stats_dir> [<fetch_dir> <write_dir>] --config C2 --episodes 8

  <stats_dir>  output of `rocprofv3 --kernel-trace --stats --output-format csv -d <stats_dir> -- python3 bench.py ...`
  <fetch_dir>  output of a separate `rocprofv3 --pmc FETCH_SIZE ...` pass of the same command
  <write_dir>  output of a separate `rocprofv3 --pmc WRITE_SIZE ...` pass

Writes profiles/<tag>_kernel_stats.csv (copy of the stats table), profiles/<tag>_summary.txt and, with the
PMC passes, profiles/pmc_traffic.json: per kernel, HBM bytes per launch =
f * FETCH_SIZE * 1024 + WRITE_SIZE * 1024 (FETCH_SIZE/WRITE_SIZE are in KiB).  f = 2 for kernels whose streamed
loads are 16 B per lane (on gfx950 FETCH_SIZE reports half the bytes of a wide coalesced read:
MI355X_MICROARCH.md §HBM; calibrated here on masked_pool_fwd (K0+K1), whose 68.9 MB stream reads 34.7 MiB raw) and f = 1
for kernels that stream with 8-B-per-lane loads (calibrated on wce_fwd: 7.1 MB of int64 labels + logits read
7.4 MiB raw, which includes the halo columns).  Raw values are kept beside the corrected figure.
"""
import argparse
import csv
import glob
import json
import os
import shutil
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

NAMES = [("upsample_loss_pair_kernel", "loss_pair_fwd"), ("head_bwd_dwn_kernel", "bwd_dwn"),
         ("head_bwd_mid_kernel", "bwd_chain_dfq"), ("head_bwd_tail_kernel", "bwd_params_dfs"),
         ("mask_resize_kernel", "mask_resize"), ("masked_sum_fwd_kernel", "masked_pool_fwd"),
         ("proto_extend_fwd_kernel", "proto_extend_fwd"), ("gate_proj_kernel", "proto_extend_fwd"), ("classify_fwd_kernel", "classify_fwd"),
         ("upsample_loss_kernel<0, true>", "wce_fwd"), ("upsample_loss_kernel<0, false>", "wce_fwd_nograd"),
         ("upsample_loss_kernel<1, false>", "focal_fwd"), ("loss_finalize_kernel", "loss_finalize"),
         ("classify_bwd_kernel", "classify_bwd"), ("reduce_parts_kernel", "pe_bwd_reduce"),
         ("proto_extend_bwd_chain_kernel", "pe_bwd_chain"), ("proto_extend_bwd_params_kernel", "pe_bwd_params"),
         ("masked_sum_bwd_kernel", "masked_sum_bwd"), ("head_fwd_tail_kernel", "head_fwd_tail"),
         ("head_bwd_tail_kernel", "head_bwd_tail")]


# read-side correction factor per kernel (see module docstring); anything else: 1, flagged uncalibrated
READ_FACTOR = {"masked_pool_fwd": 2, "masked_sum_bwd": 2, "pe_bwd_reduce": 2, "proto_extend_fwd": 2,
               "wce_fwd": 1, "focal_fwd": 1, "wce_fwd_nograd": 1, "mask_resize": 1, "loss_pair_fwd": 1}


def short(name):
    for pat, s in NAMES:
        if pat in name:
            return s
    return None


def find(d, pat):
    f = glob.glob(os.path.join(d, "**", pat), recursive=True)
    return f[0] if f else None


def pmc_mean(d, counter):
    f = find(d, "*counter_collection.csv")
    acc = defaultdict(list)
    if not f:
        return {}
    for r in csv.DictReader(open(f)):
        if r.get("Counter_Name") != counter:
            continue
        k = short(r.get("Kernel_Name", ""))
        if k:
            acc[k].append(float(r["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in acc.items()}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("tag")
    ap.add_argument("stats_dir")
    ap.add_argument("fetch_dir", nargs="?")
    ap.add_argument("write_dir", nargs="?")
    ap.add_argument("--config", default="C2")
    ap.add_argument("--episodes", type=int, default=8)
    a = ap.parse_args()
    out = os.path.join(ROOT, "profiles")
    os.makedirs(out, exist_ok=True)
    stats = find(a.stats_dir, "*kernel_stats.csv")
    lines = []
    if stats:
        shutil.copy(stats, os.path.join(out, f"{a.tag}_kernel_stats.csv"))
        tot = 0.0
        for r in csv.DictReader(open(stats)):
            nm = short(r["Name"]) or r["Name"].split("(")[0][:48]
            lines.append(f"{nm:24s} calls={int(r['Calls']):6d} avg_us={float(r['AverageNs']) / 1e3:9.2f} "
                         f"min_us={float(r['MinNs']) / 1e3:9.2f} max_us={float(r['MaxNs']) / 1e3:9.2f} "
                         f"pct={float(r['Percentage']):6.2f}")
    traffic = None
    if a.fetch_dir and a.write_dir:
        fe, wr = pmc_mean(a.fetch_dir, "FETCH_SIZE"), pmc_mean(a.write_dir, "WRITE_SIZE")
        traffic = {"config": a.config, "episodes": a.episodes,
                   "note": "hbm_bytes_per_launch = f*FETCH_SIZE*1024 + WRITE_SIZE*1024 (KiB counters); f = 2 for "
                           "16-B-per-lane streams (gfx950 FETCH_SIZE tallies 128-B requests at 64 B: "
                           "MI355X_MICROARCH.md), f = 1 for 8-B-per-lane streams; see scripts/pmc_summary.py",
                   "kernels": {}}
        lines.append("")
        lines.append("PMC (mean per launch): kernel, FETCH_SIZE KiB (raw), WRITE_SIZE KiB (raw), corrected HBM bytes")
        for k in sorted(set(fe) | set(wr)):
            f_, w_ = fe.get(k, 0.0), wr.get(k, 0.0)
            fac = READ_FACTOR.get(k, 1)
            hb = int(fac * f_ * 1024 + w_ * 1024)
            traffic["kernels"][k] = {"fetch_size_kib_raw": round(f_, 2), "write_size_kib_raw": round(w_, 2),
                                     "read_factor": fac, "calibrated": k in READ_FACTOR,
                                     "hbm_bytes_per_launch": hb}
            lines.append(f"{k:24s} fetch={f_:12.2f} write={w_:12.2f} hbm_bytes={hb}")
        json.dump(traffic, open(os.path.join(out, "pmc_traffic.json"), "w"), indent=1)
    open(os.path.join(out, f"{a.tag}_summary.txt"), "w").write("\n".join(lines) + "\n")
    print("\n".join(lines))


if __name__ == "__main__":
    main()
