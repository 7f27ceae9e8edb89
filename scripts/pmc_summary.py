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

NAMES = [("wce_chain_pair_kernel", "loss_pair_fwd"), ("upsample_loss_pair_kernel", "loss_pair_fwd"),
         ("wce_chain_kernel", "wce_fwd"), ("focal_kernel", "focal_fwd"), ("classify_fwd_wide4_kernel", "classify_fwd"), ("classify_fwd_wide4_bf16_kernel", "classify_fwd"), ("classify_fwd_wide_kernel", "classify_fwd"), ("head_bwd_dwn_kernel", "bwd_dwn"), ("head_bwd_dwn_wide_kernel", "bwd_dwn"), ("head_bwd_dwn_wide_bf16_kernel", "bwd_dwn"),
         ("head_bwd_mid_kernel", "bwd_chain_dfq"), ("head_bwd_tail_kernel", "bwd_params_dfs"),
         ("dfs_chunk_kernel", "dfs_chunk"), ("mask_resize_flat_kernel", "mask_resize"), ("mask_resize_kernel", "mask_resize"), ("masked_sum_fwd_kernel", "masked_pool_fwd"),
         ("proto_extend_fwd_kernel", "proto_extend_fwd"), ("gate_proj_kernel", "proto_extend_fwd"), ("classify_fwd_kernel", "classify_fwd"),
         ("upsample_loss_kernel<0, true>", "wce_fwd"), ("upsample_loss_kernel<0, false>", "wce_fwd_nograd"),
         ("upsample_loss_kernel<1, false>", "focal_fwd"), ("loss_finalize_kernel", "loss_finalize"),
         ("classify_bwd_kernel", "classify_bwd"), ("reduce_parts_kernel", "pe_bwd_reduce"),
         ("proto_extend_bwd_chain_kernel", "pe_bwd_chain"), ("proto_extend_bwd_params_kernel", "pe_bwd_params"),
         ("masked_sum_bwd_kernel", "masked_sum_bwd"), ("head_fwd_tail_kernel", "head_fwd_tail"),
         ("head_bwd_tail_kernel", "head_bwd_tail")]


# Read-side correction factor per kernel = FETCH_SIZE under-count of its dominant load width, from
# scripts/diag/fetch_calib.hip (profiles/r2_fetch_calibration.txt: a 1 GiB stream read with 4-, 8- and 16-byte
# loads per lane).  The width of each kernel's streamed loads is read off its ISA (scripts/isa_scan.py):
#   16 B/lane: masked_pool_fwd (Fs), bwd_dwn (Fq tiles, dlog slabs as dwords), pe_bwd_reduce, proto_extend_fwd
#    8 B/lane: classify_fwd (Fq pairs), loss_pair_fwd / wce_fwd / focal_fwd (int64 labels), mask_resize
#    4 B/lane: bwd_chain_dfq (slab rows), bwd_params_dfs (mask rows; its 80 MB are WRITES)
LOAD_WIDTH = {"masked_pool_fwd": 16, "masked_sum_bwd": 16, "pe_bwd_reduce": 16, "proto_extend_fwd": 16,
              "bwd_dwn": 16, "classify_bwd": 16, "dfs_chunk": 16,
              "classify_fwd": 8, "loss_pair_fwd": 8, "wce_fwd": 8, "focal_fwd": 8, "wce_fwd_nograd": 8, "mask_resize": 8,
              "bwd_chain_dfq": 4, "bwd_params_dfs": 4, "loss_finalize": 4}
WIDTH_FACTOR = {4: 1.0, 8: 1.0, 16: 2.0}          # overwritten from profiles/r2_fetch_calibration.json if present


def load_calibration():
    path = os.path.join(ROOT, "profiles", "r2_fetch_calibration.json")
    if os.path.exists(path):
        d = json.load(open(path))
        for w, f in d.get("factor_by_width", {}).items():
            WIDTH_FACTOR[int(w)] = float(f)


def read_factor(k):
    return WIDTH_FACTOR.get(LOAD_WIDTH.get(k, 0), 1.0)


def short(name):
    for pat, s in NAMES:
        if pat in name:
            return s
    return None


def find(d, pat):
    f = glob.glob(os.path.join(d, "**", pat), recursive=True)
    return max(f, key=os.path.getmtime) if f else None        # a directory may hold several runs: the newest


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
    ap.add_argument("--out", default=os.path.join(ROOT, "profiles"),
                    help="where to write (on the GPU box: a directory under gpurun_out/, copied into profiles/ afterwards)")
    a = ap.parse_args()
    out = a.out
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
        load_calibration()
        fe, wr = pmc_mean(a.fetch_dir, "FETCH_SIZE"), pmc_mean(a.write_dir, "WRITE_SIZE")
        path = os.path.join(out, "pmc_traffic.json")
        seed = path if os.path.exists(path) else os.path.join(ROOT, "profiles", "pmc_traffic.json")
        allruns = json.load(open(seed)) if os.path.exists(seed) else {}
        if "runs" not in allruns:
            allruns = {"runs": {}}
        allruns["note"] = ("hbm_bytes_per_launch = f*FETCH_SIZE*1024 + WRITE_SIZE*1024 (KiB counters, separate --pmc "
                           "passes); f = the FETCH_SIZE under-count of the kernel's streamed load width, calibrated in "
                           "profiles/r2_fetch_calibration.json (scripts/diag/fetch_calib.hip); see scripts/pmc_summary.py")
        traffic = {"config": a.config, "episodes": a.episodes, "tag": a.tag, "kernels": {}}
        lines.append("")
        lines.append("PMC (mean per launch): kernel, FETCH_SIZE KiB (raw), WRITE_SIZE KiB (raw), corrected HBM bytes")
        for k in sorted(set(fe) | set(wr)):
            f_, w_ = fe.get(k, 0.0), wr.get(k, 0.0)
            fac = read_factor(k)
            hb = int(fac * f_ * 1024 + w_ * 1024)
            traffic["kernels"][k] = {"fetch_size_kib_raw": round(f_, 2), "write_size_kib_raw": round(w_, 2),
                                     "read_factor": fac, "load_width": LOAD_WIDTH.get(k),
                                     "calibrated": k in LOAD_WIDTH, "hbm_bytes_per_launch": hb}
            lines.append(f"{k:24s} fetch={f_:12.2f} write={w_:12.2f} hbm_bytes={hb}")
        allruns["runs"][f"{a.config}/{a.episodes}"] = traffic
        json.dump(allruns, open(path, "w"), indent=1)
    open(os.path.join(out, f"{a.tag}_summary.txt"), "w").write("\n".join(lines) + "\n")
    print("\n".join(lines))


if __name__ == "__main__":
    main()
