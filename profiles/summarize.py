#!/usr/bin/env python3
"""Condense rocprofv3 output directories (gpurun_out/...) into the small summaries kept under profiles/.

usage: python profiles/summarize.py <round-tag> <stats_dir> [--kernel substring] [--out dir] [--pmc name=dir ...] [--calib dir]
  stats_dir : output of  rocprofv3 --kernel-trace --stats --output-format csv -d <dir> -- python bench.py ...
  pmc dirs  : outputs of rocprofv3 --kernel-trace --pmc <counters> --output-format csv -d <dir> -- python bench.py ...
"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

import numpy as np


def one(pattern):
    f = glob.glob(pattern, recursive=True)
    if not f:
        raise SystemExit("no file matches " + pattern)
    return f[0]


def kernel_stats(d):
    rows = list(csv.DictReader(open(one(os.path.join(d, "**", "*_kernel_stats.csv")))))
    return [{"name": r["Name"], "calls": int(r["Calls"]), "avg_ns": float(r["AverageNs"]), "min_ns": float(r["MinNs"]),
             "max_ns": float(r["MaxNs"]), "pct": float(r["Percentage"])} for r in rows]


def trace_stats(d, key="k_step"):
    rows = [r for r in csv.DictReader(open(one(os.path.join(d, "**", "*_kernel_trace.csv")))) if key in r["Kernel_Name"]]
    st = np.array([int(r["Start_Timestamp"]) for r in rows]); en = np.array([int(r["End_Timestamp"]) for r in rows])
    dur = en - st
    r0 = rows[0]
    return {"kernel": r0["Kernel_Name"], "dispatches": len(rows), "avg_ns": float(dur.mean()), "median_ns": float(np.median(dur)),
            "p10_ns": float(np.percentile(dur, 10)), "p90_ns": float(np.percentile(dur, 90)),
            "median_start_to_start_ns": float(np.median(np.diff(st))), "vgpr": int(r0["VGPR_Count"]),
            "accum_vgpr": int(r0["Accum_VGPR_Count"]), "sgpr": int(r0["SGPR_Count"]), "lds_bytes": int(r0["LDS_Block_Size"]),
            "scratch_bytes": int(r0["Scratch_Size"]), "grid": int(r0["Grid_Size_X"]), "workgroup": int(r0["Workgroup_Size_X"])}


def pmc(d, key="k_step"):
    acc = defaultdict(list)
    for r in csv.DictReader(open(one(os.path.join(d, "**", "*_counter_collection.csv")))):
        if key in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    return {k: {"mean_per_dispatch": float(np.mean(v)), "samples": len(v)} for k, v in acc.items()}


def main():
    tag, stats_dir = sys.argv[1], sys.argv[2]
    key, outdir = "k_step", os.path.dirname(os.path.abspath(__file__))
    for j, a in enumerate(sys.argv):
        if a == "--kernel":
            key = sys.argv[j + 1]
        if a == "--out":
            outdir = sys.argv[j + 1]
    out = {"tag": tag, "kernel_filter": key, "kernel_stats": kernel_stats(stats_dir), "step_kernel_trace": trace_stats(stats_dir, key)}
    i = 3
    while i < len(sys.argv):
        if sys.argv[i] == "--pmc":
            name, d = sys.argv[i + 1].split("=")
            out.setdefault("pmc", {})[name] = pmc(d, key)
            i += 2
        elif sys.argv[i] == "--calib":
            out["fetch_calibration"] = pmc(sys.argv[i + 1], key="")
            i += 2
        else:
            i += 1
    path = os.path.join(outdir, "%s_rocprof_summary.json" % tag)
    json.dump(out, open(path, "w"), indent=1)
    print("wrote", path)


if __name__ == "__main__":
    main()
