#!/usr/bin/env python3
"""Condense rocprofv3 output directories (gpurun_out/...) into the small summaries kept under profiles/.

usage: python profiles/summarize.py <round-tag> <stats_dir> [--kernel substring] [--out dir] [--pmc name=dir ...] [--calib dir]
                                    [--grid threads] [--cut steps]
  --grid    : keep only dispatches of this grid size (a run holds launches of other env counts too)
  --cut     : a persistent kernel's duration (and every counter) scales with its step count, which the trace does not carry.  The
              longest dispatches of a bench.py run are its 1024-step fragments; with x = value / (max value / 1024) as the step
              estimate of a dispatch (the typical longest launch, not one slow outlier, as the reference), only those within 3 % of that
              reference (cut = 1024; --tol) or with x in [0.8 cut, 1.25 cut + 8] (shorter
              runs: the +8 is the fixed cost of a launch in units of steps) are summarised; recorded as steps_per_launch
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


def select(rows, grid, cut):
    """the dispatches a summary is about: one grid size, and for a persistent kernel the cluster of longest launches"""
    if grid:
        rows = [r for r in rows if int(r["Grid_Size_X"]) == grid]
    if cut and rows:
        dur = np.array([int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rows], dtype=float)
        rows = [r for r, k in zip(rows, cut_mask(dur, cut)) if k]
    return rows


LONGEST = 1024     # --longest: the step count of the longest launches of the profiled command (bench.py --fragment)
TOL = 0.97         # --tol: a launch belongs to the longest cluster when it lasts at least this share of the longest one


def long_centre(v):
    """the typical value of the LONGEST-step launches: the centre of the most populated +-3 % window among the values of at least
    half the maximum (the maximum itself may be one slow outlier: a first launch, a clock ramp)"""
    cand = v[v >= 0.5 * v.max()]
    counts = [(np.abs(cand - d) <= 0.03 * d).sum() for d in cand]
    return float(cand[int(np.argmax(counts))])


def cut_mask(v, cut):
    c = long_centre(v)
    if cut >= LONGEST:
        return np.abs(v - c) <= (1.0 - TOL) * c
    x = v / (c / float(LONGEST))
    return (x >= 0.8 * cut) & (x <= 1.25 * cut + 8)


def trace_stats(d, key="k_step", grid=0, cut=0):
    rows = [r for r in csv.DictReader(open(one(os.path.join(d, "**", "*_kernel_trace.csv")))) if key in r["Kernel_Name"]]
    all_dispatches = len(rows)
    rows = select(rows, grid, cut)
    st = np.array([int(r["Start_Timestamp"]) for r in rows]); en = np.array([int(r["End_Timestamp"]) for r in rows])
    dur = en - st
    r0 = rows[0]
    return {"kernel": r0["Kernel_Name"], "dispatches": len(rows), "dispatches_of_this_kernel_in_the_run": all_dispatches,
            "steps_per_launch": cut or 1, "avg_ns": float(dur.mean()), "median_ns": float(np.median(dur)),
            "p10_ns": float(np.percentile(dur, 10)), "p90_ns": float(np.percentile(dur, 90)),
            "median_start_to_start_ns": float(np.median(np.diff(st))) if len(st) > 1 else None, "vgpr": int(r0["VGPR_Count"]),
            "accum_vgpr": int(r0["Accum_VGPR_Count"]), "sgpr": int(r0["SGPR_Count"]), "lds_bytes": int(r0["LDS_Block_Size"]),
            "scratch_bytes": int(r0["Scratch_Size"]), "grid": int(r0["Grid_Size_X"]), "workgroup": int(r0["Workgroup_Size_X"])}


def pmc(d, key="k_step", grid=0, cut=0):
    """mean counter value per dispatch.  With --cut only the dispatches of the longest-launch cluster of each counter are kept (cut_mask:
    within 1 - TOL of the typical largest value): the counter passes carry no duration, and a persistent kernel's counters scale
    with its step count"""
    acc = defaultdict(list)
    for r in csv.DictReader(open(one(os.path.join(d, "**", "*_counter_collection.csv")))):
        if key in r["Kernel_Name"] and (not grid or int(r["Grid_Size_X"] if "Grid_Size_X" in r else r.get("Grid_Size", 0)) == grid):
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    out = {}
    for k, v in acc.items():
        v = np.array(v)
        if cut and len(v):
            v = v[cut_mask(v, cut)]
        out[k] = {"mean_per_dispatch": float(np.mean(v)), "samples": int(len(v))}
    return out


def main():
    tag, stats_dir = sys.argv[1], sys.argv[2]
    key, outdir, grid, cut = "k_step", os.path.dirname(os.path.abspath(__file__)), 0, 0
    for j, a in enumerate(sys.argv):
        if a == "--kernel":
            key = sys.argv[j + 1]
        if a == "--out":
            outdir = sys.argv[j + 1]
        if a == "--grid":
            grid = int(sys.argv[j + 1])
        if a == "--cut":
            cut = int(sys.argv[j + 1])
        if a == "--longest":
            global LONGEST
            LONGEST = int(sys.argv[j + 1])
        if a == "--tol":
            global TOL
            TOL = float(sys.argv[j + 1])
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "mujoco-drone_amd"))
    try:    # the library the profiled command loaded was built from these sources (bench.py checks the hash before quoting a profile)
        import build as _b
        src_hash = _b.embedded_hash()
    except Exception:
        src_hash = None
    out = {"tag": tag, "kernel_filter": key, "source_hash": src_hash, "kernel_stats": kernel_stats(stats_dir),
           "step_kernel_trace": trace_stats(stats_dir, key, grid, cut)}
    i = 3
    while i < len(sys.argv):
        if sys.argv[i] == "--pmc":
            name, d = sys.argv[i + 1].split("=")
            out.setdefault("pmc", {})[name] = pmc(d, key, grid, cut)
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
