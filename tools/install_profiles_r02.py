#!/usr/bin/env python3
"""Copy the summaries tools/profile_r02.sh and tools/evidence_r02.sh left in gpurun_out/ into profiles/ (trimming the
kilobyte-long torch kernel names) and rebuild profiles/pmc_traffic.json from the FETCH_SIZE / WRITE_SIZE passes."""
import json
import os
import shutil

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src, ev, dst = (os.path.join(ROOT, p) for p in ("gpurun_out/prof_r02", "gpurun_out/evidence_r02", "profiles"))
S = {}
for f in ("r02_s20_n4096", "r02_default_n4096", "r02_pmc_n4096", "r02_pmc_n1m", "r02_sq_coop_n4096", "r02_sq_singlewave_n4096"):
    d = json.load(open(os.path.join(src, f + "_rocprof_summary.json")))
    for k in d["kernel_stats"]:
        if len(k["name"]) > 160:
            k["name"] = k["name"][:157] + "..."
    d["kernel_stats"] = d["kernel_stats"][:12]
    json.dump(d, open(os.path.join(dst, f + "_rocprof_summary.json"), "w"), indent=1)
    S[f] = d
    t = d["step_kernel_trace"]
    print("%-26s %-42s dispatches %6d avg %6.0f ns median %6.0f start-to-start %6.0f" % (f, t["kernel"][:42], t["dispatches"], t["avg_ns"], t["median_ns"], t["median_start_to_start_ns"]))
for f in ("r02_s20_kernel_stats.csv", "r02_default_kernel_stats.csv"):
    out = []
    for ln in open(os.path.join(src, f)).read().splitlines():
        if len(ln) > 400:
            ln = '"' + ln[1:150] + '..."' + ln[ln.rfind('",') + 1:]
        out.append(ln)
    open(os.path.join(dst, f), "w").write("\n".join(out) + "\n")
for f in ("bench_s20_profiled.json", "bench_default_profiled.json"):
    shutil.copy(os.path.join(src, f), os.path.join(dst, "r02_" + f))


for f in ("r02_profiler_calibration.txt", "r02_calib_empty_rocprof_summary.json"):
    if os.path.exists(os.path.join(src, f)):
        shutil.copy(os.path.join(src, f), os.path.join(dst, f))


def traffic(d):
    return (d["pmc"]["fetch"]["FETCH_SIZE"]["mean_per_dispatch"] * 1024 * 2,      # KiB, doubled per the gfx950 calibration
            d["pmc"]["write"]["WRITE_SIZE"]["mean_per_dispatch"] * 1024)


p = json.load(open(os.path.join(dst, "pmc_traffic.json")))
(f4, w4), (f1, w1) = traffic(S["r02_pmc_n4096"]), traffic(S["r02_pmc_n1m"])
p["config3"] = {"4096": f4 + w4, "1048576": f1 + w1}
p["detail"] = {"4096": {"fetch_bytes": f4, "write_bytes": w4, "per_env_step": (f4 + w4) / 4096, "kernel": S["r02_pmc_n4096"]["step_kernel_trace"]["kernel"]},
               "1048576": {"fetch_bytes": f1, "write_bytes": w1, "per_env_step": (f1 + w1) / 1048576, "kernel": S["r02_pmc_n1m"]["step_kernel_trace"]["kernel"]}}
json.dump(p, open(os.path.join(dst, "pmc_traffic.json"), "w"), indent=1)
print("traffic per env-step:", {k: round(v["per_env_step"], 1) for k, v in p["detail"].items()})
for f in ("r02_bench_default.json", "r02_bench_steps20.json", "r02_coop_vs_singlewave.txt", "r02_coop_timeline.txt", "r02_env_count_sweep.txt",
          "r02_fragment_length.txt", "r02_kernel_start_latency.txt"):
    if os.path.exists(os.path.join(ev, f)):
        shutil.copy(os.path.join(ev, f), os.path.join(dst, f))
for f in ("r02_bench_default.json", "r02_bench_steps20.json"):
    d = json.loads(open(os.path.join(dst, f)).read())
    r = d["roofline"]
    print("%-24s value %.4g  ms/step %.5f  kernel_us %.3f  frac %.4f  rocprofv3 %.3f us" % (f, d["value"], d["ms_per_step"], r["kernel_us"], r["frac"], r["rocprofv3_avg_kernel_us"] or -1))
