#!/usr/bin/env python3
"""Copy the summaries tools/profile_r04.sh left in gpurun_out/prof_r04/ into profiles/ (trimming the kilobyte-long torch kernel
names) and write profiles/current.json: the figures bench.py may quote next to its live measurement -- rocprofv3's average
duration of the dominant kernel and its HBM-side bytes per launch -- keyed by kernel / config / env count and stamped with the
source hash of the library they were taken from (bench.py drops them when another library is loaded)."""
import json
import os
import shutil

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src, dst = os.path.join(ROOT, "gpurun_out/prof_r04"), os.path.join(ROOT, "profiles")
S = {}
for f in sorted(os.listdir(src)):
    if not f.endswith("_rocprof_summary.json"):
        continue
    d = json.load(open(os.path.join(src, f)))
    for k in d["kernel_stats"]:
        if len(k["name"]) > 160:
            k["name"] = k["name"][:157] + "..."
    d["kernel_stats"] = d["kernel_stats"][:12]
    json.dump(d, open(os.path.join(dst, f), "w"), indent=1)
    S[f[:-len("_rocprof_summary.json")]] = d
    t = d["step_kernel_trace"]
    print("%-22s %-40s dispatches %4d (of %4d) avg %10.0f ns median %10.0f  steps/launch %d" % (
        f[:22], t["kernel"][:40], t["dispatches"], t.get("dispatches_of_this_kernel_in_the_run", -1), t["avg_ns"], t["median_ns"], t.get("steps_per_launch", 1)))
for f in os.listdir(src):
    if f.endswith("_kernel_stats.csv"):
        out = []
        for ln in open(os.path.join(src, f)).read().splitlines():
            if len(ln) > 400:
                ln = '"' + ln[1:150] + '..."' + ln[ln.rfind('",') + 1:]
            out.append(ln)
        open(os.path.join(dst, f), "w").write("\n".join(out) + "\n")
    elif f.startswith("bench_") and f.endswith(".json"):
        shutil.copy(os.path.join(src, f), os.path.join(dst, "r04_" + f))
    elif f.endswith(".txt"):
        shutil.copy(os.path.join(src, f), os.path.join(dst, f))


def short(name):     # "void qd::k_rollout_coop<1>(qd::KArgs, ...)" -> "qd::k_rollout_coop<1>"
    name = name.split("(")[0].replace(", ", ",").replace(",false>", ">")     # (the library's selector leaves the PID = false flag out)
    return name[5:] if name.startswith("void ") else name   # (a plain kernel like qd::k_rollout_pair comes without the "void ")


cur = {"_comment": "written by tools/install_profiles_r04.py from the rocprofv3 runs of tools/profile_r04.sh; bench.py quotes an entry only "
                   "when source_hash equals the loaded library's qd_source_hash()", "source_hash": None, "kernels": {}}
hashes = set()
for tag, ptag, qtag, conf, n in (("r04_default_n4096", "r04_pmc_n4096", "r04_sq_n4096", "config3", 4096),
                                 ("r04_pmc_config5_n8192", "r04_pmc_config5_n8192", "r04_pmc_config5_n8192", "config5", 8192),
                                 ("r04_pmc_config2_n4096", "r04_pmc_config2_n4096", "r04_pmc_config2_n4096", "config2", 4096),
                                 ("r04_pmc_config3_n1048576", "r04_pmc_config3_n1048576", "r04_pmc_config3_n1048576", "config3", 1048576),
                                 ("r04_pmc_config5_n1048576", "r04_pmc_config5_n1048576", "r04_pmc_config5_n1048576", "config5", 1048576),
                                 ("r04_pmc_config2_n1048576", "r04_pmc_config2_n1048576", "r04_pmc_config2_n1048576", "config2", 1048576)):
    if tag not in S:
        continue
    t = S[tag]["step_kernel_trace"]
    ent = {"rocprofv3_avg_kernel_us": t["avg_ns"] * 1e-3, "dispatches": t["dispatches"], "steps_per_launch": t.get("steps_per_launch", 1),
           "from": "profiles/%s_rocprof_summary.json" % tag}
    hashes.add(S[tag].get("source_hash"))
    p = S.get(ptag)
    if p and "pmc" in p:
        f = p["pmc"]["fetch"]["FETCH_SIZE"]["mean_per_dispatch"] * 1024 * 2      # KiB, doubled per the gfx950 calibration (MI355X_MICROARCH.md)
        w = p["pmc"]["write"]["WRITE_SIZE"]["mean_per_dispatch"] * 1024
        steps = n * ent["steps_per_launch"]
        ent.update(hbm_bytes_per_launch=f + w, fetch_bytes=f, write_bytes=w, hbm_bytes_per_env_step=(f + w) / steps,
                   traffic_from="profiles/%s_rocprof_summary.json" % ptag)
        hashes.add(p.get("source_hash"))
        print("%s %d envs: traffic per env-step %.1f B (fetch %.1f, write %.1f); %.3f us per step" % (
            conf, n, (f + w) / steps, f / steps, w / steps, ent["rocprofv3_avg_kernel_us"] / ent["steps_per_launch"]))
    q = (S.get(qtag) or {}).get("pmc", {}).get("sq")
    if q and "SQ_INSTS_VALU" in q and "GRBM_GUI_ACTIVE" in q:
        # VALU issue: a wave64 instruction occupies its SIMD16 for 4 cycles; the chip has 256 CUs x 4 SIMDs.  GRBM_GUI_ACTIVE is summed
        # over the 8 XCDs (each counts the launch's cycles), SQ_INSTS_VALU over all SIMDs.
        insts, cyc = q["SQ_INSTS_VALU"]["mean_per_dispatch"], q["GRBM_GUI_ACTIVE"]["mean_per_dispatch"] / 8.0
        waves = q.get("SQ_WAVES", {}).get("mean_per_dispatch")
        simds = min(1024.0, waves) if waves else 1024.0      # (persistent kernels: one wave per SIMD up to a full chip)
        ent.update(valu_insts_per_launch=insts, kernel_cycles=cyc, valu_issue_frac=insts * 4.0 / (1024.0 * cyc),
                   valu_issue_frac_of_occupied_simds=insts * 4.0 / (simds * cyc), waves_per_launch=waves,
                   wait_frac_of_wave_cycles=(q["SQ_WAIT_ANY"]["mean_per_dispatch"] / q["SQ_WAVE_CYCLES"]["mean_per_dispatch"]) if "SQ_WAIT_ANY" in q and "SQ_WAVE_CYCLES" in q else None,
                   issue_from="profiles/%s_rocprof_summary.json" % qtag)
        hashes.add(S[qtag].get("source_hash"))
        print("%s %d envs: VALU issue %.1f %% of the chip (%.1f %% of the occupied SIMDs), %.0f instructions per 64 env-steps" % (
            conf, n, 100 * ent["valu_issue_frac"], 100 * ent["valu_issue_frac_of_occupied_simds"], insts / (n / 64.0 * ent["steps_per_launch"])))
    cur["kernels"]["%s/%s/%d" % (short(t["kernel"]), conf, n)] = ent
if len(hashes) == 1:
    cur["source_hash"] = hashes.pop()
else:
    print("WARNING: the summaries come from different libraries:", hashes)
json.dump(cur, open(os.path.join(dst, "current.json"), "w"), indent=1)
print("wrote profiles/current.json for library", str(cur["source_hash"])[:12])
