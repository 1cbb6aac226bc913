#!/usr/bin/env python3
"""Turns the rocprofv3 output of tools/profile_gpu.sh (gpurun_out/prof_<tag>/) into the small, tracked
summaries under profiles/: <tag>_kernel_stats.csv (rocprofv3 --kernel-trace --stats), <tag>_pmc.json
(FETCH_SIZE / WRITE_SIZE of the row kernel, corrected as MI355X_MICROARCH.md prescribes) and an
entry in profiles/pmc_summary.json that bench.py reads for roofline.traffic.

Usage: tools/summarize_profile.py <tag> <storage f64|f32> <all|noq>
"""
import csv
import glob
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag, storage, terms = sys.argv[1], sys.argv[2], sys.argv[3]
src = os.path.join(ROOT, "gpurun_out", f"prof_{tag}")
dst = os.path.join(ROOT, "profiles")
os.makedirs(dst, exist_ok=True)

# kernel stats
stats = glob.glob(os.path.join(src, "stats", "*", "*_kernel_stats.csv"))[0]
rows = list(csv.DictReader(open(stats)))
with open(os.path.join(dst, f"{tag}_kernel_stats.csv"), "w", newline="") as f:
    w = csv.DictWriter(f, fieldnames=rows[0].keys())
    w.writeheader()
    for i, r in enumerate(rows):
        if i < 12 or "lec_" in r["Name"]:
            w.writerow(r)
# stage 1 (one lec_rowstats call) may be several kernels: the first time step on lec_rowsweep_kernel, the rest on
# lec_rowblock_kernel, then lec_qtime_kernel; the dominant one is reported by name, the call by the sum
stage1 = [r for r in rows if "lec_row" in r["Name"] or "lec_qtime" in r["Name"]]
row_k = max(stage1, key=lambda r: float(r["TotalDurationNs"]))
calls = int(row_k["Calls"])
stage1_ms = sum(float(r["TotalDurationNs"]) for r in stage1) / calls / 1e6


def counter(sub, name):
    """Counter summed over the stage-1 kernels of one lec_rowstats call (averaged over the calls of the run)."""
    f = glob.glob(os.path.join(src, sub, "*", "*_counter_collection.csv"))[0]
    per_kernel = {}
    for r in csv.DictReader(open(f)):
        if ("lec_row" in r["Kernel_Name"] or "lec_qtime" in r["Kernel_Name"]) and r["Counter_Name"] == name:
            per_kernel.setdefault(r["Kernel_Name"], []).append(float(r["Counter_Value"]))
    n = max(len(v) for v in per_kernel.values())
    return sum(sum(v) / len(v) for v in per_kernel.values()), n


fetch_kb, n = counter("pmc_fetch", "FETCH_SIZE")
write_kb, _ = counter("pmc_write", "WRITE_SIZE")
bench_pmc = json.load(open(os.path.join(src, "bench_pmc_fetch.json")))
bench_stats = json.load(open(os.path.join(src, "bench_stats.json")))
t_pmc = bench_pmc["config"]["timesteps_per_gpu"]
t_stats = bench_stats["config"]["timesteps_per_gpu"]
alg_pmc = bench_pmc["roofline"]["algorithmic_bytes_per_launch"]
# gfx950: FETCH_SIZE (KiB) tallies the 128-B requests of 16-B-per-lane streaming loads at 64 B -> x2;
# calibrated on this kernel's no-Q configuration (known byte count): 2 * FETCH_SIZE * 1024 = 1.000 x algorithmic.
read_bytes = 2.0 * fetch_kb * 1024.0
write_bytes = write_kb * 1024.0
summary = {
    "tag": tag, "storage": storage, "terms": terms,
    "kernel": row_k["Name"],
    "rocprof_avg_launch_ms": float(row_k["AverageNs"]) / 1e6, "rocprof_calls": int(row_k["Calls"]),
    "stage1_kernels_ms_per_call": {re.search(r"(lec_\w+)", r["Name"]).group(1) + re.sub(r"^[^<]*", "", r["Name"].split("(lec::")[0]):
                                   float(r["TotalDurationNs"]) / calls / 1e6 for r in stage1},
    "rocprof_stage1_ms_per_call": stage1_ms,
    "timesteps_per_launch_stats_run": t_stats,
    "bench_event_avg_launch_ms": bench_stats["roofline"]["avg_launch_ms"],
    "pmc_run_timesteps_per_launch": t_pmc, "pmc_dispatches": n,
    "FETCH_SIZE_KiB_raw": fetch_kb, "WRITE_SIZE_KiB_raw": write_kb,
    "hbm_read_bytes_per_launch_corrected": read_bytes, "hbm_write_bytes_per_launch": write_bytes,
    "algorithmic_bytes_per_launch": alg_pmc,
    "traffic_over_algorithmic": (read_bytes + write_bytes) / alg_pmc,
    "correction": "read = 2 x FETCH_SIZE x 1024 (gfx950 wide-load under-count, MI355X_MICROARCH.md HBM section; "
                  "calibrated on the T,u,v,omega-only configuration where it reproduces the algorithmic bytes to 0.1 %)",
}
json.dump(summary, open(os.path.join(dst, f"{tag}_pmc.json"), "w"), indent=1)
allp = os.path.join(dst, "pmc_summary.json")
agg = json.load(open(allp)) if os.path.exists(allp) else {}
# per-launch traffic scaled to the default bench launch (T = 64 per launch): traffic is proportional to time steps
agg[f"rowstats_{storage}_{terms}_hbm_bytes_per_timestep"] = (read_bytes + write_bytes) / t_pmc
agg[f"rowstats_{storage}_{terms}_source"] = f"profiles/{tag}_pmc.json"
json.dump(agg, open(allp, "w"), indent=1)
print(json.dumps(summary, indent=1))
