#!/usr/bin/env python3
"""Turns gpurun_out/prof_<tag>/ (written by tools/prof.sh on the GPU box) into the tracked summaries under profiles/:
<round>_<tag>_kernel_stats.csv (rocprofv3 --kernel-trace --stats, the lec_* kernels and the dozen busiest others),
<round>_<tag>_pmc.json (FETCH_SIZE of the stage-1 kernels, corrected as MI355X_MICROARCH.md prescribes: x2, requests of 128 B are
tallied at 64 B -- calibrated on configurations with a known byte count: the fixed-box conversion-terms run reads 1.00-1.03 x its
algorithmic bytes after the correction, the moving no-Q run 1.255 x = its 128-byte-line over-fetch of 488-byte rows), and an entry in
profiles/pmc_summary.json that bench.py reads for roofline.traffic.

Usage: tools/summarize_prof.py <round> <tag> <pmc_summary key or ->      (keys: rowstats_f64_all_hbm_bytes_per_timestep, rowstats_f64_noq_...,
rowstats_f32_all_..., rowstats_moving_hbm_bytes_per_timestep)"""
import csv
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
rnd, tag, key = sys.argv[1], sys.argv[2], sys.argv[3]
src = os.path.join(ROOT, "gpurun_out", f"prof_{tag}")
dst = os.path.join(ROOT, "profiles")
import re

noq_run = "--no-q" in open(os.path.join(src, "bench_stats.json")).read() or "T,u,v,omega only" in open(os.path.join(src, "bench_stats.json")).read()


def stage1(n):
    """Kernels of one lec_rowstats call of the benched configuration.  The default bench line also times the conversion-terms
    configuration (roofline.conversion_terms): those launches are lec_rowsweep_kernel<..., MODE 0, ...> and are not part of it."""
    if not any(s in n for s in ("lec_rowsweep", "lec_rowblock", "lec_rowstats", "lec_boxtile", "lec_qtime")):
        return False
    m = re.search(r"lec_rowsweep_kernel<\w+, \d+, \w+, (\d+),", n)
    return not (m and m.group(1) == "0" and not noq_run)

stats = max(glob.glob(os.path.join(src, "stats", "*", "*_kernel_stats.csv")), key=os.path.getmtime)      # gpurun merges runs: newest
rows = list(csv.DictReader(open(stats)))
with open(os.path.join(dst, f"{rnd}_{tag}_kernel_stats.csv"), "w", newline="") as f:
    w = csv.DictWriter(f, fieldnames=rows[0].keys())
    w.writeheader()
    for i, r in enumerate(rows):
        if i < 12 or "lec_" in r["Name"]:
            w.writerow(r)
s1 = [r for r in rows if stage1(r["Name"])]
main = max(s1, key=lambda r: float(r["TotalDurationNs"]))
calls = int(main["Calls"])
# (the --moving line's 8-step check launches add ~1 % to this per-call figure; the dominant kernel's own average below is exact only for the
# fixed-box configurations, where every launch of it is a timed-pass launch)
stage1_ms = sum(float(r["TotalDurationNs"]) for r in s1) / calls / 1e6
bench = json.load(open(os.path.join(src, "bench_stats.json")))
out = {"tag": tag, "bench_line": bench, "stage1_kernels_ms_per_call": stage1_ms, "dominant_kernel": main["Name"],
       "dominant_kernel_avg_ms": float(main["AverageNs"]) / 1e6, "calls": calls}
pm = glob.glob(os.path.join(src, "pmc_1", "*", "*_counter_collection.csv"))
if pm:
    per = {}
    for r in csv.DictReader(open(max(pm, key=os.path.getmtime))):
        if stage1(r["Kernel_Name"]) and r["Counter_Name"] == "FETCH_SIZE":
            per.setdefault(r["Kernel_Name"], {}).setdefault(r["Grid_Size"], []).append(float(r["Counter_Value"]))
    # a bench line may launch a kernel at other sizes too (the --moving line's check of 8 steps against the one-wave-per-row kernel): only the
    # launches of the timed pass count, i.e. per kernel the grid size it was launched with most often; kernels seen once only are not the timed pass's
    per = {k: max(g.values(), key=len) for k, g in per.items()}
    per = {k: v for k, v in per.items() if len(v) > 1 or len(per) == 1}
    fetch_kib = sum(sum(v) / len(v) for v in per.values())
    bp = json.load(open(os.path.join(src, "bench_pmc_1.json")))
    alg = bp["roofline"]["algorithmic_bytes_per_launch"]
    read_bytes = 2.0 * fetch_kib * 1024.0
    t_per_launch = bp["config"]["timesteps_per_gpu"]
    out.update(fetch_size_kib_per_call=fetch_kib, hbm_read_bytes_per_call_corrected=read_bytes, algorithmic_bytes_per_call=alg,
               traffic_over_algorithmic=read_bytes / alg, timesteps_per_call=t_per_launch)
    if key != "-":
        p = os.path.join(dst, "pmc_summary.json")
        summ = json.load(open(p)) if os.path.exists(p) else {}
        summ[key] = read_bytes / t_per_launch
        summ[key.replace("_hbm_bytes_per_timestep", "_source")] = f"profiles/{rnd}_{tag}_pmc.json"
        json.dump(summ, open(p, "w"), indent=1)
json.dump(out, open(os.path.join(dst, f"{rnd}_{tag}_pmc.json"), "w"), indent=1)
print(tag, "stage1 ms/call %.3f" % stage1_ms, "dominant", main["Name"][:70], "%.3f ms" % (float(main["AverageNs"]) / 1e6),
      "traffic/alg %.3f" % out.get("traffic_over_algorithmic", float("nan")))
