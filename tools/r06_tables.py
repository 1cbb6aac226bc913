#!/usr/bin/env python3
"""Prints the measured tables of DESIGN.md section 4.5 / 7 and profiles/r06_notes.md section 7 from the files under profiles/ (so that
the documents quote what the profile set holds).  Usage: python tools/r06_tables.py"""
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
P = os.path.join(ROOT, "profiles")


def load(name):
    with open(os.path.join(P, name)) as f:
        return json.load(f)


ROWS = [
    ("noq", "conversion terms (T,u,v,ω; `--no-q`), fp64, T = 64: BASELINE's target configuration"),
    ("all", "all 16 terms (5 fields + Q stencil), fp64, T = 64: the `bench.py` default"),
    ("f32", "all 16 terms, fp32 storage"),
    ("moving", "moving 61 × 61 box per step, box-packed, T = 512 (`lec_boxplane`)"),
    ("moving2048", "the same, T = 2048"),
    ("moving_cube", "moving, track-extent crop (`--moving-layout cube`, `lec_boxtile`), T = 512"),
]

shas = set()
print("| configuration | dominant stage-1 kernel, ms per call (kernel-trace mean × calls) | time steps/s in `bench.py` | of 8 TB/s | fabric traffic (PMC) / algorithmic |")
print("|---|---|---|---|---|")
for tag, text in ROWS:
    d = load("r06_%s_pmc.json" % tag)
    shas.add(d["csrc_sha"])
    print("| %s | %.3f × %d | %s | **%.3f** | %.3f |" % (text, d["dominant_kernel_avg_ms"], d["calls"], format(round(d["bench_line"]["value"]), ","),
                                                      d["dominant_kernel_frac_of_8TBs"], d["traffic_over_algorithmic"]))
print("sources:", sorted(shas))
print()

print("| series | consumer, ms per pass | steps/s (`value`) | of 8 TB/s | producer (`lec_ingest` gathers + `lec_dtdt`), ms | steps/s all in | CPU leg | sources |")
print("|---|---|---|---|---|---|---|---|")
for name, text in [("n1m4096", "T = 4096 packed (`lec_boxplane`)"), ("n1m4096cube", "T = 4096 cube (`lec_boxtile`)"), ("n1m512", "T = 512 packed"),
                   ("n1f2048", "fixed box, T = 2048 in chunks")]:
    fn = "r06_rehearse_%s.json" % name
    if not os.path.exists(os.path.join(P, fn)):
        continue
    d = load(fn)
    c = d["config"]
    pm = c.get("producer_ms")
    cb = d.get("cpu_baseline") or {}
    print("| %s | %.3f | %s | %.3f | %s | %s | %s | %s |" % (
        text, d["ms_per_step"], format(round(d["value"]), ","), d["roofline"]["frac"],
        ("%.1f + %.1f" % (pm["pack"], pm["dtdt"])) if pm else "inside",
        format(round(c.get("value_incl_producer", d["value"])), ","),
        ("%.2f %s, parity %s" % (cb.get("value", 0), cb.get("unit", ""), (d.get("parity") or {}).get("ok"))) if cb else "-",
        c.get("csrc_sha", "")))
print()
for name in ("g2default", "g4default", "g2m4096", "g4m4096", "g2chunk", "g3mchunk"):
    fn = "r06_rehearse_%s.json" % name
    if not os.path.exists(os.path.join(P, fn)):
        continue
    d = load(fn)
    c = d["config"]
    print("%-10s n_gpus %d backend %-5s value %12.1f ms %9.3f  speedup_vs_n1 %s stale %s  legs %s" % (
        name, d["n_gpus"], c.get("backend"), d["value"], d["ms_per_step"], c.get("speedup_vs_n1"), c.get("n1_stale"),
        [(k, l.get("timesteps_global"), round(l["value"], 1), l.get("speedup_vs_n1")) for k, l in (c.get("strong_scaling") or {}).items() if isinstance(l, dict)]))
if os.path.exists(os.path.join(P, "r06_bench_default.json")):
    d = load("r06_bench_default.json")
    r = d["roofline"]
    print("default bench: value %.1f  ms %.3f  frac %.3f  traffic_stale %s  sha %s  cpu %s  parity %s" % (
        d["value"], d["ms_per_step"], r["frac"], r.get("traffic_stale"), r.get("traffic_csrc_sha"), d["cpu_baseline"]["value"], d.get("parity", {}).get("ok")))
