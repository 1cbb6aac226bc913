#!/bin/bash
# Runs on the GPU box (via gpurun): rocprofv3 kernel-trace stats of a bench.py command, then separate PMC passes (no tracing;
# one counter set per pass), and prints a compact summary of the lec_* kernels.
# Usage: tools/prof.sh <tag> "<counter sets separated by ';' or empty>" <bench args...>
TAG=$1; SETS=$2; shift; shift
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/prof_$TAG
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $REPO/bench.py --cpu-baseline none "$@" > $OUT/bench_stats.json 2> $OUT/stats.err || { echo "stats pass failed"; tail -5 $OUT/stats.err; exit 1; }
# (gpurun copies at most 64 MiB back: of the kernel trace only the header and the lec_ launches are kept -- a moving run's trace holds
# tens of thousands of torch launches of the synthetic generator)
for f in $OUT/stats/*/*_kernel_trace.csv; do head -1 "$f" > "$f.tmp"; grep lec_ "$f" >> "$f.tmp"; mv "$f.tmp" "$f"; done
i=0
IFS=';' read -ra ARR <<< "$SETS"
for SET in "${ARR[@]}"; do
  [ -z "$SET" ] && continue
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --pmc $SET --kernel-include-regex lec_ --output-format csv -d $OUT/pmc_$i -- python3 $REPO/bench.py --cpu-baseline none --steps 2 --warmup 1 "$@" > $OUT/bench_pmc_$i.json 2> $OUT/pmc_$i.err || { echo "pmc pass $i failed"; tail -5 $OUT/pmc_$i.err; exit 1; }
done
python3 - <<PY
import csv, glob, collections, json
print("== $TAG:", "$*")
try:
    b = json.load(open("$OUT/bench_stats.json")); print("  bench: value %.1f  ms/step %.3f  roofline frac %.3f  launch ms %.3f" % (b["value"], b["ms_per_step"], b["roofline"]["frac"], b["roofline"]["avg_launch_ms"]))
except Exception as e:
    print("  bench json:", e)
for f in glob.glob("$OUT/stats/*/*_kernel_stats.csv"):
    for r in csv.DictReader(open(f)):
        if "lec_" in r["Name"]:
            print("  %-90s calls %5s avg %10.1f us" % (r["Name"][:90], r["Calls"], float(r["AverageNs"]) / 1e3))
rows = collections.defaultdict(lambda: collections.defaultdict(list)); meta = {}
for f in glob.glob("$OUT/pmc_*/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "lec_" in r["Kernel_Name"]:
            k = r["Kernel_Name"][:60]
            rows[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
            meta[k] = (r["VGPR_Count"], r["SGPR_Count"], r["LDS_Block_Size"], r["Workgroup_Size"], r["Grid_Size"])
for k in rows:
    print("  [%s] vgpr,sgpr,lds,wg,grid = %s" % (k, meta[k]))
    for c, v in sorted(rows[k].items()):
        print("      %-30s %.5g (n=%d)" % (c, sum(v) / len(v), len(v)))
PY
