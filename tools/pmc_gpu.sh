#!/bin/bash
# Collects a set of SQ counters for the rowstats kernel (separate pass, no tracing). Usage: tools/pmc_gpu.sh <tag> "<counters>" [bench args]
TAG=$1; CTRS=$2; shift; shift
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/pmc_$TAG; rm -rf $OUT
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 150 rocprofv3 --pmc $CTRS --output-format csv -d $OUT -- python3 $REPO/bench.py --timesteps 64 --steps 2 --warmup 1 --no-cpu-baseline "$@" > $OUT/bench.json 2> $OUT/err.log
python3 - <<PY
import csv, glob, collections
rows = collections.defaultdict(list)
for f in glob.glob("$OUT/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "lec_row" in r["Kernel_Name"]:
            rows[r["Counter_Name"]].append(float(r["Counter_Value"]))
            vg = r["VGPR_Count"], r["SGPR_Count"], r["LDS_Block_Size"], r["Workgroup_Size"], r["Grid_Size"]
print("$TAG", "vgpr,sgpr,lds,wg,grid=", vg if rows else "(no rows)")
for k, v in sorted(rows.items()):
    print("  %-28s %.4g (n=%d)" % (k, sum(v) / len(v), len(v)))
PY
