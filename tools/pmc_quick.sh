#!/bin/bash
# Quick PMC passes (no tracing) over a bench.py command on the GPU box; prints the counters of the lec_* stage-1 kernels.
# Usage: tools/pmc_quick.sh "<set1>;<set2>;..." <bench args...>        (LEC_LIB may select a variant build)
SETS=$1; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=/tmp/pmcq_$$; rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
IFS=';' read -ra ARR <<< "$SETS"
i=0
for SET in "${ARR[@]}"; do
  [ -z "$SET" ] && continue
  i=$((i+1))
  timeout -k 5 100 rocprofv3 --pmc $SET --kernel-include-regex lec_ --output-format csv -d $OUT/p$i -- python3 $R/bench.py --steps 2 --warmup 1 --cpu-baseline none "$@" > $OUT/b$i.json 2> $OUT/e$i.log
  rc=$?
  if [ $rc -ne 0 ]; then     # say WHICH: 124 / 137 = killed at the limit, anything else = the run itself failed; no further GPU step after it
    echo "pass $i rc=$rc ($([ $rc -eq 124 -o $rc -eq 137 ] && echo killed at its time limit || echo failed)): $SET"; tail -3 $OUT/e$i.log; break
  fi
done
python3 - <<PY
import csv, glob, collections
rows = collections.defaultdict(lambda: collections.defaultdict(list)); meta = {}
for f in glob.glob("$OUT/p*/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"]
        if "lec_" in n and (not "$LEC_PMC_ONLY" or "$LEC_PMC_ONLY" in n):
            k = n.replace("(anonymous namespace)::", "").replace("void ", "").split("(lec::")[0][:80]
            rows[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
            meta[k] = (r["VGPR_Count"], r["SGPR_Count"], r["LDS_Block_Size"], r["Workgroup_Size"], r["Grid_Size"])
for k in rows:
    print("[%s] vgpr,sgpr,lds,wg,grid = %s" % (k, meta[k]))
    for c, v in sorted(rows[k].items()):
        print("    %-30s %.6g (n=%d)" % (c, sum(v) / len(v), len(v)))
PY
rm -rf $OUT
