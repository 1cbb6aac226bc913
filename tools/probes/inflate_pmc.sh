#!/bin/bash
# PMC passes over the inflate bench (GPU box).  Usage: tools/probes/inflate_pmc.sh "<set1>;<set2>" STREAMS LEVEL NOISE
SETS=$1; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=/tmp/pmci_$$; rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
IFS=';' read -ra ARR <<< "$SETS"
i=0
for SET in "${ARR[@]}"; do
  [ -z "$SET" ] && continue
  i=$((i+1))
  timeout -k 5 200 rocprofv3 --pmc $SET --output-format csv -d $OUT/p$i -- python3 $R/tools/probes/inflate_check.py --bench-one "$@" > $OUT/b$i.log 2> $OUT/e$i.log || { echo "pass $i failed: $SET"; tail -3 $OUT/e$i.log; }
done
python3 - <<PY
import csv, glob, collections
rows = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$OUT/p*/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "inflate" in r["Kernel_Name"]:
            rows["lec_inflate_kernel"][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k in rows:
    for c, v in sorted(rows[k].items()):
        print("    %-30s %.6g (n=%d)" % (c, sum(v) / len(v), len(v)))
PY
rm -rf $OUT
