#!/bin/bash
# Why did round 4's `rocprofv3 --pmc` passes over `bench.py --moving` hit their time limits?  Runs on the GPU box (via gpurun).
# Measures, for ONE counter set and one bench command line: (a) the plain wall clock, (b) the wall clock under --pmc with every
# dispatch instrumented, with the number of dispatches the counter file lists -> cost per instrumented dispatch, (c) the same with
# collection restricted to the lec_* kernels (--kernel-include-regex).  Every leg is bounded and prints a line when it ends.
# Usage: tools/pmc_cause.sh <timesteps> [extra bench args]
T=${1:-128}; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/pmc_cause_T$T; rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
ARGS="--moving --timesteps $T --steps 2 --warmup 1 --cpu-baseline none $*"
now() { date +%s%N; }
secs() { echo "$(( ($2 - $1) / 1000000 )) ms"; }
t0=$(now); timeout -k 10 280 python3 $R/bench.py $ARGS > $OUT/plain.json 2> $OUT/plain.err; rc=$?; t1=$(now)
echo "plain           rc=$rc wall=$(secs $t0 $t1)"; [ $rc -eq 0 ] || exit 1
t0=$(now); timeout -k 10 280 rocprofv3 --pmc FETCH_SIZE GRBM_GUI_ACTIVE --output-format csv -d $OUT/all -- python3 $R/bench.py $ARGS > $OUT/all.json 2> $OUT/all.err; rc=$?; t1=$(now)
echo "pmc, every kernel rc=$rc wall=$(secs $t0 $t1)"; [ $rc -eq 0 ] || exit 1      # a leg killed at its limit ends the script: no further GPU step
t0=$(now); timeout -k 10 280 rocprofv3 --pmc FETCH_SIZE GRBM_GUI_ACTIVE --kernel-include-regex 'lec_' --output-format csv -d $OUT/lec -- python3 $R/bench.py $ARGS > $OUT/lec.json 2> $OUT/lec.err; rc=$?; t1=$(now)
echo "pmc, lec_ only   rc=$rc wall=$(secs $t0 $t1)"; [ $rc -eq 0 ] || exit 1
python3 - <<PY
import csv, glob, collections
for leg in ("all", "lec"):
    n = collections.Counter(); disp = set()
    for f in glob.glob("$OUT/%s/*/*counter_collection.csv" % leg):
        for r in csv.DictReader(open(f)):
            disp.add(r["Dispatch_Id"])
            n[r["Kernel_Name"][:70]] += 1
    print(leg, "instrumented dispatches:", len(disp))
    for k, v in n.most_common(6):
        print("    %6d rows  %s" % (v, k))
PY
