#!/bin/bash
# Runs on the GPU box (via gpurun): rocprofv3 kernel-trace stats and separate PMC passes of bench.py.
# Usage: tools/profile_gpu.sh <tag> [extra bench args]
set -e
TAG=${1:-r01}; shift || true
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $REPO/bench.py --timesteps 64 --steps 5 --warmup 1 --no-cpu-baseline "$@" > $OUT/bench_stats.json 2> $OUT/stats.err
echo "stats done"
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 $REPO/bench.py --timesteps 64 --steps 2 --warmup 1 --no-cpu-baseline "$@" > $OUT/bench_pmc_fetch.json 2> $OUT/pmc_fetch.err
echo "fetch done"
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 $REPO/bench.py --timesteps 64 --steps 2 --warmup 1 --no-cpu-baseline "$@" > $OUT/bench_pmc_write.json 2> $OUT/pmc_write.err
echo "write done"
find $OUT -name '*.csv' | head -20
