#!/bin/bash
# GPU box: (1) the round-2 bench (git archive 2ae2ab4 into tools/probes/old_r02, built there, one_pass instrumented: see profiles/r03_notes.md section 1) on the configuration of gpurun_out/g4m.json;
# (2) the ingest copy pipeline's timeline in its variants.
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
cd $R/tools/probes/old_r02 && LEC_DIST_BACKEND=gloo timeout -k 10 200 python3 bench.py --gpus 4 --moving --timesteps-global 1024 --cpu-baseline none --steps 3 --warmup 1 > $O/r03_old_g4m.json 2> $O/r03_old_g4m.err; grep "OLD rank" $O/r03_old_g4m.err | tail -8; grep -o '"ms_per_step": [0-9.]*' $O/r03_old_g4m.json
cd $R
for m in "staged 16 2" "staged 16 3" "staged 8 2" "staged 32 2" "staged 4 2" "pageable 16 2" "registered 16 2"; do
  set -- $m
  echo "== mode $1 threads $2 slots $3"
  timeout -k 10 200 python3 tools/ingest_timeline.py --src i16 --timesteps 16 --chunk 4 --mode $1 --threads $2 --slots $3 2>&1 | tail -7
done
echo "== f64 staged"; timeout -k 10 200 python3 tools/ingest_timeline.py --src f64 --timesteps 8 --chunk 2 --mode staged --threads 16 2>&1 | tail -6
nproc; lscpu | grep -i "numa\|model name\|socket" | head; numactl -H 2>/dev/null | head -12
