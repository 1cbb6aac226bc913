#!/bin/bash
# A/B on ONE box: bench.py --moving --timesteps 512 with the box-packed series and with the whole crop, alternating, N rounds.
# Prints the stage-1 launch time (HIP events, mean of the timed passes) and the pass time of every run.  Usage: tools/ab_moving_layout.sh [rounds]
R=${GRAFT_REPO_ROOT:-$(pwd)}; N=${1:-5}
for i in $(seq 1 $N); do
  for L in packed cube; do
    timeout -k 10 120 python3 $R/bench.py --moving --timesteps 512 --cpu-baseline none --moving-layout $L 2>/dev/null | python3 -c "
import json, sys
d = json.loads(sys.stdin.read())
print('round $i %-6s stage-1 launch %.4f ms  pass %.4f ms  frac %.4f  ok %s' % ('$L', d['roofline']['avg_launch_ms'], d['ms_per_step'], d['roofline']['frac'], d['config']['moving_check']['ok']))"
  done
done
