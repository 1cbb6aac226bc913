#!/bin/bash
# GPU box: alternating A/B of library builds on one bench.py argument set.  Usage: tools/probes/ab_libs.sh <rounds> "<bench args>" <lib or -> ...
ROUNDS=$1; ARGS=$2; shift; shift
for r in $(seq 1 $ROUNDS); do for L in "$@"; do
  LL=$L; [ "$L" = "-" ] && LL=""
  LEC_LIB=$LL python3 bench.py --cpu-baseline none --steps 10 --warmup 3 $ARGS 2>/dev/null | python3 -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('%-34s %-32s value %9.1f  launch ms %7.3f  frac %.4f' % ('$L', '$ARGS', d['value'], d['roofline']['avg_launch_ms'], d['roofline']['frac']))"
done; done
