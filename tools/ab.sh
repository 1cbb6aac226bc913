#!/bin/bash
# GPU box: A/B timing of library variants.  Usage: tools/ab.sh "<bench args>" lib1.so lib2.so ...   ("-" = the shipped library)
ARGS=$1; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
for round in 1 2; do
for L in "$@"; do
  if [ "$L" = "-" ]; then unset LEC_LIB; else export LEC_LIB=$R/$L; fi
  python3 $R/bench.py --cpu-baseline none --steps 10 --warmup 3 $ARGS 2>/dev/null | python3 -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('%-34s %-28s value %9.1f  launch ms %8.3f  frac %.4f' % ('$L', '$ARGS', d['value'], d['roofline']['avg_launch_ms'], d['roofline']['frac']))"
done; done
