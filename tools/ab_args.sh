#!/bin/bash
# GPU box: A/B timing of bench.py argument sets on the shipped library, alternating.  Usage: tools/ab_args.sh <rounds> "<args 1>" "<args 2>" ...
ROUNDS=$1; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
for round in $(seq 1 $ROUNDS); do
for A in "$@"; do
  python3 $R/bench.py --cpu-baseline none --steps 10 --warmup 3 $A 2>/dev/null | python3 -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1])
c=d['config']
print('%-58s value %9.1f  pass ms %7.3f  stage-1 launch ms %7.3f  frac %.4f  %s' % ('$A', d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'], d['roofline']['frac'], ('ok' if c.get('moving_check', {}).get('ok', True) else 'CHECK FAILED') + (' producer %.3f ms' % c['producer_ms']['total'] if 'producer_ms' in c else '')))"
done; done
