#!/bin/bash
# A/B of kernel knobs via environment variables; prints value + kernel ms for each combination.
# Usage: tools/ab_env.sh "<bench args>" VAR=a,b VAR2=c,d ...   (cartesian product, small)
BARGS="$1"; shift
run() {
  python bench.py $BARGS --no-cpu-baseline 2>/dev/null | python -c "
import json,sys,os
d=json.loads(sys.stdin.read()); r=d['roofline']
print(os.environ.get('AB_TAG',''), 'ts/s=%.1f' % d['value'], 'kernel_ms=%.3f' % r['avg_launch_ms'], 'frac=%.3f' % r['frac'])"
}
combos=("")
for spec in "$@"; do
  var=${spec%%=*}; vals=${spec#*=}
  new=()
  for c in "${combos[@]}"; do
    IFS=',' read -ra vs <<< "$vals"
    for v in "${vs[@]}"; do new+=("$c $var=$v"); done
  done
  combos=("${new[@]}")
done
for c in "${combos[@]}"; do
  env $c AB_TAG="$c" bash -c "$(declare -f run); BARGS='$BARGS'; run"
done
