#!/bin/bash
# GPU box: wall clock from the script's last line to the process being gone, for the three ways to end
R=${GRAFT_REPO_ROOT:-$(pwd)}
for mode in leave del hard; do
  out=$(python3 $R/tools/probes/exit_cost.py ${1:-8} $mode); end=$(python3 -c "import time; print(time.time())")
  pid=$(echo "$out" | tail -1); last=$(cat /tmp/exit_cost_$pid)
  echo "$mode: $(echo "$out" | head -n -1 | tr '\n' ';') exit took $(python3 -c "print(round($end - $last, 3))") s"
done
