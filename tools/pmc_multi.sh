#!/bin/bash
# Several PMC passes (one counter set each; no tracing) over the same bench command.
# Usage: tools/pmc_multi.sh <tag> [bench args]     counter sets: tools/pmc_sets.txt (one set per line)
TAG=$1; shift
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
i=0
while read -r SET; do
  [ -z "$SET" ] && continue
  i=$((i+1))
  echo "== $TAG pass $i: $SET"
  $REPO/tools/pmc_gpu.sh ${TAG}_$i "$SET" "$@" | tail -n +2 || exit 1
done < $REPO/tools/pmc_sets.txt
