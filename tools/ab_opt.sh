#!/bin/bash
# same-box A/B of tools/bench_trunk.py over one GS_* switch: bash tools/ab_opt.sh GS_HCONVW_PREFETCH 0 1 [rounds]
var=$1; a=$2; b=$3; rounds=${4:-2}
for r in $(seq $rounds); do
  for v in $a $b; do
    echo "$var=$v"
    env $var=$v python tools/bench_trunk.py 2>/dev/null | grep -E "forward|dgrad" | cut -c1-150
  done
done
