#!/bin/bash
# same-box A/B of the headline bench between two builds of the library: bash tools/ab_lib.sh <lib A> <lib B> [rounds] [bench args]
a=$1; b=$2; rounds=${3:-3}; shift 3
for r in $(seq $rounds); do
  for v in $a $b; do
    echo -n "$v "
    env GANSLATE_HIP_LIB=$v python bench.py --no-cpu-baseline --no-kernel-timing --no-secondary "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"
  done
done
