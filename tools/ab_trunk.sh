#!/bin/bash
# same-box A/B of tools/bench_trunk.py between two builds of the library: bash tools/ab_trunk.sh <lib A> <lib B> [rounds] [grep pattern]
a=$1; b=$2; rounds=${3:-2}; pat=${4:-"forward|dgrad \(ring"}
for r in $(seq $rounds); do
  for v in $a $b; do
    echo "== $v"
    GANSLATE_HIP_LIB=$v python tools/bench_trunk.py 2>/dev/null | grep -E "$pat" | cut -c1-140
  done
done
