#!/bin/bash
# one PMC pass over tools/bench_kernels.py: bash tools/pmc_bench_kernels.sh <out.txt> <kernel filter> <counters...> -- <bench_kernels args>
set -u
out=$1; filt=$2; shift 2
ctr=()
while [ "$1" != "--" ]; do ctr+=("$1"); shift; done
shift
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
rm -rf /tmp/pmck
timeout 600 rocprofv3 --pmc "${ctr[@]}" --kernel-trace -d /tmp/pmck -o p -- python3 tools/bench_kernels.py "$@" > /tmp/pmck.log 2>&1
db=$(find /tmp/pmck -name "*.db" | head -1)
python tools/pmc_summary.py "$db" "$filt" >> "$out"
