#!/bin/bash
# kernel-trace statistics of a tools/bench_kernels.py run: bash tools/kstats.sh <out.txt> <bench_kernels args>
out=$1; shift
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
rm -rf /tmp/kst
timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/kst -o p -f csv -- python3 tools/bench_kernels.py "$@" > /tmp/kst.log 2>&1
f=$(find /tmp/kst -name "*kernel_stats.csv" | head -1)
python3 - "$f" >> "$out" <<'P'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:12]:
    print(f"{r['Name'][:90]:90s} calls {r['Calls']:>6s} avg {float(r['AverageNs'])/1e3:9.1f} us  total {float(r['TotalDurationNs'])/1e6:8.2f} ms")
P
