#!/bin/bash
# PMC pass over one tools/bench_kernels.py case: bash tools/pmc_bench_kernel.sh <out.txt> <kernel-name filter> <case filter> [bench_kernels args]
# (counters collected with --kernel-trace only, as the pool requires)
set -u
out=$1; filt=$2; only=$3; shift 3
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
for ctr in "SQ_BUSY_CU_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU_MFMA_MOPS_BF16" "GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_INSTS_SALU"; do
  rm -rf /tmp/pmck
  timeout 600 rocprofv3 --pmc $ctr --kernel-trace -d /tmp/pmck -o p -- python3 tools/bench_kernels.py --only "$only" --iters 5 "$@" > /tmp/pmck.log 2>&1
  db=$(find /tmp/pmck -name "*.db" | head -1)
  python tools/pmc_summary.py "$db" "$filt" >> "$out"
done
