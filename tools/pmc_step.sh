#!/bin/bash
# PMC passes over the headline training step, launch by launch on one stream (counters need serialised kernels):
#   pass 1 matrix-pipe occupancy, pass 2 FETCH_SIZE, pass 3 WRITE_SIZE (separate passes: TCC slots, see MI355X_MICROARCH.md)
# usage (GPU box): bash tools/pmc_step.sh gpurun_out/pmc
set -u
out=${1:-gpurun_out/pmc}
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
export GS_SIDE_STREAM=0 GS_STEP_GRAPH=0
run() {  # name, counters...
  local name=$1; shift
  rm -rf /tmp/pmc_$name
  timeout 600 rocprofv3 --pmc "$@" --kernel-trace -d /tmp/pmc_$name -o p -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-timing --no-secondary > "$out/$name.log" 2>&1
  local db=$(find /tmp/pmc_$name -name "*.db" | head -1)
  for k in "gconv_kernel<288" hwgrad_wide hconvw_kernel hconvt_kernel hstrip_kernel hwgrad_ft inorm_bwd_apply_cg inorm_stats_act; do
    python tools/pmc_summary.py "$db" "$k"
  done > "$out/$name.txt" 2>> "$out/$name.log"
}
if [ "${2:-}" = "valu" ]; then
  run valu SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_BUSY_CU_CYCLES SQ_INSTS_VALU
  exit 0
fi
run mfma SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE
run fetch FETCH_SIZE
run write WRITE_SIZE
