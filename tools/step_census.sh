#!/bin/bash
# every launch of a steady-state headline step, by (kernel, grid): bash tools/step_census.sh <tag> [workload]   -> gpurun_out/<tag>_step_census.txt
tag=$1; wl=${2:-cyclegan}
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
rm -rf /tmp/pc
GS_SIDE_STREAM=0 GS_STEP_GRAPH=0 timeout 900 rocprofv3 --kernel-trace -d /tmp/pc -o p -- python3 bench.py --workload $wl --steps 4 --warmup 2 --no-secondary --no-cpu-baseline --no-kernel-timing > /dev/null 2> /tmp/pc.log
python tools/prof_by_grid.py "$(find /tmp/pc -name '*.db' | head -1)" --of 6 --steps 3 --top 400 > gpurun_out/${tag}_step_census.txt
head -3 gpurun_out/${tag}_step_census.txt
