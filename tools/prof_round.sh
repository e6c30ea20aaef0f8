#!/bin/bash
# Round profile of the headline bench on the GPU box: bash tools/prof_round.sh <tag>   (writes gpurun_out/<tag>_*)
#  (1) rocprofv3 --kernel-trace --stats of the bench command (graph replays, three streams) -> kernel stats CSV
#  (2) one-stream launch-by-launch trace of 6 steps -> per (kernel, grid) table of the steady state
tag=$1
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
rm -rf /tmp/pr1 /tmp/pr2
timeout 900 rocprofv3 --kernel-trace --stats -d /tmp/pr1 -o p -f csv -- python3 bench.py --no-secondary --no-cpu-baseline > gpurun_out/${tag}_bench_profiled.json 2> /tmp/pr1.log
cp "$(find /tmp/pr1 -name '*kernel_stats.csv' | head -1)" gpurun_out/${tag}_bench_kernel_stats.csv
GS_SIDE_STREAM=0 GS_STEP_GRAPH=0 timeout 900 rocprofv3 --kernel-trace -d /tmp/pr2 -o p -- python3 bench.py --steps 4 --warmup 2 --no-secondary --no-cpu-baseline --no-kernel-timing > /dev/null 2> /tmp/pr2.log
python tools/prof_by_grid.py "$(find /tmp/pr2 -name '*.db' | head -1)" --of 6 --steps 3 --top 70 > gpurun_out/${tag}_step_by_grid.txt
head -5 gpurun_out/${tag}_step_by_grid.txt
