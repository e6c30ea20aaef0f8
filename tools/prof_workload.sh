#!/bin/bash
# one-stream launch-by-launch profile of a secondary workload on the GPU box: bash tools/prof_workload.sh brats <tag>
wl=$1; tag=$2
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
rm -rf /tmp/pb
GS_SIDE_STREAM=0 GS_STEP_GRAPH=0 timeout 900 rocprofv3 --kernel-trace -d /tmp/pb -o p -- python3 bench.py --workload $wl --steps 4 --warmup 2 --no-secondary --no-cpu-baseline --no-kernel-timing > /dev/null 2> /tmp/pb.log
python tools/prof_by_grid.py "$(find /tmp/pb -name '*.db' | head -1)" --of 6 --steps 3 --top 60 > gpurun_out/${tag}_${wl}_by_grid.txt
head -12 gpurun_out/${tag}_${wl}_by_grid.txt
