#!/bin/bash
# the round's profile set (GPU box): bash tools/prof_final.sh <tag>  -> gpurun_out/<tag>_*
tag=$1
cd "$GRAFT_REPO_ROOT"
bash tools/pmc_step.sh gpurun_out/${tag}_pmc > /dev/null 2>&1
python tools/pmc_hbm_json.py gpurun_out/${tag}_pmc/fetch.txt gpurun_out/${tag}_pmc/write.txt --images 16 > gpurun_out/${tag}_trunk_hbm.json
bash tools/prof_round.sh ${tag}
python tools/conv_table.py > gpurun_out/${tag}_conv_table.txt 2> /dev/null
for wl in pix2pix brats cut; do
  bash tools/step_census.sh ${tag}_${wl} $wl > /dev/null 2>&1
done
python bench.py > gpurun_out/${tag}_bench.json 2> gpurun_out/${tag}_bench.err
tail -c 600 gpurun_out/${tag}_bench.json
