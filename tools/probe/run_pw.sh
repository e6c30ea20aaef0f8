python -m pytest tests/test_pix2pix_gpu.py tests/test_recipe_gradients_gpu.py -q -x -k "pix2pix or p2p or chunked or dropout" > /tmp/t2.log 2>&1; grep -E "passed|failed|^E |Error|assert" /tmp/t2.log | tail -8
for r in 1 2 3; do echo -n "pix2pix "; python bench.py --workload pix2pix --no-cpu-baseline --no-kernel-timing --no-secondary 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"; done
bash tools/step_census.sh r06_pix2pix_v6 pix2pix > /dev/null 2>&1
grep -n "at6native\|rocclr" gpurun_out/r06_pix2pix_v6_step_census.txt | cut -c1-150
head -3 gpurun_out/r06_pix2pix_v6_step_census.txt | cut -c1-120
