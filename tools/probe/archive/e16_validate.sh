# round validation: full GPU suite, the default bench (secondary workloads + cpu_baseline), profiled bench + step by grid
cd "$GRAFT_REPO_ROOT"
python -m pytest tests -q -x -m gpu 2>&1 | grep -E "passed|failed|error" | tail -3 > gpurun_out/r05_gpu_tests_v3.txt
cat gpurun_out/r05_gpu_tests_v3.txt
( time python bench.py ) > gpurun_out/r05_bench_v3.json 2> gpurun_out/r05_bench_v3.err
tail -3 gpurun_out/r05_bench_v3.err
cat gpurun_out/r05_bench_v3.json
bash tools/prof_round.sh r05v3
