# weight-major grid order of weight-heavy split-K launches: same-box A/B on pix2pix (GS_SPLITK_WMAJOR=1 / 0), then the full GPU suite
cd "$GRAFT_REPO_ROOT"
for r in 1 2 3; do for v in 1 0; do
  echo -n "pix2pix GS_SPLITK_WMAJOR=$v "
  GS_SPLITK_WMAJOR=$v python bench.py --workload pix2pix --no-cpu-baseline --no-kernel-timing --no-secondary 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"
done; done
python tools/conv_table.py --workload pix2pix 2>&1 | grep -v amdgpu.ids > gpurun_out/ct_p2p_v1.txt
head -30 gpurun_out/ct_p2p_v1.txt
python -m pytest tests -q -x -m gpu 2>&1 | grep -E "passed|failed|error" | tail -3 > gpurun_out/r05_gpu_tests_v2.txt
cat gpurun_out/r05_gpu_tests_v2.txt
