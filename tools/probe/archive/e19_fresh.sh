# dw_fresh hint: op tests, pix2pix tests, then same-box A/B against the previous commit's behaviour (GS_WGRAD_FRESH=0 disables the hint host-side)
cd "$GRAFT_REPO_ROOT"
python -m pytest tests/test_ops_gpu.py tests/test_pix2pix_gpu.py -q -x -m gpu -k "fresh or weight_gradient or pix2pix or wgrad" 2>&1 | grep -E "passed|failed|error" | tail -3
for r in 1 2 3; do for v in 1 0; do
  echo -n "pix2pix GS_WGRAD_FRESH=$v "
  GS_WGRAD_FRESH=$v python bench.py --workload pix2pix --no-cpu-baseline --no-kernel-timing --no-secondary 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"
done; done
