# wgrad row epilogue: parity tests, then same-box A/B on pix2pix, brats and the headline (GS_WGRAD_ROWS=1 / 0)
cd "$GRAFT_REPO_ROOT"
python -m pytest tests/test_ops_gpu.py -q -x -m gpu -k "wgrad or weight_gradient" 2>&1 | tail -5
for wl in pix2pix brats cyclegan; do
  for r in 1 2; do for v in 1 0; do
    echo -n "$wl GS_WGRAD_ROWS=$v "
    GS_WGRAD_ROWS=$v python bench.py --workload $wl --no-cpu-baseline --no-kernel-timing --no-secondary 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"
  done; done
done
