# split-K launches on a 4-stage ring: parity tests, then same-box A/B on pix2pix (GS_SPLITK_RING=1 / 0)
cd "$GRAFT_REPO_ROOT"
python -m pytest tests/test_ops_gpu.py -q -x -m gpu -k "split or conv_forward or dgrad or merged_parity" 2>&1 | tail -4
for r in 1 2 3; do for v in 1 0; do
  echo -n "pix2pix GS_SPLITK_RING=$v "
  GS_SPLITK_RING=$v python bench.py --workload pix2pix --no-cpu-baseline --no-kernel-timing --no-secondary 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"
done; done
