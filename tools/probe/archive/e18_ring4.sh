# 4-stage ring for small-grid long-K im2col launches: op tests, then same-box A/B on brats, cyclegan, cut (GS_GCONV_RING4=16 / 0)
cd "$GRAFT_REPO_ROOT"
python -m pytest tests/test_ops_gpu.py -q -x -m gpu -k "conv_forward or dgrad or merged_parity or split" 2>&1 | grep -E "passed|failed|error" | tail -3
for wl in brats cyclegan; do
  for r in 1 2; do for v in 16 0; do
    echo -n "$wl GS_GCONV_RING4=$v "
    GS_GCONV_RING4=$v python bench.py --workload $wl --no-cpu-baseline --no-kernel-timing --no-secondary 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"
  done; done
done
