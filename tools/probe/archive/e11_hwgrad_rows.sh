# wgrad_rows: 0 = accumulator-layout epilogues, 2 = rows in wgrad_kernel only, 1 = rows in wgrad_kernel and hwgrad_wide
cd "$GRAFT_REPO_ROOT"
for v in 0 2 1; do echo "wgrad_rows=$v (twin-launch geometry, batch 16)"; python tools/bench_kernels.py --only _wgrad --batch 16 --opt wgrad_rows=$v 2>&1 | grep wgrad; done
for r in 1 2 3; do for v in 0 2 1; do
  echo -n "cyclegan GS_WGRAD_ROWS=$v "
  GS_WGRAD_ROWS=$v python bench.py --no-cpu-baseline --no-kernel-timing --no-secondary 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"
done; done
