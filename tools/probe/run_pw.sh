python -m pytest tests/test_ops_gpu.py -q -k "pointwise" -x 2>&1 | tail -15
for r in 1 2; do for v in 0 8; do echo -n "brats GS_PWISE=$v "; GS_PWISE=$v python bench.py --workload brats --no-cpu-baseline --no-kernel-timing --no-secondary 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"; done; done
