python -m pytest tests/test_ops_gpu.py -q -x -k "fused_adam" > /tmp/t.log 2>&1; grep -E "passed|failed|^E |Error" /tmp/t.log | tail -8
python -m pytest tests/test_pix2pix_gpu.py -q -x > /tmp/t2.log 2>&1; grep -E "passed|failed|^E |Error|assert" /tmp/t2.log | tail -8
for r in 1 2 3; do for v in 0 1; do echo -n "pix2pix GS_WGRAD_ADAM_TR=$v "; GS_WGRAD_ADAM_TR=$v python bench.py --workload pix2pix --no-cpu-baseline --no-kernel-timing --no-secondary 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"; done; done
