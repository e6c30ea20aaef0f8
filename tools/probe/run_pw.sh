for r in 1 2; do
for m in 262144 1048576 4194304 16777216; do echo -n "inline min=$m "; GS_EARLY_ADAM=inline GS_EARLY_ADAM_MIN=$m python bench.py --workload pix2pix --no-cpu-baseline --no-kernel-timing --no-secondary 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"; done
echo -n "pix2pix GS_EARLY_ADAM=0 "; GS_EARLY_ADAM=0 python bench.py --workload pix2pix --no-cpu-baseline --no-kernel-timing --no-secondary 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"
done
