python tools/bench_kernels.py --only v64_fwd
python tools/bench_kernels.py --only v64_fwd --opt hconv2=4
python tools/bench_kernels.py --only v64_dgrad
python tools/bench_kernels.py --only v64_dgrad --opt hconv2=4
for r in 1 2; do for v in 3 4; do echo -n "brats GS_HCONV2=$v "; GS_HCONV2=$v python bench.py --workload brats --no-cpu-baseline --no-kernel-timing --no-secondary 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"; done; done
