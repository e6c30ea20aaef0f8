python tools/bench_kernels.py --only v64s_fwd
python tools/bench_kernels.py --only v64s_fwd --opt hconv2=5
python tools/bench_kernels.py --only v64s_dgrad
python tools/bench_kernels.py --only v64s_dgrad --opt hconv2=5
for r in 1 2; do for v in 4 5; do echo -n "brats GS_HCONV2=$v "; GS_HCONV2=$v python bench.py --workload brats --no-cpu-baseline --no-kernel-timing --no-secondary 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"; done; done
