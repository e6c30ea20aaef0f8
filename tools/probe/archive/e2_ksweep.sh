# fixed cost vs per-K-step cost of the im2col tiles: the residual-conv shape with 64 / 128 / 256 input channels (9 / 18 / 36 K-steps)
for k in 0 100; do
echo "== hconv_wide=0 gconv_smallk=$k"
for c in rbk64_fwd rbk128_fwd rb_fwd; do
python tools/bench_kernels.py --batch 16 --opt hconv_wide=0 --opt gconv_smallk=$k --only $c
done
done
echo "== step A/B gconv_smallk"
for r in 1 2; do for v in 0 18; do echo -n "GS_GCONV_SMALLK=$v "; GS_GCONV_SMALLK=$v python bench.py --no-cpu-baseline --no-kernel-timing --no-secondary 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"; done; done
