for k in 0 40; do
echo "== gconv_smallk=$k"
python tools/bench_kernels.py --batch 16 --opt gconv_smallk=$k --only d1_fwd
python tools/bench_kernels.py --batch 16 --opt gconv_smallk=$k --only d2_fwd
python tools/bench_kernels.py --batch 16 --opt gconv_smallk=$k --only u1_dgrad
python tools/bench_kernels.py --batch 16 --opt gconv_smallk=$k --only u2_dgrad
python tools/bench_kernels.py --batch 32 --opt gconv_smallk=$k --only dc2_fwd
python tools/bench_kernels.py --batch 32 --opt gconv_smallk=$k --only dc3_fwd
done
for k in 0 100; do
echo "== gconv_smallk=$k"
python tools/bench_kernels.py --batch 32 --opt gconv_smallk=$k --only dc4_fwd
python tools/bench_kernels.py --batch 32 --opt gconv_smallk=$k --only dc4_dgrad
done
