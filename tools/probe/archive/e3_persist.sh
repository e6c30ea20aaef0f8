for k in 0 1000; do
echo "== gconv_persist=$k"
python tools/bench_kernels.py --batch 16 --opt gconv_persist=$k --only d1_fwd
python tools/bench_kernels.py --batch 16 --opt gconv_persist=$k --only d2_fwd
python tools/bench_kernels.py --batch 16 --opt gconv_persist=$k --only u1_dgrad
python tools/bench_kernels.py --batch 16 --opt gconv_persist=$k --only u2_dgrad
python tools/bench_kernels.py --batch 32 --opt gconv_persist=$k --only dc2_fwd
python tools/bench_kernels.py --batch 32 --opt gconv_persist=$k --only dc3_fwd
python tools/bench_kernels.py --batch 32 --opt gconv_persist=$k --only dc4_fwd
python tools/bench_kernels.py --batch 32 --opt gconv_persist=$k --only dc4_dgrad
python tools/bench_kernels.py --batch 16 --opt hconv_wide=0 --opt gconv_persist=$k --only rbk64_fwd
python tools/bench_kernels.py --batch 16 --opt hconv_wide=0 --opt gconv_persist=$k --only rb_fwd
done
