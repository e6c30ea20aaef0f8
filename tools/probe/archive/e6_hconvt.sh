for k in 0 1; do
echo "== hconvt_persist=$k"
python tools/bench_kernels.py --batch 16 --opt hconvt_persist=$k --only u1_fwd
python tools/bench_kernels.py --batch 16 --opt hconvt_persist=$k --only u2_fwd
python tools/bench_kernels.py --batch 16 --opt hconvt_persist=$k --only d1_dgrad
python tools/bench_kernels.py --batch 16 --opt hconvt_persist=$k --only d2_dgrad
python tools/bench_kernels.py --batch 32 --opt hconvt_persist=$k --only dc2_dgrad
python tools/bench_kernels.py --batch 32 --opt hconvt_persist=$k --only dc3_dgrad
done
