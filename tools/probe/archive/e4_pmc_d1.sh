# what bounds the stride-2 forward (d128 at a twin batch) on the im2col kernels: HBM traffic, L2 hit rate, LDS / TA activity
o=gpurun_out/r05_pmc_d1_fwd.txt; rm -f $o
for p in 0 1000; do
echo "== gconv_persist=$p" >> $o
bash tools/pmc_bench_kernels.sh $o conv_kernel FETCH_SIZE -- --batch 16 --opt gconv_persist=$p --only d1_fwd --iters 10
bash tools/pmc_bench_kernels.sh $o conv_kernel WRITE_SIZE -- --batch 16 --opt gconv_persist=$p --only d1_fwd --iters 10
bash tools/pmc_bench_kernels.sh $o conv_kernel TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum -- --batch 16 --opt gconv_persist=$p --only d1_fwd --iters 10
bash tools/pmc_bench_kernels.sh $o conv_kernel SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT -- --batch 16 --opt gconv_persist=$p --only d1_fwd --iters 10
bash tools/pmc_bench_kernels.sh $o conv_kernel TA_BUSY_avr TCP_PENDING_STALL_CYCLES_sum TA_TA_BUSY_sum TCP_TCC_READ_REQ_sum -- --batch 16 --opt gconv_persist=$p --only d1_fwd --iters 10
done
cat $o
