# norm_ex backward reduction with four pixels in flight and select-form activations: tests, then same-box library A/B on pix2pix
cd "$GRAFT_REPO_ROOT"
python -m pytest tests/test_ops_gpu.py tests/test_pix2pix_gpu.py tests/test_gradients_gpu.py -q -x -m gpu -k "norm_ex or norm_act or pix2pix or unet or dropout" 2>&1 | grep -E "passed|failed|error" | tail -3
bash tools/ab_lib.sh $PWD/gpurun_libs/lib_prev.so $PWD/gpurun_libs/lib_new.so 3 --workload pix2pix
