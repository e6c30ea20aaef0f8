# single-image parameter gradients folded into the finalize passes: norm tests, then same-box library A/B on brats and pix2pix
cd "$GRAFT_REPO_ROOT"
python -m pytest tests/test_ops_gpu.py tests/test_volumes_gpu.py tests/test_pix2pix_gpu.py tests/test_gradients_gpu.py -q -x -m gpu -k "norm or pnorm or prelu or vnet or pix2pix or gradient" 2>&1 | grep -E "passed|failed|error" | tail -3
bash tools/ab_lib.sh $PWD/abl/lib_prev.so $PWD/abl/lib_new.so 2 --workload brats
bash tools/ab_lib.sh $PWD/abl/lib_prev.so $PWD/abl/lib_new.so 2 --workload pix2pix
