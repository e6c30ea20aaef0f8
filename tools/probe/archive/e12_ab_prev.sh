# same-box A/B: library of the previous commit vs this one (headline + pix2pix), prev has no wgrad_rows option
cd "$GRAFT_REPO_ROOT"
bash tools/ab_lib.sh $PWD/abl/lib_prev.so $PWD/abl/lib_new.so 3
bash tools/ab_lib.sh $PWD/abl/lib_prev.so $PWD/abl/lib_new.so 2 --workload pix2pix
