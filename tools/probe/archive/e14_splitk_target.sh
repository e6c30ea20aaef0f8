# pix2pix: split-K workgroup target (256 = one per CU; the 128 x 128 tile's 64 KB of LDS lets two share a CU)
cd "$GRAFT_REPO_ROOT"
for r in 1 2; do for v in 256 128 192; do
  echo -n "pix2pix GS_SPLITK_TARGET=$v "
  GS_SPLITK_TARGET=$v python bench.py --workload pix2pix --no-cpu-baseline --no-kernel-timing --no-secondary 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"
done; done
