# gemm2 tile A/B on the step's plain layers: default table vs the 128 x 128 x 8-wave slab tile, idle chip and beside 8 held CUs
cd ${GRAFT_REPO_ROOT:-/root/repo}
for side in 0 8; do
  for tile in default 2 3 1; do
    echo "== SIDE_BLOCKS=$side tile=$tile"
    if [ $tile = default ]; then unset REGNET_G2_TILE; else export REGNET_G2_TILE=$tile; fi
    SIDE_BLOCKS=$side SHAPES=2,3,4,5,6,7,8,9,10 timeout 300 python scripts/bench_gemm2_layers.py 2>&1 | grep TFLOP
  done
done
