# Stand-alone A/B of the plain-layer tile variants on the step's shapes (scripts/bench_gemm2_layers.py): REGNET_G2_TILE forces a
# gemm2_kernel tile (csrc/mlp.hip: launch_gemm2), STREAM=1 takes the persistent launch where it is supported.
cd ${GRAFT_REPO_ROOT:-/root/repo}
for t in -1 7 8; do echo "== REGNET_G2_TILE=$t"; REGNET_G2_TILE=$t SHAPES=2,3,4,5,6,7,8,9,10 REPS=100 timeout 300 python scripts/bench_gemm2_layers.py 2>&1 | tail -9; done
echo "== STREAM=1"; STREAM=1 SHAPES=2,3,4,5,6,7,8,9,10 REPS=100 timeout 300 python scripts/bench_gemm2_layers.py 2>&1 | tail -9
