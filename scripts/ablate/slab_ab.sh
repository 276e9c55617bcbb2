cd /root/repo
echo "== error budget, product (slab 8 k-tiles on the 64x128 tile)"; timeout 300 python scripts/error_budget.py 2>&1 | grep -E "^(default|reference)"
echo "== slab 16"; REGNET_HIP_LIB=/root/repo/gpurun_variant_slab16.so timeout 300 python scripts/error_budget.py 2>&1 | grep -E "^default"
echo "== no slab"; REGNET_HIP_LIB=/root/repo/gpurun_variant_noslab.so timeout 300 python scripts/error_budget.py 2>&1 | grep -E "^default"
echo "== force 64x128 tile everywhere, slab 8"; REGNET_G2_TILE=2 timeout 300 python scripts/error_budget.py 2>&1 | grep -E "^default"
for v in product noslab; do
  echo "== layers $v (idle chip)"; if [ $v = noslab ]; then export REGNET_HIP_LIB=/root/repo/gpurun_variant_noslab.so; else unset REGNET_HIP_LIB; fi
  SHAPES=2,3,4,5,6,7,8,9,10 timeout 300 python scripts/bench_gemm2_layers.py 2>&1 | grep TFLOP
  echo "== layers $v tile 64x128 forced"; REGNET_G2_TILE=2 SHAPES=2,4 timeout 300 python scripts/bench_gemm2_layers.py 2>&1 | grep TFLOP
done
