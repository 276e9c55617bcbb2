for v in A static nogather static_nogather A static_nogather; do
  if [ $v = A ]; then unset REGNET_HIP_LIB; else export REGNET_HIP_LIB=$PWD/gpurun_variant_$v.so; fi
  echo "== $v"; REPS=30 ROWS=5 python scripts/features_alone.py 8 2>&1 | grep -v amdgpu.ids
done
