#!/bin/bash
# Same-box A/B of several builds of the library: bash scripts/ablate/ab_libs.sh <steps> <rounds> <lib1.so> <lib2.so> ...
# ("tree" = the tree's own library; each other file is copied over it for its runs; the tree's library is restored afterwards)
STEPS=$1; R=$2; shift 2
LIB=regnet_for_3d_grasping_amd/csrc/libregnet_hip.so
cp $LIB /tmp/lib_tree.so
for i in $(seq 1 $R); do
  for v in tree "$@"; do
    if [ $v = tree ]; then cp /tmp/lib_tree.so $LIB; else cp $v $LIB; fi
    python bench.py --steps $STEPS --warmup 5 --cpu-scenes 0 --latency-runs 0 --train-steps 0 --no-lookahead-steps 0 --real-density-steps 0 --split-products-steps 0 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=j['roofline']
print('%-32s %d steps: %.3f ms/step %.1f scenes/s  sa_chain %.3f ms' % ('$v', j['steps'], j['ms_per_step'], j['value'], r['families_ms_per_step'].get('sa_chain_kernel', 0)))"
  done
done
cp /tmp/lib_tree.so $LIB
