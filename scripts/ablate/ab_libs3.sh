#!/bin/bash
# Same-box A/B/C of library builds: bash scripts/ablate/ab_libs3.sh <steps> <rounds> <lib B> [<lib C> ...]   (A = the tree's library)
STEPS=$1; R=$2; shift 2
LIB=regnet_for_3d_grasping_amd/csrc/libregnet_hip.so
cp $LIB /tmp/lib_A.so
for i in $(seq 1 $R); do
  for v in A "$@"; do
    if [ $v = A ]; then cp /tmp/lib_A.so $LIB; else cp $v $LIB; fi
    python bench.py --steps $STEPS --warmup 5 --cpu-scenes 0 --latency-runs 0 --train-steps 0 --no-lookahead-steps 0 --real-density-steps 0 --split-products-steps 0 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=j['roofline']
print('%-28s %d steps: %.3f ms/step %.1f scenes/s  %s' % ('$v', j['steps'], j['ms_per_step'], j['value'], r['families_ms_per_step']))"
  done
done
cp /tmp/lib_A.so $LIB
