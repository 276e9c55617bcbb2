#!/bin/bash
# Same-box A/B of two builds of the library: bash scripts/ablate/ab_lib.sh <other libregnet_hip.so> <steps> [rounds]
# (A = the tree's library, B = the other file copied over it for the run; the tree's library is restored afterwards)
OTHER=$1; STEPS=${2:-60}; R=${3:-3}
LIB=regnet_for_3d_grasping_amd/csrc/libregnet_hip.so
cp $LIB /tmp/lib_A.so
for i in $(seq 1 $R); do
  for v in A B; do
    if [ $v = A ]; then cp /tmp/lib_A.so $LIB; else cp $OTHER $LIB; fi
    python bench.py --steps $STEPS --warmup 5 --cpu-scenes 0 --latency-runs 0 --train-steps 0 --no-lookahead-steps 0 --real-density-steps 0 --split-products-steps 0 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=j['roofline']
print('$v %d steps: %.3f ms/step %.1f scenes/s  %s' % (j['steps'], j['ms_per_step'], j['value'], r['families_ms_per_step']))"
  done
done
cp /tmp/lib_A.so $LIB
