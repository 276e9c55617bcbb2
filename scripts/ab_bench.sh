#!/bin/bash
# Same-box A/B of a package switch through bench.py (alternating runs): bash scripts/ab_bench.sh <tag> <steps> "<switch A args>" "<switch B args>" [rounds]
TAG=$1; STEPS=$2; A=$3; B=$4; R=${5:-2}
mkdir -p gpurun_out/$TAG
for i in $(seq 1 $R); do
  for v in A B; do
    if [ $v = A ]; then X="$A"; else X="$B"; fi
    python bench.py --steps $STEPS --warmup 5 --cpu-scenes 0 --latency-runs 0 --train-steps 0 --no-lookahead-steps 0 --real-density-steps 0 --split-products-steps 0 $X 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=j['roofline']
print('$v [$X] %d steps: %.3f ms/step %.1f scenes/s  %s' % (j['steps'], j['ms_per_step'], j['value'], r['families_ms_per_step']))"
  done
done | tee gpurun_out/$TAG/ab.txt
