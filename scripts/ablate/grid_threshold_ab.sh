#!/bin/bash
# Same-box A/B of the grid thresholds of the level-2 searches (pn2_ext.GRID_MIN_POINTS_BALL / GRID_MIN_POINTS), alternating runs
R=${1:-4}
for i in $(seq 1 $R); do
  for X in "" "--set pn2_ext.GRID_MIN_POINTS_BALL=4096" "--set pn2_ext.GRID_MIN_POINTS=1024" "--set pn2_ext.GRID_MIN_POINTS_BALL=4096 --set pn2_ext.GRID_MIN_POINTS=1024"; do
    python bench.py --steps 20 --warmup 5 --cpu-scenes 0 --latency-runs 0 --train-steps 0 --no-lookahead-steps 0 --real-density-steps 0 --split-products-steps 0 $X 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('%-90s %.3f ms/step %.1f scenes/s' % ('[$X]', j['ms_per_step'], j['value']))"
  done
done
