#!/bin/bash
# Small-batch throughput of the forward pipeline: hipGraph replays on/off x host/device draws (bash scripts/small_batch_ab.sh [tag])
TAG=${1:-r03w}
mkdir -p gpurun_out/$TAG
for B in 1 2 4 8; do
 for g in off on; do
  for v in 0 1; do
   python bench.py --batch $B --steps 60 --warmup 10 --cpu-scenes 0 --latency-runs 0 --train-steps 0 --no-lookahead-steps 0 --real-density-steps 0 --split-products-steps 0 --graphs $g --set get_regiondataset.DEVICE_DRAWS=$v 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('B=$B graphs=$g device_draws=$v: %.3f ms/step %.1f scenes/s (hip_graphs %s)' % (j['ms_per_step'], j['value'], j['config']['hip_graphs']))"
  done
 done
done | tee gpurun_out/$TAG/small_batch.txt
