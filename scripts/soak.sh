#!/bin/bash
# Stability pass on one GPU box: the GPU suite twice more, then the driver's bench command ten times (distribution of the line).
#   bash scripts/soak.sh <tag>     (through gpurun; writes gpurun_out/<tag>/soak.txt)
TAG=${1:-soak}; OUT=gpurun_out/$TAG; mkdir -p $OUT
for i in 1 2; do python -m pytest tests -q -m gpu -x 2>&1 | tail -1; done > $OUT/soak.txt
for i in $(seq 1 10); do
  python bench.py --steps 20 --warmup 5 --cpu-scenes 0 --latency-runs 0 --train-steps 0 --no-lookahead-steps 0 --real-density-steps 0 --split-products-steps 0 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('20 steps: %.3f ms/step %.1f scenes/s late %s' % (j['ms_per_step'], j['value'], j['config']['host_late_feature_stages']))"
done >> $OUT/soak.txt
for i in $(seq 1 5); do
  python bench.py --steps 200 --warmup 5 --cpu-scenes 0 --latency-runs 0 --train-steps 0 --no-lookahead-steps 0 --real-density-steps 0 --split-products-steps 0 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('200 steps: %.3f ms/step %.1f scenes/s late %s' % (j['ms_per_step'], j['value'], j['config']['host_late_feature_stages']))"
done >> $OUT/soak.txt
cat $OUT/soak.txt
