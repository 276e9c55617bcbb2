#!/bin/bash
# what bench.py's own HIP-event brackets cost the timed region: --time-every n (every n-th call of each native op is bracketed)
cd ${GRAFT_REPO_ROOT:-/root/repo}
steps=${1:-120}
COMMON="--steps $steps --warmup 5 --cpu-scenes 0 --latency-runs 0 --train-steps 0 --no-lookahead-steps 0 --real-density-steps 0 --split-products-steps 0"
for i in 1 2; do for te in 8 1 32 100000; do
  python bench.py $COMMON --time-every $te 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('time-every $te: %.3f ms/step %.1f scenes/s' % (j['ms_per_step'], j['value']))"
done; done
