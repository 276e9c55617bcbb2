#!/bin/bash
# Same-box A/B of a module switch: bash scripts/ablate/switch_steps_ab.sh <steps> <rounds> <module.NAME=value>   (A = defaults)
STEPS=$1; R=$2; SW=$3
for i in $(seq 1 $R); do
  for v in A "$SW"; do
    if [ "$v" = A ]; then EXTRA=""; else EXTRA="--set $v"; fi
    python bench.py --steps $STEPS --warmup 5 --cpu-scenes 0 --latency-runs 0 --train-steps 0 --no-lookahead-steps 0 --real-density-steps 0 --split-products-steps 0 $EXTRA 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=j['roofline']
print('%-36s %d steps: %.3f ms/step %.1f scenes/s late %s frac %.3f  %s' % ('$v', j['steps'], j['ms_per_step'], j['value'], j['config']['host_late_feature_stages'], r['frac'], {k: v for k, v in list(r['families_ms_per_step'].items())[:6]}))"
  done
done
