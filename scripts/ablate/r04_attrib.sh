# what the refine stage and the rotating inputs cost the step: alternating runs of bench.py (20 timed steps)
cd ${GRAFT_REPO_ROOT:-/root/repo}
COMMON="--steps 20 --warmup 5 --cpu-scenes 0 --latency-runs 0 --train-steps 0 --no-lookahead-steps 0 --real-density-steps 0 --split-products-steps 0"
for rep in 1 2 3; do
  for cfg in "" "--no-region-calibration" "--distinct-batches 1" "--no-region-calibration --distinct-batches 1" "--score-only"; do
    python bench.py $COMMON $cfg 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('%-52s %8.1f scenes/s %7.3f ms  chains %s' % ('$cfg' or '(default)', d['value'], d['ms_per_step'], {k[:10]:v for k,v in list(r['families_ms_per_step'].items())[:5]}))"
  done
done
