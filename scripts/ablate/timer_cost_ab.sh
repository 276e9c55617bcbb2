# what bench.py's own HIP-event brackets cost the step: --time-every 8 (default) vs sparser, alternating
cd ${GRAFT_REPO_ROOT:-/root/repo}
for steps in 20 200; do
COMMON="--steps $steps --warmup 5 --cpu-scenes 0 --latency-runs 0 --train-steps 0 --no-lookahead-steps 0 --real-density-steps 0"
for rep in 1 2; do
  for te in 8 32 100000; do
    timeout 300 python bench.py $COMMON --time-every $te 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('steps $steps time-every %-7s %8.1f scenes/s %7.3f ms  frac %.3f launches %d' % ('$te', d['value'], d['ms_per_step'], r['frac'], r['launches']))"
  done
done
done
