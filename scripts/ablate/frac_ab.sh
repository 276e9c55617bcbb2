# same-box alternating A/B of the driver-style bench with one switch flipped, reporting the dominant kernel's launch time
cd ${GRAFT_REPO_ROOT:-/root/repo}
SW=$1
for rep in 1 2 3; do
  for v in 1 0; do
    python bench.py --steps 20 --warmup 5 --cpu-scenes 0 --latency-runs 0 --train-steps 0 --no-lookahead-steps 0 --real-density-steps 0 --split-products-steps 0 --set $SW=$v 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('$SW=$v', d['value'], d['ms_per_step'], 'frac', r['frac'], 'sa_chain ms', r['avg_launch_ms'], {k[:12]:v for k,v in list(r['families_ms_per_step'].items())[:4]})"
  done
done
