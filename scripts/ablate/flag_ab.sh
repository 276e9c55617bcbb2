# same-box alternating A/B of the driver-style bench with extra flags:  bash scripts/ablate/flag_ab.sh "--mlp-streams 2" [reps]
cd ${GRAFT_REPO_ROOT:-/root/repo}
FLAGS="$1"
for rep in $(seq 1 ${2:-3}); do
  for v in base flags; do
    if [ $v = base ]; then F=""; else F="$FLAGS"; fi
    python bench.py --steps 20 --warmup 5 --cpu-scenes 0 --latency-runs 0 --train-steps 0 --no-lookahead-steps 0 --real-density-steps 0 --split-products-steps 0 --exclusive-steps 0 $F 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('$v [$F]', d['value'], d['ms_per_step'], 'frac', r['frac'], 'sa_chain ms', r['avg_launch_ms'])"
  done
done
