# alternating A/B of the forward bench with one module switch flipped:  bash scripts/ablate/switch_ab.sh fused.HEADS_CHAIN "--batch 1"
cd ${GRAFT_REPO_ROOT:-/root/repo}
SW=$1; shift
for rep in 1 2 3; do
  for v in 1 0; do
    python bench.py "$@" --steps 60 --warmup 8 --cpu-scenes 0 --latency-runs 4 --train-steps 0 --no-lookahead-steps 0 --real-density-steps 0 --split-products-steps 0 --set $SW=$v 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$SW=$v $*', d['value'], d['ms_per_step'], 'latency', d['latency_ms_single_scene'])"
  done
done
