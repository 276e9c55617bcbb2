# alternating A/B of bench.py --train with one module switch flipped:  bash scripts/ablate/train_ab.sh get_regiondataset.BATCHED_LABELS
cd ${GRAFT_REPO_ROOT:-/root/repo}
SW=$1
for rep in 1 2 3; do
  for v in 1 0; do
    python bench.py --train --batch 8 --steps 20 --warmup 4 --set $SW=$v 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$SW=$v', d['value'], d['ms_per_step'])"
  done
done
