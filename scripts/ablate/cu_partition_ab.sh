# EXPERIMENT: queue CU masks (feature stream off r CUs per XCD, region / sampling streams on them) vs the default
cd ${GRAFT_REPO_ROOT:-/root/repo}
COMMON="--steps 20 --warmup 5 --cpu-scenes 0 --latency-runs 0 --train-steps 0 --no-lookahead-steps 0 --real-density-steps 0 --split-products-steps 0"
run() { timeout 300 python bench.py $COMMON "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('%-44s %8.1f scenes/s %7.3f ms  %s' % (' '.join(sys.argv[1:]) or '(default)', d['value'], d['ms_per_step'], {k[:10]:v for k,v in list(r['families_ms_per_step'].items())[:5]}))" "$@"; }
for rep in 1 2; do
  run
  run --reserve-cus 1 --reserve-for reg
  run --reserve-cus 1 --reserve-for reg,fps
  run --reserve-cus 2 --reserve-for reg
  run --mlp-streams 2
  run --set pipeline.LEVEL_EVENTS=1
done
echo "== 200 steps"
COMMON="--steps 200 --warmup 5 --cpu-scenes 0 --latency-runs 0 --train-steps 0 --no-lookahead-steps 0 --real-density-steps 0 --split-products-steps 0"
run
run --reserve-cus 1 --reserve-for reg,fps
run --reserve-cus 1 --reserve-for reg
