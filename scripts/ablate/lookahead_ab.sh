# bench.py --lookahead (batches whose region stage may be pending = how far the launching thread runs ahead) x --geometry-ahead,
# the driver's 20 steps, alternating on one box
cd ${GRAFT_REPO_ROOT:-/root/repo}
for i in 1 2 3; do for cfg in "3 1" "6 1" "6 2" "10 2" "10 3"; do set -- $cfg; python bench.py --steps 20 --warmup 5 --cpu-scenes 0 --latency-runs 0 --train-steps 0 --no-lookahead-steps 0 --real-density-steps 0 --split-products-steps 0 --lookahead $1 --geometry-ahead $2 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=j['roofline']
print('lookahead $1 geometry-ahead $2: %.3f ms/step %.1f scenes/s  %s' % (j['ms_per_step'], j['value'], r['families_ms_per_step']))"; done; done
