cd ${GRAFT_REPO_ROOT:-/root/repo}
for i in 1 2 3; do for ga in 1 2 3; do python bench.py --steps 20 --warmup 5 --cpu-scenes 0 --latency-runs 0 --train-steps 0 --no-lookahead-steps 0 --real-density-steps 0 --split-products-steps 0 --geometry-ahead $ga 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=j['roofline']
print('geometry-ahead $ga: %.3f ms/step %.1f scenes/s  %s' % (j['ms_per_step'], j['value'], r['families_ms_per_step']))"; done; done
