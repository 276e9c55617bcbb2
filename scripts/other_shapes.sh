for args in "--batch 16" "--batch 4" "--batch 2" "--batch 1" "--batch 8 --points 51200" "--batch 4 --score-only" "--batch 8 --points 12800"; do
python bench.py $args --steps 60 --warmup 8 --cpu-scenes 0 --latency-runs 0 --train-steps 0 --no-lookahead-steps 0 --real-density-steps 0 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$args: %.1f scenes/s, %.3f ms/step, hip_graphs %s' % (j['value'], j['ms_per_step'], j['config']['hip_graphs']))"
done
