for args in "--batch 16" "--batch 4" "--batch 2" "--batch 1" "--batch 8 --points 51200" "--batch 4 --score-only" "--batch 8 --points 12800"; do
python bench.py $args --steps 60 --warmup 8 --cpu-scenes 0 --latency-runs 0 --train-steps 0 --no-lookahead-steps 0 --real-density-steps 0 --split-products-steps 0 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$args: %.1f scenes/s, %.3f ms/step, hip_graphs %s' % (j['value'], j['ms_per_step'], j['config']['hip_graphs']))"
done
# test.py scale (test.py:68-71: ONE scene, 4000 centres, 256- / 2048-point groups, heads on 4000 rows, 4000 x 2048 box crop, refine)
python scripts/testpy_scale.py 25600 2>/dev/null | tail -7 | sed 's/^/test.py scale: /'
