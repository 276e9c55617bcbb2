for r in 1 2; do for t in 1 2 4 8 1000000 off; do
  if [ $t = off ]; then EXTRA="--set fused.SA_CHAIN_PERSISTENT=0"; unset REGNET_SA_CHAIN_TILES; else EXTRA=""; export REGNET_SA_CHAIN_TILES=$t; fi
  python bench.py --steps 120 --warmup 5 --cpu-scenes 0 --latency-runs 0 --train-steps 0 --no-lookahead-steps 0 --real-density-steps 0 --split-products-steps 0 $EXTRA 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=j['roofline']
print('tiles %-8s %d steps: %.3f ms/step %.1f scenes/s late %s frac %.3f sa_chain %.3f' % ('$t', j['steps'], j['ms_per_step'], j['value'], j['config']['host_late_feature_stages'], r['frac'], r['families_ms_per_step'].get('sa_chain_kernel', 0)))"
done; done
