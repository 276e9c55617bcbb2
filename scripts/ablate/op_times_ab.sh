#!/bin/bash
# Per-op event times (bench.py --time-every 1) of library builds on one box: bash scripts/ablate/op_times_ab.sh [steps] A|<lib> ...
STEPS=${1:-120}; shift
for v in "$@"; do
  if [ $v = A ]; then unset REGNET_HIP_LIB; else export REGNET_HIP_LIB=$PWD/$v; fi
  python bench.py --steps $STEPS --warmup 5 --time-every 1 --cpu-scenes 0 --latency-runs 0 --train-steps 0 --no-lookahead-steps 0 --real-density-steps 0 --split-products-steps 0 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('== $v', j['ms_per_step'], 'late', j['config']['host_late_feature_stages'])
agg={}
for k in j['kernels']:
    key=k['op']+' '+k['shape'].split(' flop')[0].split(' ')[0]
    a=agg.setdefault(key,[0,0.0]); a[0]+=k['calls']; a[1]+=k['total_ms']
side=sum(t for key,(c,t) in agg.items() if not key.startswith(('sa_','fp_head','sa3')))
print('   side + plain ops, event ms per step: %.3f' % (side/j['steps']))
for key,(c,t) in sorted(agg.items(), key=lambda kv:-kv[1][1])[:12]:
    print('  %-40s calls %5d  avg %.4f ms  per step %.3f' % (key, c, t/c, t/j['steps']))
"
done
