#!/bin/bash
# How much does a training iteration depend on the speed of the launching thread?  bench.py --train pinned to ONE core, alone and
# with a busy loop pinned to the same core (the launching thread then gets about half of it: host_ms_per_step roughly doubles),
# hipGraph replays on and off.     bash scripts/train_host_sensitivity.sh [core]
cd ${GRAFT_REPO_ROOT:-$(dirname $(dirname $(readlink -f $0)))}
CORE=${1:-3}
show() { python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); p=d.get('trunk_stream') or {}
print('%-34s %7.2f ms/iteration (median %s)  host %6.2f ms  region_wait %s ms  replays %s' % ('$1', d['ms_per_step'], d.get('ms_per_step_median'), d['host_ms_per_step'], p.get('region_wait_ms'), d.get('graph_replays')))"; }
for graphs in on off; do
  taskset -c $CORE python bench.py --train --batch 8 --steps 24 --warmup 5 --train-graphs $graphs --train-timeline-steps 0 2>/dev/null | show "graphs $graphs, core alone"
  taskset -c $CORE python -c "
while True: pass" &
  SPIN=$!
  taskset -c $CORE python bench.py --train --batch 8 --steps 24 --warmup 5 --train-graphs $graphs --train-timeline-steps 0 2>/dev/null | show "graphs $graphs, core shared with a spinner"
  kill $SPIN
  taskset -c $CORE python -c "
while True: pass" &
  SPIN=$!
  taskset -c $CORE python -c "
while True: pass" &
  SPIN2=$!
  taskset -c $CORE python bench.py --train --batch 8 --steps 24 --warmup 5 --train-graphs $graphs --train-timeline-steps 0 2>/dev/null | show "graphs $graphs, core shared with two spinners"
  kill $SPIN $SPIN2
done
