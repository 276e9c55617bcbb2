# What does an initialised RCCL process group cost a ONE-GPU run that issues no collective in its timed region?
# (bench.py --single-rank-group: a one-rank communicator).  Round 6: 9 % of the forward line and 3 ms of a training iteration when
# RCCL's streams are bound to hardware queues before the pipeline's; nothing once sharding.init binds the package's streams first.
run() { # label, bench arguments...
  label=$1; shift
  out=$(python bench.py "$@" 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], (d.get('trunk_stream') or {}).get('region_wait_ms'))")
  echo "$label: $out"
}
F="--steps 40 --warmup 5 --cpu-scenes 0 --train-steps 0 --latency-runs 0 --no-lookahead-steps 0 --real-density-steps 0 --split-products-steps 0"
T="--train --batch 8 --steps 24 --warmup 5"
for rep in 1 2; do
run "forward, no group" $F
run "forward, group   " $F --single-rank-group
run "train,   no group" $T
run "train,   group   " $T --single-rank-group
done
