run() { label=$1; shift; out=$(env $ENVV python bench.py "$@" 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], (d.get('trunk_stream') or {}).get('region_wait_ms'))"); echo "$label [$ENVV]: $out"; }
F="--steps 40 --warmup 5 --cpu-scenes 0 --train-steps 0 --latency-runs 0 --no-lookahead-steps 0 --real-density-steps 0 --split-products-steps 0"
T="--train --batch 8 --steps 24 --warmup 5"
for rep in 1 2; do
for e in "A=1" "GPU_MAX_HW_QUEUES=8" "GPU_MAX_HW_QUEUES=2" "HIP_FORCE_DEV_KERNARG=1"; do
ENVV=$e run "forward" $F
ENVV=$e run "train  " $T
done; done
