#!/bin/bash
# Multi-GPU lines of bench.py, for a driver with 1 / 2 / 4 / 8 MI355X on ONE node (nothing here needs more than bench.py's own
# contract: one process per GPU under torch.distributed.run, RCCL over xGMI, rendezvous on 127.0.0.1):
#     bash scripts/scale_driver.sh [out_dir] [N ...]          default: gpurun_out/scale, N = 1 2 4 8 up to the GPUs present
# Per N it runs, each as ONE JSON line into <out_dir>/<name>_n<N>.json:
#   forward         configs[2] weak scaling: 8 scenes of 25 600 points per GPU                     bench.py --gpus N
#   train           configs[3] weak scaling: 8 scenes of 25 600 points per GPU, one flat gradient all-reduce
#   train_51200     configs[4] weak scaling: 4 scenes of 51 200 points per GPU
#   train_gb16      configs[3] VERBATIM (global batch 16) at N = 2, 4        --global-batch 16 --train
#   train_gb32      configs[4] VERBATIM (global batch 32, 51 200 points) at N = 8      --global-batch 32 --train --points 51200
# and CHECKS every line: n_gpus == N, config.collective.world_size == N == distinct_ranks_by_allreduce == distinct_devices,
# backend nccl; it prints value / ms_per_step / allreduce_ms / bus_GBps.  At N = 1 the lines are produced with
# --single-rank-group: a ONE-rank RCCL communicator on which the job's collectives (the rank / device census, the
# bucket-sized probe, GradientBucket's side-stream all-reduce) are really issued -- the multi-GPU code path up to and
# including init_process_group("nccl") on a one-GPU box.  No scaling efficiency is computed here: the driver does that.
set -u
cd ${GRAFT_REPO_ROOT:-$(dirname $(dirname $(readlink -f $0)))}
OUT=${1:-gpurun_out/scale}; shift || true
mkdir -p $OUT
HAVE=$(python -c "import torch; print(torch.cuda.device_count())")
NS=${@:-1 2 4 8}
export HSA_ENABLE_IPC_MODE_LEGACY=0
PORT=29610
FAIL=0
line() {   # name N bench-arguments...
  local name=$1 n=$2; shift 2
  local out=$OUT/${name}_n${n}.json
  PORT=$((PORT + 1))
  if [ "$n" = 1 ]; then
    timeout 900 python bench.py --gpus 1 --single-rank-group "$@" > $out 2> ${out%.json}.err
  else
    timeout 1500 python -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port $PORT \
      bench.py --gpus $n "$@" > $out 2> ${out%.json}.err
  fi
  python - "$out" "$name" "$n" <<'PY' || FAIL=1
import json, sys
path, name, n = sys.argv[1], sys.argv[2], int(sys.argv[3])
try:
    d = json.loads(open(path).read().strip().splitlines()[-1])
except Exception as exc:
    print("%-12s N=%d: NO LINE (%r) -- see %s" % (name, n, exc, path.replace(".json", ".err"))); sys.exit(1)
c = (d.get("config") or {}).get("collective") or {}
ok = (d.get("n_gpus") == n and c.get("world_size") == n and c.get("distinct_ranks_by_allreduce") == n
      and c.get("distinct_devices") == n and c.get("backend") == "nccl")
t = d.get("train") or {}
print("%-12s N=%d: %9.2f %s  %8.3f ms/step  allreduce_ms %s  probe %s ms  bus %s GB/s  rccl %s  %s" % (
    name, n, d["value"], d["unit"], d["ms_per_step"], c.get("allreduce_ms", t.get("allreduce_ms")), c.get("probe_allreduce_ms"),
    c.get("bus_GBps"), c.get("rccl_version"), "ok" if ok else "CHECK FAILED: %s" % json.dumps(c)))
sys.exit(0 if ok else 1)
PY
}
for N in $NS; do
  if [ "$N" -gt "$HAVE" ]; then echo "N=$N: only $HAVE GPU(s) here, skipped"; continue; fi
  line forward     $N --steps 40 --warmup 5 --cpu-scenes 0 --latency-runs 0 --train-steps 0 --no-lookahead-steps 0 --real-density-steps 0 --split-products-steps 0
  line train       $N --train --batch 8 --steps 24 --warmup 5
  line train_51200 $N --train --batch 4 --points 51200 --steps 16 --warmup 4
  if [ "$N" = 2 ] || [ "$N" = 4 ]; then line train_gb16 $N --train --global-batch 16 --steps 24 --warmup 5; fi
  if [ "$N" = 8 ]; then line train_gb32 $N --train --global-batch 32 --points 51200 --steps 16 --warmup 4; fi
done
exit $FAIL
