#!/bin/bash
# One GPU-box pass that produces everything kept under profiles/ for a round tag:
#   bash scripts/collect_profiles.sh r01d [extra bench.py arguments, e.g. --mlp-streams 1]
#   (run through gpurun; outputs land in gpurun_out/<tag>/)
# 1. the bench line (with cpu_baseline), 2. rocprofv3 --kernel-trace --stats of the same command,
# 3. separate PMC passes (FETCH_SIZE, WRITE_SIZE) of the same command.  Summaries: scripts/rocprof_summary.py,
# scripts/collect_pmc.py.
TAG=${1:-rXX}
shift
EXTRA="$@"
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
python $REPO/bench.py $EXTRA > $OUT/bench.json 2> $OUT/bench.err
rocprofv3 --kernel-trace --stats -d $OUT/prof -o $TAG -- python $REPO/bench.py $EXTRA --cpu-scenes 0 --exclusive-steps 0 --latency-runs 0 --train-steps 0 --no-lookahead-steps 0 --real-density-steps 0 --split-products-steps 0 --steps 32 > $OUT/prof.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch -o p -- python $REPO/bench.py $EXTRA --cpu-scenes 0 --exclusive-steps 0 --latency-runs 0 --train-steps 0 --no-lookahead-steps 0 --real-density-steps 0 --split-products-steps 0 --steps 4 > $OUT/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write -o p -- python $REPO/bench.py $EXTRA --cpu-scenes 0 --exclusive-steps 0 --latency-runs 0 --train-steps 0 --no-lookahead-steps 0 --real-density-steps 0 --split-products-steps 0 --steps 4 > $OUT/pmc_write.log 2>&1
cd $REPO
DB=$(find $OUT/prof -name "*results.db" | head -1)
# (the workload the trace is of goes into the JSON: bench.py refuses to price another batch / point count with it)
WB=$(echo " $EXTRA" | sed -n 's/.* --batch \([0-9]*\).*/\1/p'); WP=$(echo " $EXTRA" | sed -n 's/.* --points \([0-9]*\).*/\1/p')
python scripts/rocprof_summary.py $DB --json $OUT/rocprof_launch_ms.json --tag $TAG --steps 37 --batch ${WB:-8} --points ${WP:-25600} > $OUT/kernel_stats.txt
python scripts/collect_pmc.py $OUT/pmc_fetch $OUT/pmc_write > $OUT/pmc_traffic.json
# the raw traces are large; keep the summaries and the trace database only
rm -rf $OUT/pmc_fetch $OUT/pmc_write
tail -c 1500 $OUT/bench.json | head -c 700; echo; head -12 $OUT/kernel_stats.txt
