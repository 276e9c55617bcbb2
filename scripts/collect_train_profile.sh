#!/bin/bash
# Steady-state kernel listing of the training iteration (configs[3] shard, 8 x 25 600):
#   bash scripts/collect_train_profile.sh r04e      (through gpurun; outputs in gpurun_out/<tag>/)
# rocprofv3 --kernel-trace of `bench.py --train` with enough iterations that the LAST 600 ms hold steady-state iterations only
# (MIOpen's find-mode trial kernels and the first touches all lie in front), summarised by scripts/trace_tail.py.
TAG=${1:-rXX}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $OUT/prof_train -o train -- python $REPO/bench.py --train --batch 8 --steps 24 --warmup 6 --train-timeline-steps 0 --train-accounting-steps 0 > $OUT/train_prof.log 2>&1
cd $REPO
python scripts/trace_tail.py $OUT/prof_train 600 60 > $OUT/train_steady_state_kernels.txt
python scripts/trace_gaps.py $OUT/prof_train 600 40 > $OUT/train_steady_state_gaps.txt
# keep the raw kernel trace (compact: start, end, stream/queue, grid, kernel name) for offline analysis
python scripts/trace_compact.py $OUT/prof_train $OUT/train_trace_compact.csv.gz
rm -rf $OUT/prof_train
head -12 $OUT/train_steady_state_kernels.txt; cat $OUT/train_steady_state_gaps.txt
