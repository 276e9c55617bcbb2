TAG=r03b; REPO=$GRAFT_REPO_ROOT; OUT=$REPO/gpurun_out/$TAG; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
X="--steps 4 --warmup 5 --cpu-scenes 0 --exclusive-steps 0 --latency-runs 0 --train-steps 0 --no-lookahead-steps 0 --real-density-steps 0 --split-products-steps 0"
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch -o p -- python $REPO/bench.py $X > $OUT/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write -o p -- python $REPO/bench.py $X > $OUT/pmc_write.log 2>&1
cd $REPO
python scripts/collect_pmc.py $OUT/pmc_fetch $OUT/pmc_write > $OUT/pmc_traffic.json
rm -rf $OUT/pmc_fetch $OUT/pmc_write
