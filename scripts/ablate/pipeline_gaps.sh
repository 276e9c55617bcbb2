# stage-boundary gap / steady-state step of the feature stream inside ForwardPipeline under variations (scripts/pipeline_boundary_gaps.py)
cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for cfg in "PIPE_TAIL=1" "PIPE_CALIB=1" "PIPE_CALIB=1 PIPE_DISTINCT=8" "PIPE_CALIB=1 PIPE_DISTINCT=8 PIPE_TIMERS=1"; do
  rm -rf /tmp/pg_x; env $cfg rocprofv3 --kernel-trace --stats -d /tmp/pg_x -o g -- python $R/scripts/pipeline_boundary_gaps.py > /tmp/pg.log 2>&1
  echo "$cfg: $(python $R/scripts/pipeline_boundary_gaps.py --report /tmp/pg_x)"
done
