# stage-boundary gap of the feature stream inside ForwardPipeline under variations (scripts/pipeline_boundary_gaps.py)
cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for cfg in "PIPE_TAIL=1" "PIPE_GEO_SAME=1" "PIPE_REGION=0 PIPE_GEO_SAME=1" "GPU_MAX_HW_QUEUES=8" "GPU_MAX_HW_QUEUES=2"; do
  rm -rf /tmp/pg_x; env $cfg rocprofv3 --kernel-trace --stats -d /tmp/pg_x -o g -- python $R/scripts/pipeline_boundary_gaps.py > /tmp/pg.log 2>&1
  echo "$cfg: $(python $R/scripts/pipeline_boundary_gaps.py --report /tmp/pg_x)"
done
