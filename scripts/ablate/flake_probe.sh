cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out/flake
for i in $(seq 1 ${1:-5}); do timeout 400 python -m pytest tests/test_gpu_train.py -x -q --tb=short > gpurun_out/flake/run_$i.log 2>&1; grep -E "passed|failed" gpurun_out/flake/run_$i.log | tail -1; done
