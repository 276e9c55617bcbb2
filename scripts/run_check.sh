cd ${GRAFT_REPO_ROOT:-/root/repo}
OUT=gpurun_out/${1:-r04k}; mkdir -p $OUT
timeout 1200 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1; tail -3 $OUT/pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 400 python bench.py --steps 20 --warmup 5 > $OUT/bench.json 2>$OUT/bench.err; tail -c 600 $OUT/bench.json
timeout 300 python bench.py --train --batch 8 --steps 20 --warmup 4 > $OUT/train_b8.json 2>/dev/null; tail -c 400 $OUT/train_b8.json
