cd ${GRAFT_REPO_ROOT:-/root/repo}
timeout 500 python -m pytest tests/test_gpu_heads_train.py tests/test_gpu_train.py -x -q 2>&1 | tail -5
for i in 1 2; do
timeout 300 python bench.py --train --batch 8 --steps 20 --warmup 4 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('native', d['ms_per_step'])"
timeout 300 python bench.py --train --batch 8 --steps 20 --warmup 4 --set heads_train.ENABLED=0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('torch ', d['ms_per_step'])"
done
