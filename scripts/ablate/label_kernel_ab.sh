cd ${GRAFT_REPO_ROOT:-/root/repo}
timeout 600 python -m pytest tests/test_gpu_train.py tests/test_abi.py -x -q 2>&1 | tail -5
for i in 1 2; do
timeout 300 python bench.py --train --batch 8 --steps 20 --warmup 4 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('new ', d['ms_per_step'])"
timeout 300 python bench.py --train --batch 8 --steps 20 --warmup 4 --set get_regiondataset.LABEL_KERNEL=0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('no label kernel ', d['ms_per_step'])"
done
