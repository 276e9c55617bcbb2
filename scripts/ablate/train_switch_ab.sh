# bash scripts/ablate/train_switch_ab.sh "<--set switch=0 ...>" [rounds]: alternating 30-step training runs with / without the switches
cd ${GRAFT_REPO_ROOT:-/root/repo}
for i in $(seq 1 ${2:-3}); do
timeout 300 python bench.py --train --batch 8 --steps 30 --warmup 4 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('default ', d['ms_per_step'])"
timeout 300 python bench.py --train --batch 8 --steps 30 --warmup 4 $1 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('switched', d['ms_per_step'])"
done
