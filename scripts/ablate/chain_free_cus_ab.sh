# alternating 20-step forward runs with N CUs left free by the persistent chain kernels: bash scripts/ablate/chain_free_cus_ab.sh "8 16" [rounds]
cd ${GRAFT_REPO_ROOT:-/root/repo}
for i in $(seq 1 ${2:-2}); do
for n in 0 $1; do
timeout 300 python bench.py --steps 20 --warmup 5 --cpu-scenes 0 --train-steps 0 --latency-runs 0 --no-lookahead-steps 0 --real-density-steps 0 --split-products-steps 0 --set fused.CHAIN_FREE_CUS=$n 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('free $n', d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'])"
done; done
