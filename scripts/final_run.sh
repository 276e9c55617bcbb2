cd ${GRAFT_REPO_ROOT:-/root/repo}
TAG=$1
OUT=gpurun_out/$TAG
mkdir -p $OUT
timeout 1500 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1; tail -2 $OUT/pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
for i in 1 2 3; do timeout 300 python -m pytest tests/test_gpu_train.py -x -q --tb=short > $OUT/pytest_train_again_$i.log 2>&1; tail -1 $OUT/pytest_train_again_$i.log; done
timeout 1200 bash scripts/collect_profiles.sh $TAG --steps 20 --warmup 5 > $OUT/collect.log 2>&1; tail -3 $OUT/collect.log
timeout 300 python bench.py --steps 20 --warmup 5 --cpu-scenes 0 --train-steps 0 > $OUT/bench_20.json 2>/dev/null
timeout 300 python bench.py --steps 200 --warmup 5 --cpu-scenes 0 --train-steps 0 > $OUT/bench_200.json 2>/dev/null
timeout 300 python bench.py --train --batch 8 --steps 20 --warmup 4 > $OUT/train_b8.json 2>/dev/null
timeout 300 python bench.py --train --batch 4 --points 51200 --steps 12 --warmup 3 > $OUT/train_51200_b4.json 2>/dev/null
timeout 600 bash scripts/collect_train_profile.sh $TAG > $OUT/train_profile.log 2>&1
timeout 600 bash scripts/other_shapes.sh > $OUT/other_shapes.txt 2>&1
timeout 900 bash scripts/collect_mfma_pmc.sh $TAG > $OUT/pmc_mfma.log 2>&1
python - <<PY
import json
for f in ("bench","bench_20","bench_200","train_b8","train_51200_b4"):
    try:
        d=json.loads(open("$OUT/%s.json"%f).read().strip().splitlines()[-1]); r=d.get("roofline") or {}
        print(f, d["value"], d["ms_per_step"], r.get("frac"), r.get("frac_rocprof"), r.get("step_frac_executed"), d.get("latency_ms_single_scene"), d.get("value_no_lookahead"), d.get("value_real_density"), (d.get("train") or {}).get("ms_per_step"))
    except Exception as e: print(f, "ERR", e)
PY
