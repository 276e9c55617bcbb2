#!/bin/bash
# What a co-running kernel that holds whole CUs costs the matrix kernels (DESIGN.md par. 10): run on the GPU box, writes
# gpurun_out/<tag>_dispatch_*.txt.  usage: bash scripts/collect_dispatch_evidence.sh r02e
tag=${1:-dispatch}
out=gpurun_out
mkdir -p $out
{ for n in 0 1 2; do echo "== level-1 sampling launches in flight beside the feature stage: $n"; FPS_STREAMS=$n ROWS=12 python scripts/features_alone.py 8 2>&1 | grep -v amdgpu; done; } > $out/${tag}_dispatch_features_beside_fps.txt
python scripts/side_load_probe.py 2>&1 | grep -v amdgpu > $out/${tag}_dispatch_side_load.txt
REGNET_HIP_LIB=scripts/ablate/libregnet_trace.so python scripts/wg_timeline.py 2>&1 | grep -v amdgpu > $out/${tag}_dispatch_wg_timeline.txt
python scripts/side_load_micro.py 2>&1 | grep -v amdgpu > $out/${tag}_dispatch_micro.txt
python scripts/queue_state_probe.py 2>&1 | grep -v amdgpu > $out/${tag}_dispatch_queue_state.txt
python scripts/clock_under_load.py 2>&1 | grep -v amdgpu > $out/${tag}_dispatch_clock.txt
{ echo "== alone"; python scripts/bench_gemm2_layers.py 2>&1 | grep -v amdgpu; echo "== 8 CUs held (one per XCD)"; SIDE_BLOCKS=8 python scripts/bench_gemm2_layers.py 2>&1 | grep -v amdgpu; } > $out/${tag}_dispatch_gemm2_layers.txt
for v in none fps plan; do { if [ $v = none ]; then python bench.py --cpu-scenes 0 --latency-runs 0; else python scripts/reuse_geometry_bench.py $v --cpu-scenes 0 --latency-runs 0; fi; } 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('geometry reuse = $v: %.3f ms per step, %s frac %.3f (%.3f ms per launch)' % (d['ms_per_step'], r['kernel'], r['frac'], r['avg_launch_ms']))"; done > $out/${tag}_dispatch_pipeline_without_sampling.txt
