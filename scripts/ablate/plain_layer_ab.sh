# In-pipeline A/B of the plain-layer launch policies over the driver's 20 steps, alternating on one box:
#   old   REGNET_G2_T8=0, fused.STREAM_LAYERS=0   (64 x 128 x 4-wave tile for the slab layers, one workgroup per tile)
#   t8    fused.STREAM_LAYERS=0                     (128 x 128 x 4 waves of 64 x 64 at two workgroups per CU)
#   strm  fused.STREAM_LAYERS=1                     (persistent launch where >= 256 tiles)
cd ${GRAFT_REPO_ROOT:-/root/repo}
R=${1:-3}; STEPS=${2:-20}
COMMON="--steps $STEPS --warmup 5 --cpu-scenes 0 --latency-runs 0 --train-steps 0 --no-lookahead-steps 0 --real-density-steps 0 --split-products-steps 0"
for i in $(seq 1 $R); do
  for v in old t8 strm; do
    case $v in
      old)  E="REGNET_G2_T8=0"; X="--set fused.STREAM_LAYERS=0";;
      t8)   E="REGNET_G2_T8=1"; X="--set fused.STREAM_LAYERS=0";;
      strm) E="REGNET_G2_T8=1"; X="--set fused.STREAM_LAYERS=1";;
    esac
    env $E python bench.py $COMMON $X 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=j['roofline']
plain=sum(k['total_ms'] for k in j['kernels'] if k['op']=='mlp_layer' and int(k['shape'].split()[0][1:])>=2048)/j['steps']
print('$v %d steps: %.3f ms/step %.1f scenes/s  plain layers %.3f ms/step  %s' % (j['steps'], j['ms_per_step'], j['value'], plain, r['families_ms_per_step']))"
  done
done
