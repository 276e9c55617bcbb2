# Where does sa_chain_split_kernel's time go?  Measurement builds of the library with -DSS_ABLATE=n (csrc/sa_split.hip), timed by
# scripts/bench_sa_split.py (results of ablated builds are wrong on purpose).   bash scripts/ablate/sa_split_ablate.sh   (GPU box)
cd ${GRAFT_REPO_ROOT:-/root/repo}
for n in 0 2 4 5 7 8 9 10; do
  lib=gpurun_variant_ss$n.so
  [ -f $lib ] || python -c "
import sys; sys.path.insert(0, 'regnet_for_3d_grasping_amd/csrc'); import build; build.build_variant('$lib', ['-DSS_ABLATE=$n'])" > /dev/null 2>&1
  echo -n "SS_ABLATE=$n: "; REGNET_HIP_LIB=$PWD/$lib python scripts/bench_sa_split.py 8 2>/dev/null | tail -1
done
