#!/bin/bash
# PMC passes over the stand-alone MLP GEMM micro-benchmark (scripts/ablate/mlp_ablate_0 <shape>).
cd /tmp && export TMPDIR=/tmp
BIN=$GRAFT_REPO_ROOT/scripts/ablate/mlp_ablate_0
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_ablate
mkdir -p $OUT
for shape in "$@"; do
  i=0
  for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT" \
             "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM GRBM_GUI_ACTIVE" \
             "TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_LATENCY_sum"; do
    i=$((i+1))
    timeout 120 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $OUT/s${shape}_g$i -o p -- $BIN $shape > $OUT/s${shape}_g$i.log 2>&1
  done
done
python3 - <<'PY'
import csv, glob, collections, os
root=os.environ["GRAFT_REPO_ROOT"]+"/gpurun_out/pmc_ablate"
for d in sorted(glob.glob(root+"/s*_g*")):
    if not os.path.isdir(d): continue
    acc=collections.defaultdict(list)
    for f in glob.glob(d+"/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "mlp_gemm" in r["Kernel_Name"]: acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    print(os.path.basename(d), {k: round(sum(v)/len(v)) for k,v in acc.items()})
PY
