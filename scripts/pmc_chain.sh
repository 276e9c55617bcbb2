#!/bin/bash
# PMC passes over the stand-alone chained set-abstraction micro-benchmark (scripts/ablate/chain_ablate_0).
cd /tmp && export TMPDIR=/tmp
BIN=$GRAFT_REPO_ROOT/scripts/ablate/chain_ablate_0
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_chain
mkdir -p $OUT
i=0
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT" \
           "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE" \
           "SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_WAVES SQ_INSTS_MFMA SQ_ACTIVE_INST_FLAT"; do
  i=$((i+1))
  timeout 120 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $OUT/g$i -o p -- $BIN > $OUT/g$i.log 2>&1
done
python3 - <<'PY'
import csv, glob, collections, os
root=os.environ["GRAFT_REPO_ROOT"]+"/gpurun_out/pmc_chain"
for d in sorted(glob.glob(root+"/g*")):
    if not os.path.isdir(d): continue
    acc=collections.defaultdict(list)
    for f in glob.glob(d+"/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "sa_chain" in r["Kernel_Name"]: acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    print(os.path.basename(d), {k: round(sum(v)/len(v)) for k,v in acc.items()})
PY
