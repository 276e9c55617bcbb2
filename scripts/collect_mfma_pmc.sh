#!/bin/bash
# MFMA-utilisation and issue counters per kernel family, of the default bench command, plus a calibration kernel.
#   bash scripts/collect_mfma_pmc.sh r02   (through gpurun; output gpurun_out/<tag>/pmc_mfma.json)
# Pass 1: SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE
# Pass 2: SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_RD GRBM_GUI_ACTIVE
# The same two passes over scripts/ablate/mfma_peak (a bare v_mfma_f32_32x32x2_f32 loop: matrix pipe ~99 % busy) give the
# normalisation of "MFMA busy": busy fraction of a kernel = (MFMA_BUSY / GUI_ACTIVE) / (the same ratio of the bare loop).
TAG=${1:-rXX}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
G1="SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE"
G2="SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_RD GRBM_GUI_ACTIVE"
timeout 120 rocprofv3 --pmc $G1 --kernel-trace --output-format csv -d $OUT/pmc_cal1 -o p -- $REPO/scripts/ablate/mfma_peak > $OUT/pmc_cal1.log 2>&1
timeout 600 rocprofv3 --pmc $G1 --kernel-trace --output-format csv -d $OUT/pmc_g1 -o p -- python $REPO/bench.py --cpu-scenes 0 --exclusive-steps 0 --latency-runs 0 --train-steps 0 --no-lookahead-steps 0 --real-density-steps 0 --split-products-steps 0 --steps 4 > $OUT/pmc_g1.log 2>&1
timeout 600 rocprofv3 --pmc $G2 --kernel-trace --output-format csv -d $OUT/pmc_g2 -o p -- python $REPO/bench.py --cpu-scenes 0 --exclusive-steps 0 --latency-runs 0 --train-steps 0 --no-lookahead-steps 0 --real-density-steps 0 --split-products-steps 0 --steps 4 > $OUT/pmc_g2.log 2>&1
cd $REPO
python scripts/summarize_mfma_pmc.py $OUT/pmc_cal1 $OUT/pmc_g1 $OUT/pmc_g2 > $OUT/pmc_mfma.json
rm -rf $OUT/pmc_cal1 $OUT/pmc_g1 $OUT/pmc_g2
head -c 3000 $OUT/pmc_mfma.json
