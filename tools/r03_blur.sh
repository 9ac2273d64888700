#!/bin/bash
# Runs ON THE GPU BOX: placements of the MFMA blur (A/B in one call) and its counters.  Usage: tools/r03_blur.sh <tag>
TAG=${1:-r03x}
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/$TAG
mkdir -p $OUT
cd $REPO
VERIFY=8 bash tools/ab_env.sh 3 "ORBHIP_BLUR_PLACE=1" "ORBHIP_BLUR_PLACE=2" "ORBHIP_BLUR_PLACE=3" "ORBHIP_BLUR_PLACE=4" > $OUT/blur_place.txt 2>&1
export ORBHIP_BLUR_PLACE=2
bash tools/pmc_gpu.sh util "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_BUSY_CYCLES" 2>&1 | grep -E "^k_(blur|fast)" > $OUT/blur_counters.txt
bash tools/pmc_gpu.sh util2 "GRBM_GUI_ACTIVE SQ_WAVES SQ_INSTS_SALU SQ_INSTS_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU_MFMA_MOPS_I8 SQ_VALU_MFMA_BUSY_CYCLES" 2>&1 | grep -E "^k_(blur|fast)" >> $OUT/blur_counters.txt
bash tools/pmc_gpu.sh fetch "FETCH_SIZE" 2>&1 | grep -E "^k_blur" >> $OUT/blur_counters.txt
bash tools/pmc_gpu.sh write "WRITE_SIZE" 2>&1 | grep -E "^k_blur" >> $OUT/blur_counters.txt
bash tools/pmc_gpu.sh l2 "TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum" 2>&1 | grep -E "^k_blur" >> $OUT/blur_counters.txt
cat $OUT/blur_place.txt $OUT/blur_counters.txt
