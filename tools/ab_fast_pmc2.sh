#!/bin/bash
# One GPU call: activity counters of k_fast for several prebuilt libraries (two passes).   tools/ab_fast_pmc2.sh lib1.so lib2.so ...
LIB=vi-orb-slam-icra2018_amd/csrc/liborbhip.so
cp $LIB /tmp/liborbhip_keep1.so
A="--steps 2 --warmup 1 --batch 1024 --cpu-frames 0 --pipelined 0 --verify 0 --host-batch 0 --configs 0 --content 0"
for v in "$@"; do
  cp $v $LIB
  echo "== $(basename $v)"
  bash tools/pmc_gpu.sh a_$(basename $v .so) "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" $A 2>&1 | grep -E "^k_fast"
  bash tools/pmc_gpu.sh b_$(basename $v .so) "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_BRANCH SQ_INST_CYCLES_SALU SQ_WAVE_CYCLES SQ_ACTIVE_INST_MISC SQ_LDS_BANK_CONFLICT" $A 2>&1 | grep -E "^k_fast"
done
cp /tmp/liborbhip_keep1.so $LIB
