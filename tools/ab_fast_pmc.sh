#!/bin/bash
# One GPU call: instruction counters of k_fast for several prebuilt libraries.   tools/ab_fast_pmc.sh lib1.so lib2.so ...
LIB=vi-orb-slam-icra2018_amd/csrc/liborbhip.so
cp $LIB /tmp/liborbhip_keep1.so
for v in "$@"; do
  cp $v $LIB
  echo "== $(basename $v)"
  bash tools/pmc_gpu.sh v_$(basename $v .so) "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES" --steps 2 --warmup 1 --batch 1024 --cpu-frames 0 --pipelined 0 --verify 0 --host-batch 0 --configs 0 --content 0 2>&1 | grep -E "^k_fast"
done
cp /tmp/liborbhip_keep1.so $LIB
