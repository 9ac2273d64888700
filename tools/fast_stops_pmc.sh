#!/bin/bash
# instruction counters of k_fast at the given ablation stops for several libraries:  tools/fast_stops_pmc.sh "1 12 2 3" lib1.so lib2.so
STOPS=$1; shift
LIB=vi-orb-slam-icra2018_amd/csrc/liborbhip.so
cp $LIB /tmp/liborbhip_keep2.so
for v in "$@"; do
  cp $v $LIB
  for p in $STOPS; do
    echo -n "$(basename $v) stop<=$p "
    ORBHIP_FAST_PHASES=$p bash tools/pmc_gpu.sh s_$(basename $v .so)_$p "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_LDS_IDX_ACTIVE" --steps 2 --warmup 1 --batch 1024 --cpu-frames 0 --pipelined 0 --verify 0 --host-batch 0 --configs 0 --content 0 --batch-sweep 0 2>&1 | grep -E "^k_fast"
  done
done
cp /tmp/liborbhip_keep2.so $LIB
