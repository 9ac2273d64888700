#!/bin/bash
# A/B/C... of prebuilt libraries inside ONE gpurun call (boxes differ by more than the gains being measured):
#   tools/ab_libs.sh rounds lib1.so lib2.so ...     prints value and stage times per library per round
N=$1; shift
LIB=vi-orb-slam-icra2018_amd/csrc/liborbhip.so
cp $LIB /tmp/liborbhip_keep.so
for i in $(seq $N); do
  for v in "$@"; do
    cp $v $LIB
    python bench.py --cpu-frames 0 --pipelined 0 --host-batch 0 --configs 0 --content 0 --batch-sweep 0 --verify ${VERIFY:-0} 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$(basename $v)', d['value'], d['stage_ms'], d['verified_frames'])"
  done
done
cp /tmp/liborbhip_keep.so $LIB
