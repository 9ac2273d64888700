#!/bin/bash
# A/B of library variants on the content classes (bench.py --content): frames/s and k_fast ms per class, inside ONE gpurun call
#   tools/ab_content.sh rounds lib...
N=$1; shift
LIB=vi-orb-slam-icra2018_amd/csrc/liborbhip.so
cp $LIB /tmp/keep_lib_content.so
for i in $(seq $N); do for v in "$@"; do cp $v $LIB; python bench.py --cpu-frames 0 --pipelined 0 --host-batch 0 --configs 0 --batch-sweep 0 --verify 0 --steps 10 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); c=d['content']
print('$(basename $v)', d['value'], d['stage_ms']['fast'], {k:(v['value'], v['k_fast_ms_per_1024_frames']) for k,v in c.items() if isinstance(v,dict)})"; done; done
cp /tmp/keep_lib_content.so $LIB
