#!/bin/bash
# k_fast time against workgroups per CU (dynamic LDS padded by ORBHIP_FAST_LDS_TOTAL):  tools/fast_occupancy.sh lib.so
LIB=vi-orb-slam-icra2018_amd/csrc/liborbhip.so
cp $LIB /tmp/liborbhip_keep3.so; cp $1 $LIB
for t in 0 20000 22000 26000 31000 39000 52000 65000; do
  ORBHIP_FAST_LDS_TOTAL=$t python bench.py --cpu-frames 0 --pipelined 0 --verify 0 --host-batch 0 --configs 0 --content 0 --steps 10 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('lds_total $t fast_ms', d['stage_ms']['fast'])"
done
cp /tmp/liborbhip_keep3.so $LIB
