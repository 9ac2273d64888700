#!/bin/bash
# k_fast phase ablation per CONTENT CLASS (bench.py --content; results are invalid below the last stop: timing only).
# Stops as tools/fast_ablate.sh: 1 staging; 2 / 3 / 4 compass + list, score, suppression of pass 0; 5 the second pass of the empty cells; 8 everything.
#   tools/fast_ablate_content.sh [lib.so]        prints k_fast ms per 1024 frames per class and stop
LIB=vi-orb-slam-icra2018_amd/csrc/liborbhip_ablation.so
if [ -n "$1" ]; then cp $LIB /tmp/keep_abl_content.so; cp $1 $LIB; fi
for p in 1 2 3 4 5 8; do
  ORBHIP_FAST_PHASES=$p python bench.py --cpu-frames 0 --pipelined 0 --verify 0 --host-batch 0 --configs 0 --no-tiling 0 --batch-sweep 0 --steps 3 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); c=d['content']
print('stop<=$p headline', d['stage_ms']['fast'], {k: v['k_fast_ms_per_1024_frames'] for k, v in c.items() if isinstance(v, dict)})"
done
if [ -n "$1" ]; then cp /tmp/keep_abl_content.so $LIB; fi
