#!/bin/bash
# k_bow_lane / k_bow_seq phase ablation (liborbhip_ablation.so; results invalid below the last stop, timing only):
# ORBHIP_BOW_PHASES (k_bow_lane) = 0 keys, 1 + sort, 2 + index lists / items, 3 + order / offsets, 4 + byte matrices, 5 + greedy (groups, lanes), 6 + cooperative rest, 9 everything
for lane in 1 0; do
for p in 0 1 2 3 4 5 6 9; do
  ORBHIP_BOW_LANE=$lane ORBHIP_BOW_PHASES=$p python bench.py --cpu-frames 0 --pipelined 0 --verify 0 --host-batch 0 --configs 0 --content 0 --batch-sweep 0 --no-tiling 0 --steps 10 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('lane=$lane stop<=$p match_ms', d['stage_ms']['last_match_kernel'], d['value'])"
done; done
