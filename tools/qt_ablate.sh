#!/bin/bash
# k_quadtree by phase / level (timing ablation, INVALID results; ablation library): the quadtree stage of the bench step alone.
run() { env ORBHIP_QT_PHASES=$1 ORBHIP_DESCRIBE_FUSED_SCHED=0 python bench.py --steps 5 --cpu-frames 0 --pipelined 0 --host-batch 0 --configs 0 --content 0 --batch-sweep 0 --tiled-check 0 --verify 0 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('QT_PHASES=$1 ($2): quadtree', d['stage_ms']['quadtree'], 'ms per 1024 frames')"; }
run 0 "everything, one launch of the whole batch"
run 256 "gather only"
for k in 1 2 3 4 5 6 7 8 10; do run $((k + 1)) "gather + roots + $k passes + winners"; done
for l in 0 1 2 3 4 5 6 7; do run $(( (l + 1) << 12 )) "level $l only"; done
