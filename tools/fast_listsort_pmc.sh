#!/bin/bash
# r05 review item 1a, measured: the ring gather of k_fast_fix's score phase on the work list as the compass phase leaves it (stops 2, 3)
# and on the same list regrouped by row (stops 13, 14: ablation-only counting sort, k_fast.hip).  Per stop: launch time and LDS counters;
# score phase = (3 - 2) resp. (14 - 13).
for p in 2 3 13 14; do
  echo -n "stop $p: "
  ORBHIP_FAST_PHASES=$p python bench.py --cpu-frames 0 --pipelined 0 --verify 0 --host-batch 0 --configs 0 --content 0 --batch-sweep 0 --no-tiling 0 --steps 10 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('fast_ms', d['stage_ms']['fast'], end='  ')"
  ORBHIP_FAST_PHASES=$p bash tools/pmc_gpu.sh sort_$p "SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT" --steps 2 --warmup 1 --batch 1024 --cpu-frames 0 --pipelined 0 --verify 0 --host-batch 0 --configs 0 --content 0 --batch-sweep 0 --no-tiling 0 2>&1 | grep -E "^k_fast"
done
