#!/bin/bash
# One ablation-library knob swept inside ONE gpurun call: tools/sweep_knob.sh rounds KNOB v1 v2 ...  (value and stage times per setting)
N=$1; K=$2; shift 2
for i in $(seq $N); do
  for v in "$@"; do
    env ORBHIP_$K=$v python bench.py --cpu-frames 0 --pipelined 0 --host-batch 0 --configs 0 --content 0 --batch-sweep 0 --tiled-check 0 --verify ${VERIFY:-0} 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$K=$v', d['value'], d['stage_ms'], d['verified_frames'])"
  done
done
