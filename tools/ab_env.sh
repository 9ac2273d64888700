#!/bin/bash
# A/B of environment settings in one call: ab_env.sh rounds "ENV1=.. ENV2=.." "..." ; prints value and stage ms
N=$1; shift
for i in $(seq $N); do
  for e in "$@"; do
    env $e python bench.py --cpu-frames 0 --pipelined 0 --host-batch 0 --configs 0 --content 0 --batch-sweep 0 --verify ${VERIFY:-8} 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('[$e]', d['value'], d['stage_ms'], d['verified_frames'])"
  done
done
