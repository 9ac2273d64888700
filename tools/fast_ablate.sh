#!/bin/bash
# k_fast phase ablation (results are invalid for phases < 5; timing only)
for p in 1 2 3 4 5; do
  ORBHIP_FAST_PHASES=$p python bench.py --cpu-frames 0 --pipelined 0 --steps 10 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('phases<=$p fast_ms', d['stage_ms']['fast'])"
done
