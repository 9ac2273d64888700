#!/bin/bash
# k_fast phase ablation (results are invalid below the last stop; timing only).  Stops: 1 staging; 2 / 3 / 4 compass + list,
# score, suppression of pass 0 (iniThFAST); 5 / 6 / 7 the same of pass 1 (minThFAST, empty cells); 8 = everything.
for p in 1 2 3 4 5 6 7 8; do
  ORBHIP_FAST_PHASES=$p python bench.py --cpu-frames 0 --pipelined 0 --verify 0 --host-batch 0 --configs 0 --content 0 --batch-sweep 0 --steps 10 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('stop<=$p fast_ms', d['stage_ms']['fast'])"
done
