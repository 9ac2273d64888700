#!/bin/bash
# k_describe phase ablation (results are invalid below the last stop; timing only).  Stops: 0 slot decode; 1 + IC_Angle moments
# (disc rows of the un-blurred level); 2 + angle / cos / sin / keypoint record; 3 = everything (+ the blurred patch and the 256 tests).
# Then the occupancy sweep of the whole kernel: unused dynamic LDS caps the workgroups per CU (8 = no cap).
for p in 0 1 2 3; do
  ORBHIP_DESCRIBE_PHASES=$p python bench.py --cpu-frames 0 --pipelined 0 --verify 0 --host-batch 0 --configs 0 --content 0 --batch-sweep 0 --steps 10 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('stop<=$p describe_ms', d['stage_ms']['describe'])"
done
for pad in 0 4096 8192 16384 32768; do
  ORBHIP_DESCRIBE_PADLDS=$pad python bench.py --cpu-frames 0 --pipelined 0 --verify 0 --host-batch 0 --configs 0 --content 0 --batch-sweep 0 --steps 10 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('lds_pad $pad describe_ms', d['stage_ms']['describe'], 'step_ms', d['ms_per_step'])"
done
