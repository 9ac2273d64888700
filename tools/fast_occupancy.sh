#!/bin/bash
# k_fast_fix against the number of resident workgroups per CU: unused dynamic LDS (ORBHIP_FAST_LDS_PAD bytes, ablation library)
# caps the residency -- the kernel's own ~17.3 KB allow 8 per CU; +3 KB -> 7, +6 KB -> 6, +10 KB -> 5, +16 KB -> 4.
for pad in 0 3072 6144 10240 16384; do
  ORBHIP_FAST_LDS_PAD=$pad python bench.py --cpu-frames 0 --pipelined 0 --verify 0 --host-batch 0 --configs 0 --content 0 --batch-sweep 0 --steps 10 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('lds_pad $pad fast_ms', d['stage_ms']['fast'])"
done
