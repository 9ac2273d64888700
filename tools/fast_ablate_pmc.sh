#!/bin/bash
# k_fast phase ablation in instruction counts (results are invalid below the last stop; counting only).  Stops as in fast_ablate.sh.
for p in 1 2 3 4 5 6 7 8; do
  echo "stop<=$p"; ORBHIP_FAST_PHASES=$p bash tools/pmc_gpu.sh ab$p "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_LDS_IDX_ACTIVE" --steps 2 --warmup 1 --batch 1024 --cpu-frames 0 --pipelined 0 --verify 0 --host-batch 0 --configs 0 --content 0 --batch-sweep 0 2>&1 | grep -E "^k_fast"
done
