#!/bin/bash
# r04: frame-per-XCD grid of k_fast_fix without the division (ablation library, ORBHIP_FAST_XCD=0 / 4), time and fabric traffic
OUT=gpurun_out/r04_fast2; mkdir -p $OUT; rm -f $OUT/*.txt
BA="--cpu-frames 0 --pipelined 0 --host-batch 0 --configs 0 --content 0 --verify 8"
for i in 1 2 3; do
  for x in 0 4; do
    echo -n "FAST_XCD=$x " >> $OUT/xcd.txt
    ORBHIP_ABLATION=1 ORBHIP_FAST_XCD=$x python bench.py $BA 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['stage_ms'], d['verified_frames'])" >> $OUT/xcd.txt
  done
done
for x in 0 4; do
  echo "FAST_XCD=$x" >> $OUT/xcd_traffic.txt
  ORBHIP_ABLATION=1 ORBHIP_FAST_XCD=$x bash tools/pmc_gpu.sh xcd$x "FETCH_SIZE" 2>&1 | grep -E "^k_fast" >> $OUT/xcd_traffic.txt
  ORBHIP_ABLATION=1 ORBHIP_FAST_XCD=$x bash tools/pmc_gpu.sh xcdv$x "SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAIT_ANY SQ_WAVE_CYCLES" 2>&1 | grep -E "^k_fast" >> $OUT/xcd_traffic.txt
done
cat $OUT/xcd.txt $OUT/xcd_traffic.txt
