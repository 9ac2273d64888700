#!/bin/bash
# r04, first GPU call: parity of the extraction path with the working-tree library, A/B of build_ab/base.so (r03) against
# build_ab/d16.so, the frame-per-XCD mapping of k_fast_fix (ablation library, ORBHIP_FAST_XCD=4) with its fabric traffic,
# and SQ_LDS_BANK_CONFLICT per ablation stop.
OUT=gpurun_out/r04_fast1; mkdir -p $OUT
python -m pytest tests/test_gpu_parity.py tests/test_gpu_parity_more.py tests/test_pipeline.py -m gpu -x -q 2>&1 | tail -4 > $OUT/parity.txt
VERIFY=8 tools/ab_libs.sh 3 build_ab/base.so build_ab/d16.so > $OUT/ab.txt 2>&1
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
done
for p in 1 2 3 4 5 8 99; do
  echo -n "stop<=$p " >> $OUT/lds_conflict.txt
  ORBHIP_FAST_PHASES=$p bash tools/pmc_gpu.sh ldsc_$p "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VALU" --steps 2 --warmup 1 --batch 1024 --cpu-frames 0 --pipelined 0 --verify 0 --host-batch 0 --configs 0 --content 0 2>&1 | grep -E "^k_fast" >> $OUT/lds_conflict.txt
done
cat $OUT/parity.txt $OUT/ab.txt $OUT/xcd.txt $OUT/xcd_traffic.txt $OUT/lds_conflict.txt
