#!/bin/bash
# Runs ON THE GPU BOX: the counter passes behind profiles/r02*/ (k_fast ablation in instruction counts, per-kernel
# utilisation / LDS / L2 counters).  Usage: tools/r02_counters.sh <tag>
TAG=${1:-r02x}
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p $OUT
cd $REPO
bash tools/fast_ablate.sh > $OUT/fast_ablate_time.txt 2>&1
bash tools/fast_ablate_pmc.sh > $OUT/fast_ablate_pmc.txt 2>&1
echo "stop 12 (compass items without the list)" >> $OUT/fast_ablate_pmc.txt
ORBHIP_FAST_PHASES=12 bash tools/pmc_gpu.sh ab12 "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_LDS_IDX_ACTIVE" 2>&1 | grep -E "^k_fast" >> $OUT/fast_ablate_pmc.txt
bash tools/pmc_gpu.sh util "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_BUSY_CYCLES" 2>&1 | grep -E "^k_" > $OUT/counters_sq.txt
bash tools/pmc_gpu.sh util2 "GRBM_GUI_ACTIVE SQ_WAVES SQ_INSTS_SALU SQ_INSTS_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT" 2>&1 | grep -E "^k_" > $OUT/counters_misc.txt
bash tools/pmc_gpu.sh l2 "TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum" 2>&1 | grep -E "^k_" > $OUT/counters_l2.txt
for p in 0 1 2 3; do echo "describe stop<=$p"; ORBHIP_DESCRIBE_PHASES=$p python bench.py --cpu-frames 0 --pipelined 0 --verify 0 --host-batch 0 --configs 0 --content 0 --steps 10 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('describe_ms', d['stage_ms']['describe'])"; done > $OUT/describe_ablate_time.txt 2>&1
cat $OUT/fast_ablate_pmc.txt $OUT/counters_sq.txt $OUT/counters_misc.txt $OUT/counters_l2.txt $OUT/describe_ablate_time.txt
