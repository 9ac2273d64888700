#!/bin/bash
# Runs ON THE GPU BOX: what is installed (OpenCV?), baseline k_fast phase ablation (time and instruction counts).
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/r02_probe
mkdir -p $OUT
cd $REPO
{
  echo "== cv2 =="; python3 -c "import cv2; print(cv2.__version__, cv2.__file__)" 2>&1 | tail -1
  echo "== opencv files =="; find / -iname '*opencv*' -not -path '/proc/*' 2>/dev/null | head -20
  echo "== nproc =="; nproc; lscpu | grep 'Model name'
  echo "== rocm-smi =="; rocm-smi --showclocks 2>/dev/null | head -20
} > $OUT/env.txt 2>&1
bash tools/fast_ablate.sh > $OUT/fast_ablate_time.txt 2>&1
bash tools/fast_ablate_pmc.sh > $OUT/fast_ablate_pmc.txt 2>&1
cat $OUT/env.txt $OUT/fast_ablate_time.txt $OUT/fast_ablate_pmc.txt
