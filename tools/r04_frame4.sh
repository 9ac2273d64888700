#!/bin/bash
OUT=gpurun_out/r04_frame4; mkdir -p $OUT
for i in 1 2 3; do bash tools/latency_native.sh 3000 2>/dev/null | head -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print({k:d[k] for k in d if k.endswith('_ms')})"; done | tee $OUT/lat.txt
