#!/bin/bash
python -m pytest tests -m gpu -x -q -k "bow or BoW or resident or dropin or vocab or shape or triang or guided or init or frame_build or pipeline" 2>&1 | tail -3
python tools/percall_latency.py 2>/dev/null | grep -E "SearchByBoW|Triangulation|Projection"
bash tools/latency_native.sh 3000 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); print({k:d[k] for k in d if k.endswith('_ms') or k=='bow_matches'})"
python bench.py --cpu-frames 0 --pipelined 0 --host-batch 0 --configs 0 --content 0 --verify 8 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['stage_ms'], d['verified_frames'])"
