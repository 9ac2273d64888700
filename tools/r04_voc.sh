#!/bin/bash
python -m pytest tests/test_vocabulary.py tests/test_frame_build.py tests/test_bench_shapes.py -m gpu -x -q 2>&1 | tail -3
python tools/percall_latency.py 2>/dev/null | grep -E "transform|frame_build"
bash tools/latency_native.sh 3000 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); print({k:d[k] for k in d if k.endswith('_ms')})"
