#!/bin/bash
python -m pytest tests/test_init_search.py tests/test_gpu_dropin.py tests/test_guided.py tests/test_triangulation.py tests/test_window_best.py tests/test_resident_sets.py -m gpu -x -q 2>&1 | tail -4
python tools/percall_latency.py 2>/dev/null | grep -E "Initialization|SearchByProjection|Triangulation|floor|empty"
