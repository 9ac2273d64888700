#!/bin/bash
python -m pytest tests -m gpu -x -q -k "triang or bow or BoW or dropin or resident or soak" > gpurun_out/tri_tests.txt 2>&1; grep -E "passed|failed" gpurun_out/tri_tests.txt | tail -2
python tools/percall_latency.py 2>/dev/null | grep -E "SearchByBoW|Triangulation"
timeout 150 python tools/soak_parity.py 100 991 2>&1 | tail -1
timeout 100 python tools/soak_parity.py 60 992 match 2>&1 | tail -1
