#!/bin/bash
OUT=gpurun_out/r04_frame2; mkdir -p $OUT
python -m pytest tests/test_frame_build.py tests/test_resident_sets.py -m gpu -x -q 2>&1 | tail -8 > $OUT/tests.txt
bash tools/latency_native.sh 3000 > $OUT/latency_native.json 2> $OUT/latency_native.err
cat $OUT/tests.txt $OUT/latency_native.json; tail -3 $OUT/latency_native.err
