#!/bin/bash
OUT=gpurun_out/r04_frame1; mkdir -p $OUT
python -m pytest tests/test_frame_build.py tests/test_resident_sets.py tests/test_vocabulary.py tests/test_gpu_dropin.py -m gpu -x -q 2>&1 | tail -15 > $OUT/tests.txt
python tools/percall_latency.py > $OUT/percall.md 2> $OUT/percall.err
cat $OUT/tests.txt $OUT/percall.md; tail -5 $OUT/percall.err
