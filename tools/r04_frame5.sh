#!/bin/bash
OUT=gpurun_out/r04_frame5; mkdir -p $OUT
python -m pytest tests/test_frame_build.py tests/test_resident_sets.py tests/test_gpu_dropin.py tests/test_vocabulary.py -m gpu -x -q 2>&1 | tail -25 > $OUT/tests.txt
cat $OUT/tests.txt
