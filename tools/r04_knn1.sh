#!/bin/bash
OUT=gpurun_out/r04_knn1; mkdir -p $OUT
cp build_ab/knn_c1l4.so vi-orb-slam-icra2018_amd/csrc/liborbhip.so
python -m pytest tests/test_gpu_parity_more.py tests/test_bench_shapes.py -m gpu -x -q -k "knn2 or brute or hamming or shape" 2>&1 | tail -4 > $OUT/tests.txt
bash tools/ab_config5.sh 2 build_ab/knn_head.so build_ab/knn_c1l4.so build_ab/knn_c2l4.so build_ab/knn_c1l3.so build_ab/knn_c2l3.so > $OUT/ab.txt 2>&1
cat $OUT/tests.txt $OUT/ab.txt
