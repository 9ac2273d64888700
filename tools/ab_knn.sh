#!/bin/bash
# A/B of library variants on the 4000 x 1M query alone (tools/knn_query.py), inside ONE gpurun call:  tools/ab_knn.sh rounds lib...
N=$1; shift
LIB=vi-orb-slam-icra2018_amd/csrc/liborbhip.so
cp $LIB /tmp/keep_lib_knn.so
for i in $(seq $N); do for v in "$@"; do cp $v $LIB; echo "$(basename $v) $(python3 tools/knn_query.py 30 2>/dev/null | tail -1)"; done; done
cp /tmp/keep_lib_knn.so $LIB
