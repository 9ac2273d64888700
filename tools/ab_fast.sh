#!/bin/bash
# One GPU call: parity tests of the extraction path with the LAST library given, then A/B rounds of all of them.
#   tools/ab_fast.sh rounds lib1.so lib2.so ...
N=$1; shift
LIB=vi-orb-slam-icra2018_amd/csrc/liborbhip.so
cp $LIB /tmp/liborbhip_keep0.so
for last; do :; done
cp $last $LIB
python -m pytest tests/test_gpu_parity.py tests/test_gpu_parity_more.py tests/test_pipeline.py -m gpu -x -q 2>&1 | tail -4
cp /tmp/liborbhip_keep0.so $LIB
VERIFY=8 tools/ab_libs.sh $N "$@"
