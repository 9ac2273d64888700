#!/bin/bash
# usage: cfg5_ab.sh rounds lib...
N=$1; shift
LIB=vi-orb-slam-icra2018_amd/csrc/liborbhip.so
cp $LIB /tmp/keep_lib.so
for i in $(seq $N); do for v in "$@"; do cp $v $LIB; python bench.py --cpu-frames 0 --pipelined 0 --host-batch 0 --batch-sweep 0 --steps 5 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); c=d['configs']['5_tum_4000feat_1M_query']; print('$(basename $v)', d['value'], c['value'], c['query_ms'], c['verified'])"; done; done
cp /tmp/keep_lib.so $LIB
