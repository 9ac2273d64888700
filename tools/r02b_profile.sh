#!/bin/bash
# Runs ON THE GPU BOX: everything behind profiles/r02b (one call, one box): the bench line, rocprofv3 kernel stats + HBM
# counter passes of the same command, k_fast phase ablation (time + instruction counts), utilisation counters, the other
# BASELINE configurations, and the per-call latencies of a C++ caller.   usage: bash tools/r02b_profile.sh [tag]
TAG=${1:-r02b}
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p $OUT
cd $REPO
python bench.py > $OUT/bench_stdout.txt 2> $OUT/bench_stderr.txt
tail -1 $OUT/bench_stdout.txt > $OUT/bench.json
bash tools/profile_gpu.sh $TAG > /dev/null 2>&1
bash tools/r02_counters.sh $TAG > /dev/null 2>&1
python tools/bench_configs.py > $OUT/configs.md 2> $OUT/configs_stderr.txt
bash tools/latency_native.sh 3000 > $OUT/latency_native.json 2>&1
bash tools/prof_native_gaps.sh > $OUT/single_frame_timeline.txt 2>&1
{ nproc; lscpu | grep 'Model name'; rocm-smi --showclocks 2>/dev/null | head -12; } > $OUT/gpu_box_env.txt 2>&1
cp $OUT/trace/*/*kernel_stats.csv $OUT/kernel_stats.csv 2>/dev/null
find $OUT -name '*.csv' -size +2M -delete
find $OUT -name '*.db' -delete
cat $OUT/bench.json; cat $OUT/summary.md | head -12; cat $OUT/latency_native.json
