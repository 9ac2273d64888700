#!/bin/bash
# Runs ON THE GPU BOX (via gpurun): kernel trace + stats, then HBM counters in separate passes
# (MI355X_MICROARCH.md: FETCH_SIZE and WRITE_SIZE do not fit one pass; never combine --pmc with
# other trace domains).  Usage: tools/profile_gpu.sh <tag> [bench args...]
set -u
TAG=${1:-r01}; shift || true
ARGS=${@:---steps 5 --warmup 2 --batch 1024 --cpu-frames 0 --pipelined 0 --host-batch 0 --configs 0 --content 0 --batch-sweep 0 --verify 0}
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $REPO/bench.py $ARGS > $OUT/trace.log 2>&1
timeout 600 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -- python3 $REPO/bench.py $ARGS > $OUT/fetch.log 2>&1
timeout 600 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/write -- python3 $REPO/bench.py $ARGS > $OUT/write.log 2>&1
python3 $REPO/tools/summarize_prof.py $OUT > $OUT/summary.md 2>&1
cat $OUT/summary.md
# keep only the small files (gpurun_out merge is capped at 64 MiB)
find $OUT -name '*.csv' -size +20M -delete
