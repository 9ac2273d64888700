#!/bin/bash
# Runs ON THE GPU BOX: one rocprofv3 --pmc pass with the given counters; prints per-kernel averages.
# Usage: tools/pmc_gpu.sh <tag> "<counters>" [bench args...]
TAG=$1; CTRS=$2; shift 2
ARGS=${@:---steps 3 --warmup 1 --batch 1024 --cpu-frames 0 --pipelined 0 --host-batch 0 --configs 0 --content 0 --batch-sweep 0 --verify 0}
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/pmc_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --pmc $CTRS --output-format csv -d $OUT/run -- python3 $REPO/bench.py $ARGS > $OUT/run.log 2>&1
python3 - "$OUT" <<'PY'
import csv,glob,sys,collections
d=sys.argv[1]
acc=collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(d+'/run/**/*counter_collection.csv',recursive=True):
    for r in csv.DictReader(open(f)):
        k=r['Kernel_Name'].split('(')[0].replace('void ','')
        acc[k][r['Counter_Name']].append(float(r['Counter_Value']))
for k,v in acc.items():
    if not k.startswith('k_'): continue
    print(k, {c: round(sum(x[len(x)//2:])/max(1,len(x[len(x)//2:])),1) for c,x in v.items()})
PY
find $OUT -name '*.csv' -size +5M -delete
