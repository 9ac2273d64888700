#!/bin/bash
# Runs ON THE GPU BOX: kernel trace + two counter passes of the 4000 x 1M query (tools/knn_query.py) -> gpurun_out/knn_pmc_<tag>/
TAG=${1:-cur}
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/knn_pmc_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
python3 $REPO/tools/knn_query.py 20 > $OUT/plain.txt 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $REPO/tools/knn_query.py 20 > $OUT/trace.log 2>&1
P1="GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_I8 SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY"
P2="SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_WAVES"
i=0
for P in "$P1" "$P2"; do
  i=$((i+1))
  rocprofv3 --pmc $P --output-format csv -d $OUT/p$i -- python3 $REPO/tools/knn_query.py 6 > $OUT/p$i.log 2>&1
done
python3 - "$OUT" <<'PY'
import csv,glob,sys,collections
d=sys.argv[1]
for f in glob.glob(d+'/trace/**/*kernel_stats.csv',recursive=True):
    for r in csv.DictReader(open(f)):
        if 'knn2' in r['Name']: print(r['Name'].split('(')[0], r['Calls'], 'avg_us', float(r['AverageNs'])/1e3, 'min_us', float(r['MinNs'])/1e3)
acc=collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(d+'/p*/**/*counter_collection.csv',recursive=True):
    for r in csv.DictReader(open(f)):
        k=r['Kernel_Name'].split('(')[0].replace('void ','')
        acc[k][r['Counter_Name']].append(float(r['Counter_Value']))
for k,v in acc.items():
    if 'knn2' in k: print(k, {c: round(sum(x)/len(x),1) for c,x in v.items()})
PY
cat $OUT/plain.txt | tail -1
find $OUT -name '*.csv' -size +5M -delete
