#!/bin/bash
# Runs ON THE GPU BOX: kernel durations of the single-frame path as a C++ caller drives it (tools/native/latency_dropin).
cd ${GRAFT_REPO_ROOT:-/root/repo}
bash tools/latency_native.sh 100 > /dev/null
R=$PWD
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/lp
rocprofv3 --kernel-trace --memory-copy-trace --stats --output-format csv -d /tmp/lp -- $R/tools/native/latency_dropin 640 480 1000 /tmp/lat_640x480.raw ${1:-400} /tmp/lat_voc.bin > /dev/null 2>&1
python3 -c "
import csv,glob
for f in glob.glob('/tmp/lp/*/*kernel_stats.csv')+glob.glob('/tmp/lp/*/*memory_copy_stats.csv'):
    for r in csv.DictReader(open(f)):
        print(r['Name'][:44].ljust(44), r['Calls'].rjust(6), 'avg_us %.2f' % (float(r['AverageNs'])/1e3), 'min %.2f' % (float(r['MinNs'])/1e3), 'max %.2f' % (float(r['MaxNs'])/1e3))
"
