#!/bin/bash
# Experiment: device timeline of one ORBextractor::operator() call WITH the host pyramid (second stream copy).
cd ${GRAFT_REPO_ROOT:-/root/repo}
bash tools/latency_native.sh 50 > /dev/null
R=$PWD
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/lp
rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d /tmp/lp -- $R/tools/native/latency_dropin 640 480 1000 /tmp/lat_640x480.raw 300 > /dev/null 2>&1
python3 - <<'PY'
import csv, glob
ev = []
for f in glob.glob('/tmp/lp/*/*kernel_trace.csv'):
    for r in csv.DictReader(open(f)):
        ev.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'][:28]))
for f in glob.glob('/tmp/lp/*/*memory_copy_trace.csv'):
    for r in csv.DictReader(open(f)):
        ev.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), 'COPY ' + r.get('Direction', r.get('Name', ''))[:22]))
ev.sort()
qi = [i for i, e in enumerate(ev) if 'quadtree' in e[2]]
k = qi[-5]          # a call near the end of the run: the pyramid-download mode
lo = k
while lo > 0 and 'describe' not in ev[lo - 1][2]: lo -= 1
hi = k
while hi < len(ev) - 1 and 'describe' not in ev[hi][2]: hi += 1
hi = min(hi + 4, len(ev) - 1)
t0 = ev[lo][0]
for s, e, n in ev[lo:hi + 1]:
    print('%-34s start %8.2f  end %8.2f  dur %7.2f' % (n, (s - t0) / 1e3, (e - t0) / 1e3, (e - s) / 1e3))
PY
