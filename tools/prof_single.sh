#!/bin/bash
# GPU timeline of single-frame calls (kernel + memory-copy trace): where the 0.2 ms of orbhip_extract go
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d /root/repo/gpurun_out/single -o single -- python3 /root/repo/tools/bench_configs.py single > /dev/null 2>&1
python3 - <<'PY'
import csv,glob
k=sorted(glob.glob('/root/repo/gpurun_out/single/**/*kernel_trace.csv',recursive=True))[-1]
m=sorted(glob.glob('/root/repo/gpurun_out/single/**/*memory_copy_trace.csv',recursive=True))[-1]
ev=[]
for r in csv.DictReader(open(k)):
    ev.append((int(r['Start_Timestamp']),int(r['End_Timestamp']),r['Kernel_Name'].split('(')[0][:28]))
for r in csv.DictReader(open(m)):
    ev.append((int(r['Start_Timestamp']),int(r['End_Timestamp']),'COPY '+r.get('Direction','')[:20]))
ev.sort()
# last call: find the last 'k_describe' and print the ~16 events before/after
idx=[i for i,e in enumerate(ev) if 'k_describe' in e[2]][-3]
t0=ev[idx-12][0]
for s,e,n in ev[idx-12:idx+6]:
    print("%9.1f us  +%7.1f  %s"%((s-t0)/1e3,(e-s)/1e3,n))
PY
