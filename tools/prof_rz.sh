cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d /root/repo/gpurun_out/rz -o rz -- python3 /root/repo/bench.py --steps 3 --warmup 1 --cpu-frames 0 --pipelined 0 > /dev/null 2>&1
python3 - <<'PY'
import csv,glob
f=sorted(glob.glob('/root/repo/gpurun_out/rz/**/*kernel_trace.csv',recursive=True))[-1]
rows=[r for r in csv.DictReader(open(f)) if 'k_resize' in r['Kernel_Name']]
rows.sort(key=lambda r:int(r['Start_Timestamp']))
for r in rows[-7:]:
    print(r['Kernel_Name'][:30], r['Grid_Size_X'] if 'Grid_Size_X' in r else '', (int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3,'us')
PY
