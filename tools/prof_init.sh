cd /root/repo
python tools/bench_configs.py ${1:-init}
mkdir -p gpurun_out/init
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/gpurun_out/init -o init -- python3 /root/repo/tools/bench_configs.py ${1:-init} > /dev/null 2>&1
python3 - <<'PY'
import csv,glob
f=glob.glob('/root/repo/gpurun_out/init/**/*kernel_stats.csv',recursive=True)
for r in list(csv.DictReader(open(sorted(f)[-1])))[:12]:
    print(r['Name'][:50], r['Calls'], r['AverageNs'], r['Percentage'])
PY
