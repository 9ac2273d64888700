# HBM traffic of the r01p rows' kernels: FETCH_SIZE and WRITE_SIZE in separate passes (MI355X_MICROARCH.md), per launch.
# usage: bash tools/pmc_rows.sh fuse|distinctive
ROW=${1:-fuse}
REPO=/root/repo
OUT=$REPO/gpurun_out/pmc_$ROW
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -- python3 $REPO/tools/bench_configs.py $ROW > $OUT/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/write -- python3 $REPO/tools/bench_configs.py $ROW > $OUT/write.log 2>&1
python3 - $OUT <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
for name in ("fetch", "write"):
    f = sorted(glob.glob(out + "/" + name + "/**/*counter_collection.csv", recursive=True))
    if not f:
        print(name, "no counter file"); continue
    acc = collections.defaultdict(lambda: [0.0, 0])
    for r in csv.DictReader(open(f[-1])):
        k = r["Kernel_Name"].split("(")[0]
        acc[k][0] += float(r["Counter_Value"]); acc[k][1] += 1
    for k, (v, n) in sorted(acc.items(), key=lambda kv: -kv[1][0])[:6]:
        print("%-6s %-28s launches %4d  counter/launch %.1f" % (name, k[:28], n, v / n))
PY
