cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_init -- python3 $GRAFT_REPO_ROOT/tools/prof_init.py > $GRAFT_REPO_ROOT/gpurun_out/prof_init.log 2>&1
cat $GRAFT_REPO_ROOT/gpurun_out/prof_init/*/*kernel_stats.csv | cut -c1-200 | head -12
for st in 1 2 3; do ORBHIP_INIT_STOP=$st python3 $GRAFT_REPO_ROOT/tools/percall_latency.py 2>/dev/null | grep -i initiali; done
