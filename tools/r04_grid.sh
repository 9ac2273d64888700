#!/bin/bash
python -m pytest tests/test_guided.py tests/test_frame_build.py tests/test_resident_sets.py tests/test_init_search.py tests/test_window_best.py tests/test_gpu_dropin.py -m gpu -x -q 2>&1 | tail -3
python tools/percall_latency.py 2>/dev/null | grep -E "Initialization|AssignFeatures|frame_build|Fuse"
cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_grid -- python3 $GRAFT_REPO_ROOT/tools/prof_init.py > /dev/null 2>&1; python3 -c "
import csv,glob
for f in glob.glob('$GRAFT_REPO_ROOT/gpurun_out/prof_grid/*/*kernel_stats.csv'):
    for r in list(csv.DictReader(open(f)))[:4]: print(r['Name'][:30], r['Calls'], r['AverageNs'])
"
