#!/bin/bash
# Runs ON THE GPU BOX: everything behind profiles/r06 (one call, one box): the bench line, rocprofv3 kernel stats + HBM counter
# passes of the same command, SQ / GRBM / LDS counter passes, k_fast phase ablation (time, instructions, LDS bank conflicts),
# per-call latencies (Python wrappers incl. the launch + sync floor, and a C++ caller incl. orbhip_frame_build).
# usage: bash tools/r06_profile.sh [tag]          then, in the build container: python tools/update_traffic_meta.py gpurun_out/prof_<tag>
TAG=${1:-r06}
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p $OUT
cd $REPO
python bench.py > $OUT/bench_stdout.txt 2> $OUT/bench_stderr.txt
tail -1 $OUT/bench_stdout.txt > $OUT/bench.json
bash tools/pmc_gpu.sh util "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_BUSY_CYCLES" 2>&1 | grep -E "^k_" > $OUT/counters_sq.txt
bash tools/pmc_gpu.sh util2 "GRBM_GUI_ACTIVE SQ_WAVES SQ_INSTS_SALU SQ_INSTS_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU_MFMA_MOPS_I8 SQ_VALU_MFMA_BUSY_CYCLES" 2>&1 | grep -E "^k_" > $OUT/counters_misc.txt
python3 - "$OUT" <<'PY'
import ast, json, sys
d = sys.argv[1]
def rows(f):
    out = {}
    for l in open(f):
        k, _, v = l.partition(" {")
        out[k.strip()] = ast.literal_eval("{" + v)
    return out
sq, misc = rows(d + "/counters_sq.txt"), rows(d + "/counters_misc.txt")
fast = [k for k in sq if k.startswith("k_fast")][0]
json.dump({"valu_wave_insts_per_launch": int(sq[fast]["SQ_INSTS_VALU"]),
           "grbm_gui_active_per_launch": int(misc[fast]["GRBM_GUI_ACTIVE"]),
           "lds_idx_active_per_launch": int(misc[fast]["SQ_LDS_IDX_ACTIVE"]),
           "lds_bank_conflict_per_launch": int(misc[fast]["SQ_LDS_BANK_CONFLICT"]),
           "valu_source": "profiles/%s/counters_sq.txt (rocprofv3 --pmc SQ_INSTS_VALU ..., own pass)" % d.rsplit("prof_", 1)[-1]},
          open(d + "/fast_valu.json", "w"), indent=1)
PY
bash tools/profile_gpu.sh $TAG > /dev/null 2>&1
python3 - "$OUT" <<'PY'
import json, sys
d = sys.argv[1]
t = json.load(open(d + "/traffic.json"))
if "grbm_gui_active_per_launch" in t and t.get("avg_launch_us"):
    t["clock_ghz"] = round(t["grbm_gui_active_per_launch"] / 8.0 / (t["avg_launch_us"] * 1e-6) / 1e9, 3)
json.dump(t, open(d + "/traffic.json", "w"), indent=1)
PY
bash tools/fast_ablate.sh > $OUT/fast_ablate_time.txt 2>&1
bash tools/fast_ablate_content.sh > $OUT/fast_ablate_content.txt 2>&1      # the same stops per content class (photographs: score + suppression)
bash tools/bow_ablate.sh > $OUT/bow_ablate_time.txt 2>&1                   # k_bow_lane / k_bow_seq by phase
# k_describe_blur by phase (stops: 0 slot decode; 1 + disc loads and moments; 2 + angles; 3 + staging and blur; 4 everything; 5 = everything
# but the disc loads), the kernel alone on the device (quadtree in front of it) and in the default schedule; then by occupancy
for sc in 0 1; do for p in 0 1 2 3 4 5; do
  ORBHIP_DESCRIBE_FUSED_SCHED=$sc ORBHIP_DESCRIBE_PHASES=$p python bench.py --cpu-frames 0 --pipelined 0 --verify 0 --host-batch 0 --configs 0 --content 0 --batch-sweep 0 --tiled-check 0 --steps 10 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('schedule $sc stop<=$p describe_ms', d['stage_ms']['describe'], 'quadtree_ms', d['stage_ms']['quadtree'], 'frames/s', d['value'])"
done; done > $OUT/describe_blur_ablate.txt 2>&1
for pad in 0 8192 16384 28000; do
  ORBHIP_DESCRIBE_FUSED_SCHED=0 ORBHIP_DESCRIBE_PADLDS=$pad python bench.py --cpu-frames 0 --pipelined 0 --verify 0 --host-batch 0 --configs 0 --content 0 --batch-sweep 0 --tiled-check 0 --steps 10 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('unused LDS $pad describe_ms (alone)', d['stage_ms']['describe'])"
done >> $OUT/describe_blur_ablate.txt 2>&1
# the two-kernel path of rounds 1-5 (k_blur + k_describe) in the same call, for the comparison
ORBHIP_DESCRIBE_FUSED=0 python bench.py --cpu-frames 0 --pipelined 0 --verify -1 --host-batch 0 --configs 0 --content 0 --batch-sweep 0 --tiled-check 0 --steps 10 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('k_blur + k_describe (ORBHIP_DESCRIBE_FUSED=0):', d['value'], 'frames/s', d['stage_ms'], 'verified', d['verified_frames'])" >> $OUT/describe_blur_ablate.txt 2>&1
for p in 1 2 3 4 5 8; do
  echo -n "stop<=$p "; ORBHIP_FAST_PHASES=$p bash tools/pmc_gpu.sh ab$p "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT" --steps 2 --warmup 1 --batch 1024 --cpu-frames 0 --pipelined 0 --verify 0 --host-batch 0 --configs 0 --content 0 --batch-sweep 0 2>&1 | grep -E "^k_fast"
done > $OUT/fast_ablate_pmc.txt 2>&1
timeout 300 bash tools/pmc_gpu.sh l2 "TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum" 2>&1 | grep -E "^k_" > $OUT/counters_l2.txt
python tools/percall_latency.py > $OUT/percall_table.md 2> $OUT/percall_stderr.txt
bash tools/latency_native.sh 3000 > $OUT/latency_native.json 2>&1
bash tools/knn_pmc.sh $TAG > $OUT/knn2_counters.txt 2>&1      # the 4000 x 1M query alone: kernel trace + two counter passes
# the three-thread matcher test's own report (tests/native/test_threads_dropin.cpp)
python - > $OUT/threads_dropin.txt 2>&1 <<'PY'
import os, subprocess, sys, tempfile
import numpy as np
sys.path.insert(0, "vi-orb-slam-icra2018_amd")
from orbhip import distributed as D, synth
with tempfile.TemporaryDirectory() as td:
    fr = synth.make_frames(91, 640, 480, 3)
    open(os.path.join(td, "f.raw"), "wb").write(np.ascontiguousarray(fr).tobytes())
    open(os.path.join(td, "v.bin"), "wb").write(D.make_synthetic_vocabulary(17, k=10, L=5))
    for ns in ("0", "1"):
        r = subprocess.run(["tests/native/test_threads_dropin", "640", "480", "1200", os.path.join(td, "f.raw"), "3", os.path.join(td, "v.bin"), "500"],
                           capture_output=True, text=True, env=dict(os.environ, ORBHIP_NO_SETS=ns))
        print("ORBHIP_NO_SETS=" + ns, "rc", r.returncode)
        print(r.stdout)
PY
{ nproc; lscpu | grep 'Model name'; lscpu | grep -i numa; rocm-smi --showclocks 2>/dev/null | head -12; } > $OUT/gpu_box_env.txt 2>&1
cp $OUT/trace/*/*kernel_stats.csv $OUT/kernel_stats.csv 2>/dev/null
find $OUT -name '*.csv' -size +2M -delete
find $OUT -name '*.db' -delete
cat $OUT/bench.json | head -c 1200; echo; cat $OUT/summary.md | head -14
