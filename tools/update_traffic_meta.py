#!/usr/bin/env python3
"""Build container, after a tools/r04_profile.sh run: completes gpurun_out/prof_<tag>/traffic.json with what ties the counters
to a kernel -- the sha256 of csrc/k_fast.hip and the issue weight of its instruction mix (static count over the ISA of the
launched instance) -- and copies the summaries into profiles/<tag>/ and profiles/traffic.json (what bench.py reads).
usage: python tools/update_traffic_meta.py gpurun_out/prof_r04 [kernel-name-substring]"""
import hashlib
import json
import os
import re
import shutil
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "vi-orb-slam-icra2018_amd", "csrc")
FAST = {"v_add_u32", "v_sub_u32", "v_subrev_u32", "v_and_b32", "v_or_b32", "v_xor_b32", "v_mov_b32", "v_lshrrev_b32", "v_ashrrev_i32",
        "v_mul_f32", "v_add_f32", "v_sub_f32", "v_fma_f32", "v_fmac_f32", "v_fmamk_f32", "v_fmaak_f32", "v_bitop3_b32"}


def issue_weight(name):
    with tempfile.TemporaryDirectory() as td:
        s = os.path.join(td, "k_fast.s")
        subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-ffp-contract=off", "-S",
                               "--cuda-device-only", "-o", s, os.path.join(CSRC, "k_fast.hip")], stderr=subprocess.DEVNULL)
        text = open(s).read().split("\n")
    start = next(i for i, l in enumerate(text) if re.match(r"^_Z\w*:", l) and name in l)
    end = next(i for i in range(start, len(text)) if text[i].startswith(".Lfunc_end"))
    f = sl = 0
    for l in text[start:end]:
        m = re.match(r"\s+(v_[a-z0-9_]+)", l)
        if m:
            op = re.sub(r"_(e32|e64|sdwa|dpp)$", "", m.group(1))
            if op in FAST:
                f += 1
            else:
                sl += 1
    return round((2 * f + 4 * sl) / (4.0 * (f + sl)), 4), f, sl


def main():
    d = sys.argv[1].rstrip("/")
    name = sys.argv[2] if len(sys.argv) > 2 else "k_fast_fixILi176ELb1E"
    tag = os.path.basename(d).replace("prof_", "")
    t = json.load(open(os.path.join(d, "traffic.json")))
    t["kernel_source_sha16"] = hashlib.sha256(open(os.path.join(CSRC, "k_fast.hip"), "rb").read()).hexdigest()[:16]
    w, f, s = issue_weight(name)
    t["issue_weight"] = w
    t["issue_weight_source"] = "tools/update_traffic_meta.py: %d two-cycle and %d four-cycle vector instructions in the ISA of %s" % (f, s, name)
    json.dump(t, open(os.path.join(d, "traffic.json"), "w"), indent=1)
    dst = os.path.join(ROOT, "profiles", tag)
    os.makedirs(dst, exist_ok=True)
    for fn in os.listdir(d):
        p = os.path.join(d, fn)
        if os.path.isfile(p) and os.path.getsize(p) < 400000 and not fn.endswith(".log"):
            shutil.copy(p, os.path.join(dst, fn))
    shutil.copy(os.path.join(d, "traffic.json"), os.path.join(ROOT, "profiles", "traffic.json"))
    print(json.dumps(t, indent=1))


if __name__ == "__main__":
    main()
