#!/bin/bash
# Runs ON THE GPU BOX: per-call latency of the extractor for a C++ caller (tools/native/latency_dropin.cpp).
cd ${GRAFT_REPO_ROOT:-/root/repo}
python - <<'PY'
import sys
sys.path.insert(0, "vi-orb-slam-icra2018_amd")
from orbhip import synth
from orbhip import distributed as D
open("/tmp/lat_640x480.raw", "wb").write(synth.make_frames(5, 640, 480, 2).tobytes())
open("/tmp/lat_752x480.raw", "wb").write(synth.make_frames(5, 752, 480, 2).tobytes())
open("/tmp/lat_voc.bin", "wb").write(D.make_synthetic_vocabulary(52, k=10, L=6))      # the shape of ORBvoc (k 10, L 6)
PY
tools/native/latency_dropin 640 480 1000 /tmp/lat_640x480.raw ${1:-2000} /tmp/lat_voc.bin
tools/native/latency_dropin 752 480 1000 /tmp/lat_752x480.raw ${1:-2000} /tmp/lat_voc.bin
