#!/usr/bin/env python3
"""Batched extraction alone (device-resident frames) at a given size / feature count: ms per 1024 frames.  For the density switch of
k_describe_blur: run with ORBHIP_DESCRIBE_FUSED=2 (always fused) and =0 (k_blur + k_describe).  usage: time_extract.py w h nfeatures [B]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "vi-orb-slam-icra2018_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import hiprt
from orbhip import synth
from orbhip.extractor import ORBextractor
w, h, nf = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
B = int(sys.argv[4]) if len(sys.argv) > 4 else 256
u = synth.make_frames(1000, w, h, 16)
frames = np.concatenate([u] * (B // 16))
ex = ORBextractor(nf, max_w=w, max_h=h, max_batch=B)
cap = ex.cap
d_img = hiprt.DevBuf.from_numpy(frames)
d_k, d_d, d_c = hiprt.DevBuf(B * cap * 28), hiprt.DevBuf(B * cap * 32), hiprt.DevBuf(B * 4)
for _ in range(3):
    ex.extract_batch_device(d_img.ptr, B, w, h, w, h * w, d_k.ptr, d_d.ptr, cap, d_c.ptr)
ex.sync()
t = time.perf_counter()
for _ in range(10):
    ex.extract_batch_device(d_img.ptr, B, w, h, w, h * w, d_k.ptr, d_d.ptr, cap, d_c.ptr)
ex.sync()
ms = (time.perf_counter() - t) / 10 * 1e3 * 1024 / B
cnt = d_c.to_numpy(np.int32, (B,))
px = sum(a * b for a, b in (ex.level_size(w, h, l) for l in range(8)))
print("%dx%d nf %d: %.3f ms per 1024 frames, %.0f keypoints per frame, %.2f per 1000 pyramid pixels (ORBHIP_DESCRIBE_FUSED=%s)"
      % (w, h, nf, ms, cnt.mean(), cnt.mean() / px * 1000, os.environ.get("ORBHIP_DESCRIBE_FUSED", "default")))
