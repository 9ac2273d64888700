#!/usr/bin/env python3
"""Stand-alone timing of the rectification kernel (orbhip_remap_device): B frames 752x480 with the EuRoC maps."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "vi-orb-slam-icra2018_amd"))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from orbhip import rectify as RC, synth  # noqa: E402
from orbhip.extractor import ORBextractor  # noqa: E402
from bench_configs import EUROC_D, EUROC_K, EUROC_P, EUROC_R  # noqa: E402

B, W, H = 128, 752, 480
ex = ORBextractor(500, max_w=320, max_h=240)
mx, my = RC.initUndistortRectifyMap(EUROC_K, EUROC_D, EUROC_R, EUROC_P, W, H)
rect = RC.Rectifier(ex, mx, my)
frames = synth.make_frames(1, W, H, 4)
src = torch.from_numpy(np.concatenate([frames] * (B // 4))).cuda()
dst = torch.empty_like(src)
torch.cuda.synchronize()
for _ in range(3):
    rect.remap_device(src.data_ptr(), B, W, H, W, W * H, dst.data_ptr(), W, W * H)
ex.sync()
n = 50
t0 = time.perf_counter()
for _ in range(n):
    rect.remap_device(src.data_ptr(), B, W, H, W, W * H, dst.data_ptr(), W, W * H)
ex.sync()
dt = (time.perf_counter() - t0) / n
print("remap %d x %dx%d: %.1f us per launch, %.2f us per frame, %.0f GB/s (1 B read + 1 B written per pixel)" %
      (B, W, H, dt * 1e6, dt * 1e6 / B, 2 * B * W * H / dt / 1e9))
