"""Kernel times of one SearchForInitialization call (run under rocprofv3 --kernel-trace --stats): 200 calls, 640x480, 1000 features, window 100."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "vi-orb-slam-icra2018_amd"), os.path.join(ROOT, "oracle")]
import numpy as np
from orbhip import guided, synth
from orbhip.extractor import ORBextractor
fr = synth.make_frames(5, 640, 480, 2)
ex = ORBextractor(1000, max_w=640, max_h=480)
(k0, d0), (k1, d1) = ex(fr[0]), ex(fr[1])
gp = guided.grid_params(0, 640, 0, 480)
prev = np.stack([k0["x"], k0["y"]], 1).astype(np.float32)
for _ in range(200):
    guided.SearchForInitialization(ex, k0, d0, k1, d1, gp, prev.copy(), 100)
