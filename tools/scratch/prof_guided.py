import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [os.path.join(ROOT, "vi-orb-slam-icra2018_amd"), os.path.join(ROOT, "oracle")]
import numpy as np
from orbhip import guided, synth
from orbhip.capi import QUERY_DTYPE
from orbhip.extractor import ORBextractor
W, H = 640, 480
fr = synth.make_frames(5, W, H, 2)
ex = ORBextractor(1000, max_w=W, max_h=H)
(k0, d0), (k1, d1) = ex(fr[0]), ex(fr[1])
rng = np.random.default_rng(0)
gp = guided.grid_params(0, W, 0, H)
sf = (np.float32(1.2) ** np.arange(8, dtype=np.float32)).astype(np.float32)
u = k0["x"] + rng.normal(0, 2, len(k0)).astype(np.float32)
v = k0["y"] + rng.normal(0, 2, len(k0)).astype(np.float32)
q = guided.queries_for_last_frame(u, v, np.full(len(k0), -1, np.float32), k0["octave"], k0["angle"], np.ones(len(k0), bool), np.zeros(len(k0), bool), 15, sf)
sig = (1 / sf ** 2).astype(np.float32)
qf = np.zeros(len(k0), QUERY_DTYPE)
qf["u"], qf["v"] = u, v
qf["radius"] = 3 * sf[k0["octave"]]
qf["min_level"], qf["max_level"], qf["flags"] = k0["octave"] - 1, k0["octave"], 1
prev = np.stack([k0["x"], k0["y"]], 1).astype(np.float32)
for _ in range(50):
    guided.SearchByProjection(ex, k1, d1, gp, q, d0, use_ratio=False, th_high=100)
    guided.SearchForInitialization(ex, k0, d0, k1, d1, gp, prev, 100)
    guided.WindowBest(ex, k1, d1, gp, qf, d0, None, sig)
