#!/usr/bin/env python3
"""Where does the device blur differ from the oracle's?  Prints, per level, the number of differing pixels and their
pattern by (x mod 32, y mod 58) -- the strip / tile coordinates of k_blur's MFMA decomposition."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "vi-orb-slam-icra2018_amd"))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import orb_oracle_py as oracle  # noqa: E402
from orbhip import synth  # noqa: E402
from orbhip.extractor import ORBextractor  # noqa: E402

W, H = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (640, 480)
f = synth.make_frames(5, W, H, 1)[0]
ex = ORBextractor(1000, max_w=W, max_h=H, max_batch=1)
ref = oracle.Extractor(1000)
ex(f)
ref(f)
tot = 0
for l in range(8):
    a = ex.blurred(l).astype(np.int32)
    b = ref.blurred(l).astype(np.int32)
    d = a != b
    tot += int(d.sum())
    print("level %d %s: %d differ, max |diff| %d" % (l, a.shape, int(d.sum()), int(np.abs(a - b).max())))
    if d.any():
        ys, xs = np.nonzero(d)
        print("   first:", [(int(y), int(x), int(a[y, x]), int(b[y, x])) for y, x in list(zip(ys, xs))[:8]])
        print("   x mod 32 histogram:", np.bincount(xs % 32, minlength=32).tolist())
        print("   y mod 58 histogram:", np.bincount(ys % 58, minlength=58).tolist())
        print("   x mod 4 :", np.bincount(xs % 4, minlength=4).tolist(), " x >= w - w%4:", int((xs >= a.shape[1] - a.shape[1] % 4).sum()))
print("TOTAL differing pixels:", tot)
