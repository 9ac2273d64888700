"""Experiment: per-step timeline of the single-frame quadtree workgroups (needs the instrumented build build_ab/lib_qtstamp.so
copied over csrc/liborbhip.so).  Prints, per level, the time between consecutive barriers with the quadtree_core.h line."""
import ctypes as C, sys
import numpy as np
sys.path.insert(0, "vi-orb-slam-icra2018_amd")
from orbhip import synth, capi
from orbhip.extractor import ORBextractor
L = capi.load()
img = synth.make_frames(5, 640, 480, 1)[0]
ex = ORBextractor(1000, max_w=640, max_h=480)
for _ in range(5):
    ex(img)
st = np.zeros((16, 512), np.uint64)
n = np.zeros(16, np.int32)
assert L.orbhip_debug_qt_stamps(st.ctypes.data_as(C.c_void_p), n.ctypes.data_as(C.c_void_p)) == 0
for l in range(int(sys.argv[1]) if len(sys.argv) > 1 else 2):
    t = (st[l, :n[l]] & np.uint64(0xFFFFFFFFFF)).astype(np.int64)
    line = (st[l, :n[l]] >> np.uint64(40)).astype(np.int64)
    print("level", l, "steps", n[l], "total us %.2f" % ((t[-1] - t[0]) / 100.0))
    agg = {}
    for i in range(1, n[l]):
        d = (t[i] - t[i - 1]) / 100.0
        agg.setdefault(int(line[i]), []).append(d)
    for k in sorted(agg):
        print("  line %4d: n %3d  sum %6.2f us  each %s" % (k, len(agg[k]), sum(agg[k]), " ".join("%.2f" % v for v in agg[k][:12])))
