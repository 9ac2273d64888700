#!/usr/bin/env python3
"""k_vocab_transform_quad alone: 1024 x 1000 random-ish descriptors (the bench's stock-shape vocabulary, levelsup 4), ms per call and a checksum
of the outputs (A/B of library variants: tools/ab_transform.sh)."""
import os, sys, time, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "vi-orb-slam-icra2018_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import hiprt
from orbhip import distributed as D
from orbhip.extractor import ORBextractor
from orbhip.vocabulary import ORBVocabulary
ex = ORBextractor(1000, max_w=640, max_h=480, max_batch=8)
ORBVocabulary(ex).loadFromBinaryBlob(D.make_synthetic_vocabulary(4242, 10, 6))
n = 1024 * ex.cap
desc = np.random.default_rng(1).integers(0, 256, (n, 32), dtype=np.uint8)
d_desc = hiprt.DevBuf.from_numpy(desc)
d_w, d_wt, d_n = hiprt.DevBuf(n * 4), hiprt.DevBuf(n * 4), hiprt.DevBuf(n * 4)
L = ex._L
for _ in range(3):
    L.orbhip_vocab_transform_device(ex.handle, d_desc.ptr, n, 4, d_w.ptr, d_wt.ptr, d_n.ptr)
ex.sync()
t = time.perf_counter()
for _ in range(20):
    L.orbhip_vocab_transform_device(ex.handle, d_desc.ptr, n, 4, d_w.ptr, d_wt.ptr, d_n.ptr)
ex.sync()
ms = (time.perf_counter() - t) / 20 * 1e3
w, nd = d_w.to_numpy(np.int32, (n,)), d_n.to_numpy(np.int32, (n,))
print("transform ms per %d descriptors: %.4f  checksum %d %d" % (n, ms, int(w.astype(np.int64).sum()), int(nd.astype(np.int64).sum())))
