#!/usr/bin/env python3
"""The 4000 x 1 000 000 brute-force query (BASELINE config 5, orbhip_hamming_knn2_device) alone, for kernel traces and counter
passes of k_knn2_mfma:   python3 tools/knn_query.py [reps] [nq] [ndb]      prints ms per query and T pairs/s."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "vi-orb-slam-icra2018_amd"))
import torch  # noqa: E402  (device buffers only)

from orbhip.extractor import ORBextractor  # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
nq = int(sys.argv[2]) if len(sys.argv) > 2 else 4000
ndb = int(sys.argv[3]) if len(sys.argv) > 3 else 1000000
gen = torch.Generator(device="cuda")
gen.manual_seed(77)
db = torch.randint(0, 256, (ndb, 32), dtype=torch.uint8, device="cuda", generator=gen)
q = db[torch.randint(0, ndb, (nq,), device="cuda", generator=gen)].clone()
q[:, :4] ^= torch.randint(0, 256, (nq, 4), dtype=torch.uint8, device="cuda", generator=gen) & 0x11
ex = ORBextractor(1000, max_w=640, max_h=480)
bi = torch.empty(nq, dtype=torch.int32, device="cuda")
bd, sd = torch.empty_like(bi), torch.empty_like(bi)
L = ex._L
args = (ex.handle, q.data_ptr(), nq, db.data_ptr(), ndb, bi.data_ptr(), bd.data_ptr(), sd.data_ptr())
for _ in range(3):
    assert L.orbhip_hamming_knn2_device(*args) == 0
ex.sync()
t0 = time.perf_counter()
for _ in range(reps):
    L.orbhip_hamming_knn2_device(*args)
ex.sync()
dt = (time.perf_counter() - t0) / reps
print("query %d x %d: %.4f ms, %.3f T pairs/s" % (nq, ndb, dt * 1e3, nq * ndb / dt / 1e12))
ex.close()
