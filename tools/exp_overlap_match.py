#!/usr/bin/env python3
"""Experiment: the matching stage (vocabulary transform + SearchByBoW) of step n on its own context / stream, beside the
extraction of step n + 1 (two sets of output buffers).  Prints frames/s of the plain loop and of the overlapped one."""
import ctypes as C
import os
import sys
import time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "vi-orb-slam-icra2018_amd"))
import numpy as np
import torch
from orbhip import distributed as D, synth
from orbhip.extractor import ORBextractor
from orbhip.vocabulary import ORBVocabulary

W, H, NF, B, STEPS = 640, 480, 1000, 1024, 20
uniq = synth.make_frames(1000, W, H, 32)
frames = np.concatenate([uniq] * (B // 32))
d_img = torch.from_numpy(np.ascontiguousarray(frames)).cuda()
blob = D.make_synthetic_vocabulary(4242, 10, 6)
exE = ORBextractor(NF, 1.2, 8, 20, 7, max_w=W, max_h=H, max_batch=B)
exM = ORBextractor(50, 1.2, 1, 20, 7, max_w=128, max_h=128, max_batch=1)
for e in (exE, exM):
    ORBVocabulary(e).loadFromBinaryBlob(blob)
cap = exE.cap
L = exE._L
i32 = dict(dtype=torch.int32, device="cuda")


def bufs():
    b = {"kps": torch.empty((B, cap, 7), **i32), "desc": torch.empty((B, cap, 32), dtype=torch.uint8, device="cuda"),
         "cnt": torch.zeros(B, **i32), "wt": torch.empty((B, cap), dtype=torch.float32, device="cuda"), "nm": torch.zeros(B, **i32)}
    for n in ("word", "node", "m12", "m21"):
        b[n] = torch.empty((B, cap), **i32)
    return b


def extract(ex, b):
    ex.extract_batch_device(d_img.data_ptr(), B, W, H, W, H * W, b["kps"].data_ptr(), b["desc"].data_ptr(), cap, b["cnt"].data_ptr())


def match(ex, b):
    assert L.orbhip_vocab_transform_device(ex.handle, b["desc"].data_ptr(), B * cap, 4, b["word"].data_ptr(), b["wt"].data_ptr(),
                                           b["node"].data_ptr()) == 0
    assert L.orbhip_search_by_bow_seq_device(ex.handle, b["desc"].data_ptr(), b["kps"].data_ptr(), b["cnt"].data_ptr(),
                                             b["node"].data_ptr(), b["wt"].data_ptr(), None, cap, B, 1, 0, C.c_float(0.7), 1,
                                             b["m12"].data_ptr(), b["m21"].data_ptr(), b["nm"].data_ptr()) == 0


B0, B1 = bufs(), bufs()
sE = torch.cuda.ExternalStream(exE.stream())
sM = torch.cuda.ExternalStream(exM.stream())


def plain(steps, readout):
    ms = (C.c_float * 6)()
    for _ in range(steps):
        extract(exE, B0)
        match(exE, B0)
        if readout:
            L.orbhip_get_stage_times(exE.handle, ms)
    exE.sync()


def overlapped(steps, readout):
    ms = (C.c_float * 6)()
    evE = [torch.cuda.Event() for _ in range(2)]
    evM = [torch.cuda.Event() for _ in range(2)]
    for i in range(steps):
        b = (B0, B1)[i & 1]
        if i >= 2:
            sE.wait_event(evM[i & 1])          # the matcher of step i - 2 has read this buffer set
        extract(exE, b)
        evE[i & 1].record(sE)
        sM.wait_event(evE[i & 1])
        match(exM, b)
        evM[i & 1].record(sM)
        if readout:
            L.orbhip_get_stage_times(exE.handle, ms)   # synchronises the extraction stream only
    exE.sync()
    exM.sync()


for name, fn in (("plain, readout per step (bench.py's loop)", lambda: plain(STEPS, True)), ("plain, no readout", lambda: plain(STEPS, False)),
                 ("matching on its own context beside the next extraction, readout per step", lambda: overlapped(STEPS, True)),
                 ("matching on its own context beside the next extraction, no readout", lambda: overlapped(STEPS, False))):
    for rep in range(2):
        plain(3, False)
        overlapped(2, False)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        fn()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        print("%-80s %.0f frames/s  %.3f ms/step" % (name, B * STEPS / dt, dt / STEPS * 1e3))
# the overlapped results equal the plain ones
plain(1, False)
ref = {k: v.clone() for k, v in B0.items()}
overlapped(4, False)
for k in ("cnt", "nm", "m12", "m21"):
    assert torch.equal(ref[k], B1[k]) and torch.equal(ref[k], B0[k]), k
print("results equal")
