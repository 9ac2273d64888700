#!/usr/bin/env python3
"""k_fast_fix, the divergent list hand-over: how many trips does a wave's append take under schemes that
spread a dense lane's survivors over other lanes?  Same geometry as tools/fast_bound.py (the kernel's runs, segments and thread
mapping on the bench's frames), per wave of 64 threads:
  cur   the kernel: a lane appends its own survivors -- trips = the largest count of a lane
  p2    lanes l and l + 32 exchange (v_permlane32_swap): l takes items 0-3 of both, l + 32 items 4-7 of both
  p2b   the same pair, even items / odd items
  p4    lanes l, l + 16, l + 32, l + 48: member k takes items 2k, 2k + 1 of all four
  full  perfect balance: ceil(total / 64) -- what a prefix-sum hand-over (lane j takes entries j, j + 64, ...) reaches
Printed as mean entries per wave and scheme.  Run on CPU: python tools/fast_append_model.py > profiles/r06/fast_append_model.txt"""
import math
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in ('tools', 'vi-orb-slam-icra2018_amd', 'oracle'):
    sys.path.insert(0, os.path.join(ROOT, p))
import numpy as np
import fast_bound as FB
from orbhip import synth
frames = synth.make_frames(1000, 640, 480, 3)
INI=FB.INI
tot = dict(waves=0, cur=0, p2=0, p4=0, full=0, surv=0, p2b=0)
for f in frames:
    for a in FB.levels(f):
        h, w = a.shape
        maxBX, maxBY = w - 16, h - 16
        width, height = float(maxBX - 16), float(maxBY - 16)
        nC, nR = int(width / 30), int(height / 30)
        wC, hC = math.ceil(width / nC), math.ceil(height / nR)
        c_ini, both = FB.compass(a, INI)
        nruns = (nC + 4) // 5
        base, extra = nC // nruns, nC % nruns
        for i in range(nR):
            iniY = 16 + i * hC
            if iniY >= maxBY - 3: continue
            y0, y1 = iniY + 3, min(iniY + hC + 6, maxBY) - 3
            j = 0
            for r in range(nruns):
                nc = base + (1 if r < extra else 0)
                X0 = 16 + j * wC
                x0, x1 = X0 + 3, min(16 + (j + nc) * wC + 6, maxBX) - 3
                j += nc
                DH, TW = y1 - y0, x1 - x0
                if DH <= 0 or TW <= 0: continue
                XA = X0 & ~15
                j0 = X0 + 3 - XA
                GPR = ((j0 + TW + 3) >> 2) - (j0 >> 2)
                S = max(1, 256 // GPR)
                seg = (DH + S - 1) // S
                cm = c_ini[y0:y1, x0:x1]
                jd0 = j0 & ~3
                grp = (np.arange(TW) + j0 - jd0) >> 2
                per_item = np.zeros((DH + 16, GPR), np.int32)
                for g in range(GPR):
                    sel = grp == g
                    if sel.any(): per_item[:DH, g] = cm[:, sel].sum(1)
                # lane item counts [256][8] (seg <= 8 here)
                assert seg <= 8
                L = np.zeros((256, 8), np.int32)
                for tid in range(256):
                    sidx, g = divmod(tid, GPR)
                    rs = sidx * seg
                    if rs < DH:
                        for it in range(seg):
                            if rs + it < DH: L[tid, it] = per_item[rs + it, g]
                for w0 in range(0, 256, 64):
                    W = L[w0:w0+64]
                    n = W.sum(1)
                    tot['waves'] += 1
                    tot['surv'] += int(n.sum())
                    tot['cur'] += int(n.max())
                    # pair split: lane l<32 takes items 0-3 of l and l+32; lane l+32 takes items 4-7 of both
                    lo = W[:, :4].sum(1); hi = W[:, 4:].sum(1)
                    p2 = np.concatenate([lo[:32] + lo[32:], hi[:32] + hi[32:]])
                    tot['p2'] += int(p2.max())
                    # pair split by alternate items (even items / odd items)
                    ev = W[:, 0::2].sum(1); od = W[:, 1::2].sum(1)
                    p2b = np.concatenate([ev[:32] + ev[32:], od[:32] + od[32:]])
                    tot['p2b'] += int(p2b.max())
                    # quad split: lanes l, l+16, l+32, l+48; member k takes items 2k, 2k+1 of all four
                    q = np.zeros(64, np.int32)
                    for k in range(4):
                        part = W[:, 2*k:2*k+2].sum(1)
                        q[16*k:16*k+16] = part[0:16] + part[16:32] + part[32:48] + part[48:64]
                    tot['p4'] += int(q.max())
                    tot['full'] += int(math.ceil(n.sum() / 64.0))
w = tot['waves']
print({k: round(v / w, 2) for k, v in tot.items() if k != 'waves'})
