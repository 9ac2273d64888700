#!/usr/bin/env python3
"""Work model of k_fast on the bench's frames (CPU, numpy): per run of <= 5 cells the number of compass survivors at
iniThFAST, the wave iterations of the score phase they need (256 threads), corners, both-polarity entries.  Uses the
oracle's pyramid only to get the level images (tools/ may use the oracle)."""
import sys, os, math
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "vi-orb-slam-icra2018_amd"))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import numpy as np
from orbhip import synth

def levels(img, n=8, sf=1.2):
    from oracle import orb_oracle_py as O
    ex = O.Extractor(1000, sf, n, 20, 7)
    ex(img)
    return [ex.pyramid(l) for l in range(n)]

def main():
    W, H = 640, 480
    frames = synth.make_frames(1000, W, H, 4)
    t = 20
    tot = dict(tiles=0, px=0, surv=0, both=0, iters=0, iters_ideal=0.0)
    for f in frames:
        for img in levels(f):
            a = img.astype(np.int32)
            h, w = a.shape
            maxBX, maxBY = w - 16, h - 16
            width, height = maxBX - 16 + 0.0, maxBY - 16 + 0.0   # (w-32+... ) the reference: maxBorder = w - EDGE + 3, min = EDGE - 3
            nC, nR = int(width / 30), int(height / 30)
            wC, hC = math.ceil(width / nC), math.ceil(height / nR)
            v = a[3:-3, 3:-3]; T = a[:-6, 3:-3]; B = a[6:, 3:-3]; L = a[3:-3, :-6]; R = a[3:-3, 6:]
            pb = np.minimum(np.maximum(T, B), np.maximum(L, R)) > v + t
            pd = np.maximum(np.minimum(T, B), np.minimum(L, R)) < v - t
            comp = np.zeros_like(a, bool); comp[3:-3, 3:-3] = pb | pd
            both = np.zeros_like(a, bool); both[3:-3, 3:-3] = pb & pd
            nruns = (nC + 4) // 5; base, extra = nC // nruns, nC % nruns
            for i in range(nR):
                y0 = 16 + i * hC + 3; y1 = min(16 + i * hC + hC + 6, maxBY) - 3
                if 16 + i * hC >= maxBY - 3: continue
                j = 0
                for r in range(nruns):
                    nc = base + (1 if r < extra else 0)
                    x0 = 16 + j * wC + 3; x1 = min(16 + (j + nc) * wC + 6, maxBX) - 3
                    j += nc
                    if x1 <= x0 or y1 <= y0: continue
                    n = int(comp[y0:y1, x0:x1].sum())
                    tot["tiles"] += 1; tot["px"] += (y1 - y0) * (x1 - x0); tot["surv"] += n
                    tot["both"] += int(both[y0:y1, x0:x1].sum())
                    tot["iters"] += (n + 63) // 64; tot["iters_ideal"] += n / 64.0
    nf = len(frames)
    print({k: v / nf for k, v in tot.items()})
    print("per 1024 frames: wave iterations %.2f M (ideal %.2f M), survivors %.1f %% of pixels, both %.1f %% of survivors" % (
        tot["iters"] / nf * 1024 / 1e6, tot["iters_ideal"] / nf * 1024 / 1e6, 100.0 * tot["surv"] / tot["px"], 100.0 * tot["both"] / max(1, tot["surv"])))

if __name__ == "__main__":
    main()
