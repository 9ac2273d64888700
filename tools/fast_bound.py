#!/usr/bin/env python3
"""Work model and instruction bound of k_fast_fix (build container, numpy; tools/ may use the oracle for the level images).

For the frames of a content class it counts, run by run with the kernel's own geometry (orb_geometry.hip), everything the
kernel's phases are proportional to -- staged bytes, compass items, compass survivors at iniThFAST, corners, suppression
survivors, cells that stay empty, the second pass's items / survivors / corners at minThFAST -- and from the layout of the
threads over a run: the trips of the divergent list append (max survivors per lane over a wave) for the kernel's mapping
(a lane owns consecutive rows) and for rows strided over the segments, and the LDS bank conflicts of the ring gather for the
list orders under discussion (MI355X_MICROARCH.md section LDS: ds_read_u8 banks like ds_read_b32, two groups of 32 lanes, bank =
(address / 4) mod 32, same dword broadcasts).  Then the bound: per phase the fewest vector instructions per unit this
formulation can take (the sequence is named), times the units, priced at the measured issue cost of each opcode class
(profiles/r03/valu_rates.txt: 2.4 cycles for add / and / or / xor / shifts right / mov / bitop3, 4.3 for everything else) on 1024
SIMDs at 2.4 GHz -> milliseconds per 1024 frames -> fraction of the 8 TB/s HBM roofline at 793 732 algorithmic bytes per frame.

  python tools/fast_bound.py [--classes textured,photographs] [--frames 4] [--out profiles/r05/fast_bound.md]
"""
import argparse
import math
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "vi-orb-slam-icra2018_amd"))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np  # noqa: E402
from orbhip import synth  # noqa: E402

W, H = 640, 480
INI, MIN = 20, 7
RING = [(0, 3), (1, 3), (2, 2), (3, 1), (3, 0), (3, -1), (2, -2), (1, -3), (0, -3), (-1, -3), (-2, -2), (-3, -1), (-3, 0), (-3, 1), (-2, 2), (-1, 3)]  # (dx, dy)
CHEAP, SLOW = 2.4, 4.3            # cycles per wave-instruction and SIMD (valu_rates.txt, 8 waves per SIMD)


def levels(img):
    import orb_oracle_py as O
    ex = O.Extractor(1000, 1.2, 8, INI, MIN)
    ex(img)
    return [ex.pyramid(l) for l in range(8)]


def score_map(a):
    """FAST-9/16 score (largest threshold at which the pixel is a corner; 0 = not even at 1) of every pixel >= 3 from the border."""
    a = a.astype(np.int16)
    h, w = a.shape
    v = a[3:h - 3, 3:w - 3]
    d = np.stack([a[3 + dy:h - 3 + dy, 3 + dx:w - 3 + dx] - v for dx, dy in RING])          # (16, h-6, w-6)
    best_b = np.zeros_like(v)
    best_d = np.zeros_like(v)
    for s in range(16):
        idx = [(s + j) % 16 for j in range(9)]
        best_b = np.maximum(best_b, d[idx].min(0))
        best_d = np.maximum(best_d, (-d[idx]).min(0))
    sc = np.maximum(best_b, best_d) - 1
    out = np.zeros((h, w), np.int16)
    out[3:h - 3, 3:w - 3] = np.maximum(sc, 0)
    return out


def compass(a, t, loose_dark=True):
    a = a.astype(np.int16)
    h, w = a.shape
    v = a[3:-3, 3:-3]; T = a[:-6, 3:-3]; B = a[6:, 3:-3]; L = a[3:-3, :-6]; R = a[3:-3, 6:]
    pb = np.minimum(np.maximum(T, B), np.maximum(L, R)) > v + t
    td = t - 1 if loose_dark else t      # the kernel's shared halving admits q - v = -t as well
    pd = np.maximum(np.minimum(T, B), np.minimum(L, R)) < v - td
    out = np.zeros((h, w), bool)
    out[3:-3, 3:-3] = pb | pd
    both = np.zeros((h, w), bool)
    both[3:-3, 3:-3] = pb & pd
    return out, both


def nms_cellwise(sc, t, x0, y0, x1, y1):
    """strict 8-neighbour maximum inside the cell's domain, among scores >= t"""
    s = np.where(sc[y0:y1, x0:x1] >= t, sc[y0:y1, x0:x1], 0).astype(np.int32)
    p = np.pad(s, 1)
    m = np.zeros_like(s)
    for dy in (0, 1, 2):
        for dx in (0, 1, 2):
            if dy == 1 and dx == 1:
                continue
            m = np.maximum(m, p[dy:dy + s.shape[0], dx:dx + s.shape[1]])
    return (s > 0) & (s > m)


def bank_conflict_cycles(addr_rows, addr_cols, pitch):
    """LDS cycles of ONE ds_read_u8 wave-instruction (two 32-lane groups) for the lanes' byte addresses row * pitch + col: per group
    the largest number of distinct dwords on one of the 32 banks."""
    cyc = 0
    a = (np.asarray(addr_rows) * pitch + np.asarray(addr_cols)) >> 2
    for g in (a[:32], a[32:]):
        if len(g) == 0:
            continue
        u = np.unique(g)
        cnt = np.bincount(u % 32, minlength=32)
        cyc += max(1, int(cnt.max()))
    return cyc


def gather_cycles(rows, cols, pitch):
    """all 17 loads of the ring gather for a list of entries taken 64 at a time (256 threads: entry e -> thread e % 256)"""
    n = len(rows)
    total = ideal = 0
    for w0 in range(0, n, 64):
        r, c = rows[w0:w0 + 64], cols[w0:w0 + 64]
        for dx, dy in RING + [(0, 0)]:
            total += bank_conflict_cycles(r + 3 + dy, c + 3 + dx, pitch)
            ideal += 1 + (len(r) > 32)
    return total, ideal


def analyse(frames, want_lds=True):
    T = dict(px=0, staged=0, runs=0, items=0, threads_busy=0.0, surv=0, both=0, corners=0, nms=0, cells=0, empty=0, px2=0, items2=0, surv2=0,
             corners2=0, nms2=0, trips_seg=0, trips_str=0, waves=0, overflow=0, lds={}, surv_hist=[])
    orders = ("kernel", "row_major", "tile8x4")
    for o in orders:
        T["lds"][o] = [0, 0]
    for f in frames:
        for img in levels(f):
            a = img
            h, w = a.shape
            maxBX, maxBY = w - 16, h - 16
            width, height = float(maxBX - 16), float(maxBY - 16)
            nC, nR = int(width / 30), int(height / 30)
            wC, hC = math.ceil(width / nC), math.ceil(height / nR)
            sc = score_map(a)
            c_ini, both = compass(a, INI)
            c_min, _ = compass(a, MIN)
            nruns = (nC + 4) // 5
            base, extra = nC // nruns, nC % nruns
            for i in range(nR):
                iniY = 16 + i * hC
                if iniY >= maxBY - 3:
                    continue
                y0, y1 = iniY + 3, min(iniY + hC + 6, maxBY) - 3
                j = 0
                for r in range(nruns):
                    nc = base + (1 if r < extra else 0)
                    X0 = 16 + j * wC
                    x0, x1 = X0 + 3, min(16 + (j + nc) * wC + 6, maxBX) - 3
                    j += nc
                    DH, TW = y1 - y0, x1 - x0
                    if DH <= 0 or TW <= 0:
                        continue
                    XA = X0 & ~15
                    j0 = X0 + 3 - XA
                    GPR = ((j0 + TW + 3) >> 2) - (j0 >> 2)
                    S = max(1, 256 // GPR)
                    seg = (DH + S - 1) // S
                    nchunk = (min(16 + (j) * wC + 6, maxBX) - XA + 15) >> 4
                    T["runs"] += 1
                    T["px"] += DH * TW
                    T["staged"] += (DH + 6) * nchunk * 16
                    T["items"] += GPR * DH
                    T["threads_busy"] += GPR * min(S, (DH + seg - 1) // seg) / 256.0
                    cm = c_ini[y0:y1, x0:x1]
                    n = int(cm.sum())
                    T["surv"] += n
                    T["surv_hist"].append(n / float(DH * TW))
                    T["both"] += int(both[y0:y1, x0:x1].sum())
                    T["overflow"] += n > 2176
                    if n > 2176:
                        # the same run in two rounds (lower / upper half of the row segments, k_fast_fix r05): survivors of each half,
                        # corners of the run (corner list: 640)
                        half_rows = ((((DH + seg - 1) // seg) + 1) >> 1) * seg
                        lo = int(cm[:half_rows].sum())
                        T.setdefault("over_runs", []).append((n, lo, n - lo, int(((sc[y0:y1, x0:x1] >= INI) & cm).sum())))
                    # the divergent append: per wave of 64 threads (tid = sidx * GPR + slot) the largest survivor count of a lane
                    jd0 = j0 & ~3
                    grp = (np.arange(TW) + j0 - jd0) >> 2                       # dword group of a domain column
                    per_item = np.zeros((DH, GPR), np.int32)
                    for g in range(GPR):
                        sel = grp == g
                        if sel.any():
                            per_item[:, g] = cm[:, sel].sum(1)
                    nseg = (DH + seg - 1) // seg
                    lane_seg = np.zeros(256, np.int32)
                    lane_str = np.zeros(256, np.int32)
                    for sidx in range(nseg):
                        for g in range(GPR):
                            tid = sidx * GPR + g
                            if tid >= 256:
                                continue
                            lane_seg[tid] = per_item[sidx * seg:(sidx + 1) * seg, g].sum()
                            lane_str[tid] = per_item[sidx::nseg, g].sum()
                    for w0 in range(0, 256, 64):
                        T["trips_seg"] += int(lane_seg[w0:w0 + 64].max())
                        T["trips_str"] += int(lane_str[w0:w0 + 64].max())
                        T["waves"] += 1
                    # corners, suppression, empty cells, second pass
                    for cj in range(nc):
                        cx0, cx1 = x0 + cj * wC, min(x0 + cj * wC + wC, x1)
                        if cx1 <= cx0:
                            continue
                        T["cells"] += 1
                        cs = sc[y0:y1, cx0:cx1]
                        ncor = int(((cs >= INI) & c_ini[y0:y1, cx0:cx1]).sum())
                        T["corners"] += ncor
                        keep = nms_cellwise(sc, INI, cx0, y0, cx1, y1)
                        T["nms"] += int(keep.sum())
                        if not keep.any():
                            T["empty"] += 1
                            T["px2"] += DH * (cx1 - cx0)
                            gA, gB = (j0 + cx0 - x0) >> 2, (j0 + cx1 - x0 - 1) >> 2
                            T["items2"] += (gB - gA + 1) * DH
                            T["surv2"] += int(c_min[y0:y1, cx0:cx1].sum())
                            T["corners2"] += int(((cs >= MIN) & c_min[y0:y1, cx0:cx1]).sum())
                            T["nms2"] += int(nms_cellwise(sc, MIN, cx0, y0, cx1, y1).sum())
                    # LDS cycles of the ring gather for three orders of the work list (first frame only: it is slow)
                    if want_lds and f is frames[0] and n > 0:
                        ys, xs = np.nonzero(cm)
                        cols = xs + j0
                        # kernel: threads in tid order append their entries (rows of the segment ascending, pixel ascending)
                        order = np.lexsort((xs & 3 if False else (cols & 3), ys % seg if False else ys, (cols - jd0) >> 2, ys // seg))
                        kr, kc = ys[order], cols[order]
                        rr = np.lexsort((cols, ys))
                        t8 = np.lexsort((cols, ys, ((cols - jd0) >> 5), ys >> 2))
                        for name, (r_, c_) in zip(orders, ((kr, kc), (ys[rr], cols[rr]), (ys[t8], cols[t8]))):
                            tot, ideal = gather_cycles(r_, c_, 176)
                            T["lds"][name][0] += tot
                            T["lds"][name][1] += ideal
    return T


def bound_table(T, nf):
    """(phase, unit, units per frame, cheap, slow instructions per unit, sequence) -> wave-instructions and SIMD cycles per frame."""
    per = lambda k: T[k] / nf                                                          # noqa: E731
    runs, waves = per("runs"), per("runs") * 4
    rows = [
        ("staging", "wave", waves, 14, 6, "address of the lane's 16-byte piece (5), <= 3 LDS-DMA issues with their address adds (6), score tile / bitmap zeroing (4), domain mask of the lane's group (5)"),
        ("compass", "item of 4 pixels", per("items"), 5, 15, "2 v_alignbyte + v_not + 12 v_lerp_u8 (one halving shared by both polarities) + 4 mask operations + v_dot4 (nibble gather); 5 LDS dwords"),
        ("list hand-over", "survivor", per("surv"), 2.5, 2.5, "per entry: find-first-bit, clear, entry add, store (the append as dense as a ballot-compacted one: 5 per entry, no idle lanes) + one returning add per 3 entries"),
        ("arc score", "survivor", per("surv") + per("both"), 12, 38, "entry read + decode (4), window address (2), 17 ring bytes, exact compass of the polarity (8), 8 v_bitop3 (pair + polarity), 21 v_pk_minimum3 / maximum3_f16, threshold + score store + corner append (7)"),
        ("suppression", "corner", per("corners"), 6, 14, "entry decode (4), 9 score bytes, 8 masked maxima, compare, bitmap OR"),
        ("second pass: compass", "item", per("items2"), 5, 15, "as above, on the cells without a survivor"),
        ("second pass: hand-over + score", "survivor", per("surv2"), 14.5, 40.5, "as above at minThFAST"),
        ("second pass: suppression", "corner", per("corners2"), 6, 14, "as above"),
        ("order + output", "cell", per("cells"), 10 / 64.0 * 64, 22, "per cell one wave: bitmap row, popcount, 10-step DPP scan, then per survivor of the fullest row 6 (find bit, position, score byte, store)"),
    ]
    out = []
    for name, unit, n, cheap, slow, seq in rows:
        lanes = 64.0
        if unit in ("wave", "cell"):
            winst = n * (cheap + slow)
            cyc = n * (cheap * CHEAP + slow * SLOW)
        else:
            winst = n * (cheap + slow) / lanes
            cyc = n * (cheap * CHEAP + slow * SLOW) / lanes
        out.append((name, unit, n, cheap + slow, winst, cyc, seq))
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--classes", default="textured,photographs")
    ap.add_argument("--frames", type=int, default=3)
    ap.add_argument("--out", default=None)
    ap.add_argument("--no-lds", action="store_true")
    args = ap.parse_args()
    lines = ["# k_fast_fix: work model and instruction bound (tools/fast_bound.py)", ""]
    for kind in args.classes.split(","):
        if kind == "textured":
            frames = synth.make_frames(1000, W, H, args.frames)
        elif kind == "photographs":
            frames = synth.photograph_frames(W, H, args.frames)
            if frames is None:
                continue
        else:
            frames = synth.make_frames_class(kind, 2000, W, H, args.frames)
        nf = len(frames)
        T = analyse(list(frames), want_lds=not args.no_lds)
        per = lambda k: T[k] / nf                                                       # noqa: E731
        sh = np.array(T["surv_hist"])
        lines += ["## %s (%d frames, per frame)" % (kind, nf), "",
                  "| quantity | value |", "|---|---|",
                  "| runs (workgroups) / cells | %.0f / %.0f |" % (per("runs"), per("cells")),
                  "| domain pixels / staged bytes | %.0f / %.0f (%.2f x) |" % (per("px"), per("staged"), per("staged") / per("px")),
                  "| compass items (4 pixels) / threads with an item | %.0f / %.1f %% |" % (per("items"), 100.0 * T["threads_busy"] / T["runs"]),
                  "| compass survivors at %d | %.0f = %.1f %% of the pixels (median run %.1f %%, 90th percentile %.1f %%); both polarities %.1f %% of them |"
                  % (INI, per("surv"), 100.0 * T["surv"] / T["px"], 100 * np.median(sh), 100 * np.percentile(sh, 90), 100.0 * T["both"] / max(T["surv"], 1)),
                  "| runs over the 2176-entry list | %.1f %% |" % (100.0 * T["overflow"] / T["runs"]),
                  "| ... of these: survivors <= 7/4 of the list / both halves fit a list / survivors (median, max) / corners (median) | %s |" % (
                      (lambda o: "%d of %d / %d / %d, %d / %d" % (sum(x[0] <= 3808 for x in o), len(o), sum(x[0] <= 3808 and x[1] <= 2176 and x[2] <= 2176 for x in o),
                                                               sorted(x[0] for x in o)[len(o) // 2], max(x[0] for x in o), sorted(x[3] for x in o)[len(o) // 2]))(T["over_runs"])
                      if T.get("over_runs") else "-"),
                  "| corners (score >= %d) / after suppression | %.0f (%.1f %% of the survivors) / %.0f |" % (INI, per("corners"), 100.0 * T["corners"] / max(T["surv"], 1), per("nms")),
                  "| cells without a survivor | %.0f = %.1f %% (pixels %.0f = %.1f %%) |" % (per("empty"), 100.0 * T["empty"] / T["cells"], per("px2"), 100.0 * T["px2"] / T["px"]),
                  "| second pass at %d: items / survivors / corners / after suppression | %.0f / %.0f (%.1f %% of its pixels) / %.0f / %.0f |"
                  % (MIN, per("items2"), per("surv2"), 100.0 * T["surv2"] / max(T["px2"], 1), per("corners2"), per("nms2")),
                  "| list append, trips per wave: a lane owns consecutive rows / rows strided over the segments | %.2f / %.2f (mean entries per lane %.2f) |"
                  % (T["trips_seg"] / T["waves"], T["trips_str"] / T["waves"], T["surv"] / (T["waves"] * 64.0)),
                  ]
        if not args.no_lds:
            for name, (tot, ideal) in T["lds"].items():
                lines.append("| ring gather, LDS cycles per conflict-free cycle, list order `%s` | %.2f |" % (name, tot / max(ideal, 1)))
        lines += ["", "### bound", "", "| phase | unit | units per frame | instructions per unit | wave-instructions per 1024 frames (M) | SIMD cycles per frame (k) | assumed sequence |", "|---|---|---|---|---|---|---|"]
        tw = tc = 0.0
        for name, unit, n, ipu, winst, cyc, seq in bound_table(T, nf):
            lines.append("| %s | %s | %.0f | %.1f | %.1f | %.0f | %s |" % (name, unit, n, ipu, winst * 1024 / 1e6, cyc / 1e3, seq))
            tw += winst
            tc += cyc
        ms = tc * 1024 / 1024.0 / 2.4e9 * 1e3               # 1024 frames over 1024 SIMDs at 2.4 GHz
        lines += ["| **total** | | | | **%.1f** | **%.0f** | |" % (tw * 1024 / 1e6, tc / 1e3), "",
                  "Issue-bound time at full vector issue: %.3f ms per 1024 frames = %.0f GB/s of algorithmic bytes = **%.3f of the HBM roofline**; at the "
                  "0.85 issue utilisation the kernel reaches (barriers, LDS round trips): %.3f ms = %.3f."
                  % (ms, 793732 * 1024 / (ms * 1e-3) / 1e9, 793732 * 1024 / (ms * 1e-3) / 1e9 / 8000.0, ms / 0.85, 793732 * 1024 / (ms / 0.85 * 1e-3) / 1e9 / 8000.0), ""]
    text = "\n".join(lines) + "\n"
    if args.out:
        with open(args.out, "w") as fh:
            fh.write(text)
    print(text)


if __name__ == "__main__":
    main()
