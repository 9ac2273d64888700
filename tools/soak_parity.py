#!/usr/bin/env python3
"""Randomised parity soak on the GPU box: HIP path vs the CPU oracle over many random geometries / parameters / seeds.
Not part of the test suite (minutes); run with `gpurun -- python tools/soak_parity.py [seconds]`.  Prints a summary, exits
non-zero on the first mismatch."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "vi-orb-slam-icra2018_amd"))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np  # noqa: E402

import orb_oracle_py as oracle  # noqa: E402
from orbhip import distributed as D, guided, synth  # noqa: E402
from orbhip.extractor import ORBextractor  # noqa: E402
from orbhip.vocabulary import ORBVocabulary  # noqa: E402


def soak_bow_seq(budget, rng):
    """Random vocabulary shapes / feature counts / thresholds for the batched SearchByBoW kernel (k_bow_seq)."""
    import ctypes as C
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import hiprt
    from orbhip.capi import check
    t0 = time.time()
    n = 0
    nmatch = 0
    while time.time() - t0 < budget:
        k, Lv = int(rng.integers(2, 11)), int(rng.integers(2, 5))
        levelsup = int(rng.integers(0, Lv + 1))
        NF = int(rng.choice([200, 600, 1000, 1500, 2500]))
        W, H = int(rng.integers(300, 900)), int(rng.integers(240, 600))
        th_mode = int(rng.integers(0, 2))
        ratio = float(rng.choice([0.6, 0.7, 0.75, 0.9]))
        seed = int(rng.integers(0, 1 << 30))
        B = 3
        frames = synth.make_frames(seed, W, H, B)
        blob = D.make_synthetic_vocabulary(seed % 1000, k=k, L=Lv)
        try:
            refx = oracle.Extractor(NF)
            feats_k = [refx(f) for f in frames]
        except Exception:
            continue
        ex = ORBextractor(NF, max_w=W, max_h=H, max_batch=B)
        ORBVocabulary(ex).loadFromBinaryBlob(blob)
        cap = ex.cap
        L = ex._L
        d_img = hiprt.DevBuf.from_numpy(frames)
        bufs = [hiprt.DevBuf(B * cap * 28), hiprt.DevBuf(B * cap * 32), hiprt.DevBuf(B * 4)] + [hiprt.DevBuf(B * cap * 4) for _ in range(5)] + \
            [hiprt.DevBuf(B * 4)]
        d_kps, d_desc, d_cnt, d_word, d_wt, d_node, d_m12, d_m21, d_nm = bufs
        valid = (rng.random((B, cap)) < 0.85).astype(np.uint8)
        d_valid = hiprt.DevBuf.from_numpy(valid)
        ex.extract_batch_device(d_img.ptr, B, W, H, W, H * W, d_kps.ptr, d_desc.ptr, cap, d_cnt.ptr)
        check(L.orbhip_vocab_transform_device(ex.handle, d_desc.ptr, B * cap, levelsup, d_word.ptr, d_wt.ptr, d_node.ptr), ex.handle)
        check_ori = int(rng.integers(0, 2))
        check(L.orbhip_search_by_bow_seq_device(ex.handle, d_desc.ptr, d_kps.ptr, d_cnt.ptr, d_node.ptr, d_wt.ptr, d_valid.ptr, cap, B,
                                                1, th_mode, C.c_float(ratio), check_ori, d_m12.ptr, d_m21.ptr, d_nm.ptr), ex.handle)
        ex.sync()
        m12 = d_m12.to_numpy(np.int32, (B, cap))
        m21 = d_m21.to_numpy(np.int32, (B, cap))
        nm = d_nm.to_numpy(np.int32, (B,))
        refv = oracle.Vocabulary(blob)
        fv = []
        for b in range(B):
            _, wt, nid = refv.transform(feats_k[b][1], levelsup)
            fv.append(oracle.feature_vector(nid, wt))
        for b in range(1, B):
            (k1, d1), (k2, d2) = feats_k[b - 1], feats_k[b]
            n1, n2 = len(k1), len(k2)
            wn, w12, w21 = oracle.search_by_bow(d1, valid[b - 1, :n1], k1["angle"], fv[b - 1], d2, valid[b, :n2] if th_mode else None,
                                                k2["angle"], fv[b], th=50, th_mode=th_mode, nnratio=ratio, check_ori=bool(check_ori))
            if nm[b] != wn or not np.array_equal(m12[b, :n1], w12) or not np.array_equal(m21[b, :n2], w21):
                print("MISMATCH bow_seq", k, Lv, levelsup, NF, W, H, th_mode, ratio, seed, b)
                sys.exit(1)
            nmatch += wn
        ex.close()
        for x in [d_img, d_valid] + bufs:
            x.free()
        n += 1
    print("soak bow_seq ok: %d random configurations in %.0f s, %d matches" % (n, time.time() - t0, nmatch))


def soak_stereo(budget, rng):
    """Random stereo rigs: ComputeStereoMatches on the resident pyramids, keypoint undistortion, rectification."""
    from orbhip import rectify
    from orbhip.extractor import ComputeStereoMatches
    t0 = time.time()
    n = 0
    stats = {"depth_points": 0, "remapped_px": 0, "undistorted": 0}
    while time.time() - t0 < budget:
        W, H = int(rng.integers(300, 1300)), int(rng.integers(200, 520))
        NF = int(rng.choice([300, 800, 1500, 2000]))
        nlev = int(rng.integers(3, 9))
        seed = int(rng.integers(0, 1 << 30))
        disp = int(rng.integers(3, 60))
        mbf = float(rng.uniform(20, 400))
        mb = float(rng.uniform(0.05, 0.6))
        try:
            L_, R_ = synth.make_stereo_pair(seed, W, H, disparity=disp)
            rL, rR = oracle.Extractor(NF, 1.2, nlev), oracle.Extractor(NF, 1.2, nlev)
            kL, dL = rL(L_)
            kR, dR = rR(R_)
        except Exception:
            continue
        exL = ORBextractor(NF, 1.2, nlev, max_w=W, max_h=H)
        exR = ORBextractor(NF, 1.2, nlev, max_w=W, max_h=H)
        try:
            gkL, gdL = exL(L_)
            gkR, gdR = exR(R_)
        except Exception as e:
            exL.close()
            exR.close()
            if "too small" in str(e):
                continue
            raise
        if gkL.tobytes() != kL.tobytes() or gkR.tobytes() != kR.tobytes():
            print("MISMATCH extract (stereo)", W, H, NF, nlev, seed)
            sys.exit(1)
        u, z, nb = ComputeStereoMatches(exL, gkL, gdL, exR, gkR, gdR, mb, mbf)
        ru, rz, rn = oracle.stereo_matches(rL, kL, dL, rR, kR, dR, mb, mbf)
        if nb != rn or u.tobytes() != ru.tobytes() or z.tobytes() != rz.tobytes():
            print("MISMATCH stereo", W, H, NF, nlev, seed, disp, mb, mbf)
            sys.exit(1)
        stats["depth_points"] += int((u >= 0).sum())
        # undistortion of the keypoints and rectification of the left image with a random mild rig
        fx, fy = float(rng.uniform(0.6, 1.2) * W), float(rng.uniform(0.6, 1.2) * W)
        K = np.array([fx, 0, W / 2 + rng.uniform(-20, 20), 0, fy, H / 2 + rng.uniform(-20, 20), 0, 0, 1.0])
        Dc = np.array([rng.uniform(-0.35, 0.1), rng.uniform(-0.1, 0.15), rng.uniform(-2e-3, 2e-3), rng.uniform(-2e-3, 2e-3), 0.0])
        a = rng.uniform(-0.02, 0.02, 3)
        Rm = np.array([[1, -a[2], a[1]], [a[2], 1, -a[0]], [-a[1], a[0], 1.0]])
        Rm, _ = np.linalg.qr(Rm)
        P = np.array([fx * 0.95, 0, K[2] + 1.5, 0, fx * 0.95, K[5] - 0.75, 0, 0, 1.0])
        pts = np.stack([kL["x"], kL["y"]], 1).astype(np.float32)
        nd = int(rng.choice([4, 5]))
        gu = rectify.undistort_points(exL, pts, K.astype(np.float32), Dc[:nd].astype(np.float32), K.astype(np.float32))
        ruu = oracle.undistort_points(pts, K.astype(np.float32), Dc[:nd].astype(np.float32), K.astype(np.float32))
        if gu.tobytes() != ruu.tobytes():
            print("MISMATCH undistort", W, H, seed)
            sys.exit(1)
        stats["undistorted"] += len(pts)
        mx, my = rectify.initUndistortRectifyMap(K, Dc, Rm.ravel(), P, W, H)
        omx, omy = oracle.init_undistort_rectify_map(K, Dc, Rm.ravel(), P, W, H)
        if not (np.array_equal(mx, omx) and np.array_equal(my, omy)):
            print("MISMATCH rectify map", W, H, seed)
            sys.exit(1)
        rect = rectify.Rectifier(exL, mx, my)(L_)
        if not np.array_equal(rect, oracle.remap_linear(L_, omx, omy)):
            print("MISMATCH remap", W, H, seed)
            sys.exit(1)
        stats["remapped_px"] += W * H
        exL.close()
        exR.close()
        n += 1
    print("soak stereo/rectify ok: %d random configurations in %.0f s, %s" % (n, time.time() - t0, stats))


def soak_match(budget, rng):
    """Random descriptor sets for the matcher primitives: knn2 (all-pairs, incl. the few-query form and ties), knn2_lists,
    SearchByBoW through the host API with random node groups, the grid and window queries."""
    from orbhip.capi import KP_DTYPE
    from orbhip.extractor import ORBmatcher
    t0 = time.time()
    n = 0
    M = ORBmatcher(0.7, True)
    ex = M._ctx
    while time.time() - t0 < budget:
        nq = int(rng.choice([1, 3, 17, 32, 33, 100, 700, 2000]))
        ndb = int(rng.choice([1, 2, 50, 333, 1000, 5000, 40000]))
        # low-entropy descriptors: many exact ties in distance, decided by the first-minimum rule
        base = rng.integers(0, 256, (int(rng.integers(1, 40)), 32), dtype=np.uint8)

        def mk(m):
            d = base[rng.integers(0, len(base), m)].copy()
            flips = rng.integers(0, 4, m)
            for i in np.nonzero(flips)[0]:
                for b in rng.integers(0, 256, flips[i]):
                    d[i, b >> 3] ^= 1 << (b & 7)
            return d
        q, db = mk(nq), mk(ndb)
        a, b_ = M.knn2(q, db), oracle.knn2(q, db)
        if not all(np.array_equal(x, y) for x, y in zip(a, b_)):
            print("MISMATCH knn2", nq, ndb)
            sys.exit(1)
        lens = rng.integers(0, min(ndb, 60) + 1, nq)
        off = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
        cand = rng.integers(0, ndb, int(off[-1])).astype(np.int32)
        a, b_ = M.knn2_lists(q, db, off, cand), oracle.knn2_lists(q, db, off, cand)
        if not all(np.array_equal(x, y) for x, y in zip(a, b_)):
            print("MISMATCH knn2_lists", nq, ndb)
            sys.exit(1)
        # SearchByBoW, host API: random node ids (sparse, large), some nodes on one side only, invalid flags
        n1, n2 = int(rng.integers(1, 1500)), int(rng.integers(1, 1500))
        d1, d2 = mk(n1), mk(n2)
        nn = int(rng.integers(1, 120))
        ids = np.sort(rng.choice(1 << 20, nn, replace=False))
        fv = []
        for m in (n1, n2):
            node = ids[rng.integers(0, nn, m)]
            wt = (rng.random(m) < 0.95).astype(np.float32)           # stopped words are left out of the FeatureVector
            fv.append(oracle.feature_vector(node, wt))
        v1, v2 = (rng.random(n1) < 0.8).astype(np.uint8), (rng.random(n2) < 0.8).astype(np.uint8)
        a1, a2 = rng.uniform(0, 360, n1).astype(np.float32), rng.uniform(0, 360, n2).astype(np.float32)
        for kf in (False, True):
            ga = M.SearchByBoW(d1, v1, a1, fv[0], d2, v2 if kf else None, a2, fv[1], kf_kf=kf)
            gb = oracle.search_by_bow(d1, v1, a1, fv[0], d2, v2 if kf else None, a2, fv[1], th=50, th_mode=1 if kf else 0, nnratio=0.7,
                                      check_ori=True)
            if ga[0] != gb[0] or not np.array_equal(ga[1], gb[1]) or not np.array_equal(ga[2], gb[2]):
                print("MISMATCH search_by_bow", n1, n2, nn, kf)
                sys.exit(1)
        # grid + windows on random (partly out-of-grid) points
        m = int(rng.integers(0, 3000))
        k = np.zeros(m, KP_DTYPE)
        k["x"], k["y"] = rng.uniform(-50, 900, m), rng.uniform(-50, 700, m)
        k["octave"] = rng.integers(0, 8, m)
        gp = guided.grid_params(float(rng.uniform(-10, 30)), float(rng.uniform(600, 860)), float(rng.uniform(-10, 30)), float(rng.uniform(400, 660)))
        if m:
            goff, gidx = guided.AssignFeaturesToGrid(ex, k, gp)
            roff, ridx = oracle.grid_build(k, gp)
            if not (np.array_equal(goff, roff) and np.array_equal(gidx, ridx)):
                print("MISMATCH grid", m)
                sys.exit(1)
            nw = 50
            x, y, r = rng.uniform(-100, 1000, nw).astype(np.float32), rng.uniform(-100, 800, nw).astype(np.float32), rng.uniform(0.5, 300, nw).astype(np.float32)
            lv = np.array([(-1, -1), (0, 3), (2, -1), (3, 4), (0, -1), (5, 5)])[rng.integers(0, 6, nw)]
            qoff, qidx = guided.GetFeaturesInArea(ex, k, gp, x, y, r, lv[:, 0], lv[:, 1])
            for i in range(nw):
                ref = oracle.features_in_area(k, (roff, ridx), gp, x[i], y[i], r[i], int(lv[i, 0]), int(lv[i, 1]))
                if not np.array_equal(qidx[qoff[i]:qoff[i + 1]], ref):
                    print("MISMATCH area", m, i)
                    sys.exit(1)
        # MapPoint::ComputeDistinctiveDescriptors over random lists of the same low-entropy descriptors
        P = int(rng.choice([1, 40, 600]))
        cnt = rng.integers(0, int(rng.choice([3, 20, 90, 300])) + 1, P)
        loff = np.concatenate([[0], np.cumsum(cnt)]).astype(np.int32)
        ld = mk(int(loff[-1])) if loff[-1] else np.zeros((0, 32), np.uint8)
        a = guided.ComputeDistinctiveDescriptors(ex, ld, loff)
        b_ = oracle.distinctive_descriptors(ld, loff)
        if not (np.array_equal(a[0], b_[0]) and np.array_equal(a[1], b_[1])):
            print("MISMATCH distinctive", P, int(loff[-1]))
            sys.exit(1)
        n += 1
    M.close()
    print("soak matcher primitives ok: %d random configurations in %.0f s" % (n, time.time() - t0))


def soak_knn2_seq(budget, rng):
    """The frame-sequence form of the brute force (orbhip_hamming_knn2_seq_device: frame b against frame b - lag, one workgroup walks
    the whole database of its frame): ragged counts, databases of one tile to several key ranges (16384 rows each), low-entropy
    descriptors (ties decide index and second best) and near-duplicates planted next to the best row."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import hiprt                                            # (tests/hiprt.py: device buffers without torch)
    from orbhip.capi import check
    from orbhip.extractor import ORBextractor
    ex = ORBextractor(1000, max_w=640, max_h=480)
    t0 = time.time()
    n = 0
    while time.time() - t0 < budget:
        cap = int(rng.choice([70, 300, 1000, 4000, 17000, 33500]))
        B = int(rng.integers(2, 5))
        lag = int(rng.integers(1, 3))
        counts = rng.integers(0, cap + 1, B).astype(np.int32)
        counts[rng.integers(0, B)] = cap
        if cap > 4000:
            counts[1:] = np.minimum(counts[1:], 1500)       # (the oracle's all-pairs loop sets the size)
            counts[0] = cap
        nbase = int(rng.integers(1, 60))
        base = rng.integers(0, 256, (nbase, 32), dtype=np.uint8)
        desc = base[rng.integers(0, nbase, (B, cap))].copy()
        noise = rng.random((B, cap)) < 0.5
        desc[noise, rng.integers(0, 32)] ^= rng.integers(0, 256, int(noise.sum()), dtype=np.uint8) & rng.integers(0, 256, int(noise.sum()), dtype=np.uint8)
        d_desc, d_cnt = hiprt.DevBuf.from_numpy(desc), hiprt.DevBuf.from_numpy(counts)
        d_bi, d_bd, d_sd = (hiprt.DevBuf(B * cap * 4) for _ in range(3))
        check(ex._L.orbhip_hamming_knn2_seq_device(ex.handle, d_desc.ptr, d_cnt.ptr, cap, B, lag, d_bi.ptr, d_bd.ptr, d_sd.ptr), ex.handle)
        ex.sync()
        bi, bd, sd = (x.to_numpy(np.int32, (B, cap)) for x in (d_bi, d_bd, d_sd))
        for b in range(B):
            nq = int(counts[b])
            if nq == 0:
                continue
            if b >= lag and counts[b - lag] > 0:
                wi, wd, ws = oracle.knn2(desc[b, :nq], desc[b - lag, :counts[b - lag]])
            else:
                wi, wd, ws = np.full(nq, -1), np.full(nq, 256), np.full(nq, 256)
            if not (np.array_equal(bi[b, :nq], wi) and np.array_equal(bd[b, :nq], wd) and np.array_equal(sd[b, :nq], ws)):
                print("MISMATCH knn2_seq", cap, B, lag, counts, b)
                sys.exit(1)
        n += 1
    ex.close()
    print("soak knn2_seq ok: %d random configurations in %.0f s" % (n, time.time() - t0))


_hip = None


def probe(step, cfg):
    """SOAK_PROBE=1: the HIP runtime's sticky last error after every step of the loop -- which call leaves one behind without reporting it?"""
    global _hip
    if not os.environ.get("SOAK_PROBE"):
        return
    if _hip is None:
        import ctypes
        _hip = ctypes.CDLL("libamdhip64.so.7")
        _hip.hipGetErrorString.restype = ctypes.c_char_p
    e = _hip.hipGetLastError()
    if e:
        print("PROBE: sticky HIP error %d (%s) after step '%s' of configuration %r" % (e, _hip.hipGetErrorString(e).decode(), step, cfg))
        sys.exit(2)


def main():
    budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
    if len(sys.argv) > 3 and sys.argv[3] == "knn2seq":
        soak_knn2_seq(budget, np.random.default_rng(int(sys.argv[2])))
        return
    if len(sys.argv) > 3 and sys.argv[3] == "match":
        soak_match(budget, np.random.default_rng(int(sys.argv[2])))
        return
    if len(sys.argv) > 3 and sys.argv[3] == "stereo":
        soak_stereo(budget, np.random.default_rng(int(sys.argv[2])))
        return
    if len(sys.argv) > 3 and sys.argv[3] == "bow":
        soak_bow_seq(budget, np.random.default_rng(int(sys.argv[2])))
        return
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 12345)
    t0 = time.time()
    blob = D.make_synthetic_vocabulary(99, k=6, L=4)
    V = oracle.Vocabulary(blob)
    n = 0
    stats = {"frames": 0, "kps": 0, "bow": 0, "proj": 0, "init": 0, "tri": 0, "fuse": 0}
    while time.time() - t0 < budget:
        w = int(rng.integers(200, 900))
        h = int(rng.integers(160, 700))
        nf = int(rng.choice([150, 400, 1000, 2000, 3500]))
        nlev = int(rng.integers(2, 9))
        scale = float(rng.choice([1.2, 1.2, 1.15, 1.3, 1.5, 1.08, 1.75, 2.0]))
        ini, mn = (20, 7) if rng.random() < 0.7 else (int(rng.integers(12, 40)), int(rng.integers(3, 12)))
        seed = int(rng.integers(0, 1 << 30))
        if rng.random() < 0.06:                       # now and then a large frame
            w, h = int(rng.integers(900, 1930)), int(rng.integers(500, 1090))
        kind = rng.choice(["scene", "scene", "scene", "noise", "blocks", "lowcontrast", "scene2x"])
        try:
            ref = oracle.Extractor(nf, scale, nlev, ini, mn)
            frames = synth.make_frames(seed, w, h, 2)
            if kind == "noise":                       # corners everywhere: the FAST list / corner-list fallbacks
                frames = np.random.default_rng(seed).integers(0, 256, frames.shape, dtype=np.uint8)
            elif kind == "blocks":                    # hard edges + a little noise: ties in scores and distances
                g = np.random.default_rng(seed)
                s_ = int(g.integers(5, 40))
                base = ((np.add.outer(np.arange(h) // s_, np.arange(w) // s_) % 2) * int(g.integers(20, 200)) + 20).astype(np.int32)
                frames = np.stack([np.clip(base + g.integers(-2, 3, base.shape), 0, 255).astype(np.uint8) for _ in range(2)])
            elif kind == "lowcontrast":               # most cells need the second threshold, many are empty
                frames = (frames.astype(np.int32) // 8 + 100).astype(np.uint8)
            elif kind == "scene2x":                   # saturating contrast
                frames = np.clip((frames.astype(np.int32) - 128) * 3 + 128, 0, 255).astype(np.uint8)
            r = [ref(f) for f in frames]
        except Exception:
            continue                                  # geometry the reference cannot handle (too small for some level)
        # every other configuration goes through the batch kernels (B >= 8: 5-cell FAST runs, 32-row resize tiles, 64-slot
        # describe workgroups); the others through the forms used for a frame or two
        reps = 4 if n % 2 else 1
        if os.environ.get("SOAK_TRACE"):
            with open(os.environ["SOAK_TRACE"], "a") as fh:     # last line = the configuration that was running
                fh.write("%d %d %d %d %g %d %d %d reps=%d kind=%s\n" % (w, h, nf, nlev, scale, ini, mn, seed, reps, kind))
        ex = None
        try:
            ex = ORBextractor(nf, scale, nlev, ini, mn, max_w=w, max_h=h, max_batch=2 * reps)
            ks, ds = ex.extract_batch(np.concatenate([frames] * reps))
            probe("extract_batch", (w, h, nf, nlev, scale, seed, reps))
            for b in range(2, 2 * reps):
                if ks[b].tobytes() != ks[b % 2].tobytes() or not np.array_equal(ds[b], ds[b % 2]):
                    print("MISMATCH batch copy", w, h, nf, nlev, scale, ini, mn, seed, b)
                    sys.exit(1)
        except Exception as e:
            if "too small" in str(e) or "too large for this number of levels" in str(e):   # documented limits
                if ex is not None:
                    ex.close()
                continue
            raise
        for b in range(2):
            if ks[b].tobytes() != r[b][0].tobytes() or not np.array_equal(ds[b], r[b][1]):
                print("MISMATCH extract", w, h, nf, nlev, scale, ini, mn, seed, b)
                sys.exit(1)
        (k0, d0), (k1, d1) = r
        stats["frames"] += 2
        stats["kps"] += len(k0) + len(k1)
        if len(k0) > 20 and len(k1) > 20:
            # vocabulary + SearchByBoW (host API)
            voc = ORBVocabulary(ex)
            voc.loadFromBinaryBlob(blob)
            g = []
            for d in (d0, d1):
                wd, wt, nid = V.transform(d, 2)
                g.append(oracle.feature_vector(nid, wt))
                hw, hwt, hnid = voc.transform_raw(d, 2)
                if not (np.array_equal(hw, wd) and np.array_equal(hwt, wt) and np.array_equal(hnid, nid)):
                    print("MISMATCH vocab", seed)
                    sys.exit(1)
            # the Frame constructor in one launch (orbhip_frame_build) on random calibrations: extraction, undistortion, grid and
            # transform against the oracle's four calls; every third time the frame then enters a matcher's resident sets from
            # the device block and is searched against the other frame
            Kc = np.array([[w * rng.uniform(0.5, 1.1), 0, w * rng.uniform(0.4, 0.6)], [0, w * rng.uniform(0.5, 1.1), h * rng.uniform(0.4, 0.6)],
                           [0, 0, 1]], np.float32)
            nd = int(rng.choice([0, 4, 4, 5, 8]))
            Dc = (rng.normal(0, 1, nd) * np.array([0.25, 0.08, 0.001, 0.001, 0.02, 0.01, 0.01, 0.01][:nd])).astype(np.float32)
            if nd and rng.random() < 0.15:
                Dc[0] = 0.0                           # the reference's shortcut: mvKeysUn = mvKeys
            lu = int(rng.choice([-1, 0, 2, 4]))
            use_grid = rng.random() < 0.85
            gpf = guided.grid_params(-0.05 * w, 1.04 * w, -0.03 * h, 1.05 * h)
            fb = ex.frame_build(frames[1], Kc, Dc, gpf if use_grid else None, min(lu, 4))
            probe("frame_build", (w, h, nf, nlev, scale, seed, reps))
            kun = k1.copy()
            if nd and Dc[0] != 0:
                xy = oracle.undistort_points(np.stack([k1["x"], k1["y"]], 1), Kc, Dc, Kc)
                kun["x"], kun["y"] = xy[:, 0], xy[:, 1]
            okf = fb["kps"].tobytes() == k1.tobytes() and fb["kps_un"].tobytes() == kun.tobytes() and np.array_equal(fb["desc"], d1)
            if okf and use_grid:
                go, gi = oracle.grid_build(kun, gpf)
                okf = np.array_equal(fb["cell_off"], go) and np.array_equal(fb["cell_idx"], gi)
            if okf and lu >= 0:
                wd, wt, nid = V.transform(d1, min(lu, 4))
                okf = np.array_equal(fb["word_id"], wd) and np.array_equal(fb["weight"], wt) and np.array_equal(fb["node_id"], nid)
            if not okf:
                print("MISMATCH frame_build", w, h, nf, nlev, scale, ini, mn, seed, nd, lu, use_grid)
                sys.exit(1)
            stats["frame_build"] = stats.get("frame_build", 0) + 1
            if n % 3 == 0 and lu >= 0:
                from orbhip.extractor import ORBmatcher
                Mf = ORBmatcher(0.7, True)
                fvb = oracle.feature_vector(fb["node_id"], fb["weight"])
                wd0, wt0, nid0 = V.transform(d0, min(lu, 4))
                fva = oracle.feature_vector(nid0, wt0)
                Mf.put_set_from_frame(21, ex, fvb)
                Mf.put_set(22, k0, d0, fva)
                v0 = (rng.random(len(k0)) < 0.8).astype(np.uint8)
                a3 = Mf.SearchByBoW_sets(22, v0, len(k0), 21, None, len(k1))
                b3 = oracle.search_by_bow(d0, v0, k0["angle"], fva, d1, None, k1["angle"], fvb, th=50, th_mode=0, nnratio=0.7, check_ori=True)
                if a3[0] != b3[0] or not np.array_equal(a3[1], b3[1]) or not np.array_equal(a3[2], b3[2]):
                    print("MISMATCH bow from frame block", w, h, nf, seed)
                    sys.exit(1)
                Mf.close()
                probe("bow from frame block", (w, h, nf, nlev, scale, seed, reps))
            # guided search / initialisation / triangulation
            gp = guided.grid_params(0, w, 0, h)
            sf = np.array(list(ref.params.mvScaleFactor)[:nlev], np.float32)
            s2 = np.array(list(ref.params.mvLevelSigma2)[:nlev], np.float32)
            q = guided.queries_for_last_frame(k0["x"] + np.float32(rng.normal(0, 2)), k0["y"] + np.float32(rng.normal(0, 2)), k0["x"],
                                              k0["octave"], k0["angle"], rng.random(len(k0)) < 0.9, rng.random(len(k0)) < 0.7,
                                              float(rng.choice([7, 15, 30])), sf)
            a = guided.SearchByProjection(ex, k1, d1, gp, q, d0, use_ratio=False, nnratio=0.9, check_ori=True)
            probe("SearchByProjection", (w, h, nf, nlev, scale, seed, reps))
            b_ = oracle.search_by_projection(k1, d1, gp, q, d0, use_ratio=False, nnratio=0.9, check_ori=True)
            if a[0] != b_[0] or not np.array_equal(a[1], b_[1]):
                print("MISMATCH proj", w, h, nf, seed)
                sys.exit(1)
            stats["proj"] += a[0]
            prev = np.stack([k0["x"], k0["y"]], 1).astype(np.float32)
            win = int(rng.choice([10, 40, 100]))
            a = guided.SearchForInitialization(ex, k0, d0, k1, d1, gp, prev, win, 0.9, True)
            probe("SearchForInitialization", (w, h, nf, nlev, scale, seed, reps))
            b_ = oracle.search_for_initialization(k0, d0, k1, d1, gp, prev, win, 0.9, True)
            if a[0] != b_[0] or not np.array_equal(a[1], b_[1]) or not np.array_equal(a[2], b_[2]):
                print("MISMATCH init", w, h, nf, seed, win)
                sys.exit(1)
            stats["init"] += a[0]
            F = np.array([[0, 0, 0], [0, 0, -1], [0, 1, 0]], np.float32) + rng.normal(0, 1e-5, (3, 3)).astype(np.float32)
            sk0, sk1 = (rng.random(len(k0)) < 0.3).astype(np.uint8), (rng.random(len(k1)) < 0.3).astype(np.uint8)
            a = guided.SearchForTriangulation(ex, k0, d0, sk0, g[0], k1, d1, sk1, g[1], F, w / 2, h / 2, sf, s2)
            probe("SearchForTriangulation", (w, h, nf, nlev, scale, seed, reps))
            b_ = oracle.search_for_triangulation(k0, d0, sk0, g[0], k1, d1, sk1, g[1], F, w / 2, h / 2, sf, s2)
            if a[0] != b_[0] or not np.array_equal(a[1], b_[1]):
                print("MISMATCH tri", w, h, nf, seed)
                sys.exit(1)
            stats["tri"] += a[0]
            # the Fuse / SearchBySim3 window search: frame 0's features as points projected into frame 1
            nq = len(k0)
            qw = np.zeros(nq, guided.QUERY_DTYPE)
            qw["u"] = (k0["x"] + rng.normal(0, 2, nq)).astype(np.float32)
            qw["v"] = (k0["y"] + rng.normal(0, 2, nq)).astype(np.float32)
            pred = np.clip(k0["octave"] + rng.integers(-1, 2, nq), 0, nlev - 1).astype(np.int32)
            qw["radius"] = (np.float32(rng.choice([3, 4, 7.5])) * sf[pred]).astype(np.float32)
            qw["proj_xr"] = (qw["u"] - rng.uniform(0, 30, nq)).astype(np.float32)
            qw["min_level"], qw["max_level"] = pred - 1, pred
            qw["flags"] = np.where(rng.random(nq) < 0.1, 0, 1)
            gate = rng.random() < 0.6
            ur = np.where(rng.random(len(k1)) < 0.5, k1["x"] - rng.uniform(0, 30, len(k1)), -1).astype(np.float32) \
                if rng.random() < 0.5 else None
            isg = (np.float32(1) / s2).astype(np.float32) if gate else None
            a = guided.WindowBest(ex, k1, d1, gp, qw, d0, ur, isg)
            probe("WindowBest", (w, h, nf, nlev, scale, seed, reps))
            b_ = oracle.window_best(k1, d1, gp, qw, d0, ur, isg)
            if not np.array_equal(a[0], b_[0]) or not np.array_equal(a[1], b_[1]):
                print("MISMATCH window_best", w, h, nf, seed, gate, ur is not None)
                sys.exit(1)
            stats["fuse"] += int((a[1] <= 50).sum())
            # the same search into a resident set, and SearchByBoW between resident sets (orbhip_set_*)
            from orbhip.extractor import ORBmatcher
            M = ORBmatcher(0.7, True, ctx=ex)
            M.put_set(11, k1, d1, g[1], gp)
            M.put_set(12, k0, d0, g[0], gp)
            a2 = guided.WindowBestSet(ex, 11, qw, d0, ur, isg)
            if not np.array_equal(a2[0], b_[0]) or not np.array_equal(a2[1], b_[1]):
                print("MISMATCH window_best_set", w, h, nf, seed)
                sys.exit(1)
            v0 = (rng.random(len(k0)) < 0.8).astype(np.uint8)
            a3 = M.SearchByBoW_sets(12, v0, len(k0), 11, None, len(k1))
            b3 = oracle.search_by_bow(d0, v0, k0["angle"], g[0], d1, None, k1["angle"], g[1], th=50, th_mode=0, nnratio=0.7, check_ori=True)
            if a3[0] != b3[0] or not np.array_equal(a3[1], b3[1]) or not np.array_equal(a3[2], b3[2]):
                print("MISMATCH bow_sets", w, h, nf, seed)
                sys.exit(1)
            M.drop_set()
            probe("sets", (w, h, nf, nlev, scale, seed, reps))
        ex.close()
        probe("close", (w, h, nf, nlev, scale, seed, reps))
        n += 1
    print("soak ok: %d random configurations in %.0f s, %s" % (n, time.time() - t0, stats))


if __name__ == "__main__":
    main()
