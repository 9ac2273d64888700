"""Rare-event coverage inside the driver-run suite.

Round 5's only wrong-result defect -- a form of k_bow_lane that was wrong once in a few thousand frame pairs -- passed every
B = 4 test and the soak and was caught by bench.py's whole-batch check alone.  These tests put that kind of coverage where the
driver runs it: fixed seeds, a time budget of a few minutes in total, counts reported in pytest's terminal summary
(tests/conftest.py: "orbhip rare-event coverage: ...").

  (a) the batched SearchByBoW (k_bow_lane / k_bow_seq; ref src/ORBmatcher.cc:159-288, :522-655 -- the greedy claims of :205-232
      are order dependent, ties are where rare bugs live) on thousands of frame pairs in batches of 256-1024: real features of
      distinct frames, and synthetic low-entropy descriptor sets with random vocabulary shapes, ragged counts, validity masks,
      both threshold modes, lag 1 / 2, with and without the rotation check;
  (b) bench.py's whole-batch check as a test: a 1024-frame batch of textured frames and photographs, every distinct frame and
      every pair against the oracle, every tiled copy against its original;
  (c) hundreds of random extraction configurations (sizes, scales, levels, thresholds, content), with the kernel variants they
      reached read back from the library (orbhip_debug_path_mask) and asserted: k_fast (generic grid), k_resize<32>, the
      global-memory quadtree.
"""
import ctypes as C
import json
import os
import subprocess
import sys
import time

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))


# ------------------------------------------------------------------------------------------------------------------------
# helpers
# ------------------------------------------------------------------------------------------------------------------------
def fast_feature_vector(node, weight):
    """oracle.feature_vector (DBoW2::FeatureVector as CSR, ascending feature index inside a node, stopped words left out) by one
    stable sort instead of a scan per node."""
    node = np.asarray(node)
    keep = np.nonzero(np.asarray(weight) > 0)[0]
    order = keep[np.argsort(node[keep], kind="stable")]
    ids, start = np.unique(node[order], return_index=True)
    off = np.concatenate([start, [len(order)]]).astype(np.int32)
    return ids.astype(np.int32), off, order.astype(np.int32)


def low_entropy_descriptors(rng, nbase, shape, max_flips=3):
    """Descriptors drawn from `nbase` base rows with 0..max_flips random bit flips each: many exact ties in distance."""
    base = rng.integers(0, 256, (nbase, 32), dtype=np.uint8)
    pick = rng.integers(0, nbase, shape)
    d = base[pick].copy()
    flat = d.reshape(-1, 32)
    for _ in range(max_flips):
        on = rng.random(len(flat)) < 0.5
        bit = rng.integers(0, 256, len(flat))
        rows = np.nonzero(on)[0]
        flat[rows, bit[rows] >> 3] ^= (1 << (bit[rows] & 7)).astype(np.uint8)
    return d, pick


def path_mask(reset=False):
    from orbhip import capi
    return int(capi.load().orbhip_debug_path_mask(1 if reset else 0))


PATH_BITS = {"k_fast_fix": 0, "k_fast": 1, "k_resize_fit": 2, "k_resize<32>/<8>": 3, "k_pyramid_chain": 4, "k_quadtree(lds)": 5,
             "k_quadtree(lds points)": 6, "k_quadtree(global)": 7, "k_bow_lane": 8, "k_bow_seq(lds)": 9, "k_bow_seq(global)": 10,
             "k_fast_fix(tall)": 11, "k_describe": 12, "k_describe_blur": 13, "k_blur": 14}


def paths_of(mask):
    return sorted(k for k, b in PATH_BITS.items() if mask >> b & 1)


# ------------------------------------------------------------------------------------------------------------------------
# (a) batched SearchByBoW on thousands of pairs
# ------------------------------------------------------------------------------------------------------------------------
def _bow_seq_device(ex, bufs, cap, B, lag, th_mode, ratio, check_ori, valid):
    import hiprt
    from orbhip.capi import check
    d_valid = hiprt.DevBuf.from_numpy(valid) if valid is not None else None
    check(ex._L.orbhip_search_by_bow_seq_device(ex.handle, bufs["desc"].ptr, bufs["kps"].ptr, bufs["cnt"].ptr, bufs["node"].ptr,
                                                bufs["wt"].ptr, d_valid.ptr if d_valid else None, cap, B, lag, th_mode,
                                                C.c_float(ratio), check_ori, bufs["m12"].ptr, bufs["m21"].ptr, bufs["nm"].ptr),
          ex.handle, "search_by_bow_seq")
    ex.sync()
    out = (bufs["m12"].to_numpy(np.int32, (B, cap)), bufs["m21"].to_numpy(np.int32, (B, cap)), bufs["nm"].to_numpy(np.int32, (B,)))
    if d_valid:
        d_valid.free()
    return out


def _check_pairs(oracle, what, desc, angle, counts, node, wt, valid, lag, th_mode, ratio, check_ori, got):
    """Every pair (b - lag, b) of the batch against the oracle's SearchByBoW; returns (pairs, matches)."""
    m12, m21, nm = got
    B = len(counts)
    fv = [fast_feature_vector(node[b, :counts[b]], wt[b, :counts[b]]) for b in range(B)]
    pairs = matches = 0
    for b in range(B):
        n2 = int(counts[b])
        if b < lag:
            assert nm[b] == 0 and (m12[b] == -1).all() and (m21[b] == -1).all(), "%s: frame %d has no predecessor" % (what, b)
            continue
        a = b - lag
        n1 = int(counts[a])
        v1 = valid[a, :n1] if valid is not None else np.ones(n1, np.uint8)
        v2 = valid[b, :n2] if (valid is not None and th_mode) else None
        wn, w12, w21 = oracle.search_by_bow(desc[a, :n1], v1, angle[a, :n1], fv[a], desc[b, :n2], v2, angle[b, :n2], fv[b], th=50,
                                            th_mode=th_mode, nnratio=ratio, check_ori=bool(check_ori))
        ok = nm[b] == wn and np.array_equal(m12[b, :n1], w12) and np.array_equal(m21[b, :n2], w21) and \
            (m12[b, n1:] == -1).all() and (m21[b, n2:] == -1).all()
        assert ok, "%s: pair (%d, %d) differs from the oracle (matches %d vs %d)" % (what, a, b, nm[b], wn)
        pairs += 1
        matches += wn
    return pairs, matches


SYNTH_RECIPES = [
    # (seed, B, cap, nbase, nnodes, th_mode, ratio, check_ori, lag, use_valid)
    (9001, 1024, 1000, 4000, 120, 0, 0.7, 1, 1, False),     # the bench's shape: ~100 shared nodes, distinct descriptors
    (9002, 768, 1000, 60, 40, 0, 0.7, 1, 1, True),          # few base rows: ties everywhere
    (9003, 512, 1000, 8, 5, 1, 0.75, 0, 1, True),           # five large nodes (> 128 candidates: the cooperative path)
    (9004, 512, 600, 500, 600, 1, 0.9, 1, 2, True),         # more nodes than features per node: lane items of 1-3 candidates
    (9005, 256, 2500, 3000, 150, 0, 0.6, 1, 1, True),       # cap 2500: NP 4096
    (9006, 384, 300, 40, 1, 0, 0.7, 1, 1, False),           # one node holds everything (levelsup above the tree's depth)
    (9007, 640, 1000, 1500, 90, 1, 0.7, 1, 1, True),        # KF-KF mode at the bench's shape
    (9008, 512, 1000, 20000, 100, 0, 0.19, 1, 1, False),    # ratio outside the byte clamp's bound: k_bow_seq
    (9009, 768, 1200, 200, 33, 0, 0.7, 0, 2, True),         # nodes of 17-64 candidates: 16-lane rows, lag 2
    (9010, 1024, 700, 2500, 1000, 0, 0.8, 1, 1, True),      # a thousand small nodes
]


@pytest.mark.parametrize("rep", [0, 1, 2])
@pytest.mark.parametrize("recipe", SYNTH_RECIPES, ids=lambda r: "seed%d-B%d-cap%d" % r[:3])
def test_bow_seq_synthetic_batches_match_oracle(oracle, recipe, rep, tally):
    import hiprt
    from orbhip.capi import KP_DTYPE
    from orbhip.extractor import ORBextractor
    seed, B, cap, nbase, nnodes, th_mode, ratio, check_ori, lag, use_valid = recipe
    rng = np.random.default_rng(seed + 100 * rep)
    ex = ORBextractor(300, max_w=320, max_h=240)
    counts = rng.integers(int(0.6 * cap), cap + 1, B).astype(np.int32)
    counts[rng.integers(0, B, 4)] = cap
    counts[rng.integers(0, B, 3)] = rng.integers(0, 3, 3)          # empty and near-empty frames
    desc, pick = low_entropy_descriptors(rng, nbase, (B, cap))
    # the node of a feature follows its base row (a vocabulary groups similar descriptors), 5 % land elsewhere
    node_ids = np.sort(rng.choice(1 << 20, nnodes, replace=False)).astype(np.int32) + 1
    node_of_base = node_ids[rng.integers(0, nnodes, nbase)]
    node = node_of_base[pick]
    stray = rng.random((B, cap)) < 0.05
    node[stray] = node_ids[rng.integers(0, nnodes, int(stray.sum()))]
    wt = np.where(rng.random((B, cap)) < 0.95, rng.uniform(0.01, 3.0, (B, cap)), 0.0).astype(np.float32)
    kps = np.zeros((B, cap), KP_DTYPE)
    # angles on a coarse lattice: equal rotation bins, and histogram maxima that tie (ComputeThreeMaxima, :1661-1669)
    kps["angle"] = (rng.integers(0, 72, (B, cap)) * 5).astype(np.float32) + np.where(rng.random((B, cap)) < 0.3, 0.0, rng.random((B, cap))).astype(np.float32)
    valid = (rng.random((B, cap)) < 0.85).astype(np.uint8) if use_valid else None
    bufs = {"desc": hiprt.DevBuf.from_numpy(desc), "kps": hiprt.DevBuf.from_numpy(kps), "cnt": hiprt.DevBuf.from_numpy(counts),
            "node": hiprt.DevBuf.from_numpy(node), "wt": hiprt.DevBuf.from_numpy(wt), "m12": hiprt.DevBuf(B * cap * 4),
            "m21": hiprt.DevBuf(B * cap * 4), "nm": hiprt.DevBuf(B * 4)}
    path_mask(reset=True)
    got = _bow_seq_device(ex, bufs, cap, B, lag, th_mode, ratio, check_ori, valid)
    mask = path_mask()
    want_lane = 50.0 < ratio * 255.0 and os.environ.get("ORBHIP_BOW_LANE", "1") != "0"   # (ablation build: k_bow_seq for every pair)
    assert bool(mask >> 8 & 1) == want_lane and bool(mask >> 9 & 3) == (not want_lane), paths_of(mask)
    pairs, matches = _check_pairs(oracle, "synthetic %r" % (recipe,), desc, kps["angle"], counts, node, wt, valid, lag, th_mode, ratio,
                                  check_ori, got)
    assert pairs >= B - lag - 0 and matches > 0
    tally("bow_seq pairs (synthetic)", pairs)
    tally("bow_seq matches", matches)
    for x in bufs.values():
        x.free()
    ex.close()


def test_bow_seq_wave_per_node_kernel_on_the_same_batches():
    """k_bow_seq, the fallback of k_bow_lane: the synthetic recipes with the lane kernel switched off (ORBHIP_BOW_LANE=0,
    ablation build, child process) -- among them 2500 slots per frame (NP 4096: descriptors in global memory), one node holding
    everything, five nodes beyond 128 candidates and a thousand small ones."""
    ids = ["%s::test_bow_seq_synthetic_batches_match_oracle[seed%d-B%d-cap%d-0]" % ((os.path.abspath(__file__),) + r[:3])
           for r in SYNTH_RECIPES if r[0] in (9001, 9003, 9005, 9006, 9010)]
    p = subprocess.run([sys.executable, "-m", "pytest", "-q", "-m", "gpu", "-x"] + ids,
                       env=dict(os.environ, ORBHIP_BOW_LANE="0"), capture_output=True, text=True, timeout=900, cwd=os.path.dirname(HERE))
    assert p.returncode == 0 and " passed" in p.stdout, p.stdout[-3000:] + p.stderr[-2000:]
    assert "5 passed" in p.stdout, p.stdout[-1500:]


@pytest.mark.parametrize("seed,B,levelsup,k,Lv,th_mode", [(9101, 1024, 4, 10, 6, 0), (9102, 1024, 2, 10, 4, 1), (9103, 512, 1, 6, 3, 0)])
def test_bow_seq_real_features_of_distinct_frames_match_oracle(oracle, seed, B, levelsup, k, Lv, th_mode, tally):
    """extract_batch_device -> vocab_transform_device -> search_by_bow_seq_device on B DISTINCT 640 x 480 frames (a stream's frames
    shifted cyclically, a different shift per round); the oracle's transform + SearchByBoW on the device's keypoints /
    descriptors (eight of the frames through the oracle's extractor too).  The first case is the bench's step and vocabulary."""
    import hiprt
    from orbhip import distributed as D, synth
    from orbhip.capi import KP_DTYPE, check
    from orbhip.extractor import ORBextractor
    from orbhip.vocabulary import ORBVocabulary
    W, H, NF = 640, 480, 1000
    U = 32
    uniq = synth.make_frames(seed, W, H, U)
    frames = np.stack([np.roll(uniq[b % U], ((5 * (b // U)) % H, (9 * (b // U)) % W), axis=(0, 1)) for b in range(B)])
    blob = D.make_synthetic_vocabulary(seed % 1000, k=k, L=Lv)
    ex = ORBextractor(NF, max_w=W, max_h=H, max_batch=B)
    ORBVocabulary(ex).loadFromBinaryBlob(blob)
    cap = ex.cap
    d_img = hiprt.DevBuf.from_numpy(frames)
    bufs = {"kps": hiprt.DevBuf(B * cap * 28), "desc": hiprt.DevBuf(B * cap * 32), "cnt": hiprt.DevBuf(B * 4), "word": hiprt.DevBuf(B * cap * 4),
            "wt": hiprt.DevBuf(B * cap * 4), "node": hiprt.DevBuf(B * cap * 4), "m12": hiprt.DevBuf(B * cap * 4), "m21": hiprt.DevBuf(B * cap * 4),
            "nm": hiprt.DevBuf(B * 4)}
    ex.extract_batch_device(d_img.ptr, B, W, H, W, H * W, bufs["kps"].ptr, bufs["desc"].ptr, cap, bufs["cnt"].ptr)
    check(ex._L.orbhip_vocab_transform_device(ex.handle, bufs["desc"].ptr, B * cap, levelsup, bufs["word"].ptr, bufs["wt"].ptr,
                                              bufs["node"].ptr), ex.handle)
    rng = np.random.default_rng(seed)
    valid = (rng.random((B, cap)) < 0.9).astype(np.uint8) if th_mode else None
    got = _bow_seq_device(ex, bufs, cap, B, 1, th_mode, 0.7, 1, valid)
    counts = bufs["cnt"].to_numpy(np.int32, (B,))
    kps = bufs["kps"].to_numpy(KP_DTYPE, (B, cap))
    desc = bufs["desc"].to_numpy(np.uint8, (B, cap, 32))
    node = bufs["node"].to_numpy(np.int32, (B, cap))
    wt = bufs["wt"].to_numpy(np.float32, (B, cap))
    refx, refv = oracle.Extractor(NF), oracle.Vocabulary(blob)
    for b in list(range(0, B, B // 8))[:8]:                    # the inputs of the comparison are themselves the oracle's
        rk, rd = refx(frames[b])
        assert counts[b] == len(rk) and kps[b, :len(rk)].tobytes() == rk.tobytes() and np.array_equal(desc[b, :len(rk)], rd)
    for b in range(B):
        n = int(counts[b])
        _, rwt, rnid = refv.transform(desc[b, :n], levelsup)
        assert np.array_equal(rwt, wt[b, :n]) and np.array_equal(rnid, node[b, :n]), "vocabulary transform of frame %d" % b
    pairs, matches = _check_pairs(oracle, "real features seed %d" % seed, desc, kps["angle"], counts, node, wt, valid, 1, th_mode, 0.7, 1, got)
    assert pairs == B - 1 and matches > 100 * pairs // 2
    tally("bow_seq pairs (real features, distinct frames)", pairs)
    tally("bow_seq matches", matches)
    for x in list(bufs.values()) + [d_img]:
        x.free()
    ex.close()


# ------------------------------------------------------------------------------------------------------------------------
# (b) the whole-batch check of bench.py as a test
# ------------------------------------------------------------------------------------------------------------------------
def test_whole_batch_textured_and_photographs_every_frame_verified(oracle, tally):
    """A 1024-frame batch through the bench's step (extract + transform + SearchByBoW, one launch each): 512 distinct frames --
    384 textured (a stream's frames, shifted per round) and 128 cut from the photographs this image holds (when present; textured
    otherwise) -- each against the oracle, frame and pair; the other 512 are tiled copies, each equal to its original."""
    import hiprt
    from orbhip import distributed as D, synth
    from orbhip.capi import KP_DTYPE, check
    from orbhip.extractor import ORBextractor
    from orbhip.vocabulary import ORBVocabulary
    W, H, NF, B, U = 640, 480, 1000, 1024, 512
    base = synth.make_frames(1000, W, H, 32)
    tex = [np.roll(base[i % 32], ((7 * (i // 32)) % H, (13 * (i // 32)) % W), axis=(0, 1)) for i in range(384)]
    ph = synth.photograph_frames(W, H, 128)
    n_photo = 0 if ph is None else len(ph)
    extra = list(ph) if ph is not None else [np.roll(base[i % 32], (11 + i, 3 * i), axis=(0, 1)) for i in range(128)]
    uniq = np.stack(tex + extra)
    assert len(uniq) == U
    frames = np.concatenate([uniq, uniq])
    blob = D.make_synthetic_vocabulary(4242, 10, 6)           # the bench's vocabulary (stock shape)
    ex = ORBextractor(NF, max_w=W, max_h=H, max_batch=B)
    ORBVocabulary(ex).loadFromBinaryBlob(blob)
    cap = ex.cap
    d_img = hiprt.DevBuf.from_numpy(frames)
    bufs = {"kps": hiprt.DevBuf(B * cap * 28), "desc": hiprt.DevBuf(B * cap * 32), "cnt": hiprt.DevBuf(B * 4), "word": hiprt.DevBuf(B * cap * 4),
            "wt": hiprt.DevBuf(B * cap * 4), "node": hiprt.DevBuf(B * cap * 4), "m12": hiprt.DevBuf(B * cap * 4), "m21": hiprt.DevBuf(B * cap * 4),
            "nm": hiprt.DevBuf(B * 4)}
    for rep in range(2):                                        # twice: the second run on buffers that already hold results
        ex.extract_batch_device(d_img.ptr, B, W, H, W, H * W, bufs["kps"].ptr, bufs["desc"].ptr, cap, bufs["cnt"].ptr)
        check(ex._L.orbhip_vocab_transform_device(ex.handle, bufs["desc"].ptr, B * cap, 4, bufs["word"].ptr, bufs["wt"].ptr, bufs["node"].ptr),
              ex.handle)
        m12, m21, nm = _bow_seq_device(ex, bufs, cap, B, 1, 0, 0.7, 1, None)
    counts = bufs["cnt"].to_numpy(np.int32, (B,))
    kps = bufs["kps"].to_numpy(KP_DTYPE, (B, cap))
    desc = bufs["desc"].to_numpy(np.uint8, (B, cap, 32))
    refx, refv = oracle.Extractor(NF), oracle.Vocabulary(blob)
    feats = []
    for b in range(U + 1):                                      # rows 0 .. U: every distinct frame, and the pair across the tile boundary
        rk, rd = refx(frames[b])
        n = len(rk)
        assert counts[b] == n, "frame %d: %d keypoints, the oracle %d" % (b, counts[b], n)
        assert kps[b, :n].tobytes() == rk.tobytes(), "keypoints of frame %d" % b
        assert np.array_equal(desc[b, :n], rd), "descriptors of frame %d" % b
        _, wt, nid = refv.transform(rd, 4)
        feats.append((rk, rd, fast_feature_vector(nid, wt)))
        if b:
            (k1, d1, f1), (k2, d2, f2) = feats[b - 1], feats[b]
            wn, w12, w21 = oracle.search_by_bow(d1, np.ones(len(d1), np.uint8), k1["angle"], f1, d2, None, k2["angle"], f2, th=50, th_mode=0,
                                                nnratio=0.7, check_ori=True)
            assert nm[b] == wn and np.array_equal(m12[b, :len(d1)], w12) and np.array_equal(m21[b, :len(d2)], w21), "SearchByBoW of pair %d" % b
    for b in range(U + 1, B):                                   # tiled copies: frame b = frame b - U, pair (b - 1, b) = pair (b - U - 1, b - U)
        o = b - U
        n = int(counts[o])
        assert counts[b] == n and kps[b, :n].tobytes() == kps[o, :n].tobytes() and np.array_equal(desc[b, :n], desc[o, :n]), "copy %d" % b
        assert nm[b] == nm[o] and np.array_equal(m12[b], m12[o]) and np.array_equal(m21[b], m21[o]), "pair of copy %d" % b
    tally("whole-batch frames verified (vs oracle + copies vs originals)", B)
    tally("whole-batch distinct frames vs oracle", U + 1)
    tally("whole-batch photographs", n_photo)
    for x in list(bufs.values()) + [d_img]:
        x.free()
    ex.close()


# ------------------------------------------------------------------------------------------------------------------------
# (c) random extraction configurations
# ------------------------------------------------------------------------------------------------------------------------
def run_extraction_configs(nconf, seed, budget_s=1e9):
    """`nconf` random extraction configurations (tools/soak_parity.py's generator): two frames each, every other configuration as a
    batch of eight (the batch kernels), against the oracle; a few per hundred aimed at the fallback paths.  Returns a summary."""
    import orb_oracle_py as oracle
    from orbhip import synth
    from orbhip.extractor import ORBextractor
    rng = np.random.default_rng(seed)
    scenes = {}
    n = skipped = kp_total = 0
    seen = 0
    aimed = {"k_fast": 0, "k_resize<32>/<8>": 0, "k_quadtree(global)": 0}
    t0 = time.time()
    while n < nconf and time.time() - t0 < budget_s:
        w, h = int(rng.integers(200, 900)), int(rng.integers(160, 700))
        nf = int(rng.choice([150, 400, 1000, 2000, 3500]))
        nlev = int(rng.integers(2, 9))
        scale = float(rng.choice([1.2, 1.2, 1.15, 1.3, 1.5, 1.08, 1.75, 2.0]))
        ini, mn = (20, 7) if rng.random() < 0.7 else (int(rng.integers(12, 40)), int(rng.integers(3, 12)))
        reps = 8 if n % 6 == 5 else (4 if n % 2 else 1)   # batches of 2, 8 and 16 frames (16: the two-stream schedule of the half-batches)
        aim = None
        r = rng.random()
        if r < 0.04:                                   # the global-memory quadtree: a per-level quota beyond the LDS tables
            nf, nlev, scale, aim = (int(rng.choice([2500, 3000])), 1, 1.2, "k_quadtree(global)") if rng.random() < 0.5 else \
                (3500, 2, 1.75, "k_quadtree(global)")
            w, h = int(rng.integers(480, 800)), int(rng.integers(270, 600))
        elif r < 0.10:                                 # cells wider / taller than the fixed-layout kernel's bounds: k_fast
            # (a run is laid out for five cells: wCell >= 38 <=> width - 32 in 38..59, 75..89, 112..119; hCell + 6 > 48 likewise)
            w, h, aim = int(rng.choice([int(rng.integers(70, 92)), int(rng.integers(107, 122)), int(rng.integers(144, 152))])), \
                int(rng.integers(75, 200)), "k_fast"
            nlev, scale, reps = int(rng.integers(1, 3)), 1.2, 4
        elif r < 0.16:                                 # large frames (now and then beyond the fitted resize tiles' window)
            w, h = int(rng.integers(900, 1930)), int(rng.integers(500, 1090))
        kind = str(rng.choice(["scene", "scene", "scene", "noise", "blocks", "lowcontrast", "scene2x"]))
        fseed = int(rng.integers(0, 1 << 30))
        try:
            ref = oracle.Extractor(nf, scale, nlev, ini, mn)
            key = fseed % 12                           # a dozen scenes, cut to size: drawing a scene costs more than extracting it
            if key not in scenes:
                scenes[key] = synth.make_scene(5000 + key, 1930, 1090)
            t = int(rng.integers(0, 50))
            sc = scenes[key]
            oy, ox = int(rng.integers(0, sc.shape[0] - h - 192)), int(rng.integers(0, sc.shape[1] - w - 192))
            sub = sc[oy:oy + h + 192, ox:ox + w + 192]
            frames = np.stack([synth.warp_frame(sub, w, h, t + i) for i in range(2)])
            g = np.random.default_rng(fseed)
            if kind == "noise":
                frames = g.integers(0, 256, frames.shape, dtype=np.uint8)
            elif kind == "blocks":
                s_ = int(g.integers(5, 40))
                b0 = ((np.add.outer(np.arange(h) // s_, np.arange(w) // s_) % 2) * int(g.integers(20, 200)) + 20).astype(np.int32)
                frames = np.stack([np.clip(b0 + g.integers(-2, 3, b0.shape), 0, 255).astype(np.uint8) for _ in range(2)])
            elif kind == "lowcontrast":
                frames = (frames.astype(np.int32) // 8 + 100).astype(np.uint8)
            elif kind == "scene2x":
                frames = np.clip((frames.astype(np.int32) - 128) * 3 + 128, 0, 255).astype(np.uint8)
            want = [ref(f) for f in frames]
        except Exception:
            skipped += 1
            continue                                   # geometry the reference cannot handle (a level too small)
        ex = None
        try:
            path_mask(reset=True)
            ex = ORBextractor(nf, scale, nlev, ini, mn, max_w=w, max_h=h, max_batch=2 * reps)
            ks, ds = ex.extract_batch(np.concatenate([frames] * reps))
        except Exception as e:
            if ex is not None:
                ex.close()
            if "too small" in str(e) or "too large for this number of levels" in str(e):   # documented limits
                skipped += 1
                continue
            raise
        m = path_mask()
        seen |= m
        cfg = (w, h, nf, nlev, scale, ini, mn, fseed, reps, kind)
        for b in range(2 * reps):
            assert ks[b].tobytes() == want[b % 2][0].tobytes() and np.array_equal(ds[b], want[b % 2][1]), \
                "extraction differs from the oracle: config %r frame %d (paths %s)" % (cfg, b, paths_of(m))
        kp_total += sum(len(k) for k in ks)
        if aim is not None and m >> PATH_BITS[aim] & 1:
            aimed[aim] += 1
        for name in ("k_fast", "k_resize<32>/<8>", "k_quadtree(global)"):
            if aim is None and m >> PATH_BITS[name] & 1 and (name != "k_resize<32>/<8>" or reps > 1):
                aimed[name] += 1
        ex.close()
        n += 1
    return {"configs": n, "skipped": skipped, "keypoints": kp_total, "paths": paths_of(seen), "reached": aimed,
            "seconds": round(time.time() - t0, 1)}


def test_random_extraction_configurations_match_oracle(oracle, tally):
    s = run_extraction_configs(800, 20261)
    assert s["configs"] == 800
    # the fallback variants were really launched (and gave the oracle's result): generic-grid FAST, tiled resize in a batch,
    # quadtree tables in global memory
    assert s["reached"]["k_fast"] >= 20 and s["reached"]["k_quadtree(global)"] >= 10 and s["reached"]["k_resize<32>/<8>"] >= 4, s
    assert "k_fast_fix" in s["paths"] and "k_resize_fit" in s["paths"] and "k_pyramid_chain" in s["paths"], s
    # batches blur inside the describe kernel (r06); a frame or two keep k_blur + k_describe
    assert "k_describe_blur" in s["paths"] and "k_describe" in s["paths"] and "k_blur" in s["paths"], s
    tally("extraction configs (shipped switches default)", s["configs"])
    tally("extraction keypoints compared", s["keypoints"])
    for k, v in s["reached"].items():
        tally("extraction configs that ran " + k, v)


@pytest.mark.parametrize("env", [{"ORBHIP_FAST_FIX": "0", "ORBHIP_NO_CHAIN": "1"}, {"ORBHIP_NO_GRAPH": "1", "ORBHIP_RESIZE_FIT": "0"},
                                 {"ORBHIP_DESCRIBE_FUSED": "0"}],
                         ids=["fast_generic+no_chain", "no_graph+resize_tiles", "blur_and_describe_as_two_kernels"])
def test_random_extraction_configurations_other_switches(env, tally):
    """The same generator in a child process under the shipped switches' other positions (k_fast for every grid, one launch per
    pyramid level, no hipGraph) and with the fitted resize tiles off (k_resize<32> for every batch; ablation build), and with the blur as a kernel of its own for batches
    too (k_blur + k_describe, the path of rounds 1-5; ablation build)."""
    p = subprocess.run([sys.executable, os.path.abspath(__file__), "100", "777"], env=dict(os.environ, **env), capture_output=True, text=True,
                       timeout=600)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-2000:]
    s = json.loads(p.stdout.strip().splitlines()[-1])
    assert s["configs"] == 100
    if "ORBHIP_FAST_FIX" in env:
        assert "k_fast" in s["paths"] and "k_fast_fix" not in s["paths"] and "k_pyramid_chain" not in s["paths"], s
    elif "ORBHIP_DESCRIBE_FUSED" in env:
        assert "k_describe_blur" not in s["paths"] and "k_describe" in s["paths"] and "k_blur" in s["paths"], s
    else:
        assert "k_resize_fit" not in s["paths"] and "k_resize<32>/<8>" in s["paths"], s
    tally("extraction configs (other switch positions)", s["configs"])


if __name__ == "__main__":          # child process of test_random_extraction_configurations_other_switches
    ROOT = os.path.dirname(HERE)
    for p_ in (os.path.join(ROOT, "vi-orb-slam-icra2018_amd"), os.path.join(ROOT, "oracle"), HERE):
        sys.path.insert(0, p_)
    print(json.dumps(run_extraction_configs(int(sys.argv[1]), int(sys.argv[2]))))


# ------------------------------------------------------------------------------------------------------------------------
# (d) k_describe_blur at the image borders (r06): the reflected columns / rows of the blur and the scalar tail of its rounding
# ------------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("w,h", [(640, 480), (641, 363), (642, 301), (643, 480), (527, 241)])
def test_describe_blur_border_keypoints_match_oracle(oracle, w, h, tally):
    """Frames with strong corners planted along all four borders of every level -- keypoints 19 to 24 pixels from an edge, whose 43 x 43
    blur neighbourhood leaves the image (BORDER_REFLECT_101, ref src/ORBextractor.cc:1104), on widths with w % 4 = 0 .. 3 (the columns
    from w - w % 4 on take the scalar tail's rounding of OpenCV's column filter) -- through the batch path (k_describe_blur) and as
    single frames (k_blur + k_describe), against the oracle; the test counts how many keypoints exercised each border path."""
    from orbhip.extractor import ORBextractor
    rng = np.random.default_rng(w * 1000 + h)
    frames = []
    for f in range(2):
        img = rng.integers(90, 110, (h, w)).astype(np.int32)
        # bright / dark squares whose corners fall 19 .. 24 pixels from the borders at several scales (so that higher levels see them too)
        for s in (1.0, 1.2, 1.44, 1.728, 2.0736, 2.488, 2.986, 3.583):
            d = int(round(21 * s)) + f
            size = max(3, int(round(5 * s)))
            for t in range(d, max(w, h), int(26 * s) + 3):
                for (y, x) in ((d, t), (h - 1 - d, t), (t, d), (t, w - 1 - d)):
                    if 0 <= y - size and y + size < h and 0 <= x - size and x + size < w:
                        img[y - size:y, x - size:x] += 70 if (t // 7) % 2 else -70
        frames.append(np.clip(img + rng.integers(-3, 4, img.shape), 0, 255).astype(np.uint8))
    frames = np.stack(frames)
    # as many features as the batch path still describes with k_describe_blur (up to 2.0 keypoints per 1000 pyramid pixels)
    P0 = oracle.params(1000)
    NF = min(1500, sum(a * b for a, b in (oracle.level_size(P0, w, h, l) for l in range(8))) // 600)
    ref = oracle.Extractor(NF)
    want = [ref(f) for f in frames]
    path_mask(reset=True)
    ex = ORBextractor(NF, max_w=w, max_h=h, max_batch=8)
    ks, ds = ex.extract_batch(np.concatenate([frames] * 4))
    assert path_mask() >> PATH_BITS["k_describe_blur"] & 1
    for b in range(8):
        assert ks[b].tobytes() == want[b % 2][0].tobytes() and np.array_equal(ds[b], want[b % 2][1]), "batch frame %d (%d x %d)" % (b, w, h)
    for b in range(2):                               # a frame at a time: k_blur + k_describe
        k1, d1 = ex(frames[b])
        assert k1.tobytes() == want[b][0].tobytes() and np.array_equal(d1, want[b][1])
    ex.close()
    # how many keypoints took which border path of k_describe_blur (level coordinates)
    left = right = tail = rows = 0
    for k, _ in want:
        for l in range(8):
            lw, lh = oracle.level_size(ref.params, w, h, l)
            sel = k[k["octave"] == l]
            sc = float(ref.params.mvScaleFactor[l])
            cx = np.rint(sel["x"] / np.float32(sc)).astype(np.int64)
            cy = np.rint(sel["y"] / np.float32(sc)).astype(np.int64)
            left += int((cx < 23).sum())
            right += int((cx + 21 >= lw).sum())
            tail += int((cx + 18 >= lw - lw % 4).sum()) if lw % 4 else 0
            rows += int(((cy < 21) | (cy + 21 >= lh)).sum())
    assert left > 3 and right > 3 and rows > 6, (left, right, tail, rows)
    tally("describe_blur border keypoints: columns reflected", left + right)
    tally("describe_blur border keypoints: rows reflected", rows)
    tally("describe_blur border keypoints: scalar-tail columns", tail)
