"""orbhip_frame_build (include/orbhip.h): the Frame constructor's device work -- extraction, UndistortKeyPoints,
AssignFeaturesToGrid, the vocabulary transform (ref: src/Frame.cc:518-572, 574-597, 739-778) -- as one captured graph must
return exactly what the four oracle calls return, on replay as on capture, for every combination of its optional stages;
and the block it leaves on the device must make the same resident set as orbhip_set_put from the host."""
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

# Examples/Monocular/EuRoC.yaml (Camera.*): the calibration the reference's monocular driver reads
K_EUROC = np.array([458.654, 0.0, 367.215, 0.0, 457.296, 248.375, 0.0, 0.0, 1.0], np.float32).reshape(3, 3)
D_EUROC = np.array([-0.28340811, 0.07395907, 0.00019359, 1.76187114e-05], np.float32)


def _want(oracle, ref, V, img, K, D, gp, levelsup):
    k, d = ref(img)
    kun = k.copy()
    if D is not None and len(D) and D[0] != 0:
        xy = oracle.undistort_points(np.stack([k["x"], k["y"]], 1), K, D, K)
        kun["x"], kun["y"] = xy[:, 0], xy[:, 1]
    out = dict(kps=k, kps_un=kun, desc=d)
    if gp is not None:
        out["cell_off"], out["cell_idx"] = oracle.grid_build(kun, gp)
    if levelsup >= 0:
        out["word_id"], out["weight"], out["node_id"] = V.transform(d, levelsup)
    return out


def _same(got, want):
    assert len(got["kps"]) == len(want["kps"]) > 100
    assert got["kps"].tobytes() == want["kps"].tobytes(), "keypoints differ"
    assert got["kps_un"].tobytes() == want["kps_un"].tobytes(), "undistorted keypoints differ"
    assert np.array_equal(got["desc"], want["desc"]), "descriptors differ"
    for f in ("cell_off", "cell_idx", "word_id", "node_id"):
        if f in want:
            assert np.array_equal(got[f], want[f]), f + " differs"
        else:
            assert got[f] is None
    if "weight" in want:
        assert got["weight"].tobytes() == np.asarray(want["weight"], np.float32).tobytes(), "weights differ"


@pytest.mark.parametrize("size", [(640, 480), (752, 480)])
def test_frame_build_equals_the_four_oracle_calls(oracle, size):
    from orbhip import distributed as Dist, synth
    from orbhip.extractor import ORBextractor
    from orbhip.vocabulary import ORBVocabulary
    W, H = size
    frames = synth.make_frames(41, W, H, 4)
    ex = ORBextractor(1000, max_w=W, max_h=H)
    ref = oracle.Extractor(1000)
    blob = Dist.make_synthetic_vocabulary(52, k=10, L=5)
    voc = ORBVocabulary(ex)
    voc.loadFromBinaryBlob(blob)
    V = oracle.Vocabulary(blob)
    # image bounds as Frame::ComputeImageBounds derives them (:780-808)
    corners = np.array([[0, 0], [W, 0], [0, H], [W, H]], np.float32)
    un = oracle.undistort_points(corners, K_EUROC, D_EUROC, K_EUROC)
    gp = oracle.grid_params(min(un[0, 0], un[2, 0]), max(un[1, 0], un[3, 0]), min(un[0, 1], un[1, 1]), max(un[2, 1], un[3, 1]))
    gp0 = oracle.grid_params(0, W, 0, H)
    D0 = np.zeros(4, np.float32)
    # (K, D, grid, levelsup): every stage on; replay with another image; the first frame of a run (no grid yet, no BoW);
    # no distortion (mvKeysUn = mvKeys); grid without BoW; back to the first parameters (a third capture)
    cases = [(K_EUROC, D_EUROC, gp, 4), (K_EUROC, D_EUROC, gp, 4), (K_EUROC, D_EUROC, gp, 4), (K_EUROC, D_EUROC, None, -1),
             (K_EUROC, D0, gp0, 3), (K_EUROC, None, gp0, -1), (K_EUROC, D_EUROC, gp, 4)]
    for i, (K, D, g, lu) in enumerate(cases):
        img = frames[i % len(frames)]
        got = ex.frame_build(img, K, D, g, lu)
        _same(got, _want(oracle, ref, V, img, K, D, g, lu))
    # the separate entry points on the same context still agree (they share the staging and the pyramid buffers)
    k, d = ex(frames[0])
    rk, rd = ref(frames[0])
    assert k.tobytes() == rk.tobytes() and np.array_equal(d, rd)
    # more than 256 replays: the eager refresh of the stage times every 256th call leaves the results alone
    for i in range(260):
        got = ex.frame_build(frames[i & 3], K_EUROC, D_EUROC, gp, 4)
    _same(got, _want(oracle, ref, V, frames[259 & 3], K_EUROC, D_EUROC, gp, 4))
    assert ex.GetTimeOfComputePyramid() >= 0
    ex.close()


def test_frame_becomes_a_resident_set_without_travelling(oracle):
    from orbhip import distributed as Dist, synth
    from orbhip.extractor import ORBextractor, ORBmatcher
    from orbhip.vocabulary import ORBVocabulary
    W, H = 640, 480
    frames = synth.make_frames(43, W, H, 3)
    ex = ORBextractor(1000, max_w=W, max_h=H)
    blob = Dist.make_synthetic_vocabulary(53, k=10, L=4)
    voc = ORBVocabulary(ex)
    voc.loadFromBinaryBlob(blob)
    gp = oracle.grid_params(0, W, 0, H)
    M = ORBmatcher(0.7, True)                      # its own context, as the drop-in's per-thread matcher context
    res, fvs = [], []
    for i, f in enumerate(frames):
        r = ex.frame_build(f, K_EUROC, D_EUROC, gp, 2)
        fv = oracle.feature_vector(r["node_id"], r["weight"])
        assert ex.frame_fingerprint() == ORBmatcher.fingerprint(r["kps_un"], r["desc"]) != 0
        M.put_set_from_frame(10 + i, ex, fv)       # device block -> device block, FeatureVector from the host
        info = M.set_info(10 + i)
        assert info == (len(r["kps"]), len(fv[0]), ORBmatcher.fingerprint(r["kps_un"], r["desc"]))
        res.append(r)
        fvs.append(fv)
    rng = np.random.default_rng(9)
    for a, b in ((0, 1), (1, 2), (2, 0)):
        ra, rb = res[a], res[b]
        v1 = (rng.random(len(ra["kps"])) < 0.8).astype(np.uint8)
        want = oracle.search_by_bow(ra["desc"], v1, ra["kps_un"]["angle"], fvs[a], rb["desc"], None, rb["kps_un"]["angle"], fvs[b],
                                    th=50, th_mode=0, nnratio=0.7, check_ori=True)
        got = M.SearchByBoW_sets(10 + a, v1, len(ra["kps"]), 10 + b, None, len(rb["kps"]))
        assert got[0] == want[0] > 30 and np.array_equal(got[1], want[1]) and np.array_equal(got[2], want[2])
    # the grid travelled too: the Fuse window search into the set equals the per-call one
    from orbhip import guided
    from orbhip.capi import QUERY_DTYPE
    sf = (np.float32(1.2) ** np.arange(8, dtype=np.float32)).astype(np.float32)
    k0, k1 = res[0]["kps_un"], res[1]["kps_un"]
    q = np.zeros(len(k0), QUERY_DTYPE)
    q["u"], q["v"] = k0["x"], k0["y"]
    q["radius"] = 3 * sf[k0["octave"]]
    q["min_level"], q["max_level"], q["flags"] = k0["octave"] - 1, k0["octave"], 1
    sig = (1 / sf ** 2).astype(np.float32)
    want = oracle.window_best(k1, res[1]["desc"], gp, q, res[0]["desc"], None, sig)
    got = guided.WindowBestSet(M._ctx, 11, q, res[0]["desc"], None, sig)
    assert np.array_equal(got[0], want[0]) and np.array_equal(got[1], want[1])
    # a set is what its fingerprint says, not what its key says: the same key with another frame's data is replaced
    assert M.set_info(10)[2] != ORBmatcher.fingerprint(res[1]["kps_un"], res[1]["desc"])
    M.close()
    ex.close()


def test_frame_build_eager_sequence(oracle):
    """ORBHIP_NO_GRAPH=1 (one of the four switches of the shipped library): the same chain without the capture."""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = r'''
import sys, numpy as np
sys.path[:0] = [%r, %r]
import orb_oracle_py as O
from orbhip import synth
from orbhip.extractor import ORBextractor
f = synth.make_frames(44, 640, 480, 2)
ex = ORBextractor(1000, max_w=640, max_h=480)
ref = O.Extractor(1000)
K = np.array([458.654, 0.0, 367.215, 0.0, 457.296, 248.375, 0.0, 0.0, 1.0], np.float32)
D = np.array([-0.28340811, 0.07395907, 0.00019359, 1.76187114e-05], np.float32)
gp = O.grid_params(0, 640, 0, 480)
for img in (f[0], f[1], f[0]):
    r = ex.frame_build(img, K, D, gp, -1)
    k, d = ref(img)
    xy = O.undistort_points(np.stack([k["x"], k["y"]], 1), K, D, K)
    ku = k.copy(); ku["x"], ku["y"] = xy[:, 0], xy[:, 1]
    off, idx = O.grid_build(ku, gp)
    assert r["kps"].tobytes() == k.tobytes(), "kps"
    assert r["kps_un"].tobytes() == ku.tobytes(), "kps_un"
    assert np.array_equal(r["desc"], d), "desc"
    assert np.array_equal(r["cell_off"], off) and np.array_equal(r["cell_idx"], idx), "grid"
print("OK")
''' % (os.path.join(root, "vi-orb-slam-icra2018_amd"), os.path.join(root, "oracle"))
    out = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, ORBHIP_NO_GRAPH="1"), capture_output=True, text=True)
    assert out.returncode == 0 and "OK" in out.stdout, out.stdout + out.stderr


def test_cpp_frame_constructor_in_one_launch(tmp_path):
    """tests/native/test_frame_dropin.cpp: the Frame constructor's sequence through the C++ drop-in classes, every step its own
    device call against ORBextractor::SetFrameBuild (one graph launch) -- identical Frame members, identical SearchByBoW; and
    the identity checks of the resident sets (an id that comes back with other data, a FeatureVector filled later)."""
    from orbhip import distributed as Dist, synth
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = os.path.join(root, "tests", "native", "test_frame_dropin")
    assert os.path.exists(exe), "tests/native/test_frame_dropin is not built (run __graft_entry__.build())"
    W, H = 752, 480
    (tmp_path / "frames.raw").write_bytes(synth.make_frames(45, W, H, 4).tobytes())
    (tmp_path / "voc.bin").write_bytes(Dist.make_synthetic_vocabulary(52, k=10, L=4))
    out = subprocess.run([exe, str(W), str(H), "1000", str(tmp_path / "frames.raw"), "4", str(tmp_path / "voc.bin")],
                         capture_output=True, text=True)
    lines = [l for l in out.stdout.splitlines() if l.strip()]
    assert out.returncode == 0 and len(lines) >= 12 and all(l.endswith(" ok") for l in lines), out.stdout + out.stderr


def test_borrowed_vocabulary_outlives_reload_and_destroy(oracle):
    """orbhip_vocab_share lends the lender's device block.  The lender loading another vocabulary, or being destroyed,
    must leave the borrower on valid (old) tables -- the block is reference-counted -- and orbhip_vocab_generation must tell the
    borrower to share again, after which its frame build runs on the new tables."""
    from orbhip import distributed as Dist, synth
    from orbhip.extractor import ORBextractor
    from orbhip.vocabulary import ORBVocabulary
    img = synth.make_frames(43, 640, 480, 1)[0]
    lender = ORBextractor(1000, max_w=640, max_h=480)
    ex = ORBextractor(1000, max_w=640, max_h=480)
    blobA = Dist.make_synthetic_vocabulary(61, k=10, L=4)
    blobB = Dist.make_synthetic_vocabulary(62, k=9, L=4)
    VA, VB = oracle.Vocabulary(blobA), oracle.Vocabulary(blobB)
    voc = ORBVocabulary(lender)
    voc.loadFromBinaryBlob(blobA)
    assert voc.generation(ex) == 0
    voc.shareWith(ex)
    genA = voc.generation()
    assert genA > 0 and voc.generation(ex) == genA
    r = ex.frame_build(img, levelsup=4)
    wantA = VA.transform(r["desc"], 4)
    assert np.array_equal(r["word_id"], wantA[0]) and np.array_equal(r["node_id"], wantA[2])
    r = ex.frame_build(img, levelsup=4)      # (the replayed graph)
    assert np.array_equal(r["word_id"], wantA[0])
    # the lender loads another vocabulary: the borrower still holds vocabulary A, and can see that it is behind
    voc.loadFromBinaryBlob(blobB)
    assert voc.generation() > genA and voc.generation(ex) == genA
    junk = [ORBextractor(500, max_w=640, max_h=480) for _ in range(2)]   # (allocations that would reuse a freed block)
    r = ex.frame_build(img, levelsup=4)
    assert np.array_equal(r["word_id"], wantA[0]) and np.array_equal(r["node_id"], wantA[2]), "borrowed tables were freed"
    voc.shareWith(ex)
    assert voc.generation(ex) == voc.generation()
    r = ex.frame_build(img, levelsup=4)
    wantB = VB.transform(r["desc"], 4)
    assert np.array_equal(r["word_id"], wantB[0]) and np.array_equal(r["node_id"], wantB[2]), "stale graph after sharing again"
    # the lender goes away altogether
    lender.close()
    for j in junk:
        j.close()
    r = ex.frame_build(img, levelsup=4)
    assert np.array_equal(r["word_id"], wantB[0]) and np.array_equal(r["node_id"], wantB[2])
    ex.close()
