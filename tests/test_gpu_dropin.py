"""GPU test of the C++ drop-in classes (include/orbhip/ORBextractor.h, ORBmatcher.h): a C++ program
that calls them the way the reference's Frame.cc / Tracking.cc / LoopClosing.cc do
(tests/native/test_dropin.cpp) against the CPU oracle."""
import os
import struct
import subprocess

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "tests", "native", "test_dropin")


def _fv(node, n):
    ids = sorted(set(int(v) for v in node[:n]))
    lists = [np.nonzero(node[:n] == k)[0] for k in ids]
    off = np.concatenate([[0], np.cumsum([len(x) for x in lists])]).astype(np.int32)
    return np.array(ids, np.int32), off, np.concatenate(lists).astype(np.int32)


def test_cpp_dropin_classes_match_oracle(oracle, tmp_path):
    from orbhip import capi, synth
    assert capi.load().orbhip_device_count() > 0
    assert os.path.exists(EXE), "tests/native/test_dropin is not built (run __graft_entry__.build())"
    W, H, NF = 752, 480, 1000
    frames = synth.make_frames(50, W, H, 2)
    rng = np.random.default_rng(51)
    # vocabulary-node assignment: frame features at random, key-frame features mostly in the node
    # of their nearest frame descriptor (so that SearchByBoW has real matches to find)
    pre = oracle.Extractor(NF)
    (_, pd1), (_, pd2) = pre(frames[0]), pre(frames[1])
    node = rng.integers(0, 90, 8192).astype(np.int32)
    nearest = oracle.knn2(pd1, pd2)[0]
    follow = rng.random(len(pd1)) < 0.85
    node[:len(pd1)][follow] = node[4096:][nearest[follow]]
    (tmp_path / "frames.raw").write_bytes(frames.tobytes())
    (tmp_path / "groups.bin").write_bytes(node.tobytes())
    out = tmp_path / "out.bin"
    from orbhip import distributed as D
    blob = D.make_synthetic_vocabulary(52, k=10, L=5)
    (tmp_path / "voc.bin").write_bytes(blob)
    subprocess.check_call([EXE, str(W), str(H), str(NF), str(tmp_path / "frames.raw"), str(tmp_path / "groups.bin"),
                           str(out), str(tmp_path / "voc.bin")])
    buf = out.read_bytes()
    pos = 0

    def take(fmt):
        nonlocal pos
        v = struct.unpack_from(fmt, buf, pos)
        pos += struct.calcsize(fmt)
        return v

    ref = oracle.Extractor(NF)
    P = ref.params
    nlev, = take("<i")
    sf, = take("<f")
    assert nlev == 8 and sf == np.float32(1.2)
    for name in ("mvScaleFactor", "mvInvScaleFactor", "mvLevelSigma2", "mvInvLevelSigma2"):
        got = np.array(take("<8f"), np.float32)
        assert np.array_equal(got, np.array(list(getattr(P, name))[:8], np.float32)), name
    res = []
    for fi in range(2):
        n, = take("<i")
        kps = np.frombuffer(buf, oracle.KP_DTYPE, n, pos).copy()
        pos += n * 28
        desc = np.frombuffer(buf, np.uint8, n * 32, pos).reshape(n, 32).copy()
        pos += n * 32
        rk, rd = ref(frames[fi])
        assert n == len(rk) and kps.tobytes() == rk.tobytes() and np.array_equal(desc, rd)
        for l in range(8):
            w, h, chk = take("<iiQ")
            lvl = ref.pyramid(l)
            assert (h, w) == lvl.shape
            flat = lvl.reshape(-1).astype(np.uint64)
            want = int((flat * (np.arange(len(flat), dtype=np.uint64) % 251 + 1)).sum())
            assert want == chk, "mvImagePyramid level %d" % l
        t = take("<3d")
        assert all(x > 0 for x in t)
        res.append((kps, desc))
    (k1, d1), (k2, d2) = res
    n1, n2 = len(k1), len(k2)
    valid1 = np.array([0 if (i % 7 == 3 or i % 11 == 5) else 1 for i in range(n1)], np.uint8)
    valid2 = np.array([0 if i % 5 == 1 else 1 for i in range(n2)], np.uint8)
    fv1, fv2 = _fv(node, n1), _fv(node[4096:], n2)
    # SearchByBoW(KeyFrame*, Frame&)
    nm, cnt = take("<ii")
    got = np.array(take("<%di" % cnt), np.int32)
    wn, w12, w21 = oracle.search_by_bow(d1, valid1, k1["angle"], fv1, d2, None, k2["angle"], fv2, th=50, th_mode=0,
                                        nnratio=0.7, check_ori=True)
    assert nm == wn and cnt == n2 and np.array_equal(got, w21) and nm > 50
    # SearchByBoW(KeyFrame*, KeyFrame*)
    nm, cnt = take("<ii")
    got = np.array(take("<%di" % cnt), np.int32)
    wn, w12, w21 = oracle.search_by_bow(d1, valid1, k1["angle"], fv1, d2, valid2, k2["angle"], fv2, th=50, th_mode=1,
                                        nnratio=0.75, check_ori=True)
    want = np.where(w12 >= 0, w12 + 4096, -1)
    assert nm == wn and cnt == n1 and np.array_equal(got, want) and nm > 50
    d, = take("<i")
    assert d == oracle.descriptor_distance(d1[0], d2[0])
    # ORBVocabulary::loadFromBinaryFile + transform(features, BowVector, FeatureVector, 4)
    ok, nwords = take("<ii")
    V = oracle.Vocabulary(blob)
    assert ok == 1 and nwords == V.nwords
    w, wt, nid = V.transform(d1, 4)
    bw, bv = V.bow(w, wt)
    nb, = take("<i")
    got_w, got_v = [], []
    for _ in range(nb):
        a, b = take("<id")
        got_w.append(a)
        got_v.append(b)
    assert got_w == bw.tolist() and got_v == bv.tolist()          # doubles, same summation order
    ids, off, idx = oracle.feature_vector(nid, wt)
    nfv, = take("<i")
    assert nfv == len(ids)
    for g in range(nfv):
        node_id, cnt = take("<ii")
        members = list(take("<%di" % cnt))
        assert node_id == ids[g] and members == idx[off[g]:off[g + 1]].tolist()
    assert pos == len(buf)


def test_cpp_stereo_frame_matches_oracle(oracle, tmp_path):
    """Two drop-in extractors on two host threads + Frame::ComputeStereoMatches (src/Frame.cc:413-437)."""
    from orbhip import synth
    from orbhip.capi import KP_DTYPE
    exe = os.path.join(ROOT, "tests", "native", "test_stereo_dropin")
    assert os.path.exists(exe), "tests/native/test_stereo_dropin is not built (run __graft_entry__.build())"
    W, H, NF, mb, mbf = 1241, 376, 2000, 0.53716, 386.1448
    L, R = synth.make_stereo_pair(60, W, H, disparity=23)
    (tmp_path / "pair.raw").write_bytes(L.tobytes() + R.tobytes())
    out = tmp_path / "stereo.bin"
    subprocess.check_call([exe, str(W), str(H), str(NF), repr(mb), repr(mbf), str(tmp_path / "pair.raw"), str(out)])
    buf = out.read_bytes()
    n, nr = struct.unpack_from("<ii", buf, 0)
    kps = np.frombuffer(buf, KP_DTYPE, n, 8)
    u = np.frombuffer(buf, np.float32, n, 8 + 28 * n)
    z = np.frombuffer(buf, np.float32, n, 8 + 32 * n)
    oL, oR = oracle.Extractor(NF), oracle.Extractor(NF)
    kL, dL = oL(L)
    kR, dR = oR(R)
    assert n == len(kL) and nr == len(kR) and kps.tobytes() == kL.tobytes()
    ru, rz, rn = oracle.stereo_matches(oL, kL, dL, oR, kR, dR, np.float32(mb), np.float32(mbf))
    assert u.tobytes() == ru.tobytes() and z.tobytes() == rz.tobytes() and (ru >= 0).sum() > 300
