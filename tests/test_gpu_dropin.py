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
    # SearchForTriangulation(&kf, &kf2, F12, vMatchedIndices, false): the wrapper's epipole restated (gemm: double
    # accumulation, one rounding), then the oracle
    f32 = np.float32
    ang = f32(0.01)
    import ctypes
    libm = ctypes.CDLL("libm.so.6")
    libm.cosf.restype = libm.sinf.restype = ctypes.c_float
    libm.cosf.argtypes = libm.sinf.argtypes = [ctypes.c_float]
    ca, sa = f32(libm.cosf(ang)), f32(libm.sinf(ang))
    R2 = np.array([[ca, 0, sa], [0, 1, 0], [-sa, 0, ca]], f32)
    t2 = np.array([-0.2, 0.01, 0.05], f32)
    C2 = np.array([f32(sum(np.float64(R2[r, k]) * 0.0 for k in range(3)) + np.float64(t2[r])) for r in range(3)], f32)
    invz = f32(f32(1.0) / C2[2])
    exx = f32(f32(f32(f32(458.654) * C2[0]) * invz) + f32(367.215))
    eyy = f32(f32(f32(f32(457.296) * C2[1]) * invz) + f32(248.375))
    Frows = np.array([[0, 0, 0], [0, 0, -1], [0, 1, 0]], f32)
    skip1 = np.array([0 if i % 7 == 3 else 1 for i in range(n1)], np.uint8)     # GetMapPoint(i) != NULL (bad ones too)
    skip2 = np.array([0 if i % 5 == 1 else 1 for i in range(n2)], np.uint8)
    sfv = np.array(list(P.mvScaleFactor)[:8], f32)
    s2v = np.array(list(P.mvLevelSigma2)[:8], f32)
    for ori in (False, True):
        nm, cnt = take("<ii")
        pairs = np.array(take("<%di" % (2 * cnt)), np.int32).reshape(-1, 2)
        wn, wm = oracle.search_for_triangulation(k1, d1, skip1, fv1, k2, d2, skip2, fv2, Frows, exx, eyy, sfv, s2v,
                                                 u_right1=np.full(n1, -1, f32), u_right2=np.full(n2, -1, f32), check_ori=ori)
        want = np.stack([np.nonzero(wm >= 0)[0], wm[wm >= 0]], 1)
        assert nm == wn and cnt == wn and np.array_equal(pairs, want) and wn > 3
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


def test_cpp_grid_and_search_by_projection_match_oracle(oracle, tmp_path):
    """Frame::AssignFeaturesToGrid / GetFeaturesInArea and both ORBmatcher::SearchByProjection drop-ins, driven
    the way Tracking drives them (tests/native/test_guided_dropin.cpp), against the oracle; the pose arithmetic
    of the wrapper is restated here in numpy (double accumulation, one rounding, as cv::gemm does for floats)."""
    from orbhip import guided, synth
    exe = os.path.join(ROOT, "tests", "native", "test_guided_dropin")
    assert os.path.exists(exe), "tests/native/test_guided_dropin is not built (run __graft_entry__.build())"
    f32 = np.float32
    W, H, NF = 640, 480, 1200
    frames = synth.make_frames(80, W, H, 2)
    ex = oracle.Extractor(NF)
    (k0, d0), (k1, d1) = ex(frames[0]), ex(frames[1])
    n0, n1 = len(k0), len(k1)
    rng = np.random.default_rng(81)
    fx, fy, cx, cy = f32(517.3), f32(516.5), f32(318.6), f32(255.3)
    bounds = (f32(0), f32(W), f32(0), f32(H))
    mb, mbf, th_last, th_local = f32(0.08), f32(40.0), f32(15), f32(3)
    ang = 0.004
    Rc = np.array([[np.cos(ang), 0, np.sin(ang)], [0, 1, 0], [-np.sin(ang), 0, np.cos(ang)]]).astype(f32)
    tc = np.array([0.012, -0.006, 0.03], f32)
    Tl = np.eye(4, dtype=f32)
    Tc = np.eye(4, dtype=f32)
    Tc[:3, :3], Tc[:3, 3] = Rc, tc
    z = rng.uniform(2, 10, n0).astype(f32)
    Xw = np.stack([(k0["x"] - cx) / fx * z, (k0["y"] - cy) / fy * z, z], 1).astype(f32)
    Xw[5, 2] = f32(-1.0)                                            # behind the camera -> skipped
    nobs = rng.integers(0, 4, n0)
    outlier = rng.random(n0) < 0.05
    rec = np.zeros((n0, 14), f32)
    rec[:, 0:3], rec[:, 3], rec[:, 4] = Xw, nobs, outlier
    rec[:, 5] = k0["x"] + rng.normal(0, 1.5, n0)                    # mTrackProjX / Y / XR
    rec[:, 6] = k0["y"] + rng.normal(0, 1.5, n0)
    rec[:, 7] = rec[:, 5] - 10
    rec[:, 8] = rng.uniform(0.99, 1.0, n0)                          # mTrackViewCos
    rec[:, 9] = k0["octave"]                                        # mnTrackScaleLevel
    rec[:, 10] = rng.random(n0) < 0.9                               # mbTrackInView
    rec[:, 11] = rng.random(n0) < 0.03                              # isBad
    # scale-invariance distances (MapPoint::UpdateNormalAndDepth): maxDistance = dist * 1.2^octave; here relative to the
    # CURRENT camera centre, with a factor that keeps PredictScale away from its rounding boundaries
    Ow = np.array([f32(-np.sum(Rc[:, r].astype(np.float64) * tc.astype(np.float64))) for r in range(3)], f32)
    PO = (Xw - Ow).astype(f32)
    dist3D = np.sqrt((PO.astype(np.float64) ** 2).sum(1)).astype(f32)
    sfac = np.array(list(ex.params.mvScaleFactor)[:8], f32)
    rec[:, 13] = dist3D * sfac[k0["octave"]] * rng.uniform(0.85, 0.98, n0).astype(f32)
    rec[:, 12] = rec[:, 13] / sfac[7]
    far = rng.random(n0) < 0.05
    rec[far, 13] *= f32(0.05)                                       # dist3D > 1.2 * maxDistance -> skipped
    dist = np.array([-0.05, 0.012, 2e-4, -3e-4], f32)              # mDistCoef: keypoints move by up to a few pixels
    hdr = np.concatenate([[fx, fy, cx, cy, 0, 0, 0, 0, mb, mbf, th_last, th_local, 1], Tl.ravel(), Tc.ravel(),
                          dist]).astype(f32)
    (tmp_path / "frames.raw").write_bytes(frames.tobytes())
    (tmp_path / "params.bin").write_bytes(hdr.tobytes() + rec.tobytes())
    out = tmp_path / "guided.bin"
    subprocess.check_call([exe, str(W), str(H), str(NF), str(tmp_path / "frames.raw"), str(tmp_path / "params.bin"), str(out)])
    buf = np.frombuffer(out.read_bytes(), np.int32)
    pos = 0

    def take(n=1):
        nonlocal pos
        v = buf[pos:pos + n]
        pos += n
        return v

    # Frame::ComputeImageBounds and UndistortKeyPoints (src/Frame.cc:748-808)
    Kmat = np.array([fx, 0, cx, 0, fy, cy, 0, 0, 1], f32)
    corners = oracle.undistort_points(np.array([[0, 0], [W, 0], [0, H], [W, H]], f32), Kmat, dist, Kmat)
    bounds = (min(corners[0, 0], corners[2, 0]), max(corners[1, 0], corners[3, 0]), min(corners[0, 1], corners[1, 1]),
              max(corners[2, 1], corners[3, 1]))
    assert take(4).view(f32).tolist() == [float(b) for b in bounds]
    assert take()[0] == n1
    from orbhip.capi import KP_DTYPE
    kun = take(7 * n1).view(KP_DTYPE)
    und = oracle.undistort_points(np.stack([k1["x"], k1["y"]], 1), Kmat, dist, Kmat)
    k1u = k1.copy()
    k1u["x"], k1u["y"] = und[:, 0], und[:, 1]
    assert kun.tobytes() == k1u.tobytes() and np.abs(k1u["x"] - k1["x"]).max() > 0.5
    und0 = oracle.undistort_points(np.stack([k0["x"], k0["y"]], 1), Kmat, dist, Kmat)
    k0u = k0.copy()
    k0u["x"], k0u["y"] = und0[:, 0], und0[:, 1]
    k1 = k1u                                                       # everything below runs on mvKeysUn
    gp = oracle.grid_params(*bounds)
    roff, ridx = oracle.grid_build(k1, gp)
    for c in range(64 * 48):
        cnt = take()[0]
        assert cnt == roff[c + 1] - roff[c] and np.array_equal(take(cnt), ridx[roff[c]:roff[c + 1]])
    for (x, y, r, mn, mx) in [(310.5, 200.25, 45.0, 1, 3), (20.0, 470.0, 60.0, -1, -1)]:
        cnt = take()[0]
        assert np.array_equal(take(cnt), oracle.features_in_area(k1, (roff, ridx), gp, x, y, r, mn, mx))

    # KeyFrame grid twin: copied grid, IsInImage, GetFeaturesInArea without a level filter
    assert take()[0] == 1 and take(2).tolist() == [1, 0]
    for (x, y, r) in [(310.5, 200.25, 45.0), (20.0, 470.0, 60.0)]:
        cnt = take()[0]
        assert np.array_equal(take(cnt), oracle.features_in_area(k1, (roff, ridx), gp, x, y, r, -1, -1))

    # --- SearchByProjection(CurrentFrame, LastFrame, th, bMono): the wrapper's projection, restated ---
    def affine(R, x, t):                                           # float(double sum + double t)
        s = np.zeros(x.shape[:-1] + (3,), np.float64)
        for r in range(3):
            acc = np.zeros(x.shape[:-1], np.float64)
            for k in range(3):
                acc = acc + np.float64(R[r, k]) * x[..., k].astype(np.float64)
            s[..., r] = acc + np.float64(t[r])
        return s.astype(f32)
    xc = affine(Rc, Xw, tc)
    with np.errstate(divide="ignore"):
        invz = (1.0 / xc[:, 2].astype(np.float64)).astype(f32)
    u = fx * xc[:, 0] * invz + cx
    v = fy * xc[:, 1] * invz + cy
    has_point = (np.arange(n0) % 9 != 4)
    valid = has_point & ~outlier & ~(invz < 0) & ~(u < bounds[0]) & ~(u > bounds[1]) & ~(v < bounds[2]) & ~(v > bounds[3])
    sf = (f32(1.2) ** np.arange(8)).astype(f32)
    sf = np.array(list(ex.params.mvScaleFactor)[:8], f32)
    qB = guided.queries_for_last_frame(u, v, u - mbf * invz, k0["octave"], k0u["angle"], valid, nobs > 0, th_last, sf)
    assert valid.sum() > 800 and not valid[5]
    rn, rm = oracle.search_by_projection(k1, d1, gp, qB, d0, use_ratio=False, nnratio=0.9, check_ori=True)
    n = take()[0]
    got = take(n1)
    exp = np.where(rm >= 0, rm, -1)
    assert n == rn and rn > 500 and np.array_equal(got, exp)

    # --- SearchByProjection(F, vpMapPoints, th) on top of it: points in reverse order ---
    order = np.arange(n0)[::-1]
    qA = guided.queries_for_map_points(rec[order, 5], rec[order, 6], rec[order, 7], rec[order, 8], rec[order, 9].astype(np.int32),
                                       (rec[order, 10] != 0) & (rec[order, 11] == 0), nobs[order] > 0, th_local, sf)
    occupied = np.array([1 if (e >= 0 and nobs[e] > 0) else 0 for e in exp], np.uint8)
    rn2, rm2 = oracle.search_by_projection(k1, d1, gp, qA, d0[order], occupied=occupied, use_ratio=True, nnratio=0.8)
    n2 = take()[0]
    got2 = take(n1)
    exp2 = np.where(rm2 >= 0, order[np.maximum(rm2, 0)], exp)
    assert n2 == rn2 and rn2 > 50 and np.array_equal(got2, exp2)

    # --- SearchByProjection(CurrentFrame, pKF, sAlreadyFound, 10, 100): relocalisation on top of that state ---
    import ctypes
    libm = ctypes.CDLL("libm.so.6")
    libm.logf.restype, libm.logf.argtypes = ctypes.c_float, [ctypes.c_float]
    state = exp2.copy()
    state[np.arange(n1) % 3 == 0] = -1
    found = set(state[state >= 0].tolist())
    logs = f32(libm.logf(f32(sfac[1])))
    assert sfac[1] == f32(1.2)
    qC = np.zeros(n0, guided.QUERY_DTYPE)
    nactive = 0
    for i in range(n0):
        if not has_point[i] or rec[i, 11] != 0 or i in found:
            continue
        if u[i] < bounds[0] or u[i] > bounds[1] or v[i] < bounds[2] or v[i] > bounds[3]:
            continue
        if dist3D[i] < f32(0.8) * rec[i, 12] or dist3D[i] > f32(1.2) * rec[i, 13]:
            continue
        ratio = f32(rec[i, 13] / dist3D[i])
        lv = f32(libm.logf(ratio)) / logs
        assert abs(lv - np.round(lv)) > 0.02                       # PredictScale is not at a rounding boundary
        lvl = min(max(int(np.ceil(lv)), 0), 7)
        qC[i] = (u[i], v[i], f32(10) * sfac[lvl], 0, lvl - 1, lvl + 1, k0u["angle"][i], 3)
        nactive += 1
    assert nactive > 300 and far.sum() > 10
    rn4, rm4 = oracle.search_by_projection(k1, d1, gp, qC, d0, occupied=(state >= 0).astype(np.uint8), use_ratio=False,
                                           nnratio=0.9, check_ori=True, th_high=100)
    n4 = take()[0]
    got4 = take(n1)
    exp4 = np.where(rm4 >= 0, rm4, np.where(rm4 == -2, -1, state))
    assert n4 == rn4 and rn4 > 100 and np.array_equal(got4, exp4)

    # --- SearchForInitialization(mInitialFrame, mCurrentFrame, mvbPrevMatched, mvIniMatches, 100), twice ---
    prev = np.stack([k0u["x"], k0u["y"]], 1).astype(f32)
    for _ in range(2):
        rn3, rm3, prev = oracle.search_for_initialization(k0u, d0, k1, d1, gp, prev, 100, 0.9, True)
        n3 = take()[0]
        assert n3 == rn3 and rn3 > 100 and np.array_equal(take(n0), rm3)
        assert take(2 * n0).view(f32).tobytes() == prev.tobytes()
    assert pos == len(buf)


@pytest.mark.gpu
@pytest.mark.parametrize("no_sets", ["0", "1"])
def test_cpp_fuse_sim3_and_keyframe_projection_match_host_restatement(tmp_path, no_sets):
    """ORBmatcher::Fuse (both forms), SearchByProjection(KeyFrame*, Scw, ...) and SearchBySim3 through the drop-in
    classes (tests/native/test_fuse_dropin.cpp): two identical worlds, one through the HIP path and one through the
    routines restated on the host in that program (src/ORBmatcher.cc:290-403, 825-1326); map states must be equal.
    Also the batched ComputeDistinctiveDescriptors helper against src/MapPoint.cc:283-349 restated.
    Both with the key frames resident on the device (orbhip_set_*, the default) and with ORBHIP_NO_SETS=1 (upload per call)."""
    from orbhip import synth
    exe = os.path.join(ROOT, "tests", "native", "test_fuse_dropin")
    assert os.path.exists(exe), "tests/native/test_fuse_dropin is not built (run __graft_entry__.build())"
    W, H = 640, 480
    frame = synth.make_frames(90, W, H, 1)[0]
    (tmp_path / "frame.raw").write_bytes(frame.tobytes())
    r = subprocess.run([exe, str(W), str(H), "1500", str(tmp_path / "frame.raw")], capture_output=True, text=True,
                       env=dict(os.environ, ORBHIP_NO_SETS=no_sets))
    assert r.returncode == 0, r.stdout + r.stderr
    assert r.stdout.count(": ok") == 13 and "all ok" in r.stdout, r.stdout


def test_cpp_dropins_do_not_throw_and_load_text_vocabularies(oracle, tmp_path):
    """hiperror.h: a failed device call becomes "nothing found" + a message, never an exception (the reference's callers
    have no try block); ORBVocabulary::loadFromTextFile + transform against the committed text fixture."""
    from orbhip import distributed as D, synth
    exe = os.path.join(ROOT, "tests", "native", "test_nothrow_dropin")
    assert os.path.exists(exe), "tests/native/test_nothrow_dropin is not built (run __graft_entry__.build())"
    W, H = 640, 480
    frame = synth.make_frames(53, W, H, 1)[0]
    (tmp_path / "frame.raw").write_bytes(frame.tobytes())
    gold = os.path.join(ROOT, "tests", "golden")
    g = np.load(os.path.join(gold, "vocab_k4_L2_text.npz"))
    (tmp_path / "desc.bin").write_bytes(g["desc"].tobytes())
    r = subprocess.run([exe, str(W), str(H), str(tmp_path / "frame.raw"), os.path.join(gold, "vocab_k4_L2.txt"),
                        str(tmp_path / "desc.bin")], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    kv, bow, fv = {}, [], {}
    for line in r.stdout.splitlines():
        key, _, val = line.partition("=")
        if key == "bow":
            w, v = val.split()
            bow.append((int(w), float(v)))
        elif key == "fv":
            t = [int(x) for x in val.split()]
            fv[t[0]] = t[1:]
        else:
            kv[key] = val
    assert kv["done"] == "1"
    assert (kv["tiny_keys"], kv["tiny_desc_rows"], kv["tiny_errors"]) == ("0", "0", "1") and "ORBextractor" in kv["tiny_msg"]
    assert "[orbhip]" in r.stderr                                   # logged once on stderr
    rk, rd = oracle.Extractor(1000)(frame)
    assert int(kv["good_keys"]) == len(rk) == int(kv["good_desc_rows"]) and kv["empty_keys_unchanged"] == "1"
    assert (kv["bad_keys"], kv["bad_desc_rows"]) == ("0", "0")
    assert (kv["init_short_prev"], kv["init_assigned"]) == ("0", "0") and int(kv["init_size"]) == len(rk)
    assert (kv["voc_missing"], kv["voc_missing_txt"], kv["voc_bad_txt"], kv["voc_text"], kv["voc_words"]) == ("0", "0", "0", "1", "16")
    assert [w for w, _ in bow] == g["bow_word"].tolist() and np.array_equal(np.array([v for _, v in bow]), g["bow_value"])
    ofv = oracle.feature_vector(g["node"])
    assert sorted(fv) == ofv[0].tolist()
    for i, nid in enumerate(ofv[0]):
        assert fv[int(nid)] == ofv[2][ofv[1][i]:ofv[1][i + 1]].tolist()
    assert int(kv["errors_total"]) >= 3


@pytest.mark.gpu
@pytest.mark.parametrize("no_sets", ["0", "1"])
def test_cpp_matcher_from_three_threads_over_the_same_keyframes(tmp_path, no_sets):
    """The reference runs ORBmatcher from Tracking, LocalMapping and LoopClosing at once on shared KeyFrames (src/System.cc:365-375;
    include/ORBmatcher.h:100-101 are plain stack objects).  tests/native/test_threads_dropin.cpp: thread T (SearchByBoW(KF, F) +
    SearchByProjection(F, F)), M (SearchForTriangulation + Fuse) and L (SearchByBoW(KF, KF) + SearchBySim3) over the same 30 key
    frames for 500 rounds, eight resident sets per thread (every round evicts), ORBmatcher::DropResidentSets() from T mid-run;
    every result equal to the single-threaded host restatement (tests/native/host_restate.h).  Once more with ORBHIP_NO_SETS=1."""
    from orbhip import distributed as D, synth
    exe = os.path.join(ROOT, "tests", "native", "test_threads_dropin")
    assert os.path.exists(exe), "tests/native/test_threads_dropin is not built (run __graft_entry__.build())"
    W, H, NF = 640, 480, 3
    frames = synth.make_frames(91, W, H, NF)
    (tmp_path / "frames.raw").write_bytes(np.ascontiguousarray(frames).tobytes())
    (tmp_path / "voc.bin").write_bytes(D.make_synthetic_vocabulary(17, k=10, L=5))
    r = subprocess.run([exe, str(W), str(H), "1200", str(tmp_path / "frames.raw"), str(NF), str(tmp_path / "voc.bin"), "500"],
                       capture_output=True, text=True, env=dict(os.environ, ORBHIP_NO_SETS=no_sets), timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert r.stdout.count(": ok") == 6 and "all ok" in r.stdout, r.stdout
