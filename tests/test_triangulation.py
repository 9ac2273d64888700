"""ORBmatcher::SearchForTriangulation (src/ORBmatcher.cc:657-827, CheckDistEpipolarLine :140-157): the oracle against a
definition-level Python model on CPU, the HIP path against the oracle on the GPU."""
import numpy as np
import pytest

F_ROWS = np.array([[0, 0, 0], [0, 0, -1], [0, 1, 0]], np.float32)      # pure horizontal translation: l = (0, 1, -y1)


def _scene(oracle, seed=5, nf=1500, w=640, h=480):
    from orbhip import distributed as D, synth
    L, R = synth.make_stereo_pair(seed, w, h, disparity=21)
    ex = oracle.Extractor(nf)
    (k1, d1), (k2, d2) = ex(L), ex(R)
    voc = oracle.Vocabulary(D.make_synthetic_vocabulary(77, k=6, L=4))
    g = []
    for d in (d1, d2):
        _, wt, nid = voc.transform(d, 2)
        g.append(oracle.feature_vector(nid, wt))
    sf = np.array(list(ex.params.mvScaleFactor)[:8], np.float32)
    s2 = np.array(list(ex.params.mvLevelSigma2)[:8], np.float32)
    return k1, d1, g[0], k2, d2, g[1], sf, s2


def _model(k1, d1, skip1, g1, k2, d2, skip2, g2, F, ex, ey, sf, s2, ur1, ur2, only_stereo, check_ori):
    f32 = np.float32
    b1, b2 = np.unpackbits(d1, axis=1).astype(np.int32), np.unpackbits(d2, axis=1).astype(np.int32)
    m12 = np.full(len(k1), -1, np.int32)
    hist = [[] for _ in range(30)]
    n = 0
    n2map = {int(v): j for j, v in enumerate(g2[0])}
    for j1, node in enumerate(g1[0]):
        if int(node) not in n2map:
            continue
        j2 = n2map[int(node)]
        for i1 in g1[2][g1[1][j1]:g1[1][j1 + 1]]:
            if skip1[i1]:
                continue
            st1 = ur1 is not None and ur1[i1] >= 0
            if only_stereo and not st1:
                continue
            x1, y1 = f32(k1["x"][i1]), f32(k1["y"][i1])
            a = f32(f32(x1 * F[0, 0]) + f32(y1 * F[1, 0])) + F[2, 0]
            b = f32(f32(x1 * F[0, 1]) + f32(y1 * F[1, 1])) + F[2, 1]
            c = f32(f32(x1 * F[0, 2]) + f32(y1 * F[1, 2])) + F[2, 2]
            best, bi = 50, -1
            for i2 in g2[2][g2[1][j2]:g2[1][j2 + 1]]:
                if skip2[i2]:
                    continue
                st2 = ur2 is not None and ur2[i2] >= 0
                if only_stereo and not st2:
                    continue
                dist = int(np.abs(b1[i1] - b2[i2]).sum())
                if dist > 50 or dist > best:
                    continue
                x2, y2 = f32(k2["x"][i2]), f32(k2["y"][i2])
                if not st1 and not st2:
                    dx, dy = f32(ex) - x2, f32(ey) - y2
                    if f32(f32(dx * dx) + f32(dy * dy)) < f32(100) * sf[k2["octave"][i2]]:
                        continue
                num = f32(f32(f32(a * x2) + f32(b * y2)) + c)
                den = f32(f32(a * a) + f32(b * b))
                if den == 0:
                    continue
                dsqr = f32(f32(num * num) / den)
                if float(dsqr) < 3.84 * float(s2[k2["octave"][i2]]):
                    best, bi = dist, int(i2)
            if bi >= 0:
                m12[i1] = bi
                n += 1
                if check_ori:
                    rot = f32(k1["angle"][i1]) - f32(k2["angle"][bi])
                    if rot < 0:
                        rot = f32(rot + f32(360.0))
                    v = float(f32(rot * f32(1.0 / 30)))
                    bn = int(np.floor(abs(v) + 0.5))
                    hist[0 if bn == 30 else bn].append(i1)
    if check_ori:
        from test_init_search import _three_maxima
        keep = _three_maxima([len(x) for x in hist])
        for bn in range(30):
            if bn not in keep:
                for i1 in hist[bn]:
                    m12[i1] = -1
                    n -= 1
    return n, m12


def _flags(n, seed, p):
    return (np.random.default_rng(seed).random(n) < p).astype(np.uint8)


@pytest.mark.parametrize("mono,only_stereo,check_ori", [(True, False, True), (False, False, True), (False, True, False)])
def test_oracle_matches_python_model(oracle, mono, only_stereo, check_ori):
    k1, d1, g1, k2, d2, g2, sf, s2 = _scene(oracle, nf=800)
    skip1, skip2 = _flags(len(k1), 1, 0.3), _flags(len(k2), 2, 0.3)
    rng = np.random.default_rng(3)
    ur1 = None if mono else np.where(rng.random(len(k1)) < 0.6, k1["x"] - 20, -1).astype(np.float32)
    ur2 = None if mono else np.where(rng.random(len(k2)) < 0.6, k2["x"] - 20, -1).astype(np.float32)
    args = (k1, d1, skip1, g1, k2, d2, skip2, g2, F_ROWS, 300.0, 240.0, sf, s2)
    n, m = oracle.search_for_triangulation(*args, u_right1=ur1, u_right2=ur2, only_stereo=only_stereo, check_ori=check_ori)
    rn, rm = _model(*args, ur1, ur2, only_stereo, check_ori)
    assert n == rn and np.array_equal(m, rm) and n == (m >= 0).sum() and n > 40
    assert not skip1[m >= 0].any() and not skip2[m[m >= 0]].any()


def test_oracle_tie_and_epipole_semantics(oracle):
    """Equal distances: the LAST candidate of the node wins (:719 dist>bestDist); a candidate within 10 px * scale of the
    epipole is passed over in the monocular case; a feature of key frame 2 may serve two features of key frame 1."""
    from orbhip.capi import KP_DTYPE
    k1 = np.zeros(2, KP_DTYPE)
    k1["x"], k1["y"] = [50, 60], [100, 100]
    k2 = np.zeros(3, KP_DTYPE)
    k2["x"], k2["y"] = [30, 40, 45], [100, 100.5, 100]
    d1 = np.zeros((2, 32), np.uint8)
    d2 = np.zeros((3, 32), np.uint8)
    d2[:, 0] = 0x01                                            # all three at distance 1 from both
    g1 = (np.array([7], np.int32), np.array([0, 2], np.int32), np.array([0, 1], np.int32))
    g2 = (np.array([7], np.int32), np.array([0, 3], np.int32), np.array([0, 1, 2], np.int32))
    sf, s2 = np.array([1.0], np.float32), np.array([1.0], np.float32)
    z2, z3 = np.zeros(2, np.uint8), np.zeros(3, np.uint8)
    n, m = oracle.search_for_triangulation(k1, d1, z2, g1, k2, d2, z3, g2, F_ROWS, 500.0, 400.0, sf, s2, check_ori=False)
    assert n == 2 and m.tolist() == [2, 2]
    n, m = oracle.search_for_triangulation(k1, d1, z2, g1, k2, d2, z3, g2, F_ROWS, 47.0, 100.0, sf, s2, check_ori=False)
    assert m.tolist() == [0, 0]                                # features 1 and 2 are within 10 px of the epipole
    ur = np.array([5.0, 5.0, 5.0], np.float32)                 # stereo on side 2: the epipole test does not apply
    n, m = oracle.search_for_triangulation(k1, d1, z2, g1, k2, d2, z3, g2, F_ROWS, 47.0, 100.0, sf, s2, u_right2=ur,
                                           check_ori=False)
    assert m.tolist() == [2, 2]
    k2["y"][2] = 103                                           # 3 px off the line: 9 > 3.84
    n, m = oracle.search_for_triangulation(k1, d1, z2, g1, k2, d2, z3, g2, F_ROWS, 500.0, 400.0, sf, s2, check_ori=False)
    assert m.tolist() == [1, 1]


@pytest.mark.gpu
@pytest.mark.parametrize("mono,only_stereo,check_ori", [(True, False, True), (False, False, True), (False, True, True),
                                                       (True, False, False)])
def test_hip_search_for_triangulation_matches_oracle(oracle, mono, only_stereo, check_ori):
    from orbhip import guided
    from orbhip.extractor import ORBextractor
    ex = ORBextractor(500, max_w=320, max_h=240)
    k1, d1, g1, k2, d2, g2, sf, s2 = _scene(oracle, seed=8, nf=2000)
    skip1, skip2 = _flags(len(k1), 4, 0.35), _flags(len(k2), 5, 0.35)
    rng = np.random.default_rng(6)
    ur1 = None if mono else np.where(rng.random(len(k1)) < 0.6, k1["x"] - 20, -1).astype(np.float32)
    ur2 = None if mono else np.where(rng.random(len(k2)) < 0.6, k2["x"] - 20, -1).astype(np.float32)
    # a general F (rows slightly rotated) besides the row-aligned one
    Fg = np.array([[1e-6, 2e-5, -0.004], [-2e-5, 1e-6, -1.0], [0.003, 1.0, 0.05]], np.float32)
    for F, (exx, eyy) in ((F_ROWS, (300.0, 240.0)), (Fg, (-80.0, 200.0))):
        args = (k1, d1, skip1, g1, k2, d2, skip2, g2, F, exx, eyy, sf, s2)
        n, m = guided.SearchForTriangulation(ex, *args, u_right1=ur1, u_right2=ur2, only_stereo=only_stereo, check_ori=check_ori)
        rn, rm = oracle.search_for_triangulation(*args, u_right1=ur1, u_right2=ur2, only_stereo=only_stereo, check_ori=check_ori)
        assert n == rn and np.array_equal(m, rm) and rn > (20 if only_stereo else 60)
    # empty inputs
    e = (np.zeros(0, np.int32), np.zeros(1, np.int32), np.zeros(0, np.int32))
    n, m = guided.SearchForTriangulation(ex, k1, d1, skip1, e, k2, d2, skip2, g2, F_ROWS, 0, 0, sf, s2)
    assert n == 0 and (m == -1).all()
    n, m = guided.SearchForTriangulation(ex, k1[:0], d1[:0], skip1[:0], e, k2, d2, skip2, g2, F_ROWS, 0, 0, sf, s2)
    assert n == 0 and len(m) == 0
    ex.close()


@pytest.mark.gpu
def test_hip_matcher_entry_points_reject_bad_arguments(oracle):
    """Negative C return codes surface as OrbHipError (never a crash, never a silent wrong answer)."""
    from orbhip import guided
    from orbhip.capi import KP_DTYPE, OrbHipError
    from orbhip.extractor import ORBextractor
    ex = ORBextractor(500, max_w=320, max_h=240)
    k = np.zeros(4, KP_DTYPE)
    k["x"], k["y"] = [10, 20, 30, 40], [10, 20, 30, 40]
    d = np.zeros((4, 32), np.uint8)
    z = np.zeros(4, np.uint8)
    sf, s2 = np.ones(8, np.float32), np.ones(8, np.float32)
    g_ok = (np.array([1], np.int32), np.array([0, 4], np.int32), np.arange(4, dtype=np.int32))
    g_bad = (np.array([1], np.int32), np.array([0, 4], np.int32), np.array([0, 1, 2, 9], np.int32))   # index 9 of 4
    with pytest.raises(OrbHipError):
        guided.SearchForTriangulation(ex, k, d, z, g_bad, k, d, z, g_ok, F_ROWS, 0, 0, sf, s2)
    kb = k.copy()
    kb["octave"][2] = 11                                                                                # no such level
    with pytest.raises(OrbHipError):
        guided.SearchForTriangulation(ex, k, d, z, g_ok, kb, d, z, g_ok, F_ROWS, 0, 0, sf, s2)
    with pytest.raises(OrbHipError):                                                                    # degenerate grid
        guided.SearchForInitialization(ex, k, d, k, d, (np.float32(0), np.float32(0), np.float32(0), np.float32(0.1)),
                                       np.zeros((4, 2), np.float32), 10)
    with pytest.raises(ValueError):                                                                     # vbPrevMatched too short
        guided.SearchForInitialization(ex, k, d, k, d, guided.grid_params(0, 640, 0, 480), np.zeros((3, 2), np.float32), 10)
    n, m = guided.SearchForTriangulation(ex, k, d, z, g_ok, k, d, z, g_ok, F_ROWS, 500, 500, sf, s2, check_ori=False)
    assert n == 4 and m.tolist() == [0, 1, 2, 3]          # the valid call still works on the same context (each feature finds itself)
    ex.close()
