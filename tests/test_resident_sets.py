"""Resident feature sets (orbhip_set_*, include/orbhip.h): a key frame's descriptors / keypoints / FeatureVector / grid stay on
the device across calls; SearchByBoW and the Fuse window search between sets must return exactly what the per-call entry
points return (and the oracle), also after replacement and least-recently-used eviction."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _frames_with_fv(oracle, n=3, nf=1000):
    from orbhip import distributed as D, synth
    fr = synth.make_frames(31, 640, 480, n)
    ref = oracle.Extractor(nf)
    V = oracle.Vocabulary(D.make_synthetic_vocabulary(32, k=10, L=4))
    out = []
    for f in fr:
        k, d = ref(f)
        _, wt, nid = V.transform(d, 2)
        out.append((k, d, oracle.feature_vector(nid, wt)))
    return out


def test_search_by_bow_between_resident_sets(oracle):
    from orbhip.capi import OrbHipError
    from orbhip.extractor import ORBmatcher
    fs = _frames_with_fv(oracle)
    M = ORBmatcher(0.7, True)
    rng = np.random.default_rng(3)
    for i, (k, d, fv) in enumerate(fs):
        M.put_set(100 + i, k, d, fv)
        assert M.has_set(100 + i, len(k)) and not M.has_set(100 + i, len(k) + 1)
    assert not M.has_set(999, 10)
    for a, b, kfkf in ((0, 1, False), (1, 2, True), (2, 0, False)):
        (k1, d1, f1), (k2, d2, f2) = fs[a], fs[b]
        v1 = (rng.random(len(k1)) < 0.8).astype(np.uint8)
        v2 = (rng.random(len(k2)) < 0.9).astype(np.uint8) if kfkf else None
        want = oracle.search_by_bow(d1, v1, k1["angle"], f1, d2, v2, k2["angle"], f2, th=50, th_mode=1 if kfkf else 0, nnratio=0.7,
                                    check_ori=True)
        got = M.SearchByBoW_sets(100 + a, v1, len(k1), 100 + b, v2, len(k2), kf_kf=kfkf)
        per_call = M.SearchByBoW(d1, v1, k1["angle"], f1, d2, v2, k2["angle"], f2, kf_kf=kfkf)
        assert got[0] == want[0] == per_call[0] > 30
        assert np.array_equal(got[1], want[1]) and np.array_equal(got[2], want[2])
    # replacing a set: the new content is what is matched
    k0, d0, f0 = fs[0]
    M.put_set(101, k0, d0, f0)
    v = np.ones(len(k0), np.uint8)
    got = M.SearchByBoW_sets(100, v, len(k0), 101, None, len(k0))
    assert got[0] > 0.9 * len(k0) * 0.5 and (got[1][got[1] >= 0] == np.nonzero(got[1] >= 0)[0]).mean() > 0.95   # a frame against itself
    # eviction: 96 sets per context, least recently used out; key 100 is kept alive by use
    for j in range(200):
        M.put_set(1000 + j, k0[:50], d0[:50])
        if j % 20 == 0:
            assert M.has_set(100, len(k0))
    assert M.has_set(100, len(k0)) and not M.has_set(1000, 50) and M.has_set(1199, 50)
    with pytest.raises(OrbHipError, match="unknown set"):
        M.SearchByBoW_sets(100, v, len(k0), 1000, None, 50)
    M.drop_set(100)
    assert not M.has_set(100, len(k0))
    M.drop_set()
    assert not M.has_set(1199, 50)
    M.close()


def test_window_best_into_a_resident_key_frame(oracle):
    from orbhip import guided
    from orbhip.capi import QUERY_DTYPE, OrbHipError
    from orbhip.extractor import ORBmatcher
    fs = _frames_with_fv(oracle, 2)
    (k0, d0, _), (k1, d1, _) = fs
    gp = guided.grid_params(0, 640, 0, 480)
    sf = (np.float32(1.2) ** np.arange(8, dtype=np.float32)).astype(np.float32)
    sig = (1 / sf ** 2).astype(np.float32)
    rng = np.random.default_rng(5)
    q = np.zeros(len(k0), QUERY_DTYPE)
    q["u"] = k0["x"] + rng.normal(0, 2, len(k0)).astype(np.float32)
    q["v"] = k0["y"] + rng.normal(0, 2, len(k0)).astype(np.float32)
    q["radius"] = 3 * sf[k0["octave"]]
    q["min_level"], q["max_level"], q["flags"] = k0["octave"] - 1, k0["octave"], 1
    q["flags"][::7] = 0                                            # skipped points
    q["proj_xr"] = q["u"] - 5
    ur = np.where(rng.random(len(k1)) < 0.5, k1["x"] - 5, -1).astype(np.float32)
    M = ORBmatcher(0.7, True)
    M.put_set(7, k1, d1, None, gp)
    for u_right, gate in ((None, None), (None, sig), (ur, sig)):
        want = oracle.window_best(k1, d1, gp, q, d0, u_right, gate)
        got = guided.WindowBestSet(M._ctx, 7, q, d0, u_right, gate)
        assert np.array_equal(got[0], want[0]) and np.array_equal(got[1], want[1])
        assert (got[0] >= 0).sum() > 300
    M.put_set(8, k1, d1)                                           # no grid
    with pytest.raises(OrbHipError, match="without a grid"):
        guided.WindowBestSet(M._ctx, 8, q, d0)
    M.close()


def test_set_limit_evicts_least_recently_used(oracle):
    """orbhip_set_limit: the limit is clamped to 4..96 and the oldest sets leave first."""
    from orbhip.extractor import ORBmatcher
    k, d, fv = _frames_with_fv(oracle, n=1, nf=300)[0]
    M = ORBmatcher(0.7, True)
    assert M.set_limit(1) == 4 and M.set_limit(1000) == 96 and M.set_limit(4) == 4
    for j in range(6):
        M.put_set(500 + j, k[:40], d[:40])
    assert [M.has_set(500 + j, 40) for j in range(6)] == [False, False, True, True, True, True]
    assert M.has_set(502, 40)                       # (has_set is a use: 502 is now the most recent)
    M.put_set(510, k[:40], d[:40])
    assert M.has_set(502, 40) and not M.has_set(503, 40) and M.has_set(510, 40)
    M.drop_set()
    M.close()
