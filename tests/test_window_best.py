"""The per-point window search shared by ORBmatcher::Fuse (src/ORBmatcher.cc:825-975, :977-1100) and SearchBySim3
(:1102-1326): the oracle against a definition-level Python model on CPU; the HIP path against the oracle on the GPU."""
import ctypes as C

import numpy as np
import pytest


def _scene(oracle, seed=5, nf=1000):
    from orbhip import synth
    fr = synth.make_frames(seed, 640, 480, 2)
    ex = oracle.Extractor(nf)
    (k0, d0), (k1, d1) = ex(fr[0]), ex(fr[1])
    return k0, d0, k1, d1


def _queries(rng, k_src, d_src, nq, th, scale=1.2, jitter=2.0, inactive=0.1):
    """Points of another key frame projected near where they were seen: (queries, qdesc)."""
    from orbhip.capi import QUERY_DTYPE, Q_ACTIVE
    pick = rng.integers(0, len(k_src), nq)
    q = np.zeros(nq, QUERY_DTYPE)
    q["u"] = (k_src["x"][pick] + rng.normal(0, jitter, nq)).astype(np.float32)
    q["v"] = (k_src["y"][pick] + rng.normal(0, jitter, nq)).astype(np.float32)
    pred = np.clip(k_src["octave"][pick] + rng.integers(-1, 2, nq), 0, 7).astype(np.int32)
    q["radius"] = (np.float32(th) * np.float32(scale) ** pred).astype(np.float32)
    q["proj_xr"] = (q["u"] - rng.uniform(2, 40, nq)).astype(np.float32)
    q["min_level"], q["max_level"] = pred - 1, pred
    q["flags"] = np.where(rng.random(nq) < inactive, 0, Q_ACTIVE)
    qd = d_src[pick].copy()
    flips = rng.integers(0, 30, nq)
    for i in range(nq):
        for b in rng.integers(0, 256, flips[i]):
            qd[i, b >> 3] ^= 1 << (b & 7)
    return q, qd


def _model(oracle, kps, desc, gp, q, qd, u_right, sig):
    from orbhip.capi import Q_ACTIVE
    grid = oracle.grid_build(kps, gp)
    bits = np.unpackbits(desc, axis=1).astype(np.int32)
    qb = np.unpackbits(qd, axis=1).astype(np.int32)
    f = np.float32
    bi = np.full(len(q), -1, np.int32)
    bd = np.full(len(q), 256, np.int32)
    for i in range(len(q)):
        if not q["flags"][i] & Q_ACTIVE:
            continue
        for idx in oracle.features_in_area(kps, grid, gp, q["u"][i], q["v"][i], q["radius"][i], -1, -1):
            lvl = kps["octave"][idx]
            if lvl < q["min_level"][i] or lvl > q["max_level"][i]:
                continue
            if sig is not None:
                ex, ey = f(q["u"][i] - kps["x"][idx]), f(q["v"][i] - kps["y"][idx])
                e2 = f(f(ex * ex) + f(ey * ey))
                if u_right is not None and u_right[idx] >= 0:
                    er = f(q["proj_xr"][i] - u_right[idx])
                    if float(f(f(e2 + f(er * er)) * sig[lvl])) > 7.8:
                        continue
                elif float(f(e2 * sig[lvl])) > 5.99:
                    continue
            d = int(np.abs(bits[idx] - qb[i]).sum())
            if d < bd[i]:
                bd[i], bi[i] = d, idx
    return bi, bd


def _sigma(nlev=8, s=1.2):
    return (1.0 / (np.float32(s) ** np.arange(nlev, dtype=np.float32)) ** 2).astype(np.float32)


@pytest.mark.parametrize("gate,stereo", [(False, False), (True, False), (True, True)])
def test_oracle_matches_python_model(oracle, gate, stereo):
    k0, d0, k1, d1 = _scene(oracle)
    rng = np.random.default_rng(7)
    gp = oracle.grid_params(0, 640, 0, 480)
    q, qd = _queries(rng, k0, d0, 400, 3.0)
    ur = None
    if stereo:
        ur = np.where(rng.random(len(k1)) < 0.6, k1["x"] - rng.uniform(2, 40, len(k1)), -1).astype(np.float32)
    sig = _sigma() if gate else None
    bi, bd = oracle.window_best(k1, d1, gp, q, qd, ur, sig)
    rbi, rbd = _model(oracle, k1, d1, gp, q, qd, ur, sig)
    assert np.array_equal(bi, rbi) and np.array_equal(bd, rbd)
    assert (bi >= 0).sum() > 100 and ((bi < 0) == (bd == 256)).all()


def test_oracle_gate_and_level_semantics(oracle):
    """Hand-made case: level window [pred - 1, pred], first of equal distances wins, the mono gate 5.99 and the
    stereo gate 7.8 (taken when the right coordinate is >= 0, zero included)."""
    from orbhip.capi import KP_DTYPE, QUERY_DTYPE, Q_ACTIVE
    k = np.zeros(5, KP_DTYPE)
    k["x"], k["y"] = [100, 101, 102, 100.5, 103], [100, 100, 100, 101, 100]
    k["octave"] = [0, 1, 2, 1, 3]
    d = np.zeros((5, 32), np.uint8)
    d[0, 0], d[1, 0], d[2, 0], d[3, 0], d[4, 0] = 0x01, 0x03, 0x03, 0x03, 0x00
    q = np.zeros(1, QUERY_DTYPE)
    q["u"], q["v"], q["radius"], q["flags"] = 100, 100, 10, Q_ACTIVE
    q["min_level"], q["max_level"] = 1, 2
    qd = np.zeros((1, 32), np.uint8)
    gp = oracle.grid_params(0, 640, 0, 480)
    bi, bd = oracle.window_best(k, d, gp, q, qd)
    assert bd[0] == 2 and bi[0] in (1, 2, 3)                  # levels 0 and 3 are outside the window
    order = oracle.features_in_area(k, oracle.grid_build(k, gp), gp, 100, 100, 10, -1, -1).tolist()
    assert bi[0] == [i for i in order if i in (1, 2, 3)][0]   # first in the grid's order
    sig = np.ones(4, np.float32)
    # mono gate: e2 = 1 (feature 1), 4 (feature 2), 1.25 (feature 3) pass 5.99; with sigma2 1/4 on level 1 feature 2 stays
    bi, bd = oracle.window_best(k, d, gp, q, qd, None, np.array([1, 6, 1, 1], np.float32))
    assert bi[0] == 2                                         # 1 * 6 > 5.99 and 1.25 * 6 > 5.99 removed features 1 and 3
    # stereo gate: u_right 0 counts as a stereo feature (>= 0); er = proj_xr - 0
    q["proj_xr"] = 2.7
    ur = np.array([-1, 0, -1, -1, -1], np.float32)
    bi, bd = oracle.window_best(k, d, gp, q, qd, ur, sig)     # feature 1: 1 + 7.29 = 8.29 > 7.8 -> out
    assert bi[0] in (2, 3) and 1 not in bi
    q["proj_xr"] = 2.6
    bi, bd = oracle.window_best(k, d, gp, q, qd, ur, sig)     # 1 + 6.76 = 7.76 <= 7.8 -> back in
    assert bi[0] == [i for i in order if i in (1, 2, 3)][0]
    q["flags"] = 0
    bi, bd = oracle.window_best(k, d, gp, q, qd)
    assert bi[0] == -1 and bd[0] == 256


@pytest.mark.gpu
@pytest.mark.parametrize("nf,th,gate,stereo", [(1000, 3.0, True, False), (1000, 3.0, True, True), (2000, 7.5, False, False),
                                               (4000, 20.0, False, True)])
def test_hip_window_best_matches_oracle(oracle, nf, th, gate, stereo):
    from orbhip import guided
    from orbhip.extractor import ORBextractor
    ex = ORBextractor(500, max_w=320, max_h=240)
    k0, d0, k1, d1 = _scene(oracle, seed=60 + nf // 1000, nf=nf)
    rng = np.random.default_rng(nf)
    ur = None
    if stereo:
        ur = np.where(rng.random(len(k1)) < 0.6, k1["x"] - rng.uniform(0, 40, len(k1)), -1).astype(np.float32)
        ur[:5] = 0
    sig = _sigma() if gate else None
    for gp, kk in [(guided.grid_params(0, 640, 0, 480), k1), (guided.grid_params(22.5, 611.25, 19.75, 452.5), None)]:
        if kk is None:                                        # undistorted coordinates, partly outside the grid
            kk = k1.copy()
            kk["x"] = (k1["x"] * np.float32(1.013) - np.float32(5.3)).astype(np.float32)
            kk["y"] = (k1["y"] * np.float32(1.011) - np.float32(3.1)).astype(np.float32)
        q, qd = _queries(rng, k0, d0, 3000, th)
        bi, bd = guided.WindowBest(ex, kk, d1, gp, q, qd, ur, sig)
        rbi, rbd = oracle.window_best(kk, d1, gp, q, qd, ur, sig)
        assert np.array_equal(bi, rbi) and np.array_equal(bd, rbd)
        assert (rbi >= 0).sum() > 300
    # empty sides
    bi, bd = guided.WindowBest(ex, k1[:0], d1[:0], gp, q, qd)
    assert (bi == -1).all() and (bd == 256).all() and len(bi) == len(q)
    bi, bd = guided.WindowBest(ex, k1, d1, gp, q[:0], qd[:0])
    assert len(bi) == 0
    ex.close()


@pytest.mark.gpu
@pytest.mark.parametrize("seed", [1, 2, 3])
def test_hip_window_best_ties_and_gate_edges(oracle, seed):
    """Few distinct descriptors (many equal distances: the grid's visiting order decides) on integer coordinates, where
    the chi-square products land exactly on and next to the limits."""
    from orbhip import guided
    from orbhip.capi import KP_DTYPE
    from orbhip.extractor import ORBextractor
    rng = np.random.default_rng(seed)
    n = 1500
    base = rng.integers(0, 256, (6, 32), dtype=np.uint8)
    k = np.zeros(n, KP_DTYPE)
    k["x"], k["y"] = rng.integers(150, 260, n), rng.integers(120, 200, n)
    k["octave"] = rng.integers(0, 4, n)
    d = base[rng.integers(0, len(base), n)].copy()
    ur = np.where(rng.random(n) < 0.5, k["x"] - rng.integers(0, 4, n), -1).astype(np.float32)
    q, qd = _queries(rng, k, d, 2000, 4.0, jitter=0.0, inactive=0.05)
    q["u"], q["v"] = np.round(q["u"]), np.round(q["v"])
    q["proj_xr"] = np.round(q["u"] - rng.integers(0, 4, len(q)))
    qd = base[rng.integers(0, len(base), len(q))].copy()
    sig = np.array([1.0, 0.599, 0.78, 5.99 / 8], np.float32)
    gp = guided.grid_params(0, 640, 0, 480)
    ex = ORBextractor(500, max_w=320, max_h=240)
    for u, s in [(None, None), (None, sig), (ur, sig)]:
        bi, bd = guided.WindowBest(ex, k, d, gp, q, qd, u, s)
        rbi, rbd = oracle.window_best(k, d, gp, q, qd, u, s)
        assert np.array_equal(bi, rbi) and np.array_equal(bd, rbd)
    ex.close()


@pytest.mark.gpu
def test_hip_window_best_batched_device_form(oracle):
    """B key frames with their own points in one launch equal B single calls."""
    import hiprt
    from orbhip import capi, guided
    from orbhip.capi import GRID_COLS, GRID_ROWS
    from orbhip.extractor import ORBextractor
    ex = ORBextractor(500, max_w=320, max_h=240)
    L = capi.load()
    rng = np.random.default_rng(11)
    B, cap, capq = 3, 1200, 900
    gp = guided.grid_params(0, 640, 0, 480)
    sig = _sigma()
    kps = np.zeros((B, cap), capi.KP_DTYPE)
    desc = np.zeros((B, cap, 32), np.uint8)
    cnt = np.zeros(B, np.int32)
    qs = np.zeros((B, capq), capi.QUERY_DTYPE)
    qds = np.zeros((B, capq, 32), np.uint8)
    nq = np.array([900, 0, 517], np.int32)
    per = []
    for b in range(B):
        k0, d0, k1, d1 = _scene(oracle, seed=70 + b, nf=1000)
        n = min(len(k1), cap)
        kps[b, :n], desc[b, :n], cnt[b] = k1[:n], d1[:n], n
        q, qd = _queries(rng, k0, d0, int(nq[b]), 3.0)
        qs[b, :nq[b]], qds[b, :nq[b]] = q, qd
        per.append((k1[:n], d1[:n], q, qd))
    D = hiprt.DevBuf
    t_k, t_d, t_c, t_q, t_qd, t_nq = (D.from_numpy(x) for x in (kps, desc, cnt, qs, qds, nq))
    t_off, t_idx = D(B * (GRID_COLS * GRID_ROWS + 1) * 4), D(B * cap * 4)
    t_bi, t_bd = D(B * capq * 4), D(B * capq * 4)
    capi.check(L.orbhip_grid_build_device(ex.handle, t_k.ptr, t_c.ptr, cap, B, gp[0], gp[1], gp[2], gp[3], t_off.ptr, t_idx.ptr),
               ex.handle, "grid")
    capi.check(L.orbhip_window_best_device(ex.handle, t_k.ptr, t_d.ptr, cap, B, None, sig.ctypes.data_as(C.c_void_p), len(sig),
                                           gp[0], gp[1], gp[2], gp[3], t_off.ptr, t_idx.ptr, t_q.ptr, t_qd.ptr, t_nq.ptr, capq,
                                           t_bi.ptr, t_bd.ptr), ex.handle, "window_best_device")
    ex.sync()
    bi = t_bi.to_numpy(np.int32, (B, capq))
    bd = t_bd.to_numpy(np.int32, (B, capq))
    for b, (k, d, q, qd) in enumerate(per):
        rbi, rbd = oracle.window_best(k, d, gp, q, qd, None, sig)
        assert np.array_equal(bi[b, :nq[b]], rbi) and np.array_equal(bd[b, :nq[b]], rbd)
        assert (bi[b, nq[b]:] == -1).all() and (bd[b, nq[b]:] == 256).all()
    ex.close()


@pytest.mark.gpu
def test_hip_window_best_bad_arguments(oracle):
    from orbhip import guided
    from orbhip.capi import OrbHipError
    from orbhip.extractor import ORBextractor
    ex = ORBextractor(500, max_w=320, max_h=240)
    k0, d0, k1, d1 = _scene(oracle, nf=500)
    q, qd = _queries(np.random.default_rng(0), k0, d0, 10, 3.0)
    gp = guided.grid_params(0, 640, 0, 480)
    with pytest.raises(OrbHipError):                          # octave 7 present, 4 sigma entries
        guided.WindowBest(ex, k1, d1, gp, q, qd, None, _sigma(4))
    with pytest.raises(OrbHipError):
        guided.WindowBest(ex, k1, d1, gp, q, qd, None, np.ones(17, np.float32))
    with pytest.raises(OrbHipError):
        guided.WindowBest(ex, k1, d1, (0.0, 0.0, 0.0, 0.1), q, qd)
    ex.close()
