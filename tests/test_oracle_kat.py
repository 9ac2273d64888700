"""Known-answer and definition-level tests that pin the CPU oracle.

The reference ships no tests or golden vectors, so the pins are: (1) first-principles known
answers, (2) brute-force python models written from the definitions (tests/pymodels.py),
(3) the derived geometry tables of SURVEY.md Appendix B.
"""
import hashlib
import os
import struct

import numpy as np
import pytest

import pymodels as M

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PATTERN_SHA = "7e645581387b82784797e8adddb9b6f0c12611859fda09ca8a9bec96d767a05f"


def _pattern_from_header(path):
    txt = open(path).read()
    body = txt[txt.index("{") + 1:txt.index("};")]
    return [int(v) for v in body.replace("\n", " ").split(",") if v.strip()]


@pytest.mark.parametrize("path", ["oracle/orb_pattern_data.h",
                                  "vi-orb-slam-icra2018_amd/csrc/orb_pattern_data.h"])
def test_pattern_table_sha256(path):
    vals = _pattern_from_header(os.path.join(ROOT, path))
    assert len(vals) == 1024
    assert hashlib.sha256(struct.pack("<1024i", *vals)).hexdigest() == PATTERN_SHA


def test_constructor_tables(oracle):
    P = oracle.params(1000, 1.2, 8, 20, 7)
    assert list(P.umax) == [15, 15, 15, 15, 14, 14, 14, 13, 13, 12, 11, 10, 9, 8, 6, 3]
    assert list(P.mnFeaturesPerLevel)[:8] == [217, 181, 151, 126, 105, 87, 73, 60]
    assert list(oracle.params(2000).mnFeaturesPerLevel)[:8] == [434, 362, 302, 251, 209, 175, 145, 122]
    assert list(oracle.params(4000).mnFeaturesPerLevel)[:8] == [869, 724, 603, 503, 419, 349, 291, 242]
    sf = np.array(list(P.mvScaleFactor)[:8], np.float32)
    want = np.array([1, 1.2000000477, 1.4400000572, 1.7280001640, 2.0736002922, 2.4883203506,
                     2.9859845638, 3.5831816196], np.float32)
    assert np.array_equal(sf, want)
    # float(double(prev) * double(1.2f)) chain, include/ORBextractor.h:116
    s = np.float32(1.0)
    for i in range(1, 8):
        s = np.float32(np.float64(s) * np.float64(np.float32(1.2)))
        assert s == sf[i]


@pytest.mark.parametrize("wh,sizes", [
    ((752, 480), [(752, 480), (627, 400), (522, 333), (435, 278), (363, 231), (302, 193), (252, 161), (210, 134)]),
    ((640, 480), [(640, 480), (533, 400), (444, 333), (370, 278), (309, 231), (257, 193), (214, 161), (179, 134)]),
    ((1241, 376), [(1241, 376), (1034, 313), (862, 261), (718, 218), (598, 181), (499, 151), (416, 126), (346, 105)]),
])
def test_level_sizes(oracle, wh, sizes):
    P = oracle.params()
    assert [oracle.level_size(P, wh[0], wh[1], l) for l in range(8)] == sizes


def test_cvround_half_even(oracle):
    assert [oracle.cvround(v) for v in (0.5, 1.5, 2.5, -0.5, -1.5, 2.4999, 2.5001)] == [0, 2, 2, 0, -2, 2, 3]


def test_fast_atan2(oracle):
    assert oracle.fast_atan2(0, 0) == 0.0
    assert oracle.fast_atan2(0, 1) == 0.0
    assert abs(oracle.fast_atan2(1, 0) - 90) < 1e-4
    assert abs(oracle.fast_atan2(0, -1) - 180) < 1e-4
    assert abs(oracle.fast_atan2(-1, 0) - 270) < 1e-4
    rng = np.random.default_rng(1)
    for _ in range(2000):
        y, x = rng.integers(-200000, 200000, 2)
        if x == 0 and y == 0:
            continue
        want = np.degrees(np.arctan2(float(y), float(x))) % 360.0
        got = oracle.fast_atan2(y, x)
        d = abs(got - want)
        assert min(d, 360 - d) < 0.02, (y, x, got, want)   # 7th-order poly: ~0.01 deg
        assert 0.0 <= got <= 360.0


def test_hamming_known_answers(oracle):
    z = np.zeros(32, np.uint8)
    o = np.full(32, 255, np.uint8)
    assert oracle.descriptor_distance(z, o) == 256
    assert oracle.descriptor_distance(z, z) == 0
    for bit in (0, 7, 8, 100, 255):
        a = z.copy()
        a[bit >> 3] = 1 << (bit & 7)
        assert oracle.descriptor_distance(a, z) == 1
    rng = np.random.default_rng(0)
    for _ in range(500):
        a = rng.integers(0, 256, 32, dtype=np.uint8)
        b = rng.integers(0, 256, 32, dtype=np.uint8)
        assert oracle.descriptor_distance(a, b) == M.hamming(a, b)


def test_knn2_semantics(oracle):
    rng = np.random.default_rng(3)
    db = rng.integers(0, 256, (300, 32), dtype=np.uint8)
    db[17] = db[5]          # exact duplicate: lowest index must win, second == best
    q = db[[5, 17, 40]].copy()
    q[2, 0] ^= 0x0F
    bi, bd, sd = oracle.knn2(q, db)
    assert list(bi) == [5, 5, 40] and list(bd) == [0, 0, 4] and sd[0] == 0 and sd[1] == 0
    # brute force by definition
    for i in range(3):
        d = [M.hamming(q[i], r) for r in db]
        order = sorted(range(len(d)), key=lambda j: (d[j], j))
        assert bi[i] == order[0] and bd[i] == d[order[0]] and sd[i] == d[order[1]]
    # empty database: initial values of the reference loops
    bi, bd, sd = oracle.knn2(q, np.zeros((0, 32), np.uint8))
    assert list(bi) == [-1] * 3 and list(bd) == [256] * 3 and list(sd) == [256] * 3


def _ring_image(center, ring_vals):
    img = np.full((7, 7), center, np.uint8)
    for (dx, dy), v in zip(M.RING, ring_vals):
        img[3 + dy, 3 + dx] = v
    return img


@pytest.mark.parametrize("arc,expect", [(8, False), (9, True), (10, True), (16, True)])
@pytest.mark.parametrize("start", [0, 5, 11, 15])
@pytest.mark.parametrize("bright", [True, False])
def test_fast_arc_lengths(oracle, arc, expect, start, bright):
    ring = [100] * 16
    for j in range(arc):
        ring[(start + j) % 16] = 150 if bright else 50
    img = _ring_image(100, ring)
    got = oracle.fast9_16(img, 20)
    if expect:
        assert len(got) == 1 and (got[0]["x"], got[0]["y"], got[0]["score"]) == (3, 3, 49)
        assert oracle.fast9_16(img, 49)[0]["score"] == 49   # corner at t <=> score >= t
        assert len(oracle.fast9_16(img, 50)) == 0
    else:
        assert len(got) == 0


def test_fast_matches_definition_on_random_images(oracle):
    rng = np.random.default_rng(5)
    for trial in range(6):
        h, w = rng.integers(9, 26), rng.integers(9, 30)
        base = rng.integers(0, 256, (h, w)).astype(np.uint8)
        if trial % 2:   # blocky image: many true corners
            base = np.kron(rng.integers(0, 2, (h // 4 + 1, w // 4 + 1)) * 120 + 60,
                           np.ones((4, 4)))[:h, :w].astype(np.uint8)
            base = (base + rng.integers(0, 6, (h, w))).astype(np.uint8)
        for th in (7, 20):
            got = [(int(c["x"]), int(c["y"]), int(c["score"])) for c in oracle.fast9_16(base, th)]
            assert got == M.fast9_16(base, th)


def test_fast_score_is_max_threshold(oracle):
    rng = np.random.default_rng(6)
    img = (np.kron(rng.integers(0, 2, (6, 6)) * 150 + 40, np.ones((5, 5))) + rng.integers(0, 9, (30, 30))).astype(np.uint8)
    n = 0
    for y in range(3, 27):
        for x in range(3, 27):
            s = M.fast_score(img, x, y)
            if s >= 1:
                assert oracle.fast_corner_score(img, x, y, 1) == s
                n += 1
    assert n > 5


def test_level_candidates_match_model(oracle):
    rng = np.random.default_rng(7)
    img = (np.kron(rng.integers(0, 2, (14, 17)) * 100 + 70, np.ones((8, 8)))[:100, :130]
           + rng.integers(0, 10, (100, 130))).astype(np.uint8)
    img[:, 70:] = (img[:, 70:].astype(np.int32) // 8 + 100).astype(np.uint8)   # low contrast half -> minTh cells
    got = [(int(c["x"]), int(c["y"]), int(c["score"])) for c in oracle.level_candidates(img, 20, 7)]
    want = M.level_candidates(img, 20, 7)
    assert got == want and len(got) > 20
    scores = np.array([g[2] for g in got])
    assert (scores < 20).any() and (scores >= 20).any()


def test_resize_known_answers(oracle):
    rng = np.random.default_rng(8)
    img = rng.integers(0, 256, (48, 64), dtype=np.uint8)
    assert np.array_equal(oracle.resize_linear(img, 64, 48), img)           # identity
    for c in (0, 1, 77, 254, 255):
        assert (oracle.resize_linear(np.full((40, 60), c, np.uint8), 50, 33) == c).all()
    # independent float bilinear (pixel-centre aligned): fixed point differs by <= 1
    src = (np.add.outer(np.arange(60) * 3, np.arange(90) * 2) % 256).astype(np.uint8)
    src = rng.integers(0, 256, (60, 90), dtype=np.uint8)
    dw, dh = 75, 50
    out = oracle.resize_linear(src, dw, dh).astype(np.float64)
    fx = np.clip((np.arange(dw) + 0.5) * (90 / dw) - 0.5, 0, 89)
    fy = np.clip((np.arange(dh) + 0.5) * (60 / dh) - 0.5, 0, 59)
    x0 = np.minimum(fx.astype(int), 88)
    y0 = np.minimum(fy.astype(int), 58)
    ax, ay = fx - x0, fy - y0
    s = src.astype(np.float64)
    ref = ((s[y0][:, x0] * (1 - ax) + s[y0][:, x0 + 1] * ax) * (1 - ay)[:, None]
           + (s[y0 + 1][:, x0] * (1 - ax) + s[y0 + 1][:, x0 + 1] * ax) * ay[:, None])
    assert np.abs(out - ref).max() <= 1.0


def test_resize_fixed_point_formula(oracle):
    """Vectorised numpy restatement of the cv::resize 8U INTER_LINEAR fixed-point scheme
     -- independent of the C loops."""
    rng = np.random.default_rng(9)
    for (sw, sh, dw, dh) in [(640, 480, 533, 400), (179, 134, 149, 112), (90, 61, 75, 51)]:
        src = rng.integers(0, 256, (sh, sw), dtype=np.uint8)

        def coeffs(dn, sn):
            scale = 1.0 / (dn / sn)
            f = ((np.arange(dn) + 0.5) * scale - 0.5).astype(np.float32)
            s = np.floor(f).astype(np.int32)
            f = (f - s.astype(np.float32)).astype(np.float32)
            a0 = np.rint(((np.float32(1) - f) * np.float32(2048)).astype(np.float64)).astype(np.int32)
            a1 = np.rint((f * np.float32(2048)).astype(np.float64)).astype(np.int32)
            return s, a0, a1
        sx, a0, a1 = coeffs(dw, sw)
        sy, b0, b1 = coeffs(dh, sh)
        assert sx.min() >= 0 and sx.max() + 1 < sw and sy.min() >= 0 and sy.max() + 1 < sh
        S = src.astype(np.int32)
        H = S[:, sx] * a0 + S[:, sx + 1] * a1
        out = (((b0[:, None] * (H[sy] >> 4)) >> 16) + ((b1[:, None] * (H[sy + 1] >> 4)) >> 16) + 2) >> 2
        assert np.array_equal(oracle.resize_linear(src, dw, dh), out.astype(np.uint8))


def test_blur_known_answers(oracle):
    # kernel {18,34,49,55,49,34,18} sums to 257 and is not renormalised: slight gain
    for c, want in [(0, 0), (50, 50), (64, 65), (100, 101), (200, 202), (254, 255), (255, 255)]:
        out = oracle.gaussian_blur7(np.full((20, 23), c, np.uint8))
        assert (out == want).all(), (c, want, np.unique(out))
    k = np.array([18, 34, 49, 55, 49, 34, 18])
    img = np.zeros((21, 24), np.uint8)
    img[10, 12] = 255
    out = oracle.gaussian_blur7(img)
    want = np.zeros((21, 24), np.int64)
    want[7:14, 9:16] = (np.outer(k, k) * 255 + 32768) >> 16     # no exact .5 ties here
    assert np.array_equal(out, want.astype(np.uint8))


def test_blur_matches_numpy_restatement(oracle):
    """Independent numpy restatement incl. reflect-101 borders and the SSE2 tie rule
    (half-to-even for x < w - w%4, half-up for the scalar tail)."""
    rng = np.random.default_rng(10)
    k = np.array([18, 34, 49, 55, 49, 34, 18], np.int64)
    for (h, w) in [(30, 41), (33, 40), (17, 22), (134, 179)]:
        img = rng.integers(0, 256, (h, w), dtype=np.uint8)
        p = np.pad(img.astype(np.int64), 3, mode="reflect")
        r = sum(k[j] * p[:, j:j + w] for j in range(7))
        s = sum(k[j] * r[j:j + h, :] for j in range(7))
        up = (s + 32768) >> 16
        even = up - (((s & 0xFFFF) == 0x8000) & ((up & 1) == 1))
        want = up.copy()
        wv = w - w % 4
        want[:, :wv] = even[:, :wv]
        assert np.array_equal(oracle.gaussian_blur7(img), np.clip(want, 0, 255).astype(np.uint8))


def test_blur_tie_rule_is_exercised(oracle):
    """Force an exact .5 tie: value v with 55*55*v... use a direct construction by search."""
    k = np.array([18, 34, 49, 55, 49, 34, 18], np.int64)
    rng = np.random.default_rng(11)
    found = 0
    for _ in range(400):
        img = rng.integers(0, 256, (7, 8), dtype=np.uint8)
        p = np.pad(img.astype(np.int64), 3, mode="reflect")
        r = sum(k[j] * p[:, j:j + 8] for j in range(7))
        s = sum(k[j] * r[j:j + 7, :] for j in range(7))
        if ((s & 0xFFFF) == 0x8000).any():
            found += 1
            up = (s + 32768) >> 16
            even = up - (((s & 0xFFFF) == 0x8000) & ((up & 1) == 1))
            assert np.array_equal(oracle.gaussian_blur7(img), np.clip(even, 0, 255).astype(np.uint8))
    # ties have probability 2^-16 per pixel; the numpy restatement test above covers the rule
    assert found >= 0


def test_ic_angle(oracle):
    umax = [15, 15, 15, 15, 14, 14, 14, 13, 13, 12, 11, 10, 9, 8, 6, 3]
    flat = np.full((41, 41), 90, np.uint8)
    assert oracle.ic_angle(flat, 20, 20, umax) == 0.0
    xr = np.tile(np.arange(41, dtype=np.uint8) * 3, (41, 1))
    assert oracle.ic_angle(xr, 20, 20, umax) == 0.0                       # m01 = 0, m10 > 0
    assert abs(oracle.ic_angle(xr[:, ::-1].copy(), 20, 20, umax) - 180) < 1e-4
    assert abs(oracle.ic_angle(xr.T.copy(), 20, 20, umax) - 90) < 1e-4
    assert abs(oracle.ic_angle(xr.T[::-1].copy(), 20, 20, umax) - 270) < 1e-4
    # brute-force moments over the disc
    rng = np.random.default_rng(12)
    img = rng.integers(0, 256, (41, 41), dtype=np.uint8)
    m10 = m01 = 0
    for v in range(-15, 16):
        d = umax[abs(v)]
        for u in range(-d, d + 1):
            m10 += u * int(img[20 + v, 20 + u])
            m01 += v * int(img[20 + v, 20 + u])
    assert oracle.ic_angle(img, 20, 20, umax) == oracle.fast_atan2(m01, m10)


def test_brief_known_answers(oracle):
    pat = np.array(_pattern_from_header(os.path.join(ROOT, "oracle/orb_pattern_data.h"))).reshape(256, 4)
    flat = np.full((41, 41), 77, np.uint8)
    assert (oracle.brief(flat, 20, 20, 33.0) == 0).all()
    ramp = np.tile(np.arange(41, dtype=np.uint8) * 5, (41, 1))            # value = 5*col
    d0 = oracle.brief(ramp, 20, 20, 0.0)
    bits0 = np.unpackbits(d0, bitorder="little")
    assert np.array_equal(bits0, (pat[:, 0] < pat[:, 2]).astype(np.uint8))
    d90 = oracle.brief(ramp, 20, 20, 90.0)                                  # col offset = -y
    assert np.array_equal(np.unpackbits(d90, bitorder="little"), (-pat[:, 1] < -pat[:, 3]).astype(np.uint8))
    d180 = oracle.brief(ramp, 20, 20, 180.0)
    assert np.array_equal(np.unpackbits(d180, bitorder="little"), (-pat[:, 0] < -pat[:, 2]).astype(np.uint8))


def test_octtree_small_known_answer(oracle):
    # one root (w/h ~ 1); four points, one per quadrant; children pushed to the FRONT in order
    # n1..n4 => list order n4,n3,n2,n1 = BR, BL, UR, UL
    c = np.array([(10, 10, 5), (90, 12, 6), (12, 80, 7), (85, 88, 8)], dtype=oracle.CAND_DTYPE)
    assert list(oracle.distribute_octtree(c, 100, 100, 4)) == [3, 2, 1, 0]
    assert list(oracle.distribute_octtree(c, 100, 100, 1)) == [3, 2, 1, 0]   # first split already >= N
    # max response, first wins ties
    c2 = np.array([(10, 10, 5), (11, 11, 9), (12, 12, 9), (80, 80, 3)], dtype=oracle.CAND_DTYPE)
    assert list(oracle.distribute_octtree(c2, 100, 100, 2)) == [3, 1]


def test_octtree_matches_list_model(oracle):
    rng = np.random.default_rng(13)
    for trial in range(40):
        w, h = int(rng.integers(60, 1300)), int(rng.integers(60, 500))
        if round(w / h) < 1:
            continue
        n = int(rng.integers(1, 900))
        xs = rng.integers(0, w, n)
        ys = rng.integers(0, h, n)
        if trial % 3 == 0:      # clustered
            xs = np.clip((rng.normal(w / 2, w / 12, n)).astype(int), 0, w - 1)
            ys = np.clip((rng.normal(h / 2, h / 12, n)).astype(int), 0, h - 1)
        pts = sorted(set(zip(ys.tolist(), xs.tolist())))        # unique, raster order
        sc = rng.integers(7, 120, len(pts))
        c = np.array([(x, y, s) for (y, x), s in zip(pts, sc)], dtype=oracle.CAND_DTYPE)
        N = int(rng.integers(1, 500))
        got = list(oracle.distribute_octtree(c, w, h, N))
        want = M.distribute_octtree([(int(a["x"]), int(a["y"]), int(a["score"])) for a in c], w, h, N)
        assert got == want, (trial, w, h, len(c), N)
        assert len(set(got)) == len(got)
        assert len(got) >= min(N, len(c)) or len(got) == len(c)


def test_three_maxima(oracle):
    assert oracle.three_maxima([0] * 30) == (-1, -1, -1)
    s = [0] * 30
    s[3], s[7], s[9] = 50, 20, 4
    assert oracle.three_maxima(s) == (3, 7, -1)       # third < 10% of first
    s[9] = 5
    assert oracle.three_maxima(s) == (3, 7, 9)
    s[7] = 4
    assert oracle.three_maxima(s) == (3, 9, -1)       # max2 = 5 is not < 0.1*50; max3 = 4 is
    s[9] = 4
    assert oracle.three_maxima(s) == (3, -1, -1)
    s = [0] * 30
    s[0] = s[1] = s[2] = s[3] = 10
    assert oracle.three_maxima(s) == (0, 1, 2)        # strict '>' keeps the first of equals


def _random_fv(rng, n, nnodes):
    node = rng.integers(0, nnodes, n)
    fv = {}
    for i in range(n):
        fv.setdefault(int(node[i]), []).append(i)
    ids = sorted(fv)
    off = np.cumsum([0] + [len(fv[k]) for k in ids]).astype(np.int32)
    idx = np.concatenate([fv[k] for k in ids]).astype(np.int32) if ids else np.zeros(0, np.int32)
    return fv, (np.array(ids, np.int32), off, idx)


@pytest.mark.parametrize("mode", [0, 1])
def test_search_by_bow_matches_model(oracle, mode):
    rng = np.random.default_rng(20 + mode)
    for trial in range(6):
        n1, n2 = int(rng.integers(50, 300)), int(rng.integers(50, 300))
        d2 = rng.integers(0, 256, (n2, 32), dtype=np.uint8)
        src = rng.integers(0, n2, n1)
        d1 = d2[src].copy()
        flips = rng.integers(0, 256, (n1, 32), dtype=np.uint8) & rng.integers(0, 256, (n1, 32), dtype=np.uint8) \
            & rng.integers(0, 256, (n1, 32), dtype=np.uint8)
        d1 ^= flips                                       # ~32 bit flips: around TH_LOW
        a2 = rng.uniform(0, 360, n2).astype(np.float32)
        a1 = (a2[src] + rng.choice([0, 0, 0, 95, 200], n1) + rng.uniform(-5, 5, n1)).astype(np.float32) % np.float32(360)
        v1 = (rng.random(n1) < 0.8).astype(np.uint8)
        v2 = (rng.random(n2) < 0.9).astype(np.uint8) if mode else None
        fv2, g2 = _random_fv(rng, n2, 12)
        node_of2 = {i: k for k, v in fv2.items() for i in v}
        fv1 = {}
        for i in range(n1):
            k = node_of2[int(src[i])] if rng.random() < 0.85 else int(rng.integers(0, 15))
            fv1.setdefault(k, []).append(i)
        ids = sorted(fv1)
        g1 = (np.array(ids, np.int32), np.cumsum([0] + [len(fv1[k]) for k in ids]).astype(np.int32),
              np.concatenate([fv1[k] for k in ids]).astype(np.int32))
        for ori in (True, False):
            n, m12, m21 = oracle.search_by_bow(d1, v1, a1, g1, d2, v2, a2, g2, th=50, th_mode=mode,
                                               nnratio=0.75, check_ori=ori)
            wn, w12, w21 = M.search_by_bow(d1, v1, a1, fv1, d2, v2, a2, fv2, 50, bool(mode), 0.75, ori)
            assert n == wn and list(m12) == w12 and list(m21) == w21
            assert n == int((m12 >= 0).sum()) and n > 5


def test_extract_pipeline_consistency(oracle):
    from orbhip import synth
    img = synth.make_frames(3, 640, 480, 1)[0]
    ex = oracle.Extractor(1000, 1.2, 8, 20, 7)
    kps, desc = ex(img)
    assert len(kps) >= 1000 and len(kps) <= 1000 + 3 * 8 and desc.shape == (len(kps), 32)
    assert np.array_equal(ex.pyramid(0), img)
    P = ex.params
    off = 0
    for l in range(8):
        lk = ex.level_keypoints(l)
        c = ex.level_cands(l)
        lvl = ex.pyramid(l)
        assert lvl.shape == tuple(reversed(oracle.level_size(P, 640, 480, l)))
        if l:
            assert np.array_equal(lvl, oracle.resize_linear(ex.pyramid(l - 1), lvl.shape[1], lvl.shape[0]))
        assert np.array_equal(ex.blurred(l), oracle.gaussian_blur7(lvl))
        # candidates live in [3, w-32-3) x [3, h-32-3) relative to (16,16)
        assert c["x"].min() >= 3 and c["x"].max() < lvl.shape[1] - 35
        assert c["y"].min() >= 3 and c["y"].max() < lvl.shape[0] - 35
        out = kps[off:off + len(lk)]
        assert (out["octave"] == l).all() and (out["class_id"] == -1).all()
        assert (out["size"] == np.float32(int(np.float32(31) * np.float32(P.mvScaleFactor[l])))).all()
        sc = np.float32(P.mvScaleFactor[l]) if l else np.float32(1)
        assert np.array_equal(out["x"], lk["x"] * sc) and np.array_equal(out["y"], lk["y"] * sc)
        for i in (0, len(lk) // 2, len(lk) - 1):
            x, y = int(lk["x"][i]), int(lk["y"][i])
            assert lk["angle"][i] == oracle.ic_angle(lvl, x, y, list(P.umax))
            assert np.array_equal(desc[off + i], oracle.brief(ex.blurred(l), x, y, float(lk["angle"][i])))
        off += len(lk)
    assert off == len(kps)
