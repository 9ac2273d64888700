"""ctypes binding of the CPU ORACLE (oracle/liborb_oracle.so).

Test infrastructure only: import from tests/, __graft_entry__.smoke() and the cpu_baseline leg
of bench.py.  Never from the product package.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "liborb_oracle.so")

KP_DTYPE = np.dtype([("x", "<f4"), ("y", "<f4"), ("size", "<f4"), ("angle", "<f4"),
                     ("response", "<f4"), ("octave", "<i4"), ("class_id", "<i4")])
CAND_DTYPE = np.dtype([("x", "<i4"), ("y", "<i4"), ("score", "<i4")])
MAX_LEVELS = 16


class Params(C.Structure):
    _fields_ = [("nfeatures", C.c_int), ("nlevels", C.c_int), ("iniThFAST", C.c_int),
                ("minThFAST", C.c_int), ("scaleFactor", C.c_double),
                ("mvScaleFactor", C.c_float * MAX_LEVELS),
                ("mvInvScaleFactor", C.c_float * MAX_LEVELS),
                ("mvLevelSigma2", C.c_float * MAX_LEVELS),
                ("mvInvLevelSigma2", C.c_float * MAX_LEVELS),
                ("mnFeaturesPerLevel", C.c_int * MAX_LEVELS),
                ("umax", C.c_int * 16)]


def build(force=False):
    if force or not os.path.exists(_SO):
        subprocess.check_call(["make", "-C", _HERE, "all"], stdout=subprocess.DEVNULL)
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_SO)
        u8p = C.POINTER(C.c_uint8)
        i32p = C.POINTER(C.c_int32)
        f32p = C.POINTER(C.c_float)
        vp = C.c_void_p
        L.orbo_params_init.argtypes = [C.POINTER(Params), C.c_int, C.c_float, C.c_int, C.c_int, C.c_int]
        L.orbo_level_size.argtypes = [C.POINTER(Params), C.c_int, C.c_int, C.c_int, i32p, i32p]
        L.orbo_cvround.argtypes = [C.c_double]
        L.orbo_fast_atan2.argtypes = [C.c_float, C.c_float]
        L.orbo_fast_atan2.restype = C.c_float
        L.orbo_resize_linear_u8.argtypes = [vp, C.c_int, C.c_int, C.c_int, vp, C.c_int, C.c_int, C.c_int]
        L.orbo_resize_linear_u8.restype = None
        L.orbo_gaussian_blur7_u8.argtypes = [vp, C.c_int, C.c_int, C.c_int, vp, C.c_int]
        L.orbo_gaussian_blur7_u8.restype = None
        L.orbo_fast9_16.argtypes = [vp, C.c_int, C.c_int, C.c_int, C.c_int, vp, C.c_int]
        L.orbo_fast_corner_score.argtypes = [vp, C.c_int, C.c_int]
        L.orbo_level_candidates.argtypes = [vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, vp, C.c_int]
        L.orbo_distribute_octtree.argtypes = [vp, C.c_int, C.c_int, C.c_int, C.c_int, vp, C.c_int]
        L.orbo_ic_angle.argtypes = [vp, C.c_int, C.c_int, C.c_int, i32p]
        L.orbo_ic_angle.restype = C.c_float
        L.orbo_brief.argtypes = [vp, C.c_int, C.c_int, C.c_int, C.c_float, vp]
        L.orbo_brief.restype = None
        L.orbo_create.argtypes = [C.c_int, C.c_float, C.c_int, C.c_int, C.c_int]
        L.orbo_create.restype = vp
        L.orbo_destroy.argtypes = [vp]
        L.orbo_destroy.restype = None
        L.orbo_get_params.argtypes = [vp]
        L.orbo_get_params.restype = C.POINTER(Params)
        L.orbo_extract.argtypes = [vp, vp, C.c_int, C.c_int, C.c_int, vp, vp, C.c_int]
        for name in ("orbo_pyramid_level", "orbo_blurred_level"):
            f = getattr(L, name)
            f.argtypes = [vp, C.c_int, i32p, i32p, i32p]
            f.restype = vp
        L.orbo_level_cands.argtypes = [vp, C.c_int, C.POINTER(vp)]
        L.orbo_level_keypoints.argtypes = [vp, C.c_int, C.POINTER(vp)]
        L.orbo_descriptor_distance.argtypes = [vp, vp]
        L.orbo_knn2.argtypes = [vp, C.c_int, vp, C.c_int, vp, vp, vp]
        L.orbo_knn2.restype = None
        L.orbo_knn2_lists.argtypes = [vp, C.c_int, vp, vp, vp, vp, vp, vp]
        L.orbo_knn2_lists.restype = None
        L.orbo_three_maxima.argtypes = [i32p, C.c_int, i32p, i32p, i32p]
        L.orbo_three_maxima.restype = None
        L.orbo_search_by_bow.argtypes = [vp, C.c_int, vp, vp, vp, vp, vp, C.c_int,
                                         vp, C.c_int, vp, vp, vp, vp, vp, C.c_int,
                                         C.c_int, C.c_int, C.c_float, C.c_int, vp, vp]
        L.orbo_vocab_load.argtypes = [vp, C.c_size_t]
        L.orbo_vocab_load.restype = vp
        L.orbo_vocab_free.argtypes = [vp]
        L.orbo_vocab_free.restype = None
        L.orbo_vocab_info.argtypes = [vp, i32p, i32p, i32p, i32p, i32p, i32p]
        L.orbo_vocab_transform.argtypes = [vp, vp, C.c_int, C.c_int, vp, vp, vp]
        L.orbo_vocab_transform.restype = None
        L.orbo_vocab_bow.argtypes = [vp, vp, vp, C.c_int, vp, vp]
        _lib = L
    return _lib


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def _u8(img):
    img = np.ascontiguousarray(img, dtype=np.uint8)
    assert img.ndim == 2
    return img


def params(nfeatures=1000, scale=1.2, nlevels=8, ini=20, mn=7):
    P = Params()
    rc = lib().orbo_params_init(C.byref(P), nfeatures, scale, nlevels, ini, mn)
    if rc != 0:
        raise ValueError("orbo_params_init failed")
    return P


def level_size(P, cols, rows, level):
    w, h = C.c_int32(), C.c_int32()
    lib().orbo_level_size(C.byref(P), cols, rows, level, C.byref(w), C.byref(h))
    return w.value, h.value


def cvround(v):
    return lib().orbo_cvround(float(v))


def fast_atan2(y, x):
    return lib().orbo_fast_atan2(float(y), float(x))


def resize_linear(src, dw, dh):
    src = _u8(src)
    dst = np.empty((dh, dw), np.uint8)
    lib().orbo_resize_linear_u8(_p(src), src.shape[1], src.shape[0], src.strides[0], _p(dst), dw, dh, dw)
    return dst


def gaussian_blur7(src):
    src = _u8(src)
    dst = np.empty_like(src)
    lib().orbo_gaussian_blur7_u8(_p(src), src.shape[1], src.shape[0], src.strides[0], _p(dst), dst.strides[0])
    return dst


def fast9_16(img, th):
    img = _u8(img)
    cap = img.size // 2 + 16
    out = np.empty(cap, CAND_DTYPE)
    n = lib().orbo_fast9_16(_p(img), img.strides[0], img.shape[1], img.shape[0], th, _p(out), cap)
    assert n >= 0
    return out[:n].copy()


def fast_corner_score(img, x, y, th):
    img = _u8(img)
    ptr = img.ctypes.data + y * img.strides[0] + x
    return lib().orbo_fast_corner_score(C.c_void_p(ptr), img.strides[0], th)


def level_candidates(img, ini=20, mn=7):
    img = _u8(img)
    cap = img.size // 2 + 64
    out = np.empty(cap, CAND_DTYPE)
    n = lib().orbo_level_candidates(_p(img), img.shape[1], img.shape[0], img.strides[0], ini, mn, _p(out), cap)
    if n < 0:
        raise ValueError("orbo_level_candidates rc=%d" % n)
    return out[:n].copy()


def distribute_octtree(cands, width, height, N):
    cands = np.ascontiguousarray(cands, dtype=CAND_DTYPE)
    out = np.empty(len(cands) + 1, np.int32)
    n = lib().orbo_distribute_octtree(_p(cands), len(cands), width, height, N, _p(out), len(out))
    if n < 0:
        raise ValueError("orbo_distribute_octtree rc=%d" % n)
    return out[:n].copy()


def ic_angle(img, x, y, umax):
    img = _u8(img)
    um = (C.c_int32 * 16)(*umax)
    return lib().orbo_ic_angle(_p(img), img.strides[0], x, y, um)


def brief(blurred, x, y, angle_deg):
    blurred = _u8(blurred)
    d = np.empty(32, np.uint8)
    lib().orbo_brief(_p(blurred), blurred.strides[0], x, y, angle_deg, _p(d))
    return d


class Extractor:
    """Mirror of ORBextractor (include/ORBextractor.h:69-103) over the oracle."""

    def __init__(self, nfeatures=1000, scale=1.2, nlevels=8, ini=20, mn=7):
        self._h = lib().orbo_create(nfeatures, scale, nlevels, ini, mn)
        if not self._h:
            raise ValueError("orbo_create failed")
        self.nfeatures, self.nlevels = nfeatures, nlevels
        self.params = lib().orbo_get_params(self._h).contents

    def close(self):
        if self._h:
            lib().orbo_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __call__(self, img):
        img = _u8(img)
        cap = 4 * self.nfeatures + 64
        kps = np.zeros(cap, KP_DTYPE)
        desc = np.zeros((cap, 32), np.uint8)
        n = lib().orbo_extract(self._h, _p(img), img.shape[1], img.shape[0], img.strides[0], _p(kps), _p(desc), cap)
        if n < 0:
            raise ValueError("orbo_extract rc=%d" % n)
        return kps[:n].copy(), desc[:n].copy()

    def _level_img(self, fn, level):
        w, h, s = C.c_int32(), C.c_int32(), C.c_int32()
        ptr = fn(self._h, level, C.byref(w), C.byref(h), C.byref(s))
        if not ptr:
            return None
        buf = (C.c_uint8 * (s.value * h.value)).from_address(ptr)
        return np.frombuffer(buf, np.uint8).reshape(h.value, s.value)[:, :w.value].copy()

    def pyramid(self, level):
        return self._level_img(lib().orbo_pyramid_level, level)

    def blurred(self, level):
        return self._level_img(lib().orbo_blurred_level, level)

    def level_cands(self, level):
        ptr = C.c_void_p()
        n = lib().orbo_level_cands(self._h, level, C.byref(ptr))
        if n <= 0:
            return np.empty(0, CAND_DTYPE)
        buf = (C.c_uint8 * (n * CAND_DTYPE.itemsize)).from_address(ptr.value)
        return np.frombuffer(buf, CAND_DTYPE).copy()

    def level_keypoints(self, level):
        ptr = C.c_void_p()
        n = lib().orbo_level_keypoints(self._h, level, C.byref(ptr))
        if n <= 0:
            return np.empty(0, KP_DTYPE)
        buf = (C.c_uint8 * (n * KP_DTYPE.itemsize)).from_address(ptr.value)
        return np.frombuffer(buf, KP_DTYPE).copy()


def descriptor_distance(a, b):
    a = np.ascontiguousarray(a, np.uint8)
    b = np.ascontiguousarray(b, np.uint8)
    return lib().orbo_descriptor_distance(_p(a), _p(b))


def knn2(q, db):
    q = np.ascontiguousarray(q, np.uint8).reshape(-1, 32)
    db = np.ascontiguousarray(db, np.uint8).reshape(-1, 32)
    bi = np.empty(len(q), np.int32)
    bd = np.empty(len(q), np.int32)
    sd = np.empty(len(q), np.int32)
    lib().orbo_knn2(_p(q), len(q), _p(db), len(db), _p(bi), _p(bd), _p(sd))
    return bi, bd, sd


def knn2_lists(q, db, off, cand):
    q = np.ascontiguousarray(q, np.uint8).reshape(-1, 32)
    db = np.ascontiguousarray(db, np.uint8).reshape(-1, 32)
    off = np.ascontiguousarray(off, np.int32)
    cand = np.ascontiguousarray(cand, np.int32)
    bi = np.empty(len(q), np.int32)
    bd = np.empty(len(q), np.int32)
    sd = np.empty(len(q), np.int32)
    lib().orbo_knn2_lists(_p(q), len(q), _p(db), _p(off), _p(cand), _p(bi), _p(bd), _p(sd))
    return bi, bd, sd


def three_maxima(sizes):
    arr = (C.c_int32 * len(sizes))(*[int(s) for s in sizes])
    a, b, c = C.c_int32(), C.c_int32(), C.c_int32()
    lib().orbo_three_maxima(arr, len(sizes), C.byref(a), C.byref(b), C.byref(c))
    return a.value, b.value, c.value


def search_by_bow(desc1, valid1, angle1, groups1, desc2, valid2, angle2, groups2, th=50,
                  th_mode=0, nnratio=0.7, check_ori=True):
    """groups = (node_ids[ng], offsets[ng+1], idx[...]) (a DBoW2::FeatureVector in CSR form)."""
    desc1 = np.ascontiguousarray(desc1, np.uint8).reshape(-1, 32)
    desc2 = np.ascontiguousarray(desc2, np.uint8).reshape(-1, 32)
    n1, n2 = len(desc1), len(desc2)
    valid1 = np.ascontiguousarray(valid1, np.uint8)
    valid2 = None if valid2 is None else np.ascontiguousarray(valid2, np.uint8)
    angle1 = np.ascontiguousarray(angle1, np.float32)
    angle2 = np.ascontiguousarray(angle2, np.float32)
    g1 = [np.ascontiguousarray(a, np.int32) for a in groups1]
    g2 = [np.ascontiguousarray(a, np.int32) for a in groups2]
    m12 = np.empty(n1, np.int32)
    m21 = np.empty(n2, np.int32)
    n = lib().orbo_search_by_bow(_p(desc1), n1, _p(valid1), _p(angle1), _p(g1[0]), _p(g1[1]), _p(g1[2]), len(g1[0]),
                                 _p(desc2), n2, _p(valid2), _p(angle2), _p(g2[0]), _p(g2[1]), _p(g2[2]), len(g2[0]),
                                 th, th_mode, nnratio, 1 if check_ori else 0, _p(m12), _p(m21))
    return n, m12, m21


class Vocabulary:
    """ORBVocabulary (include/ORBVocabulary.h) over the oracle: loadFromBinaryFile + transform."""

    def __init__(self, blob):
        blob = bytes(blob)
        self._h = lib().orbo_vocab_load(blob, len(blob))
        if not self._h:
            raise ValueError("bad vocabulary blob")
        vals = [C.c_int32() for _ in range(6)]
        lib().orbo_vocab_info(self._h, *[C.byref(v) for v in vals])
        self.k, self.L, self.scoring, self.weighting, self.nnodes, self.nwords = [v.value for v in vals]

    def close(self):
        if self._h:
            lib().orbo_vocab_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def transform(self, desc, levelsup=4):
        desc = np.ascontiguousarray(desc, np.uint8).reshape(-1, 32)
        n = len(desc)
        w = np.empty(n, np.int32)
        wt = np.empty(n, np.float32)
        nid = np.empty(n, np.int32)
        lib().orbo_vocab_transform(self._h, _p(desc), n, levelsup, _p(w), _p(wt), _p(nid))
        return w, wt, nid

    def bow(self, word, weight):
        word = np.ascontiguousarray(word, np.int32)
        weight = np.ascontiguousarray(weight, np.float32)
        ow = np.empty(len(word) + 1, np.int32)
        ov = np.empty(len(word) + 1, np.float64)
        m = lib().orbo_vocab_bow(self._h, _p(word), _p(weight), len(word), _p(ow), _p(ov))
        return ow[:m].copy(), ov[:m].copy()


def vocabulary_text_to_blob(text):
    """Restatement of ORBVocabulary::loadFromTextFile (Thirdparty/DBoW2/DBoW2/TemplatedVocabulary.h:1564-1647) followed by
    saveToBinaryFile (:1727-1751) -- the conversion tools/bin_vocabulary.cc performs: returns (blob, weights) with the
    weights as the doubles the text loader stores in Node::weight (node ids 1..n).  Pure Python; test sizes only.
    Canonical choice: a line without tokens is skipped (the reference's `while(!f.eof())` loop turns the empty string
    after the last newline into a node made of uninitialised memory).  Raises ValueError where the reference returns
    false (:1585-1589) or would index out of range."""
    import struct
    lines = bytes(text).decode("ascii").split("\n")
    head = lines[0].split()
    k, L, n1, n2 = (int(t) for t in head[:4])                          # :1576-1583
    if k < 0 or k > 20 or L < 1 or L > 10 or n1 < 0 or n1 > 5 or n2 < 0 or n2 > 3:
        raise ValueError("Vocabulary loading failure: This is not a correct text file!")   # :1585-1589
    recs, weights = [], []
    for ln in lines[1:]:
        tok = ln.split()
        if not tok:
            continue
        pid, leaf = int(tok[0]), int(tok[1])                           # :1611-1617
        if pid < 0 or pid > len(recs):
            raise ValueError("parent id out of range")
        d = bytes(int(t) & 0xFF for t in tok[2:34])                    # FORB::fromString, FORB.cpp:117-131
        w = float(tok[34])                                             # ssnode >> m_nodes[nid].weight (double), :1628
        recs.append(struct.pack("<i32sfB", pid, d, np.float32(w), 1 if leaf > 0 else 0))
        weights.append(w)
    blob = struct.pack("<IIiiii", len(recs) + 1, 41, k, L, n1, n2) + b"".join(recs)
    return blob, np.array(weights, np.float64)


def bow_vector64(word, weight64, scoring, weighting):
    """BowVector of TemplatedVocabulary::transform (:1167-1258) from per-feature word ids and DOUBLE weights (a
    text-loaded vocabulary), features in ascending index order: the same steps as orbo_vocab_bow."""
    acc = {}
    accumulate = weighting in (0, 1)
    for w, v in zip(word, weight64):
        if not v > 0:
            continue
        if int(w) in acc:
            if accumulate:
                acc[int(w)] += float(v)
        else:
            acc[int(w)] = float(v)
    keys = sorted(acc)
    vals = [acc[k] for k in keys]
    must = scoring != 5
    if accumulate and vals and not must:
        vals = [v / float(len(vals)) for v in vals]
    if must and vals:
        norm = 0.0
        if scoring == 1:
            for v in vals:
                norm += v * v
            norm = norm ** 0.5
        else:
            for v in vals:
                norm += abs(v)
        if norm > 0.0:
            vals = [v / norm for v in vals]
    return np.array(keys, np.int32), np.array(vals, np.float64)


def feature_vector(node_id, weight=None):
    """DBoW2::FeatureVector in CSR form from per-feature node ids (ascending index inside a node);
    features with weight <= 0 are "stopped" and left out (TemplatedVocabulary.h:1334-1338)."""
    node_id = np.asarray(node_id)
    keep = np.arange(len(node_id)) if weight is None else np.nonzero(np.asarray(weight) > 0)[0]
    ids = sorted(set(int(v) for v in node_id[keep]))
    lists = [keep[node_id[keep] == k] for k in ids]
    off = np.concatenate([[0], np.cumsum([len(x) for x in lists])]).astype(np.int32)
    idx = (np.concatenate(lists) if lists else np.zeros(0)).astype(np.int32)
    return np.array(ids, np.int32), off, idx


def stereo_matches(exL, kL, dL, exR, kR, dR, mb, mbf):
    """Frame::ComputeStereoMatches (src/Frame.cc:810-984) on two oracle Extractors that have just
    processed the left / right image.  Returns (mvuRight, mvDepth, n_before_cut)."""
    L = lib()
    L.orbo_stereo_matches.argtypes = [C.c_void_p] * 2 + [C.c_int] + [C.c_void_p] * 2 + [C.c_int] + [C.c_void_p] * 6 + \
        [C.c_float, C.c_float, C.c_void_p, C.c_void_p]
    nl = exL.nlevels
    pl = [np.ascontiguousarray(exL.pyramid(l)) for l in range(nl)]
    pr = [np.ascontiguousarray(exR.pyramid(l)) for l in range(nl)]
    lw = np.array([p.shape[1] for p in pl], np.int32)
    lh = np.array([p.shape[0] for p in pl], np.int32)
    PL = (C.c_void_p * nl)(*[p.ctypes.data for p in pl])
    PR = (C.c_void_p * nl)(*[p.ctypes.data for p in pr])
    P = exL.params
    sf = np.array(list(P.mvScaleFactor)[:nl], np.float32)
    isf = np.array(list(P.mvInvScaleFactor)[:nl], np.float32)
    kL = np.ascontiguousarray(kL)
    kR = np.ascontiguousarray(kR)
    dL = np.ascontiguousarray(dL, np.uint8)
    dR = np.ascontiguousarray(dR, np.uint8)
    u = np.empty(max(len(kL), 1), np.float32)
    z = np.empty(max(len(kL), 1), np.float32)
    n = L.orbo_stereo_matches(_p(kL), _p(dL), len(kL), _p(kR), _p(dR), len(kR), PL, PR, _p(lw), _p(lh), _p(sf), _p(isf),
                              mb, mbf, _p(u), _p(z))
    return u[:len(kL)].copy(), z[:len(kL)].copy(), n


# ---- frame grid + guided search ----
Q_ACTIVE, Q_OBSERVED = 1, 2
QUERY_DTYPE = np.dtype([("u", "<f4"), ("v", "<f4"), ("radius", "<f4"), ("proj_xr", "<f4"), ("min_level", "<i4"),
                        ("max_level", "<i4"), ("angle", "<f4"), ("flags", "<i4")])
GRID_COLS, GRID_ROWS = 64, 48


def grid_params(min_x, max_x, min_y, max_y):
    """(mnMinX, mnMinY, mfGridElementWidthInv, mfGridElementHeightInv) as float32 -- src/Frame.cc:556-557."""
    f = np.float32
    return f(min_x), f(min_y), f(GRID_COLS) / (f(max_x) - f(min_x)), f(GRID_ROWS) / (f(max_y) - f(min_y))


def grid_build(kps, gp):
    """Frame::AssignFeaturesToGrid (src/Frame.cc:574-589) as CSR: cell id = ix * 48 + iy."""
    L = lib()
    L.orbo_grid_build.argtypes = [C.c_void_p, C.c_int] + [C.c_float] * 4 + [C.c_void_p] * 2
    kps = np.ascontiguousarray(kps)
    off = np.zeros(GRID_COLS * GRID_ROWS + 1, np.int32)
    idx = np.zeros(max(len(kps), 1), np.int32)
    L.orbo_grid_build(_p(kps), len(kps), gp[0], gp[1], gp[2], gp[3], _p(off), _p(idx))
    return off, idx[:off[-1]].copy()


def features_in_area(kps, grid, gp, x, y, r, min_level=-1, max_level=-1):
    """Frame::GetFeaturesInArea (src/Frame.cc:671-724)."""
    L = lib()
    L.orbo_features_in_area.argtypes = [C.c_void_p] * 3 + [C.c_float] * 7 + [C.c_int] * 2 + [C.c_void_p, C.c_int]
    L.orbo_features_in_area.restype = C.c_int
    kps = np.ascontiguousarray(kps)
    off, idx = grid
    out = np.zeros(max(len(kps), 1), np.int32)
    idx = np.ascontiguousarray(idx if len(idx) else np.zeros(1, np.int32))
    n = L.orbo_features_in_area(_p(kps), _p(off), _p(idx), gp[0], gp[1], gp[2], gp[3], x, y, r, min_level, max_level,
                                _p(out), len(out))
    return out[:n].copy()


def search_by_projection(kps, desc, gp, queries, qdesc, u_right=None, occupied=None, use_ratio=True, nnratio=0.8,
                         check_ori=True, th_high=100):
    """Both ORBmatcher::SearchByProjection variants over prepared queries (src/ORBmatcher.cc:45-129 with
    use_ratio, :1341-1498 without).  Returns (nmatches, match[feature] = query index or -1)."""
    L = lib()
    L.orbo_search_by_projection.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p] + [C.c_float] * 4 + \
        [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_float, C.c_int, C.c_int, C.c_void_p]
    L.orbo_search_by_projection.restype = C.c_int
    kps = np.ascontiguousarray(kps)
    desc = np.ascontiguousarray(desc, np.uint8)
    queries = np.ascontiguousarray(queries, QUERY_DTYPE)
    qdesc = np.ascontiguousarray(qdesc, np.uint8)
    ur = None if u_right is None else np.ascontiguousarray(u_right, np.float32)
    occ = None if occupied is None else np.ascontiguousarray(occupied, np.uint8)
    match = np.empty(max(len(kps), 1), np.int32)
    n = L.orbo_search_by_projection(_p(kps), _p(desc), len(kps), None if ur is None else _p(ur),
                                    None if occ is None else _p(occ), gp[0], gp[1], gp[2], gp[3], _p(queries), _p(qdesc),
                                    len(queries), 1 if use_ratio else 0, nnratio, 1 if check_ori else 0, th_high, _p(match))
    return n, match[:len(kps)].copy()


def distinctive_descriptors(desc, off):
    """MapPoint::ComputeDistinctiveDescriptors (src/MapPoint.cc:283-349) for the points whose observed descriptors are
    rows off[p]..off[p+1] of desc: (best row within each list or -1, its median distance)."""
    L = lib()
    L.orbo_distinctive_descriptors.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
    L.orbo_distinctive_descriptors.restype = None
    desc = np.ascontiguousarray(desc, np.uint8).reshape(-1, 32)
    off = np.ascontiguousarray(off, np.int32)
    P = len(off) - 1
    best = np.empty(max(P, 1), np.int32)
    med = np.empty(max(P, 1), np.int32)
    if len(desc) == 0:
        desc = np.zeros((1, 32), np.uint8)
    L.orbo_distinctive_descriptors(_p(desc), _p(off), P, _p(best), _p(med))
    return best[:P].copy(), med[:P].copy()


def window_best(kps, desc, gp, queries, qdesc, u_right=None, inv_level_sigma2=None):
    """Per-query best feature in the window: the inner loop of ORBmatcher::Fuse (src/ORBmatcher.cc:887-950 with the
    chi-square gate = inv_level_sigma2 given; :1044-1075 without) and SearchBySim3 (:1190-1224).
    Returns (best_idx[nq], best_dist[nq])."""
    L = lib()
    L.orbo_window_best.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p] + [C.c_float] * 4 + \
        [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
    L.orbo_window_best.restype = None
    kps = np.ascontiguousarray(kps)
    desc = np.ascontiguousarray(desc, np.uint8)
    queries = np.ascontiguousarray(queries, QUERY_DTYPE)
    qdesc = np.ascontiguousarray(qdesc, np.uint8)
    ur = None if u_right is None else np.ascontiguousarray(u_right, np.float32)
    sg = None if inv_level_sigma2 is None else np.ascontiguousarray(inv_level_sigma2, np.float32)
    bi = np.empty(max(len(queries), 1), np.int32)
    bd = np.empty(max(len(queries), 1), np.int32)
    L.orbo_window_best(_p(kps), _p(desc), len(kps), None if ur is None else _p(ur), None if sg is None else _p(sg),
                       gp[0], gp[1], gp[2], gp[3], _p(queries), _p(qdesc), len(queries), _p(bi), _p(bd))
    return bi[:len(queries)].copy(), bd[:len(queries)].copy()


def search_for_initialization(kps1, desc1, kps2, desc2, gp, prev_matched, window_size=100, nnratio=0.9, check_ori=True,
                              th_low=50):
    """ORBmatcher::SearchForInitialization (src/ORBmatcher.cc:405-520).  Returns (nmatches, matches12, prev_matched')."""
    L = lib()
    L.orbo_search_for_initialization.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int] + \
        [C.c_float] * 4 + [C.c_void_p, C.c_int, C.c_float, C.c_int, C.c_int, C.c_void_p]
    L.orbo_search_for_initialization.restype = C.c_int
    kps1 = np.ascontiguousarray(kps1)
    kps2 = np.ascontiguousarray(kps2)
    desc1 = np.ascontiguousarray(desc1, np.uint8)
    desc2 = np.ascontiguousarray(desc2, np.uint8)
    prev = np.array(prev_matched, np.float32).reshape(-1, 2).copy()
    assert len(prev) == len(kps1)
    m12 = np.empty(max(len(kps1), 1), np.int32)
    n = L.orbo_search_for_initialization(_p(kps1), _p(desc1), len(kps1), _p(kps2), _p(desc2), len(kps2), gp[0], gp[1], gp[2],
                                         gp[3], _p(prev), int(window_size), nnratio, 1 if check_ori else 0, th_low, _p(m12))
    return n, m12[:len(kps1)].copy(), prev


def search_for_triangulation(kps1, desc1, skip1, groups1, kps2, desc2, skip2, groups2, F12, ex, ey, scale_factors2,
                             level_sigma2_2, u_right1=None, u_right2=None, only_stereo=False, check_ori=True, th_low=50):
    """ORBmatcher::SearchForTriangulation (src/ORBmatcher.cc:657-827).  groups = FeatureVector in CSR form (node ids,
    offsets, feature indices).  Returns (nmatches, matches12)."""
    L = lib()
    vp = C.c_void_p
    L.orbo_search_for_triangulation.argtypes = [vp, vp, C.c_int, vp, vp, vp, vp, vp, C.c_int] * 2 + \
        [vp, C.c_float, C.c_float, vp, vp, C.c_int, C.c_int, C.c_int, vp]
    L.orbo_search_for_triangulation.restype = C.c_int
    kps1, kps2 = np.ascontiguousarray(kps1), np.ascontiguousarray(kps2)
    desc1 = np.ascontiguousarray(desc1, np.uint8).reshape(-1, 32)
    desc2 = np.ascontiguousarray(desc2, np.uint8).reshape(-1, 32)
    skip1, skip2 = np.ascontiguousarray(skip1, np.uint8), np.ascontiguousarray(skip2, np.uint8)
    g1 = [np.ascontiguousarray(a, np.int32) for a in groups1]
    g2 = [np.ascontiguousarray(a, np.int32) for a in groups2]
    ur1 = None if u_right1 is None else np.ascontiguousarray(u_right1, np.float32)
    ur2 = None if u_right2 is None else np.ascontiguousarray(u_right2, np.float32)
    F = np.ascontiguousarray(F12, np.float32).reshape(9)
    sf = np.ascontiguousarray(scale_factors2, np.float32)
    s2 = np.ascontiguousarray(level_sigma2_2, np.float32)
    m12 = np.empty(max(len(kps1), 1), np.int32)
    n = L.orbo_search_for_triangulation(_p(kps1), _p(desc1), len(kps1), _p(skip1), None if ur1 is None else _p(ur1),
                                        _p(g1[0]), _p(g1[1]), _p(g1[2]), len(g1[0]), _p(kps2), _p(desc2), len(kps2), _p(skip2),
                                        None if ur2 is None else _p(ur2), _p(g2[0]), _p(g2[1]), _p(g2[2]), len(g2[0]), _p(F),
                                        float(ex), float(ey), _p(sf), _p(s2), 1 if only_stereo else 0, 1 if check_ori else 0,
                                        th_low, _p(m12))
    return n, m12[:len(kps1)].copy()


# ---- undistortion / rectification ----
def undistort_points(xy, K, D, P=None):
    """cv::undistortPoints(xy, K, D, Mat(), P) -- Frame::UndistortKeyPoints (src/Frame.cc:748-778)."""
    L = lib()
    L.orbo_undistort_points.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
    xy = np.ascontiguousarray(xy, np.float32).reshape(-1, 2)
    K = np.ascontiguousarray(K, np.float32).reshape(9)
    D = np.ascontiguousarray(D, np.float32).ravel()
    Pm = None if P is None else np.ascontiguousarray(P, np.float32).reshape(9)
    out = np.empty_like(xy)
    L.orbo_undistort_points(_p(xy), len(xy), _p(K), _p(D) if len(D) else None, len(D), None if Pm is None else _p(Pm), _p(out))
    return out


def init_undistort_rectify_map(K, D, R, P, w, h):
    L = lib()
    L.orbo_init_undistort_rectify_map.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_int,
                                                  C.c_void_p, C.c_void_p]
    K = np.ascontiguousarray(K, np.float64).reshape(9)
    D = np.ascontiguousarray(D, np.float64).ravel()
    R = np.ascontiguousarray(R, np.float64).reshape(9)
    P = np.ascontiguousarray(np.asarray(P, np.float64).reshape(3, -1)[:, :3]).reshape(9)
    mx = np.empty((h, w), np.float32)
    my = np.empty((h, w), np.float32)
    L.orbo_init_undistort_rectify_map(_p(K), _p(D), len(D), _p(R), _p(P), w, h, _p(mx), _p(my))
    return mx, my


def remap_prepare(mapx, mapy):
    L = lib()
    L.orbo_remap_prepare.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
    mapx = np.ascontiguousarray(mapx, np.float32)
    mapy = np.ascontiguousarray(mapy, np.float32)
    h, w = mapx.shape
    xy = np.empty((h, w, 2), np.int16)
    fr = np.empty((h, w), np.uint16)
    L.orbo_remap_prepare(_p(mapx), _p(mapy), w, h, _p(xy), _p(fr))
    return xy, fr


def remap_linear(src, mapx, mapy):
    """cv::remap(src, dst, mapx, mapy, INTER_LINEAR), BORDER_CONSTANT 0."""
    L = lib()
    L.orbo_remap_linear_u8.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_int,
                                       C.c_void_p, C.c_int]
    src = _u8(src)
    xy, fr = remap_prepare(mapx, mapy)
    h, w = fr.shape
    dst = np.empty((h, w), np.uint8)
    L.orbo_remap_linear_u8(_p(src), src.shape[1], src.shape[0], src.strides[0], _p(xy), _p(fr), w, h, _p(dst), w)
    return dst
