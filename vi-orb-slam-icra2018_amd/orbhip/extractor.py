"""Python mirror of the reference's operator interface for the hot path, over the C ABI.

ORBextractor mirrors include/ORBextractor.h:69-103 (constructor arguments, operator(), the scale
getters, the three timing getters and mvImagePyramid); ORBmatcher mirrors the in-scope part of
include/ORBmatcher.h:41-89 (DescriptorDistance applied to sets, SearchByBoW, TH_LOW/TH_HIGH/
HISTO_LENGTH).  The real drop-in classes are C++ (include/orbhip/); these wrappers exist so that
the parity tests and bench.py read like the reference's own call sites.
"""
import ctypes as C

import numpy as np

from . import capi
from .capi import CAND_DTYPE, KP_DTYPE, OrbHipError, _p, check


class ORBextractor:
    def __init__(self, nfeatures=1000, scaleFactor=1.2, nlevels=8, iniThFAST=20, minThFAST=7,
                 max_w=1280, max_h=720, max_batch=1, device=0):
        self._L = capi.load()
        self._h = self._L.orbhip_create(device, nfeatures, scaleFactor, nlevels, iniThFAST, minThFAST,
                                        max_w, max_h, max_batch)
        if not self._h:
            raise OrbHipError("orbhip_create: " + capi.last_error(None))
        self.nfeatures, self.nlevels, self.max_batch = nfeatures, nlevels, max_batch
        self.cap = self._L.orbhip_max_keypoints(self._h)
        self._timings = (C.c_float * 3)()
        self._shape = None
        n = C.c_int()
        sf = C.c_double()
        arr = [np.zeros(nlevels, np.float32) for _ in range(4)]
        per = np.zeros(nlevels, np.int32)
        um = np.zeros(16, np.int32)
        check(self._L.orbhip_get_tables(self._h, C.byref(n), C.byref(sf), _p(arr[0]), _p(arr[1]), _p(arr[2]),
                                        _p(arr[3]), _p(per), _p(um)), self._h, "orbhip_get_tables")
        self.scaleFactor = sf.value
        self.mvScaleFactor, self.mvInvScaleFactor, self.mvLevelSigma2, self.mvInvLevelSigma2 = arr
        self.mnFeaturesPerLevel, self.umax = per, um

    # -- lifetime --
    def close(self):
        if getattr(self, "_h", None):
            self._L.orbhip_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- getters of include/ORBextractor.h:81-101, :51-53 --
    def GetLevels(self):
        return self.nlevels

    def GetScaleFactor(self):
        return float(np.float32(self.scaleFactor))

    def GetScaleFactors(self):
        return self.mvScaleFactor.copy()

    def GetInverseScaleFactors(self):
        return self.mvInvScaleFactor.copy()

    def GetScaleSigmaSquares(self):
        return self.mvLevelSigma2.copy()

    def GetInverseScaleSigmaSquares(self):
        return self.mvInvLevelSigma2.copy()

    def GetTimeOfComputePyramid(self):
        return float(self._timings[0])

    def GetTimeOfComputeKeyPointsOctTree(self):
        return float(self._timings[1])

    def GetTImeOfComputeDescriptor(self):   # sic, include/ORBextractor.h:53
        return float(self._timings[2])

    # -- operator() --
    def __call__(self, image):
        """(keypoints, descriptors) of one 8-bit image; keypoints as KP_DTYPE records."""
        if image is None or image.size == 0:
            return np.empty(0, KP_DTYPE), np.empty((0, 32), np.uint8)   # silent return, :1048-1049
        img = np.ascontiguousarray(image, np.uint8)
        assert img.ndim == 2, "CV_8UC1 expected (src/ORBextractor.cc:1052)"
        kps = np.zeros(self.cap, KP_DTYPE)
        desc = np.zeros((self.cap, 32), np.uint8)
        n = C.c_int()
        check(self._L.orbhip_extract(self._h, _p(img), img.shape[1], img.shape[0], img.strides[0], _p(kps),
                                     _p(desc), self.cap, C.byref(n), self._timings), self._h, "orbhip_extract")
        self._shape = img.shape
        return kps[:n.value].copy(), desc[:n.value].copy()

    def frame_build(self, image, K=None, dist_coef=None, gp=None, levelsup=-1):
        """The Frame constructor's device work in one launch (orbhip_frame_build; ref: src/Frame.cc:518-572, 739-746):
        extraction, UndistortKeyPoints with (K, dist_coef), AssignFeaturesToGrid with gp = (min_x, min_y, inv_w, inv_h) and,
        for levelsup >= 0, the vocabulary transform.  Returns a dict: kps, kps_un, desc, cell_off, cell_idx (None without
        gp), word_id, weight, node_id (None without levelsup)."""
        img = np.ascontiguousarray(image, np.uint8)
        assert img.ndim == 2
        P = capi.FrameParams()
        Kf = np.eye(3, dtype=np.float32).ravel() if K is None else np.ascontiguousarray(K, np.float32).ravel()
        D = np.zeros(0, np.float32) if dist_coef is None else np.ascontiguousarray(dist_coef, np.float32).ravel()
        for i in range(9):
            P.K[i] = float(Kf[i])
        for i in range(len(D)):
            P.dist[i] = float(D[i])
        P.ndist = len(D)
        if gp is not None:
            P.min_x, P.min_y, P.inv_w, P.inv_h = [float(v) for v in gp]
        P.levelsup = int(levelsup)
        cap = self.cap
        kps, kun = np.zeros(cap, KP_DTYPE), np.zeros(cap, KP_DTYPE)
        desc = np.zeros((cap, 32), np.uint8)
        off = np.zeros(capi.GRID_COLS * capi.GRID_ROWS + 1, np.int32) if gp is not None else None
        idx = np.zeros(cap, np.int32) if gp is not None else None
        word = np.zeros(cap, np.int32) if levelsup >= 0 else None
        wt = np.zeros(cap, np.float32) if levelsup >= 0 else None
        node = np.zeros(cap, np.int32) if levelsup >= 0 else None
        n = C.c_int()
        check(self._L.orbhip_frame_build(self._h, _p(img), img.shape[1], img.shape[0], img.strides[0], C.byref(P), _p(kps), _p(kun),
                                         _p(desc), cap, C.byref(n), _p(off), _p(idx), _p(word), _p(wt), _p(node)),
              self._h, "orbhip_frame_build")
        self._shape = img.shape
        m = n.value
        cut = lambda a: None if a is None else a[:m].copy()
        if idx is not None:
            idx = idx[:off[-1]].copy()   # (features outside the image bounds are in no cell, :583-587)
        return dict(kps=kps[:m].copy(), kps_un=kun[:m].copy(), desc=desc[:m].copy(), cell_off=off, cell_idx=idx,
                    word_id=cut(word), weight=cut(wt), node_id=cut(node))

    def frame_fingerprint(self):
        return int(self._L.orbhip_frame_fingerprint(self._h))

    def extract_batch(self, images):
        """images: (B, H, W) uint8.  Returns lists of per-frame keypoints and descriptors."""
        imgs = np.ascontiguousarray(images, np.uint8)
        B, H, W = imgs.shape
        ptrs = (C.c_void_p * B)(*[imgs[b].ctypes.data for b in range(B)])
        kps = np.zeros((B, self.cap), KP_DTYPE)
        desc = np.zeros((B, self.cap, 32), np.uint8)
        n = np.zeros(B, np.int32)
        check(self._L.orbhip_extract_batch(self._h, ptrs, B, W, H, imgs.strides[1], _p(kps), _p(desc), self.cap,
                                           _p(n)), self._h, "orbhip_extract_batch")
        self._shape = (H, W)
        return [kps[b, :n[b]].copy() for b in range(B)], [desc[b, :n[b]].copy() for b in range(B)]

    def extract_batch_device(self, d_imgs, B, W, H, stride, frame_stride, d_kps, d_desc, cap, d_counts):
        """Everything resident on the device (raw pointers as ints); asynchronous."""
        check(self._L.orbhip_extract_batch_device(self._h, d_imgs, B, W, H, stride, frame_stride, d_kps, d_desc,
                                                  cap, d_counts), self._h, "orbhip_extract_batch_device")
        self._shape = (H, W)

    # -- host-fed pipeline (orbhip_pipe_*): batches from host memory, copies overlapped with the kernels --
    def pipe_create(self, depth, B, W, H):
        check(self._L.orbhip_pipe_create(self._h, depth, B, W, H), self._h, "orbhip_pipe_create")
        self._shape = (H, W)

    def pipe_destroy(self):
        check(self._L.orbhip_pipe_destroy(self._h), self._h, "orbhip_pipe_destroy")

    def pipe_submit(self, frames):
        """frames: (B, H, W) uint8 array (pinned: see host_frames) or (address, B, stride, frame_stride)."""
        if isinstance(frames, tuple):
            addr, B, stride, fs = frames
        else:
            assert frames.dtype == np.uint8 and frames.ndim == 3 and frames.strides[2] == 1
            addr, B, stride, fs = frames.ctypes.data, frames.shape[0], frames.strides[1], frames.strides[0]
        check(self._L.orbhip_pipe_submit(self._h, addr, B, stride, fs), self._h, "orbhip_pipe_submit")

    def pipe_wait(self, copy=True):
        """Results of the oldest outstanding batch: lists of per-frame keypoints / descriptors (copies), or with
        copy=False the raw views (kps (B, cap), desc (B, cap, 32), counts (B,)) into the pinned result block -- valid until
        the NEXT pipe_wait on this extractor (submits in between do not touch it: include/orbhip.h)."""
        k, d, n = C.c_void_p(), C.c_void_p(), C.c_void_p()
        B, cap = C.c_int(), C.c_int()
        check(self._L.orbhip_pipe_wait(self._h, C.byref(k), C.byref(d), C.byref(n), C.byref(B), C.byref(cap)), self._h,
              "orbhip_pipe_wait")
        B, cap = B.value, cap.value
        cnt = np.frombuffer((C.c_int32 * B).from_address(n.value), np.int32)
        kps = np.frombuffer((C.c_uint8 * (B * cap * 28)).from_address(k.value), KP_DTYPE).reshape(B, cap)
        desc = np.frombuffer((C.c_uint8 * (B * cap * 32)).from_address(d.value), np.uint8).reshape(B, cap, 32)
        if not copy:
            return kps, desc, cnt
        return [kps[b, :cnt[b]].copy() for b in range(B)], [desc[b, :cnt[b]].copy() for b in range(B)]

    def pipe_enable_bow(self, levelsup=4, nnratio=0.7, check_ori=True):
        check(self._L.orbhip_pipe_enable_bow(self._h, levelsup, nnratio, 1 if check_ori else 0), self._h, "orbhip_pipe_enable_bow")

    def pipe_matches(self, B, cap):
        """(match12 (B, cap), match21 (B, cap), nmatches (B,)) views for the batch the last pipe_wait returned."""
        a, b, n = C.c_void_p(), C.c_void_p(), C.c_void_p()
        check(self._L.orbhip_pipe_matches(self._h, C.byref(a), C.byref(b), C.byref(n)), self._h, "orbhip_pipe_matches")
        m12 = np.frombuffer((C.c_int32 * (B * cap)).from_address(a.value), np.int32).reshape(B, cap)
        m21 = np.frombuffer((C.c_int32 * (B * cap)).from_address(b.value), np.int32).reshape(B, cap)
        return m12, m21, np.frombuffer((C.c_int32 * B).from_address(n.value), np.int32)

    def host_frames(self, shape):
        """A pinned (page-locked) uint8 array of the given shape (orbhip_host_alloc); free with host_free(array)."""
        nbytes = int(np.prod(shape))
        ptr = self._L.orbhip_host_alloc(nbytes)
        if not ptr:
            raise OrbHipError("orbhip_host_alloc(%d) failed" % nbytes)
        arr = np.frombuffer((C.c_uint8 * nbytes).from_address(ptr), np.uint8).reshape(shape)
        self._pinned = getattr(self, "_pinned", {})
        self._pinned[arr.ctypes.data] = ptr
        return arr

    def host_free(self, arr):
        ptr = getattr(self, "_pinned", {}).pop(arr.ctypes.data, None)
        if ptr:
            self._L.orbhip_host_free(ptr)

    def sync(self):
        check(self._L.orbhip_sync(self._h), self._h, "orbhip_sync")

    def stream(self):
        return self._L.orbhip_stream(self._h)

    # -- mvImagePyramid (include/ORBextractor.h:103) and stage read-back --
    def _level(self, fn, level, frame):
        w, h = C.c_int(), C.c_int()
        check(fn(self._h, frame, level, None, 0, C.byref(w), C.byref(h)), self._h, "level size")
        out = np.empty((h.value, w.value), np.uint8)
        check(fn(self._h, frame, level, _p(out), w.value, C.byref(w), C.byref(h)), self._h, "level copy")
        return out

    def image_pyramid(self, level, frame=0):
        return self._level(self._L.orbhip_get_pyramid_level, level, frame)

    def set_host_pyramid(self, on=True):
        """Following host-pointer extract calls also land levels 1.. in page-locked host memory (orbhip_set_host_pyramid)."""
        check(self._L.orbhip_set_host_pyramid(self._h, 1 if on else 0), self._h, "set_host_pyramid")

    def host_pyramid(self, level, frame=0):
        """Level `level` of frame `frame` of the last extract call as a copy of the page-locked block the drop-in's
        mvImagePyramid headers point into (orbhip_host_pyramid_level); raises if the level is not staged on the host."""
        ptr, st, w, h = C.c_void_p(), C.c_int(), C.c_int(), C.c_int()
        check(self._L.orbhip_host_pyramid_level(self._h, frame, level, C.byref(ptr), C.byref(st), C.byref(w), C.byref(h)),
              self._h, "host_pyramid_level")
        rows = np.ctypeslib.as_array(C.cast(ptr, C.POINTER(C.c_uint8)), shape=(h.value * st.value,))
        return rows.reshape(h.value, st.value)[:, :w.value].copy()

    @property
    def mvImagePyramid(self):
        return [self.image_pyramid(l) for l in range(self.nlevels)]

    def blurred(self, level, frame=0):
        return self._level(self._L.orbhip_debug_get_blurred_level, level, frame)

    def level_candidates(self, level, frame=0):
        n = C.c_int()
        check(self._L.orbhip_debug_get_candidates(self._h, frame, level, None, 0, C.byref(n)), self._h, "cands")
        out = np.zeros(max(n.value, 1), CAND_DTYPE)
        check(self._L.orbhip_debug_get_candidates(self._h, frame, level, _p(out), len(out), C.byref(n)), self._h,
              "cands")
        return out[:n.value].copy()

    def level_keypoints(self, level, frame=0):
        n = C.c_int()
        check(self._L.orbhip_debug_get_level_keypoints(self._h, frame, level, None, 0, C.byref(n)), self._h, "lkps")
        out = np.zeros(max(n.value, 1), KP_DTYPE)
        check(self._L.orbhip_debug_get_level_keypoints(self._h, frame, level, _p(out), len(out), C.byref(n)),
              self._h, "lkps")
        return out[:n.value].copy()

    def level_size(self, w, h, level):
        lw, lh = C.c_int(), C.c_int()
        check(self._L.orbhip_level_size(self._h, w, h, level, C.byref(lw), C.byref(lh)), self._h, "level_size")
        return lw.value, lh.value

    @property
    def handle(self):
        return self._h


class ORBmatcher:
    TH_LOW = 50          # src/ORBmatcher.cc:38
    TH_HIGH = 100        # :37
    HISTO_LENGTH = 30    # :39

    def __init__(self, nnratio=0.6, checkOri=True, ctx=None):
        self.mfNNratio = float(nnratio)
        self.mbCheckOrientation = bool(checkOri)
        self._own = None
        if ctx is None:
            # matching needs a device context; a tiny extractor context provides stream + scratch
            self._own = ORBextractor(max_w=128, max_h=128, nfeatures=50, nlevels=1)
            ctx = self._own
        self._ctx = ctx
        self._L = capi.load()

    def close(self):
        if self._own is not None:
            self._own.close()
            self._own = None

    @staticmethod
    def DescriptorDistance(a, b):
        """Single pair, host popcount (the static member is a scalar helper in the reference too)."""
        a = np.ascontiguousarray(a, np.uint8).reshape(32)
        b = np.ascontiguousarray(b, np.uint8).reshape(32)
        return int(np.unpackbits(a ^ b).sum())

    def knn2(self, q, db):
        q = np.ascontiguousarray(q, np.uint8).reshape(-1, 32)
        db = np.ascontiguousarray(db, np.uint8).reshape(-1, 32)
        bi = np.empty(len(q), np.int32)
        bd = np.empty(len(q), np.int32)
        sd = np.empty(len(q), np.int32)
        check(self._L.orbhip_hamming_knn2(self._ctx.handle, _p(q), len(q), _p(db), len(db), _p(bi), _p(bd), _p(sd)),
              self._ctx.handle, "orbhip_hamming_knn2")
        return bi, bd, sd

    def knn2_lists(self, q, db, off, cand):
        q = np.ascontiguousarray(q, np.uint8).reshape(-1, 32)
        db = np.ascontiguousarray(db, np.uint8).reshape(-1, 32)
        off = np.ascontiguousarray(off, np.int32)
        cand = np.ascontiguousarray(cand, np.int32)
        bi = np.empty(len(q), np.int32)
        bd = np.empty(len(q), np.int32)
        sd = np.empty(len(q), np.int32)
        check(self._L.orbhip_hamming_knn2_lists(self._ctx.handle, _p(q), len(q), _p(db), len(db), _p(off), _p(cand),
                                                _p(bi), _p(bd), _p(sd)), self._ctx.handle, "orbhip_hamming_knn2_lists")
        return bi, bd, sd

    def SearchByBoW(self, desc1, valid1, angle1, fv1, desc2, valid2, angle2, fv2, kf_kf=False):
        """fv = (node_ids, offsets, indices) CSR FeatureVectors.  kf_kf=False: SearchByBoW(KeyFrame*,
        Frame&) (:159-288); True: SearchByBoW(KeyFrame*, KeyFrame*) (:522-655).
        Returns (nmatches, match12, match21)."""
        desc1 = np.ascontiguousarray(desc1, np.uint8).reshape(-1, 32)
        desc2 = np.ascontiguousarray(desc2, np.uint8).reshape(-1, 32)
        n1, n2 = len(desc1), len(desc2)
        valid1 = np.ascontiguousarray(valid1, np.uint8)
        valid2 = None if valid2 is None else np.ascontiguousarray(valid2, np.uint8)
        angle1 = np.ascontiguousarray(angle1, np.float32)
        angle2 = np.ascontiguousarray(angle2, np.float32)
        g1 = [np.ascontiguousarray(a, np.int32) for a in fv1]
        g2 = [np.ascontiguousarray(a, np.int32) for a in fv2]
        m12 = np.empty(max(n1, 1), np.int32)
        m21 = np.empty(max(n2, 1), np.int32)
        nm = C.c_int()
        check(self._L.orbhip_search_by_bow(self._ctx.handle, _p(desc1), n1, _p(valid1), _p(angle1), _p(g1[0]), _p(g1[1]),
                                           _p(g1[2]), len(g1[0]), _p(desc2), n2, _p(valid2), _p(angle2), _p(g2[0]),
                                           _p(g2[1]), _p(g2[2]), len(g2[0]), self.TH_LOW, 1 if kf_kf else 0,
                                           self.mfNNratio, 1 if self.mbCheckOrientation else 0, _p(m12), _p(m21),
                                           C.byref(nm)), self._ctx.handle, "orbhip_search_by_bow")
        return nm.value, m12[:n1].copy(), m21[:n2].copy()


    # -- resident feature sets (orbhip_set_*): a key frame's data stays on the device across calls --
    def put_set(self, key, kps, desc, fv=None, gp=None):
        """Keeps (kps, desc, FeatureVector CSR fv = (node, off, idx), grid of gp = (min_x, min_y, inv_w, inv_h)) under `key`."""
        kps = np.ascontiguousarray(kps, KP_DTYPE)
        desc = np.ascontiguousarray(desc, np.uint8).reshape(-1, 32)
        g = [None, None, None] if fv is None else [np.ascontiguousarray(a, np.int32) for a in fv]
        gp = (0.0, 0.0, 0.0, 0.0) if gp is None else gp
        check(self._L.orbhip_set_put(self._ctx.handle, key, _p(kps), _p(desc), len(kps), _p(g[0]), _p(g[1]), _p(g[2]),
                                     0 if fv is None else len(g[0]), gp[0], gp[1], gp[2], gp[3]), self._ctx.handle, "orbhip_set_put")

    def put_set_from_frame(self, key, src, fv=None):
        """The frame that extractor `src` built last (frame_build) as resident set `key` of this matcher's context: only the
        FeatureVector CSR travels."""
        g = [None, None, None] if fv is None else [np.ascontiguousarray(a, np.int32) for a in fv]
        check(self._L.orbhip_set_put_from_frame(self._ctx.handle, key, src.handle, _p(g[0]), _p(g[1]), _p(g[2]),
                                                0 if fv is None else len(g[0])), self._ctx.handle, "orbhip_set_put_from_frame")

    def has_set(self, key, n):
        return bool(self._L.orbhip_set_has(self._ctx.handle, key, n))

    def set_info(self, key):
        """(n, ng, fingerprint) of a resident set, or None."""
        n, ng, fp = C.c_int(), C.c_int(), C.c_uint64()
        if not self._L.orbhip_set_info(self._ctx.handle, key, C.byref(n), C.byref(ng), C.byref(fp)):
            return None
        return n.value, ng.value, fp.value

    @staticmethod
    def fingerprint(kps, desc):
        kps = np.ascontiguousarray(kps, KP_DTYPE)
        desc = np.ascontiguousarray(desc, np.uint8).reshape(-1, 32)
        return int(capi.load().orbhip_set_fingerprint(_p(kps), _p(desc), len(kps)))

    def set_limit(self, max_sets):
        """At most `max_sets` resident sets (clamped to 4..96), least recently used out; returns the limit in force."""
        return int(self._L.orbhip_set_limit(self._ctx.handle, int(max_sets)))

    def drop_set(self, key=0):
        check(self._L.orbhip_set_drop(self._ctx.handle, key), self._ctx.handle, "orbhip_set_drop")

    def SearchByBoW_sets(self, key1, valid1, n1, key2, valid2, n2, kf_kf=False):
        """SearchByBoW between two resident sets; returns (nmatches, match12, match21) like SearchByBoW."""
        valid1 = np.ascontiguousarray(valid1, np.uint8)
        valid2 = None if valid2 is None else np.ascontiguousarray(valid2, np.uint8)
        # the entry point sizes everything from the resident sets: the caller's idea of them must agree (ORBHIP_E_ARG otherwise)
        i1, i2 = self.set_info(key1), self.set_info(key2)
        if i1 is None or i2 is None or i1[0] != n1 or i2[0] != n2 or len(valid1) != n1 or (valid2 is not None and len(valid2) != n2):
            raise OrbHipError("SearchByBoW_sets: unknown set, or feature counts that differ from the resident sets' (%s, %s vs %d, %d)"
                              % (i1, i2, n1, n2))
        m12 = np.empty(max(n1, 1), np.int32)
        m21 = np.empty(max(n2, 1), np.int32)
        nm = C.c_int()
        check(self._L.orbhip_search_by_bow_sets(self._ctx.handle, key1, _p(valid1), key2, _p(valid2), self.TH_LOW, 1 if kf_kf else 0,
                                                self.mfNNratio, 1 if self.mbCheckOrientation else 0, _p(m12), _p(m21), C.byref(nm)),
              self._ctx.handle, "orbhip_search_by_bow_sets")
        return nm.value, m12[:n1].copy(), m21[:n2].copy()


def ComputeStereoMatches(exL, kpsL, descL, exR, kpsR, descR, mb, mbf):
    """Frame::ComputeStereoMatches (src/Frame.cc:810-984) on the pyramids the two extractor contexts still
    hold from their last operator() call.  Returns (mvuRight, mvDepth, n_before_median_cut)."""
    kpsL = np.ascontiguousarray(kpsL, KP_DTYPE)
    kpsR = np.ascontiguousarray(kpsR, KP_DTYPE)
    descL = np.ascontiguousarray(descL, np.uint8).reshape(-1, 32)
    descR = np.ascontiguousarray(descR, np.uint8).reshape(-1, 32)
    n = len(kpsL)
    u = np.empty(max(n, 1), np.float32)
    z = np.empty(max(n, 1), np.float32)
    nm = C.c_int()
    check(exL._L.orbhip_stereo_match(exL.handle, exR.handle, _p(kpsL), _p(descL), n, _p(kpsR), _p(descR), len(kpsR),
                                     mb, mbf, _p(u), _p(z), C.byref(nm)), exL.handle, "orbhip_stereo_match")
    return u[:n].copy(), z[:n].copy(), nm.value
