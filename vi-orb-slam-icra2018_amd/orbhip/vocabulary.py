"""Python mirror of ORBVocabulary (include/ORBVocabulary.h = DBoW2::TemplatedVocabulary<FORB>) for the
part that feeds SearchByBoW: loadFromBinaryFile and transform."""
import ctypes as C

import numpy as np

from . import capi
from .capi import _p, check


class ORBVocabulary:
    def __init__(self, ctx):
        """ctx: an ORBextractor (its device context holds the vocabulary tables)."""
        self._ctx = ctx
        self._L = capi.load()
        self.k = self.L = self.scoring = self.weighting = self.nnodes = self.nwords = 0
        self._word_weight64 = None      # text-loaded vocabularies keep double weights (Node::weight), by word id

    def _info(self):
        v = [C.c_int() for _ in range(6)]
        check(self._L.orbhip_vocab_info(self._ctx.handle, *[C.byref(x) for x in v]), self._ctx.handle, "vocab_info")
        self.k, self.L, self.scoring, self.weighting, self.nnodes, self.nwords = [x.value for x in v]

    def loadFromBinaryBlob(self, blob):
        blob = bytes(blob)
        self._word_weight64 = None
        check(self._L.orbhip_vocab_load(self._ctx.handle, blob, len(blob)), self._ctx.handle, "orbhip_vocab_load")
        self._info()
        return True

    def loadFromBinaryFile(self, filename):           # ref: TemplatedVocabulary.h:1680
        with open(filename, "rb") as f:
            return self.loadFromBinaryBlob(f.read())

    def loadFromTextFile(self, filename):             # ref: TemplatedVocabulary.h:1564-1647 (src/System.cc:335-336)
        with open(filename, "rb") as f:
            return self.loadFromText(f.read())

    def loadFromText(self, text):
        blob, w64 = text_to_binary(text)
        if blob is None:
            return False
        self.loadFromBinaryBlob(blob)
        nodes = np.frombuffer(blob, np.dtype([("parent", "<i4"), ("desc", "u1", 32), ("weight", "<f4"), ("leaf", "u1")]),
                              offset=24)
        self._word_weight64 = w64[nodes["leaf"] != 0]   # words are numbered in leaf order (:1633-1640)
        return True

    def loadFromDeviceBlob(self, d_ptr, nbytes):
        self._word_weight64 = None
        check(self._L.orbhip_vocab_load_device(self._ctx.handle, d_ptr, nbytes), self._ctx.handle,
              "orbhip_vocab_load_device")
        self._info()
        return True

    def shareWith(self, other_ctx):
        """orbhip_vocab_share: another context of the same device (an ORBextractor) runs on these tables too -- what the C++
        ORBextractor does before orbhip_frame_build.  The block is reference-counted; generation() tells a borrower to share
        again after this object has loaded another vocabulary."""
        check(self._L.orbhip_vocab_share(other_ctx.handle, self._ctx.handle), other_ctx.handle, "orbhip_vocab_share")

    def generation(self, ctx=None):
        return int(self._L.orbhip_vocab_generation((ctx or self._ctx).handle))

    def transform_raw(self, desc, levelsup=4):
        """(word_id, weight, node_id) per descriptor -- TemplatedVocabulary.h:1443-1485."""
        desc = np.ascontiguousarray(desc, np.uint8).reshape(-1, 32)
        n = len(desc)
        w = np.empty(max(n, 1), np.int32)
        wt = np.empty(max(n, 1), np.float32)
        nid = np.empty(max(n, 1), np.int32)
        check(self._L.orbhip_vocab_transform(self._ctx.handle, _p(desc), n, levelsup, _p(w), _p(wt), _p(nid)),
              self._ctx.handle, "orbhip_vocab_transform")
        return w[:n], wt[:n], nid[:n]

    def transform(self, desc, levelsup=4):
        """(BowVector as (word ids, values), FeatureVector as CSR) -- TemplatedVocabulary.h:1167-1258 with
        features visited in ascending index order (canonical)."""
        w, wt, nid = self.transform_raw(desc, levelsup)
        if self._word_weight64 is not None:                            # text-loaded: the doubles of the text file
            wt = self._word_weight64[w]
        keep = np.nonzero(wt > 0)[0]                                   # "not stopped"
        accumulate = self.weighting in (0, 1)                          # TF_IDF, TF
        words = np.unique(w[keep])
        vals = np.zeros(len(words), np.float64)
        pos = np.searchsorted(words, w[keep])
        if accumulate:
            for p, i in zip(pos, keep):                                # ascending feature order (double sums)
                vals[p] += float(wt[i])
        else:
            seen = np.zeros(len(words), bool)
            for p, i in zip(pos, keep):
                if not seen[p]:
                    vals[p] = float(wt[i])
                    seen[p] = True
        must = self.scoring != 5
        if accumulate and len(vals) and not must:
            vals /= float(len(vals))
        if must and len(vals):
            if self.scoring == 1:
                s = 0.0
                for v in vals:
                    s += v * v
                norm = s ** 0.5
            else:
                norm = 0.0
                for v in vals:
                    norm += abs(v)
            if norm > 0:
                vals = vals / norm
        ids = sorted(set(int(v) for v in nid[keep]))
        lists = [keep[nid[keep] == k] for k in ids]
        off = np.concatenate([[0], np.cumsum([len(x) for x in lists])]).astype(np.int32)
        idx = (np.concatenate(lists) if lists else np.zeros(0)).astype(np.int32)
        return (words.astype(np.int32), vals), (np.array(ids, np.int32), off, idx)


def text_to_binary(text):
    """orbhip_vocab_text_to_binary: the text vocabulary -> (binary blob, double weight per node id 1..n), or
    (None, None) for a malformed text.  Host only (no device, no context)."""
    L = capi.load()
    text = bytes(text)
    need = C.c_size_t()
    if L.orbhip_vocab_text_to_binary(text, len(text), None, 0, C.byref(need), None, 0) != 0:
        return None, None
    n = (need.value - 24) // 41
    blob = (C.c_uint8 * need.value)()
    w64 = np.zeros(max(n, 1), np.float64)
    rc = L.orbhip_vocab_text_to_binary(text, len(text), blob, need.value, C.byref(need), _p(w64), n)
    if rc != 0:
        return None, None
    return bytes(blob), w64[:n]
