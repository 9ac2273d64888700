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

    def _info(self):
        v = [C.c_int() for _ in range(6)]
        check(self._L.orbhip_vocab_info(self._ctx.handle, *[C.byref(x) for x in v]), self._ctx.handle, "vocab_info")
        self.k, self.L, self.scoring, self.weighting, self.nnodes, self.nwords = [x.value for x in v]

    def loadFromBinaryBlob(self, blob):
        blob = bytes(blob)
        check(self._L.orbhip_vocab_load(self._ctx.handle, blob, len(blob)), self._ctx.handle, "orbhip_vocab_load")
        self._info()
        return True

    def loadFromBinaryFile(self, filename):           # ref: TemplatedVocabulary.h:1680
        with open(filename, "rb") as f:
            return self.loadFromBinaryBlob(f.read())

    def loadFromDeviceBlob(self, d_ptr, nbytes):
        check(self._L.orbhip_vocab_load_device(self._ctx.handle, d_ptr, nbytes), self._ctx.handle,
              "orbhip_vocab_load_device")
        self._info()
        return True

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
