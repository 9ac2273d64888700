"""Checker of the streams mode (BASELINE config 4): sampled frames of a stream against the CPU oracle, bit for bit.
Also the rank program of the two-process GPU test: `python tests/streams_check.py <orbhip.streams arguments>`."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "vi-orb-slam-icra2018_amd"), os.path.join(ROOT, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)

from orbhip.streams import LEVELSUP, NFEAT, NNRATIO  # noqa: E402


def verify_samples(runner, got, blob, nfeat=NFEAT):
    """Sampled frames of a stream against the CPU oracle, bit for bit ."""
    import orb_oracle_py as oracle
    ref, V = oracle.Extractor(nfeat, 1.2, 8, 20, 7), oracle.Vocabulary(blob)
    cache = {}

    def ref_frame(t):
        if t not in cache:
            k, d = ref(runner.frame(t))
            _, wt, nid = V.transform(d, LEVELSUP)
            cache[t] = (k, d, oracle.feature_vector(nid, wt))
        return cache[t]
    for t, rec in sorted(got.items()):
        k, d, fv = ref_frame(t)
        if rec["n"] != len(k) or rec["kps"] != k.tobytes() or not np.array_equal(rec["desc"], d):
            raise AssertionError("stream frame %d: keypoints / descriptors differ from the oracle" % t)
        if t >= 1:
            pk, pd, pfv = ref_frame(t - 1)
            nm, m12, m21 = oracle.search_by_bow(pd, np.ones(len(pd), np.uint8), pk["angle"], pfv, d, None, k["angle"], fv,
                                                th=50, th_mode=0, nnratio=NNRATIO, check_ori=True)
            if rec["n_prev"] != len(pk) or rec["nm"] != nm or not np.array_equal(rec["m12"], m12) or \
                    not np.array_equal(rec["m21"], m21):
                raise AssertionError("stream frame %d: SearchByBoW against frame %d differs from the oracle" % (t, t - 1))
    return len(got)



if __name__ == "__main__":
    from orbhip import streams
    streams.main(sys.argv[1:], verifier=verify_samples)
