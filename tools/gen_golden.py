#!/usr/bin/env python3
"""Generate tests/golden/*.npz: small seeded inputs with the ORACLE's outputs.

The reference has no tests, fixtures or golden vectors of its own and cannot
be built here (OpenCV 2.4 absent), so these are regression vectors of the CPU restatement, not
outputs of the reference: they freeze the oracle's behaviour so that (1) an accidental change of
the oracle is caught on CPU and (2) the HIP path can be checked against committed data as well as
against the live oracle.  Run:  python3 tools/gen_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, os.path.join(ROOT, "vi-orb-slam-icra2018_amd"))
import orb_oracle_py as oracle  # noqa: E402
from orbhip import synth  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")
EXTRACT_CASES = [
    # name, seed, w, h, nfeatures, nlevels, iniTh, minTh
    ("extract_320x240_f300_l4", 101, 320, 240, 300, 4, 20, 7),
    ("extract_376x241_f500_l5", 102, 376, 241, 500, 5, 20, 7),
    ("extract_300x200_f150_l3_th30_10", 103, 300, 200, 150, 3, 30, 10),
]


def main():
    os.makedirs(OUT, exist_ok=True)
    for name, seed, w, h, nf, nl, ini, mn in EXTRACT_CASES:
        img = synth.make_frames(seed, w, h, 1)[0]
        ex = oracle.Extractor(nf, 1.2, nl, ini, mn)
        kps, desc = ex(img)
        ncand = np.array([len(ex.level_cands(l)) for l in range(nl)], np.int32)
        nkp = np.array([len(ex.level_keypoints(l)) for l in range(nl)], np.int32)
        lvl_sum = np.array([int(ex.pyramid(l).astype(np.uint64).sum()) for l in range(nl)], np.uint64)
        blur_sum = np.array([int(ex.blurred(l).astype(np.uint64).sum()) for l in range(nl)], np.uint64)
        np.savez_compressed(os.path.join(OUT, name + ".npz"), image=img, params=np.array([nf, nl, ini, mn], np.int32),
                            keypoints=kps, descriptors=desc, level_candidates=ncand, level_keypoints=nkp,
                            level_pixel_sum=lvl_sum, blurred_pixel_sum=blur_sum)
        print(name, len(kps), ncand.tolist())
    # matching: descriptor sets with structure (duplicates, near matches), knn2 + SearchByBoW results
    rng = np.random.default_rng(104)
    db = synth.make_descriptor_db(105, 600)
    db[400] = db[20]
    q, _ = synth.make_queries(106, db, 250)
    q[3] = db[400]
    bi, bd, sd = oracle.knn2(q, db)
    n1, n2 = len(q), len(db)
    node2 = rng.integers(0, 12, n2).astype(np.int32)
    node1 = np.where(rng.random(n1) < 0.8, node2[bi], rng.integers(0, 14, n1)).astype(np.int32)
    a2 = rng.uniform(0, 360, n2).astype(np.float32)
    a1 = ((a2[bi] + rng.choice([0, 0, 0, 100], n1) + rng.uniform(-4, 4, n1)) % 360).astype(np.float32)
    v1 = (rng.random(n1) < 0.85).astype(np.uint8)
    v2 = (rng.random(n2) < 0.9).astype(np.uint8)

    def fv(node):
        ids = sorted(set(int(v) for v in node))
        lists = [np.nonzero(node == k)[0] for k in ids]
        off = np.concatenate([[0], np.cumsum([len(x) for x in lists])]).astype(np.int32)
        return np.array(ids, np.int32), off, np.concatenate(lists).astype(np.int32)
    fv1, fv2 = fv(node1), fv(node2)
    r0 = oracle.search_by_bow(q, v1, a1, fv1, db, None, a2, fv2, th=50, th_mode=0, nnratio=0.7, check_ori=True)
    r1 = oracle.search_by_bow(q, v1, a1, fv1, db, v2, a2, fv2, th=50, th_mode=1, nnratio=0.75, check_ori=True)
    np.savez_compressed(os.path.join(OUT, "matching_q250_db600.npz"), q=q, db=db, best_idx=bi, best_d=bd, second_d=sd,
                        node1=node1, node2=node2, angle1=a1, angle2=a2, valid1=v1, valid2=v2,
                        bow_kf_f_n=np.int32(r0[0]), bow_kf_f_m12=r0[1], bow_kf_f_m21=r0[2],
                        bow_kf_kf_n=np.int32(r1[0]), bow_kf_kf_m12=r1[1], bow_kf_kf_m21=r1[2])
    print("matching", r0[0], r1[0])


if __name__ == "__main__":
    main()
