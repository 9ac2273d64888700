"""-m gpu parity at the sizes the bench and BASELINE.json's configs are quoted on (VERDICT r01 "next round" item 1):
  * the bench step's own shape: 640x480 / 1000 features, k=10, L=6, levelsup 4 vocabulary, SearchByBoW(0.7, checkOri) of
    every frame against its predecessor, no validity mask (ref: src/Frame.cc:739-746, src/ORBmatcher.cc:159-288,
    src/Tracking.cc:1881-1885);
  * config 5: 4000 queries against 1 000 000 descriptors (exact on a query subset, size-independent properties on all),
    the <= 32-query few-query path at the same database size, and extraction at 4000 features;
  * the multi-GPU C path on the one GPU a box has: RCCL communicator with one rank, vocabulary broadcast into a device
    buffer and load from it, all-gather + device merge of database-sharded brute force."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_bench_step_shape_l6_vocabulary_search_by_bow(oracle):
    import hiprt
    from orbhip import distributed as D, synth
    from orbhip.capi import check
    from orbhip.extractor import ORBextractor
    from orbhip.vocabulary import ORBVocabulary
    W, H, NF, B, LEVELSUP = 640, 480, 1000, 6, 4
    frames = synth.make_frames(1000, W, H, B)                   # rank 0's stream in bench.py
    blob = D.make_synthetic_vocabulary(4242, 10, 6)             # bench.py's vocabulary: 1 111 110 nodes, 45.6 MB
    ex = ORBextractor(NF, 1.2, 8, 20, 7, max_w=W, max_h=H, max_batch=B)
    voc = ORBVocabulary(ex)
    voc.loadFromBinaryBlob(blob)
    assert (voc.k, voc.L, voc.nnodes) == (10, 6, 1111111)
    cap = ex.cap
    d_img = hiprt.DevBuf.from_numpy(frames)
    d_kps, d_desc, d_cnt = hiprt.DevBuf(B * cap * 28), hiprt.DevBuf(B * cap * 32), hiprt.DevBuf(B * 4)
    d_word, d_wt, d_node = hiprt.DevBuf(B * cap * 4), hiprt.DevBuf(B * cap * 4), hiprt.DevBuf(B * cap * 4)
    d_m12, d_m21, d_nm = hiprt.DevBuf(B * cap * 4), hiprt.DevBuf(B * cap * 4), hiprt.DevBuf(B * 4)
    L = ex._L
    ex.extract_batch_device(d_img.ptr, B, W, H, W, H * W, d_kps.ptr, d_desc.ptr, cap, d_cnt.ptr)
    check(L.orbhip_vocab_transform_device(ex.handle, d_desc.ptr, B * cap, LEVELSUP, d_word.ptr, d_wt.ptr, d_node.ptr), ex.handle)
    check(L.orbhip_search_by_bow_seq_device(ex.handle, d_desc.ptr, d_kps.ptr, d_cnt.ptr, d_node.ptr, d_wt.ptr, None, cap, B, 1, 0,
                                            C.c_float(0.7), 1, d_m12.ptr, d_m21.ptr, d_nm.ptr), ex.handle, "search_by_bow_seq")
    ex.sync()
    cnt = d_cnt.to_numpy(np.int32, (B,))
    kps = d_kps.to_numpy(np.uint8, (B, cap, 28))
    desc = d_desc.to_numpy(np.uint8, (B, cap, 32))
    word = d_word.to_numpy(np.int32, (B, cap))
    wt = d_wt.to_numpy(np.float32, (B, cap))
    node = d_node.to_numpy(np.int32, (B, cap))
    m12, m21 = d_m12.to_numpy(np.int32, (B, cap)), d_m21.to_numpy(np.int32, (B, cap))
    nm = d_nm.to_numpy(np.int32, (B,))
    refx, refv = oracle.Extractor(NF, 1.2, 8, 20, 7), oracle.Vocabulary(blob)
    prev = None
    total = 0
    for b in range(B):
        k, d = refx(frames[b])
        n = len(k)
        assert cnt[b] == n and kps[b, :n].tobytes() == k.tobytes() and np.array_equal(desc[b, :n], d)
        w, wgt, nid = refv.transform(d, LEVELSUP)
        assert np.array_equal(word[b, :n], w) and np.array_equal(wt[b, :n], wgt) and np.array_equal(node[b, :n], nid)
        fv = oracle.feature_vector(nid, wgt)
        assert 50 <= len(fv[0]) <= 100                           # level-2 nodes of a k=10 tree (src/Frame.cc:744)
        if prev is not None:
            pk, pd, pfv = prev
            wn, w12, w21 = oracle.search_by_bow(pd, np.ones(len(pd), np.uint8), pk["angle"], pfv, d, None, k["angle"], fv,
                                                th=50, th_mode=0, nnratio=0.7, check_ori=True)
            assert nm[b] == wn and wn > 100
            assert np.array_equal(m12[b, :len(pk)], w12) and np.array_equal(m21[b, :n], w21)
            total += wn
        prev = (k, d, fv)
    assert total > 500
    ex.close()
    for x in (d_img, d_kps, d_desc, d_cnt, d_word, d_wt, d_node, d_m12, d_m21, d_nm):
        x.free()


def test_config5_4000_queries_against_a_million_descriptors(oracle):
    from orbhip import synth
    from orbhip.extractor import ORBextractor, ORBmatcher
    NDB, NQ = 1_000_000, 4000
    ex = ORBextractor(4000, 1.2, 8, 20, 7, max_w=640, max_h=480)
    # TUM geometry at 4000 features (config 5's extraction leg)
    frame = synth.make_frames(5, 640, 480, 1)[0]
    k, d = ex(frame)
    rk, rd = oracle.Extractor(4000, 1.2, 8, 20, 7)(frame)
    assert len(k) == len(rk) > 3000 and k.tobytes() == rk.tobytes() and np.array_equal(d, rd)
    m = ORBmatcher(0.7, True, ctx=ex)
    db = synth.make_descriptor_db(51, NDB)
    db[900_001] = db[17]                                         # duplicate rows far apart: the lowest index wins
    q, rows = synth.make_queries(52, db, NQ, max_flips=40)
    q[3] = db[900_001]
    rows[3] = 17
    bi, bd, sd = m.knn2(q, db)
    # exact against the oracle on a query subset (the oracle needs ~10 ns per pair)
    sub = np.r_[0:48, NQ - 16:NQ]
    wi, wd, ws = oracle.knn2(q[sub], db)
    assert np.array_equal(bi[sub], wi) and np.array_equal(bd[sub], wd) and np.array_equal(sd[sub], ws)
    assert bi[3] == 17 and bd[3] == 0 and sd[3] == 0
    # size-independent properties on all 4000 queries: the reported distance is the distance to the reported row, it is
    # at most the flip count of the row the query was made from, second >= best, and no row of a random sample is closer
    assert np.array_equal(np.unpackbits(q ^ db[bi], axis=1).sum(1), bd)
    assert (bd <= np.unpackbits(q ^ db[rows], axis=1).sum(1)).all() and (sd >= bd).all()
    rng = np.random.default_rng(53)
    samp = db[rng.integers(0, NDB, 2000)]
    dmin = np.array([np.unpackbits(samp ^ qq, axis=1).sum(1).min() for qq in q[::40]])
    assert (bd[::40] <= dmin).all()
    # the few-query path (<= 32 queries: the database is split over the whole chip) at the same database size
    for nq in (1, 8, 32):
        gi, gd, gs = m.knn2(q[:nq], db)
        assert np.array_equal(gi, wi[:nq]) and np.array_equal(gd, wd[:nq]) and np.array_equal(gs, ws[:nq])
    ex.close()


def test_rccl_one_rank_broadcast_vocabulary_load_and_sharded_merge(oracle):
    import hiprt
    from orbhip import distributed as D, synth
    from orbhip.capi import check
    from orbhip.extractor import ORBextractor
    from orbhip.vocabulary import ORBVocabulary
    ex = ORBextractor(500, max_w=320, max_h=240)
    L = ex._L
    uid = (C.c_uint8 * 128)()
    check(L.orbhip_comm_unique_id(uid), None, "orbhip_comm_unique_id")
    assert any(uid)
    check(L.orbhip_comm_init(ex.handle, 0, 1, uid), ex.handle, "orbhip_comm_init")
    # vocabulary: broadcast of the reference's binary blob in a device buffer (root 0 of 1), then the load from that buffer
    blob = D.make_synthetic_vocabulary(31, k=10, L=4)
    arr = np.frombuffer(blob, np.uint8)
    d_blob = hiprt.DevBuf.from_numpy(arr)
    check(L.orbhip_bcast_blob_device(ex.handle, d_blob.ptr, len(blob), 0), ex.handle, "orbhip_bcast_blob_device")
    ex.sync()
    assert d_blob.to_numpy(np.uint8, (len(blob),)).tobytes() == blob
    voc = ORBVocabulary(ex)
    voc.loadFromDeviceBlob(d_blob.ptr, len(blob))
    ref = oracle.Vocabulary(blob)
    desc = synth.make_descriptor_db(32, 2000)
    for a, b in zip(voc.transform_raw(desc, 2), ref.transform(desc, 2)):
        assert np.array_equal(a, b)
    assert L.orbhip_bcast_blob_device(ex.handle, d_blob.ptr, len(blob), 3) != 0       # bad root is an error, not a hang
    # database-sharded brute force: three shards merged by the device kernel = one pass over the whole database
    db = synth.make_descriptor_db(33, 3000)
    db[2500] = db[10]
    db[1500] = db[10]
    q, _ = synth.make_queries(34, db, 300)
    q[0] = db[10]
    bounds = [(0, 1000), (1000, 2200), (2200, 3000)]
    parts = []
    for lo, hi in bounds:
        li, ld, ls = oracle.knn2(q, db[lo:hi])
        parts.append(np.concatenate([li, ld, ls, [lo]]).astype(np.int32))
    d_parts = hiprt.DevBuf.from_numpy(np.stack(parts))
    d_o = [hiprt.DevBuf(len(q) * 4) for _ in range(3)]
    check(L.orbhip_knn2_merge_device(ex.handle, d_parts.ptr, 3, len(q), d_o[0].ptr, d_o[1].ptr, d_o[2].ptr), ex.handle, "merge")
    ex.sync()
    want = oracle.knn2(q, db)
    for o, w in zip(d_o, want):
        assert np.array_equal(o.to_numpy(np.int32, (len(q),)), w)
    # the same through the RCCL all-gather of a one-rank communicator, fed by the HIP brute force on "this rank's" rows
    d_q, d_db = hiprt.DevBuf.from_numpy(q), hiprt.DevBuf.from_numpy(db[1000:])
    d_l = [hiprt.DevBuf(len(q) * 4) for _ in range(3)]
    check(L.orbhip_hamming_knn2_device(ex.handle, d_q.ptr, len(q), d_db.ptr, 2000, d_l[0].ptr, d_l[1].ptr, d_l[2].ptr), ex.handle)
    check(L.orbhip_knn2_allgather_merge_device(ex.handle, d_l[0].ptr, d_l[1].ptr, d_l[2].ptr, len(q), 1000, d_o[0].ptr, d_o[1].ptr,
                                               d_o[2].ptr), ex.handle, "allgather_merge")
    ex.sync()
    wi, wd, ws = oracle.knn2(q, db[1000:])
    assert np.array_equal(d_o[0].to_numpy(np.int32, (len(q),)), np.where(wi >= 0, wi + 1000, -1))
    assert np.array_equal(d_o[1].to_numpy(np.int32, (len(q),)), wd) and np.array_equal(d_o[2].to_numpy(np.int32, (len(q),)), ws)
    check(L.orbhip_comm_destroy(ex.handle), ex.handle, "orbhip_comm_destroy")
    check(L.orbhip_bcast_blob_device(ex.handle, d_blob.ptr, len(blob), 0), ex.handle)   # no communicator, one rank: a no-op
    for x in [d_blob, d_parts, d_q, d_db] + d_o + d_l:
        x.free()
    ex.close()
