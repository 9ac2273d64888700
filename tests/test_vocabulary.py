"""ORB vocabulary tree: the oracle against a definition-level Python model
on CPU; the HIP transform and the batched SearchByBoW against the oracle on the GPU."""
import numpy as np
import pytest


def _py_transform(voc, feat, levelsup):
    """TemplatedVocabulary.h:1443-1485 with python ints; voc = unpack_vocabulary(blob)."""
    nodes = voc["nodes"]
    n = len(nodes) + 1
    children = [[] for _ in range(n)]
    for i, p in enumerate(nodes["parent"]):
        children[int(p)].append(i + 1)
    word = {}
    for i, leaf in enumerate(nodes["leaf"]):
        if leaf:
            word[i + 1] = len(word)
    nid_level = voc["L"] - levelsup
    final_id, level, nid = 0, 0, 0
    while True:
        level += 1
        best = None
        for c in children[final_id]:
            d = int(np.unpackbits(nodes["desc"][c - 1] ^ feat).sum())
            if best is None or d < best:
                best, nxt = d, c
        final_id = nxt
        if level == nid_level:
            nid = final_id
        if nodes["leaf"][final_id - 1]:
            break
    return word[final_id], float(nodes["weight"][final_id - 1]), nid


def test_oracle_vocabulary_matches_python_model(oracle):
    from orbhip import distributed as D, synth
    blob = D.make_synthetic_vocabulary(5, k=4, L=3)
    voc = D.unpack_vocabulary(blob)
    V = oracle.Vocabulary(blob)
    assert (V.k, V.L, V.nnodes, V.nwords) == (4, 3, 1 + 4 + 16 + 64, 64)
    desc = synth.make_descriptor_db(6, 60)
    desc[7] = voc["nodes"]["desc"][30]                      # exact hit on an inner/leaf node descriptor
    for levelsup in (0, 1, 2, 3, 5):
        w, wt, nid = V.transform(desc, levelsup)
        for i in range(len(desc)):
            assert (int(w[i]), float(wt[i]), int(nid[i])) == _py_transform(voc, desc[i], levelsup)
    # BowVector: TF-IDF accumulation + L1 normalisation
    w, wt, nid = V.transform(desc, 1)
    bw, bv = V.bow(w, wt)
    acc = {}
    for i in range(len(desc)):
        acc[int(w[i])] = acc.get(int(w[i]), 0.0) + float(wt[i])
    keys = sorted(acc)
    norm = sum(abs(acc[k]) for k in keys)
    assert list(bw) == keys and np.allclose(bv, [acc[k] / norm for k in keys], rtol=0, atol=1e-15)
    assert abs(bv.sum() - 1.0) < 1e-12
    with pytest.raises(ValueError):
        oracle.Vocabulary(blob[:-3])


@pytest.mark.gpu
def test_hip_vocab_transform_matches_oracle(oracle):
    from orbhip import distributed as D, synth
    from orbhip.extractor import ORBextractor
    from orbhip.vocabulary import ORBVocabulary
    ex = ORBextractor(300, max_w=320, max_h=240)
    for (k, L, seed) in [(10, 3, 11), (4, 5, 12), (10, 4, 13)]:
        blob = D.make_synthetic_vocabulary(seed, k=k, L=L)
        ref = oracle.Vocabulary(blob)
        voc = ORBVocabulary(ex)
        voc.loadFromBinaryBlob(blob)
        assert (voc.k, voc.L, voc.nnodes, voc.nwords) == (ref.k, ref.L, ref.nnodes, ref.nwords)
        desc = synth.make_descriptor_db(seed + 100, 3000)
        for levelsup in (0, 2, L - 2, L, L + 3):
            w, wt, nid = voc.transform_raw(desc, levelsup)
            rw, rwt, rnid = ref.transform(desc, levelsup)
            assert np.array_equal(w, rw) and np.array_equal(wt, rwt) and np.array_equal(nid, rnid)
        (bw, bv), fv = voc.transform(desc[:500], 2)
        rw, rwt, rnid = ref.transform(desc[:500], 2)
        obw, obv = ref.bow(rw, rwt)
        assert np.array_equal(bw, obw) and np.array_equal(bv, obv)        # doubles, same summation order
        ofv = oracle.feature_vector(rnid, rwt)
        assert all(np.array_equal(a, b) for a, b in zip(fv, ofv))
    from orbhip.capi import OrbHipError
    with pytest.raises(OrbHipError):
        ORBVocabulary(ex).loadFromBinaryBlob(blob[:100])
    ex.close()


@pytest.mark.gpu
@pytest.mark.parametrize("th_mode,k,Lv,levelsup,NF,W,H", [
    (0, 10, 3, 1, 1000, 640, 480), (1, 10, 3, 1, 1000, 640, 480), (0, 3, 3, 1, 1000, 640, 480), (1, 4, 4, 2, 1000, 640, 480),
    (0, 6, 3, 0, 1000, 640, 480), (0, 3, 4, 4, 1000, 640, 480),
    (0, 10, 3, 1, 2000, 1241, 376),      # KITTI size: descriptor sets too large for LDS -> the global-descriptor variant
    (1, 10, 3, 1, 300, 320, 240)])       # few features: the sort covers 512 keys
def test_hip_batched_search_by_bow_matches_oracle(oracle, th_mode, k, Lv, levelsup, NF, W, H):
    """extract_batch_device -> vocab_transform_device -> search_by_bow_seq_device, all resident on the
    device, against extractor + vocabulary + SearchByBoW of the oracle, frame pair by frame pair."""
    import ctypes as C
    import hiprt
    from orbhip import distributed as D, synth
    from orbhip.capi import check
    from orbhip.extractor import ORBextractor
    from orbhip.vocabulary import ORBVocabulary
    B = 4
    frames = synth.make_frames(70, W, H, B)
    # node sets from one node holding everything (levelsup >= L) over a few large nodes to ~200 small ones: the
    # matcher treats large and small nodes differently (whole wave / 16-lane group), in cost order
    blob = D.make_synthetic_vocabulary(71, k=k, L=Lv)
    ex = ORBextractor(NF, max_w=W, max_h=H, max_batch=B)
    ORBVocabulary(ex).loadFromBinaryBlob(blob)
    cap = ex.cap
    d_img = hiprt.DevBuf.from_numpy(frames)
    d_kps, d_desc, d_cnt = hiprt.DevBuf(B * cap * 28), hiprt.DevBuf(B * cap * 32), hiprt.DevBuf(B * 4)
    d_word, d_wt, d_node = hiprt.DevBuf(B * cap * 4), hiprt.DevBuf(B * cap * 4), hiprt.DevBuf(B * cap * 4)
    d_m12, d_m21, d_nm = hiprt.DevBuf(B * cap * 4), hiprt.DevBuf(B * cap * 4), hiprt.DevBuf(B * 4)
    rng = np.random.default_rng(72)
    valid = (rng.random((B, cap)) < 0.85).astype(np.uint8)
    d_valid = hiprt.DevBuf.from_numpy(valid)
    L = ex._L
    ex.extract_batch_device(d_img.ptr, B, W, H, W, H * W, d_kps.ptr, d_desc.ptr, cap, d_cnt.ptr)
    check(L.orbhip_vocab_transform_device(ex.handle, d_desc.ptr, B * cap, levelsup, d_word.ptr, d_wt.ptr, d_node.ptr), ex.handle)
    # nnratio 0.19: TH_LOW = 50 is no longer below nnratio * 255, the bound under which k_bow_lane's distance bytes (clamped at 255)
    # are exact -- the launcher then takes k_bow_seq, the wave-per-node kernel of rounds 1-4
    for check_ori, nnratio in ((1, 0.7), (0, 0.7), (1, 0.19)):
        check(L.orbhip_search_by_bow_seq_device(ex.handle, d_desc.ptr, d_kps.ptr, d_cnt.ptr, d_node.ptr, d_wt.ptr,
                                                d_valid.ptr, cap, B, 1, th_mode, C.c_float(nnratio), check_ori, d_m12.ptr,
                                                d_m21.ptr, d_nm.ptr), ex.handle, "search_by_bow_seq")
        ex.sync()
        cnt = d_cnt.to_numpy(np.int32, (B,))
        m12 = d_m12.to_numpy(np.int32, (B, cap))
        m21 = d_m21.to_numpy(np.int32, (B, cap))
        nm = d_nm.to_numpy(np.int32, (B,))
        refx = oracle.Extractor(NF)
        refv = oracle.Vocabulary(blob)
        feats = []
        for b in range(B):
            k, d = refx(frames[b])
            w, wt, nid = refv.transform(d, levelsup)
            feats.append((k, d, oracle.feature_vector(nid, wt)))
            assert cnt[b] == len(k)
        assert nm[0] == 0 and (m12[0] == -1).all() and (m21[0] == -1).all()
        for b in range(1, B):
            (k1, d1, fv1), (k2, d2, fv2) = feats[b - 1], feats[b]
            n1, n2 = len(k1), len(k2)
            wn, w12, w21 = oracle.search_by_bow(d1, valid[b - 1, :n1], k1["angle"], fv1, d2,
                                                valid[b, :n2] if th_mode else None, k2["angle"], fv2, th=50,
                                                th_mode=th_mode, nnratio=nnratio, check_ori=bool(check_ori))
            assert nm[b] == wn and (nnratio < 0.5 or wn > (100 if NF >= 1000 else 20))
            assert np.array_equal(m12[b, :n1], w12) and (m12[b, n1:] == -1).all()
            assert np.array_equal(m21[b, :n2], w21) and (m21[b, n2:] == -1).all()
    ex.close()
    for x in (d_img, d_kps, d_desc, d_cnt, d_word, d_wt, d_node, d_m12, d_m21, d_nm, d_valid):
        x.free()


@pytest.mark.gpu
def test_hip_batched_search_by_bow_rejects_more_than_4096_slots():
    """The per-pair tables of k_bow_seq live in LDS: a clear error, not a failed launch."""
    import ctypes as C
    import hiprt
    from orbhip.capi import OrbHipError, check
    from orbhip.extractor import ORBextractor
    ex = ORBextractor(500, max_w=320, max_h=240)
    d = hiprt.DevBuf(4096)
    with pytest.raises(OrbHipError, match="4096 feature slots"):
        check(ex._L.orbhip_search_by_bow_seq_device(ex.handle, d.ptr, d.ptr, d.ptr, d.ptr, d.ptr, None, 5000, 2, 1, 0, C.c_float(0.7), 1,
                                                    d.ptr, d.ptr, d.ptr), ex.handle, "search_by_bow_seq")
    d.free()
    ex.close()


# ---- text vocabularies: ORBVocabulary::loadFromTextFile (TemplatedVocabulary.h:1564-1647; src/System.cc:335-336) ----
def _golden(name):
    import os
    return os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", name)


def test_text_vocabulary_conversion_oracle_c_abi_and_fixture(oracle):
    """The committed text fixture through the oracle's restatement and through the C ABI's host-only converter: the same
    binary blob (= what tools/bin_vocabulary.cc would write) and the same double weights; edge cases of the format."""
    from orbhip import distributed as D
    from orbhip.vocabulary import text_to_binary
    text = open(_golden("vocab_k4_L2.txt"), "rb").read()
    g = np.load(_golden("vocab_k4_L2_text.npz"))
    ob, ow = oracle.vocabulary_text_to_blob(text)
    cb, cw = text_to_binary(text)
    assert ob == cb == g["blob"].tobytes() and np.array_equal(ow, cw) and np.array_equal(ow, g["node_weight64"])
    voc = D.unpack_vocabulary(cb)
    assert (voc["k"], voc["L"], len(voc["nodes"])) == (4, 2, 20) and int(voc["nodes"]["leaf"].sum()) == 16
    # the text keeps 6 significant digits: the double is NOT the float of the tree the text was written from
    src = D.unpack_vocabulary(D.make_synthetic_vocabulary(211, k=4, L=2))["nodes"]
    assert np.array_equal(voc["nodes"]["desc"], src["desc"]) and np.array_equal(voc["nodes"]["parent"], src["parent"])
    assert np.allclose(cw, src["weight"], rtol=1e-5) and not np.array_equal(cw.astype(np.float32), src["weight"])
    assert np.array_equal(voc["nodes"]["weight"], cw.astype(np.float32))         # saveToBinaryFile narrows to float
    # oracle transform + BowVector with the double weights reproduce the golden values
    V = oracle.Vocabulary(ob)
    w, wt, nid = V.transform(g["desc"], 1)
    assert np.array_equal(w, g["word"]) and np.array_equal(nid, g["node"])
    bw, bv = oracle.bow_vector64(w, ow[voc["nodes"]["leaf"] != 0][w], V.scoring, V.weighting)
    assert np.array_equal(bw, g["bow_word"]) and np.array_equal(bv, g["bow_value"])
    # format edge cases: no final newline, blank lines, CRLF, leaf flag > 1 -- all the same tree
    for variant in (text.rstrip(b"\n"), text + b"\n\n", text.replace(b"\n", b"\r\n"), text.replace(b"\n0 1 ", b"\n0 7 ")):
        vb, vw = text_to_binary(variant)
        assert vb == cb and np.array_equal(vw, cw)
        assert oracle.vocabulary_text_to_blob(variant.replace(b"\r", b""))[0] == cb
    # malformed: header out of the ranges of :1585-1589, a parent that does not exist yet, a truncated node line
    lines = text.split(b"\n")
    for bad in (b"25 2  0 0\n" + b"\n".join(lines[1:]), b"4 0  0 0\n" + b"\n".join(lines[1:]), b"4 2  6 0\n" + b"\n".join(lines[1:]),
                lines[0] + b"\n" + lines[1].replace(b"0 0 ", b"9 0 ", 1) + b"\n", lines[0] + b"\n" + b" ".join(lines[1].split()[:20]) + b"\n",
                b""):
        assert text_to_binary(bad) == (None, None)
    with pytest.raises(ValueError):
        oracle.vocabulary_text_to_blob(b"25 2  0 0\n")


@pytest.mark.gpu
def test_hip_text_vocabulary_matches_oracle(oracle, tmp_path):
    from orbhip import distributed as D, synth
    from orbhip.extractor import ORBextractor
    from orbhip.vocabulary import ORBVocabulary
    ex = ORBextractor(300, max_w=320, max_h=240)
    g = np.load(_golden("vocab_k4_L2_text.npz"))
    voc = ORBVocabulary(ex)
    assert voc.loadFromTextFile(_golden("vocab_k4_L2.txt")) and (voc.k, voc.L, voc.nnodes, voc.nwords) == (4, 2, 21, 16)
    (bw, bv), fv = voc.transform(g["desc"], 1)
    assert np.array_equal(bw, g["bow_word"]) and np.array_equal(bv, g["bow_value"])       # doubles from the text
    w, wt, nid = voc.transform_raw(g["desc"], 1)
    assert np.array_equal(w, g["word"]) and np.array_equal(nid, g["node"])
    # a larger tree written as text: k = 10, L = 3, weights with few digits
    blob = D.make_synthetic_vocabulary(61, k=10, L=3)
    text = D.vocabulary_to_text(blob)
    (tmp_path / "voc.txt").write_bytes(text)
    assert voc.loadFromTextFile(str(tmp_path / "voc.txt")) and voc.nnodes == 1111
    ob, ow = oracle.vocabulary_text_to_blob(text)
    V = oracle.Vocabulary(ob)
    desc = synth.make_descriptor_db(62, 1500)
    (bw, bv), fv = voc.transform(desc, 2)
    rw, rwt, rnid = V.transform(desc, 2)
    leaf = np.frombuffer(ob, D.VOC_NODE_DTYPE, offset=24)["leaf"] != 0
    obw, obv = oracle.bow_vector64(rw, ow[leaf][rw], V.scoring, V.weighting)
    assert np.array_equal(bw, obw) and np.array_equal(bv, obv)
    assert all(np.array_equal(a, b) for a, b in zip(fv, oracle.feature_vector(rnid, ow[leaf][rw])))
    assert not voc.loadFromText(b"not a vocabulary")
    ex.close()
