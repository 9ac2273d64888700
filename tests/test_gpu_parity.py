"""GPU parity tests: the HIP path, called through the C ABI, against the CPU oracle on the same
seeded inputs.  Bar: bit-exact (integer/byte/index work; the float fields angle/x/y/size are
compared as bit patterns too).  Run with -m gpu on the MI355X box."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def hip():
    from orbhip import capi
    L = capi.load()
    assert L.orbhip_device_count() > 0, "no HIP device: the gpu tests need the real MI355X"
    return capi


def _same_kps(a, b):
    return len(a) == len(b) and a.tobytes() == b.tobytes()


CONFIGS = [
    # (w, h, nfeatures, seed)            geometry of BASELINE.json configs
    (640, 480, 1000, 0),                 # TUM / headline metric
    (752, 480, 1000, 1),                 # EuRoC
    (1241, 376, 2000, 2),                # KITTI
    (640, 480, 4000, 3),                 # TUM at 4000 features
    (333, 251, 300, 4),                  # odd size, stride != width
    (200, 180, 100, 5),                  # small: few cells per level (4 levels)
]


@pytest.mark.parametrize("w,h,nf,seed", CONFIGS)
def test_extract_stage_by_stage(hip, oracle, w, h, nf, seed):
    from orbhip import synth
    from orbhip.extractor import ORBextractor
    nlev = 8 if min(w, h) >= 376 else 4
    img = synth.make_frames(seed, w, h, 1)[0]
    ex = ORBextractor(nf, 1.2, nlev, 20, 7, max_w=w, max_h=h, max_batch=1)
    ref = oracle.Extractor(nf, 1.2, nlev, 20, 7)
    k, d = ex(img)
    rk, rd = ref(img)
    assert list(ex.mnFeaturesPerLevel) == list(ref.params.mnFeaturesPerLevel)[:nlev]
    assert np.array_equal(ex.mvScaleFactor, np.array(list(ref.params.mvScaleFactor)[:nlev], np.float32))
    for l in range(nlev):
        assert np.array_equal(ex.image_pyramid(l), ref.pyramid(l)), "pyramid level %d" % l
    for l in range(nlev):
        gc, rc = ex.level_candidates(l), ref.level_cands(l)
        assert len(gc) == len(rc) and gc.tobytes() == rc.tobytes(), "FAST candidates level %d" % l
    for l in range(nlev):
        assert _same_kps(ex.level_keypoints(l), ref.level_keypoints(l)), "quadtree/angle level %d" % l
    for l in range(nlev):
        if len(ref.level_keypoints(l)):
            assert np.array_equal(ex.blurred(l), ref.blurred(l)), "blur level %d" % l
    assert _same_kps(k, rk), "final keypoints"
    assert np.array_equal(d, rd), "descriptors"
    assert ex.GetTimeOfComputePyramid() > 0 and ex.GetTImeOfComputeDescriptor() > 0
    ex.close()


def test_extract_strided_input_and_repeatability(hip, oracle):
    from orbhip import synth
    from orbhip.extractor import ORBextractor
    big = synth.make_frames(9, 700, 500, 1)[0]
    view = big[10:490, 30:670]                     # 640x480 view, stride 700
    ex = ORBextractor(1000, max_w=640, max_h=480)
    ref = oracle.Extractor(1000)
    import ctypes as C
    from orbhip.capi import KP_DTYPE, _p, check
    kps = np.zeros(ex.cap, KP_DTYPE)
    desc = np.zeros((ex.cap, 32), np.uint8)
    n = C.c_int()
    ptr = C.c_void_p(big.ctypes.data + 10 * 700 + 30)
    check(ex._L.orbhip_extract(ex.handle, ptr, 640, 480, 700, _p(kps), _p(desc), ex.cap, C.byref(n), None))
    rk, rd = ref(np.ascontiguousarray(view))
    assert _same_kps(kps[:n.value], rk) and np.array_equal(desc[:n.value], rd)
    k2, d2 = ex(np.ascontiguousarray(view))
    assert _same_kps(k2, rk) and np.array_equal(d2, rd)
    ex.close()


def test_extract_batch_matches_per_frame_oracle(hip, oracle):
    from orbhip import synth
    from orbhip.extractor import ORBextractor
    frames = synth.make_frames(21, 640, 480, 6)
    ex = ORBextractor(1000, max_w=640, max_h=480, max_batch=6)
    ref = oracle.Extractor(1000)
    ks, ds = ex.extract_batch(frames)
    for b in range(6):
        rk, rd = ref(frames[b])
        assert _same_kps(ks[b], rk) and np.array_equal(ds[b], rd), "frame %d" % b
        assert np.array_equal(ex.image_pyramid(3, frame=b), ref.pyramid(3))
    # a smaller batch and a different size on the same context
    ks2, ds2 = ex.extract_batch(frames[:2, :400, :600].copy())
    for b in range(2):
        rk, rd = ref(np.ascontiguousarray(frames[b, :400, :600]))
        assert _same_kps(ks2[b], rk) and np.array_equal(ds2[b], rd)
    ex.close()


def test_extract_batch_device_resident(hip, oracle):
    import hiprt
    from orbhip import synth
    from orbhip.extractor import ORBextractor
    B, W, H = 5, 752, 480
    frames = synth.make_frames(33, W, H, B)
    for stride in (768, 756):                       # 16-byte aligned rows (aliased) and unaligned (repacked)
        host = np.zeros((B, H, stride), np.uint8)
        host[:, :, :W] = frames
        d_img = hiprt.DevBuf.from_numpy(host)
        ex = ORBextractor(1000, max_w=W, max_h=H, max_batch=B)
        cap = ex.cap
        d_kps, d_desc, d_cnt = hiprt.DevBuf(B * cap * 28), hiprt.DevBuf(B * cap * 32), hiprt.DevBuf(B * 4)
        d_bi, d_bd, d_sd = hiprt.DevBuf(B * cap * 4), hiprt.DevBuf(B * cap * 4), hiprt.DevBuf(B * cap * 4)
        ex.extract_batch_device(d_img.ptr, B, W, H, stride, H * stride, d_kps.ptr, d_desc.ptr, cap, d_cnt.ptr)
        from orbhip.capi import check
        check(ex._L.orbhip_hamming_knn2_seq_device(ex.handle, d_desc.ptr, d_cnt.ptr, cap, B, 1, d_bi.ptr, d_bd.ptr,
                                                   d_sd.ptr), ex.handle, "knn2_seq")
        ex.sync()
        cnt = d_cnt.to_numpy(np.int32, (B,))
        kps = d_kps.to_numpy(np.uint8, (B, cap, 28))
        desc = d_desc.to_numpy(np.uint8, (B, cap, 32))
        bi, bd, sd = (x.to_numpy(np.int32, (B, cap)) for x in (d_bi, d_bd, d_sd))
        ref = oracle.Extractor(1000)
        prev = None
        for b in range(B):
            rk, rd = ref(frames[b])
            assert cnt[b] == len(rk)
            assert kps[b, :cnt[b]].tobytes() == rk.tobytes()
            assert np.array_equal(desc[b, :cnt[b]], rd)
            if prev is not None:                    # sequence matching: frame b against frame b-1
                wi, wd, ws = oracle.knn2(rd, prev)
                n = cnt[b]
                assert np.array_equal(bi[b, :n], wi) and np.array_equal(bd[b, :n], wd) and np.array_equal(sd[b, :n], ws)
            else:
                assert (bi[b, :cnt[b]] == -1).all() and (bd[b, :cnt[b]] == 256).all()
            prev = rd
        ms = (__import__("ctypes").c_float * 6)()
        check(ex._L.orbhip_get_stage_times(ex.handle, ms), ex.handle, "stage times")
        assert all(m > 0 for m in ms)
        ex.close()
        for x in (d_img, d_kps, d_desc, d_cnt, d_bi, d_bd, d_sd):
            x.free()


def test_errors_are_reported(hip):
    from orbhip.extractor import ORBextractor
    from orbhip.capi import OrbHipError
    ex = ORBextractor(1000, max_w=640, max_h=480)
    with pytest.raises(OrbHipError):
        ex(np.zeros((90, 90), np.uint8))             # level 7 has no 30-px cell: the reference divides by 0
    with pytest.raises(OrbHipError):
        ex(np.zeros((600, 800), np.uint8))           # larger than the context
    with pytest.raises(OrbHipError):
        ORBextractor(1000, 1.0, 8, 20, 7, max_w=640, max_h=480)   # scale factor 1: the reference's own quota formula is 0 / 0
    k, d = ex(np.zeros((480, 640), np.uint8))        # flat image: no corners at all
    assert len(k) == 0 and d.shape == (0, 32)
    k, d = ex(np.zeros((0, 0), np.uint8))            # empty image: silent return (src/ORBextractor.cc:1048)
    assert len(k) == 0
    ex.close()


def test_knn2_matches_oracle(hip, oracle):
    from orbhip import synth
    from orbhip.extractor import ORBextractor, ORBmatcher
    ex = ORBextractor(1000, max_w=640, max_h=480)
    m = ORBmatcher(0.7, True, ctx=ex)
    db = synth.make_descriptor_db(1, 20000)
    db[777] = db[12]                                  # duplicate rows: lowest index must win
    q, _ = synth.make_queries(2, db, 700)
    q[5] = db[777]
    # (1|8|9|32, 20000) take the few-query path (lanes own database rows), the others the many-query path
    for nq, ndb in [(700, 20000), (1, 20000), (8, 20000), (9, 20000), (32, 20000), (33, 20000), (700, 1), (257, 63),
                    (3, 0)]:
        got = m.knn2(q[:nq], db[:ndb])
        want = oracle.knn2(q[:nq], db[:ndb])
        for g, w in zip(got, want):
            assert np.array_equal(g, w), (nq, ndb)
    # size-independent property at a larger size: every query's best distance is its flip count bound
    q2, rows = synth.make_queries(3, db, 2000, max_flips=20)
    bi, bd, sd = m.knn2(q2, db)
    d_src = np.unpackbits(q2 ^ db[rows], axis=1).sum(1)
    assert (bd <= d_src).all() and (sd >= bd).all()
    same = np.unpackbits(q2 ^ db[bi], axis=1).sum(1)
    assert np.array_equal(same, bd)
    ex.close()


def test_knn2_lists_matches_oracle(hip, oracle):
    from orbhip import synth
    from orbhip.extractor import ORBextractor, ORBmatcher
    rng = np.random.default_rng(4)
    ex = ORBextractor(1000, max_w=640, max_h=480)
    m = ORBmatcher(ctx=ex)
    db = synth.make_descriptor_db(5, 3000)
    q, _ = synth.make_queries(6, db, 500)
    lens = rng.integers(0, 50, 500)
    lens[::17] = 0                                     # empty candidate lists
    off = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
    cand = rng.integers(0, 3000, off[-1]).astype(np.int32)
    got = m.knn2_lists(q, db, off, cand)
    want = oracle.knn2_lists(q, db, off, cand)
    for g, w in zip(got, want):
        assert np.array_equal(g, w)
    ex.close()


def _fv(rng, n, nnodes, assign=None):
    node = rng.integers(0, nnodes, n) if assign is None else assign
    ids = sorted(set(int(v) for v in node))
    lists = [np.nonzero(node == k)[0] for k in ids]
    off = np.concatenate([[0], np.cumsum([len(x) for x in lists])]).astype(np.int32)
    idx = (np.concatenate(lists) if lists else np.zeros(0)).astype(np.int32)
    return np.array(ids, np.int32), off, idx


@pytest.mark.parametrize("kf_kf", [False, True])
def test_search_by_bow_matches_oracle(hip, oracle, kf_kf):
    from orbhip.extractor import ORBextractor, ORBmatcher
    rng = np.random.default_rng(30 + kf_kf)
    ex = ORBextractor(1000, max_w=640, max_h=480)
    for trial, (n1, n2, nnodes) in enumerate([(1000, 1100, 100), (300, 200, 7), (50, 2000, 3), (10, 10, 40)]):
        d2 = rng.integers(0, 256, (n2, 32), dtype=np.uint8)
        src = rng.integers(0, n2, n1)
        d1 = d2[src].copy()
        d1 ^= (rng.integers(0, 256, (n1, 32), dtype=np.uint8) & rng.integers(0, 256, (n1, 32), dtype=np.uint8)
               & rng.integers(0, 256, (n1, 32), dtype=np.uint8))
        a2 = rng.uniform(0, 360, n2).astype(np.float32)
        a1 = ((a2[src] + rng.choice([0, 0, 0, 95, 200], n1) + rng.uniform(-5, 5, n1)) % 360).astype(np.float32)
        v1 = (rng.random(n1) < 0.85).astype(np.uint8)
        v2 = (rng.random(n2) < 0.9).astype(np.uint8) if kf_kf else None
        node2 = rng.integers(0, nnodes, n2)
        node1 = np.where(rng.random(n1) < 0.85, node2[src], rng.integers(0, nnodes + 3, n1))
        fv1, fv2 = _fv(rng, n1, nnodes, node1), _fv(rng, n2, nnodes, node2)
        for ori in (True, False):
            m = ORBmatcher(0.75, ori, ctx=ex)
            n, m12, m21 = m.SearchByBoW(d1, v1, a1, fv1, d2, v2, a2, fv2, kf_kf=kf_kf)
            wn, w12, w21 = oracle.search_by_bow(d1, v1, a1, fv1, d2, v2, a2, fv2, th=50, th_mode=int(kf_kf),
                                                nnratio=0.75, check_ori=ori)
            assert n == wn and np.array_equal(m12, w12) and np.array_equal(m21, w21), (trial, ori)
    ex.close()


def test_full_size_properties_batch(hip, oracle):
    """BASELINE full size in batched mode: size-independent properties + spot checks vs oracle."""
    from orbhip import synth
    from orbhip.extractor import ORBextractor
    B = 16
    frames = synth.make_frames(40, 640, 480, B)
    ex = ORBextractor(1000, max_w=640, max_h=480, max_batch=B)
    ks, ds = ex.extract_batch(frames)
    ks2, ds2 = ex.extract_batch(frames)
    ref = oracle.Extractor(1000)
    for b in range(B):
        assert _same_kps(ks[b], ks2[b]) and np.array_equal(ds[b], ds2[b])      # idempotent / deterministic
        k = ks[b]
        assert 1000 <= len(k) <= 1000 + 3 * 8
        assert (np.diff(k["octave"]) >= 0).all()                               # levels concatenated in order
        assert (k["angle"] >= 0).all() and (k["angle"] <= 360).all()
        assert (k["response"] >= 7).all()
        xy = np.stack([k["x"], k["y"], k["octave"]], 1)
        assert len(np.unique(xy, axis=0)) == len(k)                            # no duplicate keypoints
    for b in (0, 7, 15):
        rk, rd = ref(frames[b])
        assert _same_kps(ks[b], rk) and np.array_equal(ds[b], rd)
    ex.close()


def test_noise_image_overflows_the_fast_lists(hip, oracle):
    """White noise: most pixels pass the compass test and many are corners, so the FAST kernel's
    LDS work/corner lists overflow and the exact fallback paths run."""
    from orbhip.extractor import ORBextractor
    rng = np.random.default_rng(77)
    img = rng.integers(0, 256, (240, 320), dtype=np.uint8)
    img[:, 160:] = (img[:, 160:] // 4 + 90).astype(np.uint8)      # half of it low contrast
    ex = ORBextractor(500, 1.2, 4, 20, 7, max_w=320, max_h=240)
    ref = oracle.Extractor(500, 1.2, 4, 20, 7)
    k, d = ex(img)
    rk, rd = ref(img)
    for l in range(4):
        gc, rc = ex.level_candidates(l), ref.level_cands(l)
        assert len(gc) == len(rc) and gc.tobytes() == rc.tobytes(), "FAST candidates level %d" % l
    assert _same_kps(k, rk) and np.array_equal(d, rd)
    ex.close()


def test_forced_tiny_fast_lists_take_the_fallback_paths(hip):
    """ORBHIP_FAST_LISTCAP=8 makes every tile overflow both lists; results must not change."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = (
        "import sys, numpy as np\n"
        "sys.path[:0] = [%r, %r]\n"
        "import orb_oracle_py as oracle\n"
        "from orbhip import synth\n"
        "from orbhip.extractor import ORBextractor\n"
        "img = synth.make_frames(5, 640, 480, 1)[0]\n"
        "ex = ORBextractor(1000, max_w=640, max_h=480); ref = oracle.Extractor(1000)\n"
        "k, d = ex(img); rk, rd = ref(img)\n"
        "assert k.tobytes() == rk.tobytes() and np.array_equal(d, rd)\n"
        "print('fallback ok', len(k))\n"
    ) % (os.path.join(root, "vi-orb-slam-icra2018_amd"), os.path.join(root, "oracle"))
    env = dict(os.environ, ORBHIP_FAST_LISTCAP="8")
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True)
    assert out.returncode == 0 and "fallback ok" in out.stdout, out.stdout + out.stderr


def test_two_contexts_on_two_host_threads(hip, oracle):
    """Stereo pattern of the reference (src/Frame.cc:422-425): left and right extractor instances run
    concurrently on two std::threads.  Here: two contexts, two Python threads (ctypes drops the GIL)."""
    import threading
    from orbhip import synth
    from orbhip.extractor import ORBextractor
    frames = synth.make_frames(90, 1241, 376, 2)
    exs = [ORBextractor(2000, max_w=1241, max_h=376) for _ in range(2)]
    res = [None, None]

    def work(i):
        out = []
        for _ in range(6):
            out.append(exs[i](frames[i]))
        res[i] = out
    ts = [threading.Thread(target=work, args=(i,)) for i in range(2)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    ref = oracle.Extractor(2000)
    for i in range(2):
        rk, rd = ref(frames[i])
        for k, d in res[i]:
            assert _same_kps(k, rk) and np.array_equal(d, rd)
    for e in exs:
        e.close()
