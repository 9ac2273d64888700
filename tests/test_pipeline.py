"""Host-fed pipeline (orbhip_pipe_*): batches from pinned and from pageable host memory, copies overlapped with the
kernels of the neighbouring batches; results must equal the oracle's frame by frame, in submission order.
The reference's frames always come from host memory (Examples/Monocular/mono_euroc.cc:73 -> src/Frame.cc:591-597)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _same(a, b):
    return len(a) == len(b) and a.tobytes() == b.tobytes()


@pytest.mark.parametrize("W,H,NF,B,depth", [(640, 480, 1000, 8, 2), (752, 480, 1000, 3, 3), (321, 243, 400, 9, 2)])
def test_pipelined_batches_match_oracle(oracle, W, H, NF, B, depth):
    from orbhip import synth
    from orbhip.capi import OrbHipError
    from orbhip.extractor import ORBextractor
    nb = 5                                                  # batches: more than the ring is deep, the last one short
    frames = synth.make_frames(91, W, H, 6)
    ref = oracle.Extractor(NF)
    want = [ref(f) for f in frames]
    ex = ORBextractor(NF, max_w=W, max_h=H, max_batch=B)
    ex.pipe_create(depth, B, W, H)
    pinned = [ex.host_frames((B, H, W)) for _ in range(depth)]
    order = []                                              # frame index of every submitted image
    sizes = []
    rng = np.random.default_rng(92)
    done = 0
    for n in range(nb):
        if n >= depth:                                      # ring full: collect the oldest batch first
            ks, ds = ex.pipe_wait()
            for i, (k, d) in enumerate(zip(ks, ds)):
                rk, rd = want[order[done + i]]
                assert _same(k, rk) and np.array_equal(d, rd), (n, i)
            done += len(ks)
            assert len(ks) == sizes[n - depth]
        nbatch = B if n < nb - 1 else max(1, B // 2)
        idx = rng.integers(0, len(frames), nbatch)
        buf = pinned[n % depth]
        buf[:nbatch] = frames[idx]
        if n == 1:
            pageable = np.ascontiguousarray(frames[idx])               # pageable memory is accepted too; like every submitted
            ex.pipe_submit(pageable)                                   # buffer it must stay alive until its wait returns
        else:
            ex.pipe_submit(buf[:nbatch])
        order += list(idx)
        sizes.append(nbatch)
    with pytest.raises(OrbHipError, match="not collected"):
        for _ in range(depth + 1):
            ex.pipe_submit(pinned[0][:1])
    while done < len(order):
        ks, ds = ex.pipe_wait()
        for i, (k, d) in enumerate(zip(ks, ds)):
            if done + i < len(order):
                rk, rd = want[order[done + i]]
                assert _same(k, rk) and np.array_equal(d, rd)
        done += len(ks)
    # strided rows / frames (a ROI of a larger pinned buffer)
    big = ex.host_frames((2, H + 10, W + 24))
    big[:] = 0
    big[:, 5:5 + H, 8:8 + W] = frames[:2]
    roi = big[:, 5:5 + H, 8:8 + W]
    ex.pipe_destroy()
    ex.pipe_create(2, B, W, H)
    ex.pipe_submit((roi.ctypes.data, 2, roi.strides[1], roi.strides[0]))
    ks, ds = ex.pipe_wait()
    for i in range(2):
        assert _same(ks[i], want[i][0]) and np.array_equal(ds[i], want[i][1])
    with pytest.raises(OrbHipError, match="nothing submitted"):
        ex.pipe_wait()
    for a in pinned + [big]:
        ex.host_free(a)
    ex.close()


def test_pipeline_with_bow_matching_stage(oracle):
    """orbhip_pipe_enable_bow: Frame::ComputeBoW + SearchByBoW of every frame against its predecessor inside the batch,
    results travelling back with the keypoints (what bench.py's host_fed figure runs)."""
    from orbhip import distributed as D, synth
    from orbhip.extractor import ORBextractor
    from orbhip.vocabulary import ORBVocabulary
    W, H, NF, B = 640, 480, 1000, 5
    frames = synth.make_frames(93, W, H, B)
    blob = D.make_synthetic_vocabulary(94, k=10, L=4)
    ex = ORBextractor(NF, max_w=W, max_h=H, max_batch=B)
    ORBVocabulary(ex).loadFromBinaryBlob(blob)
    ex.pipe_create(2, B, W, H)
    ex.pipe_enable_bow(2, 0.7, True)
    buf = ex.host_frames((B, H, W))
    buf[:] = frames
    ex.pipe_submit(buf)
    ex.pipe_submit(buf[:3])
    ref, V = oracle.Extractor(NF), oracle.Vocabulary(blob)
    for nb in (B, 3):
        kps, desc, cnt = ex.pipe_wait(copy=False)
        cap = kps.shape[1]
        m12, m21, nm = ex.pipe_matches(nb, cap)
        prev = None
        for b in range(nb):
            k, d = ref(frames[b])
            assert cnt[b] == len(k) and kps[b, :len(k)].tobytes() == k.tobytes() and np.array_equal(desc[b, :len(k)], d)
            w, wt, nid = V.transform(d, 2)
            fv = oracle.feature_vector(nid, wt)
            if prev is None:
                assert nm[b] == 0 and (m12[b] == -1).all()
            else:
                pk, pd, pfv = prev
                wn, w12, w21 = oracle.search_by_bow(pd, np.ones(len(pd), np.uint8), pk["angle"], pfv, d, None, k["angle"], fv,
                                                    th=50, th_mode=0, nnratio=0.7, check_ori=True)
                assert nm[b] == wn > 50 and np.array_equal(m12[b, :len(pk)], w12) and np.array_equal(m21[b, :len(k)], w21)
            prev = (k, d, fv)
    ex.host_free(buf)
    ex.close()


def test_results_of_a_wait_survive_the_submits_that_follow(oracle):
    """include/orbhip.h: the pointers orbhip_pipe_wait returns stay valid until the NEXT wait.  With a full ring (depth 2:
    submit 0, 1; wait 0) the next submit used to land in the very block the caller was still reading; the ring
    now has depth + 1 host result blocks."""
    import time
    from orbhip import synth
    from orbhip.extractor import ORBextractor
    W, H, NF, B, depth = 640, 480, 1000, 4, 2
    frames = synth.make_frames(95, W, H, 3 * B)
    ex = ORBextractor(NF, max_w=W, max_h=H, max_batch=B)
    ex.pipe_create(depth, B, W, H)
    bufs = [ex.host_frames((B, H, W)) for _ in range(3)]
    for i, b in enumerate(bufs):
        b[:] = frames[i * B:(i + 1) * B]
    ex.pipe_submit(bufs[0])
    ex.pipe_submit(bufs[1])
    kps, desc, cnt = ex.pipe_wait(copy=False)                # views into the pinned block of batch 0
    snap = (kps.copy(), desc.copy(), cnt.copy())
    ex.pipe_submit(bufs[2])                                  # the ring admits exactly one more batch
    ex.sync()
    time.sleep(0.3)                                          # ... and its copy-out has long finished
    assert np.array_equal(cnt, snap[2]) and kps.tobytes() == snap[0].tobytes() and np.array_equal(desc, snap[1])
    ref = oracle.Extractor(NF)
    for b in range(B):
        rk, rd = ref(frames[b])
        assert cnt[b] == len(rk) and kps[b, :len(rk)].tobytes() == rk.tobytes() and np.array_equal(desc[b, :len(rk)], rd)
    for batch in (1, 2):                                     # the later batches are intact too
        ks, ds = ex.pipe_wait()
        for b in range(B):
            rk, rd = ref(frames[batch * B + b])
            assert _same(ks[b], rk) and np.array_equal(ds[b], rd)
    for a in bufs:
        ex.host_free(a)
    ex.close()


@pytest.mark.gpu
def test_stage_timing_modes_change_no_result():
    """orbhip_set_stage_timing: 2 records every stage's events, 1 only the pair around the FAST launch (the other stage times
    read 0), 0 none; keypoints and descriptors of a batch are the same in all three."""
    import ctypes as C
    from orbhip import synth
    from orbhip.extractor import ORBextractor
    W, H, B = 400, 300, 16
    frames = synth.make_frames(5, W, H, B)
    ex = ORBextractor(600, 1.2, 6, 20, 7, max_w=W, max_h=H, max_batch=B)
    L = ex._L
    want = None
    # (stage 3 -- k_blur -- reads 0 since round 6: a batch blurs inside the describe kernel and never launches it)
    for mode, nonzero in ((2, {0, 1, 2, 4}), (1, {1}), (0, set())):
        assert L.orbhip_set_stage_timing(ex.handle, mode) == 0
        ks, ds = ex.extract_batch(frames)
        ms = (C.c_float * 6)()
        assert L.orbhip_get_stage_times(ex.handle, ms) == 0
        for i in range(5):
            assert (ms[i] > 0) == (i in nonzero), (mode, i, ms[i])
        got = [(k.tobytes(), d.tobytes()) for k, d in zip(ks, ds)]
        if want is None:
            want = got
        assert got == want, mode
    assert L.orbhip_set_stage_timing(ex.handle, 3) != 0
    ex.close()
