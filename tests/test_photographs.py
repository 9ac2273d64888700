"""The only camera pictures this image holds -- the sample photographs scikit-learn and matplotlib install (china.jpg, flower.jpg
427 x 640, grace_hopper.jpg 600 x 512; orbhip/synth.py load_photographs reads them where they lie) -- through the same stage-by-
stage comparison as the synthetic frames: every other frame under tests/ is orbhip/synth.py's own drawing, one texture family.
Skipped where the packages (or PIL) are absent."""
import numpy as np
import pytest


def _photos():
    from orbhip import synth
    p = synth.load_photographs()
    if not p:
        pytest.skip("no sample photographs in this image")
    return p


def test_photographs_load_and_are_pictures(oracle):
    """(CPU) the photographs decode to grey pictures the oracle finds its full quota of features in."""
    for g in _photos():
        assert g.dtype == np.uint8 and g.ndim == 2 and min(g.shape) >= 400 and g.std() > 20
        k, d = oracle.Extractor(1000)(g)
        assert 900 <= len(k) <= 1100 and d.shape == (len(k), 32)


@pytest.mark.gpu
@pytest.mark.parametrize("nf", [1000, 2000])
def test_photographs_stage_by_stage(oracle, nf):
    from orbhip.extractor import ORBextractor
    for g in _photos():
        h, w = g.shape
        ex = ORBextractor(nf, 1.2, 8, 20, 7, max_w=w, max_h=h, max_batch=1)
        ref = oracle.Extractor(nf, 1.2, 8, 20, 7)
        k, d = ex(g)
        rk, rd = ref(g)
        for l in range(8):
            assert np.array_equal(ex.image_pyramid(l), ref.pyramid(l)), "pyramid level %d" % l
            gc, rc = ex.level_candidates(l), ref.level_cands(l)
            assert len(gc) == len(rc) and gc.tobytes() == rc.tobytes(), "FAST candidates level %d" % l
            assert ex.level_keypoints(l).tobytes() == ref.level_keypoints(l).tobytes(), "quadtree / angle level %d" % l
            if len(ref.level_keypoints(l)):
                assert np.array_equal(ex.blurred(l), ref.blurred(l)), "blur level %d" % l
        assert len(k) == len(rk) > 900 and k.tobytes() == rk.tobytes() and np.array_equal(d, rd)
        ex.close()


@pytest.mark.gpu
def test_photograph_frames_batched_with_matching(oracle):
    """640 x 480 frames cut from the photographs, as a device batch: extraction of every frame, then brute-force best / second of
    each frame against the one before it (the matrix-pipe kernel on real descriptors)."""
    import hiprt
    from orbhip import synth
    from orbhip.capi import check
    from orbhip.extractor import ORBextractor
    frames = synth.photograph_frames(640, 480, 12)
    if frames is None:
        pytest.skip("no sample photographs in this image")
    B = len(frames)
    ex = ORBextractor(1000, max_w=640, max_h=480, max_batch=B)
    ref = oracle.Extractor(1000)
    ks, ds = ex.extract_batch(frames)
    want = [ref(f) for f in frames]
    for k, d, (rk, rd) in zip(ks, ds, want):
        assert k.tobytes() == rk.tobytes() and np.array_equal(d, rd)
    cap = ex.cap
    desc = np.zeros((B, cap, 32), np.uint8)
    counts = np.array([len(k) for k, _ in want], np.int32)
    for b, (_, d) in enumerate(want):
        desc[b, :len(d)] = d
    d_desc, d_cnt = hiprt.DevBuf.from_numpy(desc), hiprt.DevBuf.from_numpy(counts)
    d_bi, d_bd, d_sd = (hiprt.DevBuf(B * cap * 4) for _ in range(3))
    check(ex._L.orbhip_hamming_knn2_seq_device(ex.handle, d_desc.ptr, d_cnt.ptr, cap, B, 1, d_bi.ptr, d_bd.ptr, d_sd.ptr), ex.handle)
    ex.sync()
    bi, bd, sd = (x.to_numpy(np.int32, (B, cap)) for x in (d_bi, d_bd, d_sd))
    for b in range(1, B):
        n = counts[b]
        wi, wd, ws = oracle.knn2(desc[b, :n], desc[b - 1, :counts[b - 1]])
        assert np.array_equal(bi[b, :n], wi) and np.array_equal(bd[b, :n], wd) and np.array_equal(sd[b, :n], ws), b
    ex.close()
