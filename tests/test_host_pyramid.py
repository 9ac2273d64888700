"""mvImagePyramid on the host without a copy per level (orbhip_set_host_pyramid / orbhip_host_pyramid_level): the
page-locked block that the single-frame chain fills beside the kernels must hold the same bytes as the oracle's pyramid
(ref: src/ORBextractor.cc:1128-1161; read by src/Frame.cc:817) -- for the eager first call, for graph replays with a
different frame, for a stereo pair, for a batch, and it must stay off when not asked for."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("W,H,NF", [(640, 480, 1000), (752, 480, 1200), (321, 243, 400)])
def test_single_frame_host_pyramid_equals_oracle(oracle, W, H, NF):
    from orbhip import synth
    from orbhip.extractor import ORBextractor
    frames = synth.make_frames(17, W, H, 4)
    ref = oracle.Extractor(NF)
    ex = ORBextractor(NF, max_w=W, max_h=H)
    ex.set_host_pyramid(True)
    for rep, f in enumerate([0, 1, 2, 3, 1]):               # call 0 runs eagerly, the others replay the graph
        k, d = ex(frames[f])
        rk, rd = ref(frames[f])
        assert k.tobytes() == rk.tobytes() and np.array_equal(d, rd), rep
        for l in range(ex.nlevels):
            assert np.array_equal(ex.host_pyramid(l), ref.pyramid(l)), (rep, l)
            assert np.array_equal(ex.host_pyramid(l), ex.image_pyramid(l)), (rep, l)
    ex.close()


def test_pair_and_batch_host_pyramid(oracle):
    from orbhip import synth
    from orbhip.capi import OrbHipError
    from orbhip.extractor import ORBextractor
    W, H, NF = 640, 480, 800
    frames = synth.make_frames(23, W, H, 9)
    ref = oracle.Extractor(NF)
    ex = ORBextractor(NF, max_w=W, max_h=H, max_batch=9)
    ex.set_host_pyramid(True)
    for B in (2, 9, 2):                                      # stereo pair (graph path), batch (eager path), pair again
        ks, ds = ex.extract_batch(frames[:B])
        for b in range(B):
            rk, rd = ref(frames[b])
            assert ks[b].tobytes() == rk.tobytes() and np.array_equal(ds[b], rd), (B, b)
            for l in range(1, ex.nlevels):
                assert np.array_equal(ex.host_pyramid(l, b), ref.pyramid(l)), (B, b, l)
            if B < 8:
                assert np.array_equal(ex.host_pyramid(0, b), frames[b])
            else:
                with pytest.raises(OrbHipError):            # level 0 of a batch is the caller's own memory
                    ex.host_pyramid(0, b)
        with pytest.raises(OrbHipError):
            ex.host_pyramid(1, B)                            # frame out of range
    ex.close()


def test_host_pyramid_off_by_default_and_switchable(oracle):
    from orbhip import synth
    from orbhip.capi import OrbHipError
    from orbhip.extractor import ORBextractor
    W, H = 376, 241
    f = synth.make_frames(29, W, H, 2)
    ref = oracle.Extractor(500)
    ex = ORBextractor(500, max_w=W, max_h=H)
    ex(f[0])
    with pytest.raises(OrbHipError):
        ex.host_pyramid(1)
    ex.set_host_pyramid(True)                                # the captured chain is rebuilt with the copy in it
    ex(f[1])
    ref(f[1])
    assert np.array_equal(ex.host_pyramid(3), ref.pyramid(3))
    ex.set_host_pyramid(False)
    k, d = ex(f[0])
    rk, rd = ref(f[0])
    assert k.tobytes() == rk.tobytes() and np.array_equal(d, rd)
    with pytest.raises(OrbHipError):
        ex.host_pyramid(3)
    ex.close()
