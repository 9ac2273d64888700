"""More parity cases for the extraction path: unusual pyramid parameters (the non-staged resize path, one level,
twelve levels), a large frame, frames without corners, and a batch that mixes them -- single-frame kernels and
the batch variants (8 frames and more) alike.  Bit-exact against the oracle."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _same(a, b):
    return len(a) == len(b) and a.tobytes() == b.tobytes()


@pytest.mark.parametrize("scale,nlev,ini,mn,nf,w,h", [
    (1.1, 12, 20, 7, 1500, 640, 480),      # many shallow levels
    (1.5, 5, 20, 7, 800, 640, 480),        # source window of a resize tile exceeds the LDS stage: generic path
    (2.0, 3, 25, 5, 500, 752, 480),
    (1.2, 1, 20, 7, 300, 320, 240),        # a single level: no resize at all
    (1.3, 3, 40, 12, 200, 400, 300),       # high thresholds: many cells fall back to minThFAST
    # the whole threshold domain the reference's constructor accepts (cv::FAST clamps to [0, 255]):
    (1.2, 4, 20, 0, 400, 400, 300),        # minThFAST 0 = 1: a zero-score corner never survives the strict suppression
    (1.2, 4, 0, -5, 400, 320, 240),        # both below 1
    (1.2, 4, 7, 20, 400, 400, 300),        # iniThFAST < minThFAST: the second run cannot add anything
    (1.2, 4, 20, 20, 400, 400, 300),       # equal
    (1.2, 3, 255, 30, 300, 400, 300),      # nothing can pass 255: every cell takes the second threshold
    (1.2, 3, 300, 254, 300, 400, 300),     # clamped to 255 / almost nothing passes 254
    (1.2, 3, 21, 8, 300, 400, 300),        # thresholds of the other parities (the byte-compare constants depend on t & 1)
    (1.2, 3, 33, 11, 300, 400, 300),
])
def test_unusual_pyramid_parameters(oracle, scale, nlev, ini, mn, nf, w, h):
    from orbhip import synth
    from orbhip.extractor import ORBextractor
    frames = synth.make_frames(300 + nlev, w, h, 2)
    ref = oracle.Extractor(nf, scale, nlev, ini, mn)
    ex = ORBextractor(nf, scale, nlev, ini, mn, max_w=w, max_h=h)
    k, d = ex(frames[0])
    rk, rd = ref(frames[0])
    for l in range(nlev):
        assert np.array_equal(ex.image_pyramid(l), ref.pyramid(l)), "pyramid level %d" % l
    assert _same(k, rk) and np.array_equal(d, rd) and (len(rk) > 50 or ini >= 254)
    for l in range(nlev):
        gc, rc = ex.level_candidates(l), ref.level_cands(l)
        assert len(gc) == len(rc) and gc.tobytes() == rc.tobytes(), "FAST candidates level %d" % l
    ex.close()
    # the same through the batch kernels (>= 8 frames)
    exb = ORBextractor(nf, scale, nlev, ini, mn, max_w=w, max_h=h, max_batch=8)
    ks, ds = exb.extract_batch(np.stack([frames[0], frames[1]] * 4))
    rk1, rd1 = ref(frames[1])
    for b in range(8):
        assert _same(ks[b], rk if b % 2 == 0 else rk1) and np.array_equal(ds[b], rd if b % 2 == 0 else rd1), "frame %d" % b
    exb.close()


def test_full_hd_frame_4000_features(oracle):
    from orbhip import synth
    from orbhip.extractor import ORBextractor
    img = synth.make_frames(77, 1920, 1080, 1)[0]
    ex = ORBextractor(4000, max_w=1920, max_h=1080)
    ref = oracle.Extractor(4000)
    k, d = ex(img)
    rk, rd = ref(img)
    assert _same(k, rk) and np.array_equal(d, rd) and len(rk) >= 4000
    ex.close()


def test_frames_without_corners_and_mixed_batch(oracle):
    from orbhip import synth
    from orbhip.extractor import ORBextractor
    W, H = 640, 480
    rng = np.random.default_rng(5)
    flat = np.full((H, W), 117, np.uint8)
    ramp = np.tile(np.linspace(0, 255, W).astype(np.uint8), (H, 1))                 # gradient: no FAST corners
    faint = (118 + 5 * ((np.add.outer(np.arange(H) // 24, np.arange(W) // 24)) % 2)).astype(np.uint8)   # contrast 5 < minTh
    weak = (110 + 12 * ((np.add.outer(np.arange(H) // 17, np.arange(W) // 19)) % 2)).astype(np.uint8)   # only minThFAST fires
    weak = np.clip(weak.astype(np.int32) + rng.integers(-1, 2, (H, W)), 0, 255).astype(np.uint8)
    normal = synth.make_frames(6, W, H, 4)
    ref = oracle.Extractor(1000)
    ex = ORBextractor(1000, max_w=W, max_h=H)
    for name, img in (("flat", flat), ("ramp", ramp), ("faint", faint), ("weak", weak)):
        k, d = ex(img)
        rk, rd = ref(img)
        assert _same(k, rk) and d.shape == rd.shape and np.array_equal(d, rd), name
    assert len(ref(flat)[0]) == 0 and len(ref(weak)[0]) > 0
    ex.close()
    batch = np.stack([normal[0], flat, normal[1], weak, ramp, normal[2], faint, normal[3], flat])
    exb = ORBextractor(1000, max_w=W, max_h=H, max_batch=len(batch))
    ks, ds = exb.extract_batch(batch)
    for b, img in enumerate(batch):
        rk, rd = ref(img)
        assert _same(ks[b], rk) and np.array_equal(ds[b].reshape(-1, 32), rd.reshape(-1, 32)), "frame %d" % b
    exb.close()


@pytest.mark.gpu
def test_tall_cells_with_dense_corners_in_a_batch(oracle):
    """Found by tools/soak_parity.py: 880 x 345 at scale 1.3 has a level with two cell-rows of 45 pixels; with 5-cell FAST runs a
    workgroup then scores 45 x 155 = 6975 pixels, and when the corners are dense enough for the score-every-pixel fallback the
    row / column split of a flat pixel index must still be exact at the end of that range (a 20-bit integer reciprocal was not:
    a corner landed one column outside the run and its rank went out of bounds)."""
    from orbhip import synth
    from orbhip.extractor import ORBextractor
    w, h, nf, nlev, scale, ini, mn, seed = 880, 345, 150, 7, 1.3, 38, 7, 772947031
    frames = synth.make_frames(seed, w, h, 2)
    ref = oracle.Extractor(nf, scale, nlev, ini, mn)
    want = [ref(f) for f in frames]
    ex = ORBextractor(nf, scale, nlev, ini, mn, max_w=w, max_h=h, max_batch=8)
    ks, ds = ex.extract_batch(np.concatenate([frames] * 4))
    for b in range(8):
        assert _same(ks[b], want[b % 2][0]) and np.array_equal(ds[b], want[b % 2][1])
    ex.close()


@pytest.mark.parametrize("nf,scale,nlev,w,h", [(3500, 1.75, 2, 483, 276), (2500, 1.2, 1, 640, 480)])
def test_large_per_level_quota_uses_the_global_memory_quadtree_tables(oracle, nf, scale, nlev, w, h):
    """More than ~2000 features on ONE level: the quadtree's node tables no longer fit the 160 KB of LDS and live in a
    global scratch block (k_quadtree<.., true>).  The reference has no such limit; same results."""
    from orbhip import synth
    from orbhip.extractor import ORBextractor
    rng = np.random.default_rng(9)
    frames = synth.make_frames(310, w, h, 2)
    frames[1] = rng.integers(0, 256, (h, w), dtype=np.uint8)          # noise: enough corners to fill the quota
    ref = oracle.Extractor(nf, scale, nlev, 20, 7)
    ex = ORBextractor(nf, scale, nlev, 20, 7, max_w=w, max_h=h, max_batch=8)
    for f in frames:
        k, d = ex(f)
        rk, rd = ref(f)
        assert _same(k, rk) and np.array_equal(d, rd)
    assert len(rk) > 2000
    ks, ds = ex.extract_batch(np.stack([frames[0], frames[1]] * 4))     # batch kernels
    for b in range(8):
        rk, rd = ref(frames[b % 2])
        assert _same(ks[b], rk) and np.array_equal(ds[b], rd)
    ex.close()


@pytest.mark.parametrize("env", [
    {"ORBHIP_NO_CHAIN": "1"},                 # pyramid of a single frame: one k_resize launch per level
    {"ORBHIP_CHAIN_DEPTH": "7"},              # ... all levels in one k_pyramid_chain launch
    {"ORBHIP_CHAIN_DEPTH": "1"},              # ... one level per chain launch
    {"ORBHIP_NO_GRAPH": "1"},                 # eager launches instead of the captured chain
    {"ORBHIP_FAST_PITCH": "0"},               # k_fast with the run-time LDS pitch
    {"ORBHIP_QT_LDSPTS": "256"},              # quadtree of a single frame: candidates in memory (levels over 256 of them)
    {"ORBHIP_QT_THREADS_SMALL": "256"},
])
def test_alternative_code_paths_give_the_same_results(env):
    """Every switchable implementation choice of the single-frame path (they are read once per process) must leave keypoints,
    descriptors and every pyramid level unchanged: three geometries, a frame and a stereo pair, against the oracle."""
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
        "for (W, H, NF, NL, SF) in [(640, 480, 1000, 8, 1.2), (752, 480, 1500, 8, 1.2), (333, 251, 600, 5, 1.31)]:\n"
        "    f = synth.make_frames(7, W, H, 2)\n"
        "    ex = ORBextractor(NF, SF, NL, max_w=W, max_h=H, max_batch=2); ref = oracle.Extractor(NF, SF, NL)\n"
        "    ex.set_host_pyramid(True)\n"
        "    for rep in range(3):\n"
        "        k, d = ex(f[rep & 1]); rk, rd = ref(f[rep & 1])\n"
        "        assert k.tobytes() == rk.tobytes() and np.array_equal(d, rd), (W, rep)\n"
        "        for l in range(NL):\n"
        "            assert np.array_equal(ex.image_pyramid(l), ref.pyramid(l)), (W, rep, l)\n"
        "            if l: assert np.array_equal(ex.host_pyramid(l), ref.pyramid(l)), (W, rep, l)\n"
        "    ks, ds = ex.extract_batch(f)\n"
        "    for b in range(2):\n"
        "        rk, rd = ref(f[b])\n"
        "        assert ks[b].tobytes() == rk.tobytes() and np.array_equal(ds[b], rd), (W, 'pair', b)\n"
        "    ex.close()\n"
        "print('paths ok')\n"
    ) % (os.path.join(root, "vi-orb-slam-icra2018_amd"), os.path.join(root, "oracle"))
    out = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, **env), capture_output=True, text=True)
    assert out.returncode == 0 and "paths ok" in out.stdout, out.stdout + out.stderr


@pytest.mark.parametrize("env", [
    {"ORBHIP_FAST_FIX": "0"},                 # the generic FAST kernel (any cell grid) instead of the fixed-layout one
    {"ORBHIP_FAST_DEFER": "0"},               # fixed-layout kernel: both polarities of a work-list entry in place
    {"ORBHIP_FAST_TILE_CELLS": "5"},          # runs of five cells (wave 0 finishes two cells)
    {"ORBHIP_FAST_TILE_CELLS": "2"},
    {"ORBHIP_FAST_LISTCAP": "24"},            # every list overflows: dense scoring, score-tile scans, the wave-local fallbacks of pass 1
    {"ORBHIP_FAST_LISTCAP": "24", "ORBHIP_FAST_FIX": "0"},
    {"ORBHIP_DESCRIBE_AX4": "0"},             # angle phase of the batch describe kernel: five dword loads + masks instead of one dwordx4 load + constant weights
    {"ORBHIP_DESCRIBE_KPW": "32"},            # ... 32 slots per workgroup (always the dword form)
    {"ORBHIP_NO_SPLIT": "1"},                 # batch schedules: no half-batch split; blur beside FAST
    {"ORBHIP_BLUR_PLACE": "1"},
    {"ORBHIP_RESIZE_FIT": "0"},               # k_resize<32> (128 x 32 tiles), what a level keeps when no fitted geometry passes the window check
    {"ORBHIP_FUSE_BLUR": "1"},                # levels 1.. and their blurred twins from one kernel per level (k_resize_blur)
])
def test_alternative_batch_code_paths_give_the_same_results(env):
    """The switchable choices of the BATCH path (read once per process): 16-frame batches of two geometries -- one inside the
    fixed-layout FAST kernel's bounds, one with a low second threshold and many empty cells -- against the oracle, keypoints and
    descriptors bit for bit."""
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
        "for (W, H, NF, NL, SF, INI, MN) in [(640, 480, 1000, 8, 1.2, 20, 7), (500, 333, 700, 6, 1.25, 45, 3)]:\n"
        "    f = synth.make_frames(11, W, H, 4)\n"
        "    f[3] = (f[3] // 6 + 100).astype(np.uint8)      # a low-contrast frame: most cells go to the second pass\n"
        "    ex = ORBextractor(NF, SF, NL, INI, MN, max_w=W, max_h=H, max_batch=16); ref = oracle.Extractor(NF, SF, NL, INI, MN)\n"
        "    want = [ref(x) for x in f]\n"
        "    ks, ds = ex.extract_batch(np.concatenate([f] * 4))\n"
        "    for b in range(16):\n"
        "        rk, rd = want[b %% 4]\n"
        "        assert ks[b].tobytes() == rk.tobytes() and np.array_equal(ds[b], rd), (W, b)\n"
        "    ex.close()\n"
        "print('paths ok')\n"
    ) % (os.path.join(root, "vi-orb-slam-icra2018_amd"), os.path.join(root, "oracle"))
    out = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, **env), capture_output=True, text=True)
    assert out.returncode == 0 and "paths ok" in out.stdout, out.stdout + out.stderr


@pytest.mark.gpu
@pytest.mark.parametrize("mfma", ["1", "0"])
def test_brute_force_matching_on_the_matrix_pipe_and_on_the_vector_pipe(mfma):
    """k_knn2_mfma / k_knn2_seq_mfma (default) and the scalar k_knn2 / k_knn2_seq (ORBHIP_KNN2_MFMA=0) against the oracle: low-entropy
    descriptors (few distinct distances: ties decide the index and the second-best), all-zero / all-one rows, duplicate rows, query and
    database counts around the 128-query workgroup, the 32-row matrix tile and the 64-row staged tile, the per-frame sequence
    form with ragged counts, and second-best rows planted inside the winner's own group of the matrix tile (the rescan)."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = (
        "import sys, ctypes as C, numpy as np\n"
        "sys.path[:0] = [%r, %r, %r]\n"
        "import orb_oracle_py as oracle\n"
        "import hiprt\n"
        "from orbhip.capi import check\n"
        "from orbhip.extractor import ORBextractor, ORBmatcher\n"
        "rng = np.random.default_rng(77)\n"
        "ex = ORBextractor(1000, max_w=640, max_h=480); m = ORBmatcher(0.7, True, ctx=ex)\n"
        "def lowent(n):\n"
        "    d = np.zeros((n, 32), np.uint8)\n"
        "    d[:, :2] = rng.integers(0, 256, (n, 2), dtype=np.uint8) & rng.integers(0, 256, (n, 2), dtype=np.uint8)\n"
        "    d[rng.random(n) < 0.05] = 255\n"
        "    d[rng.random(n) < 0.05] = 0\n"
        "    return d\n"
        "for nq, ndb in [(127, 31), (128, 32), (129, 33), (130, 63), (131, 64), (132, 65), (300, 1000), (1000, 4097), (513, 20001)]:\n"
        "    for kind in (0, 1):\n"
        "        db = lowent(ndb) if kind == 0 else rng.integers(0, 256, (ndb, 32), dtype=np.uint8)\n"
        "        q = lowent(nq) if kind == 0 else db[rng.integers(0, ndb, nq)] ^ (rng.integers(0, 256, (nq, 32), dtype=np.uint8) & rng.integers(0, 256, (nq, 32), dtype=np.uint8) & rng.integers(0, 256, (nq, 32), dtype=np.uint8))\n"
        "        if ndb > 40: db[37] = db[5]\n"
        "        got = m.knn2(q, db); want = oracle.knn2(q, db)\n"
        "        for g, w in zip(got, want):\n"
        "            assert np.array_equal(g, w), (nq, ndb, kind)\n"
        "# the per-frame sequence form: frame b against frame b - 1, ragged counts (incl. an empty frame)\n"
        "cap, B = 700, 6\n"
        "counts = np.array([700, 650, 0, 129, 64, 700], np.int32)\n"
        "desc = np.stack([lowent(cap) if b %% 2 else rng.integers(0, 256, (cap, 32), dtype=np.uint8) for b in range(B)])\n"
        "d_desc = hiprt.DevBuf.from_numpy(desc); d_cnt = hiprt.DevBuf.from_numpy(counts)\n"
        "d_bi, d_bd, d_sd = (hiprt.DevBuf(B * cap * 4) for _ in range(3))\n"
        "check(ex._L.orbhip_hamming_knn2_seq_device(ex.handle, d_desc.ptr, d_cnt.ptr, cap, B, 1, d_bi.ptr, d_bd.ptr, d_sd.ptr), ex.handle)\n"
        "ex.sync()\n"
        "bi = d_bi.to_numpy(np.int32, (B, cap)); bd = d_bd.to_numpy(np.int32, (B, cap)); sd = d_sd.to_numpy(np.int32, (B, cap))\n"
        "for b in range(B):\n"
        "    n = counts[b]\n"
        "    if n == 0: continue\n"
        "    wi, wd, ws = oracle.knn2(desc[b, :n], desc[b - 1, :counts[b - 1]]) if b >= 1 else (np.full(n, -1), np.full(n, 256), np.full(n, 256))\n"
        "    assert np.array_equal(bi[b, :n], wi) and np.array_equal(bd[b, :n], wd) and np.array_equal(sd[b, :n], ws), b\n"
        "# a database longer than one key range (16384 rows): the matrix kernel walks it in chunks\n"
        "cap2 = 17000; counts2 = np.array([17000, 4300], np.int32)\n"
        "desc2 = np.stack([lowent(cap2), lowent(cap2)]); desc2[0, 16500] = desc2[0, 17]; desc2[1, :600] = desc2[0, rng.integers(16384, 17000, 600)]\n"
        "d_desc = hiprt.DevBuf.from_numpy(desc2); d_cnt = hiprt.DevBuf.from_numpy(counts2)\n"
        "d_bi, d_bd, d_sd = (hiprt.DevBuf(2 * cap2 * 4) for _ in range(3))\n"
        "check(ex._L.orbhip_hamming_knn2_seq_device(ex.handle, d_desc.ptr, d_cnt.ptr, cap2, 2, 1, d_bi.ptr, d_bd.ptr, d_sd.ptr), ex.handle)\n"
        "ex.sync()\n"
        "wi, wd, ws = oracle.knn2(desc2[1, :4300], desc2[0, :17000])\n"
        "assert np.array_equal(d_bi.to_numpy(np.int32, (2, cap2))[1, :4300], wi) and np.array_equal(d_bd.to_numpy(np.int32, (2, cap2))[1, :4300], wd) and np.array_equal(d_sd.to_numpy(np.int32, (2, cap2))[1, :4300], ws)\n"
        "assert (wi >= 16384).sum() > 100, 'the second chunk must win for some queries'\n"
        "# the second best inside the winner's own group of the matrix tile (a lane holds rows 8 g + 4 h + e of a 32-row tile and only its\n"
        "# smallest key enters the running pair: the other 15 rows are looked at again after the chunk), both forms\n"
        "ndb, nq = 5000, 600\n"
        "db = rng.integers(0, 256, (ndb, 32), dtype=np.uint8)\n"
        "rows = rng.permutation(ndb // 32 - 1)[:nq // 4].repeat(4) * 32 + rng.integers(0, 32, nq)\n"
        "q = db[rows].copy()\n"
        "for i, r in enumerate(rows):\n"
        "    partner = (r & ~31) + ((r & 31) ^ (1, 2, 3, 8, 16, 24, 9)[i %% 7])   # same tile, same h (bit 2 of the row kept)\n"
        "    db[partner] = q[i]; db[partner, i %% 32] ^= 1 << (i %% 8)\n"
        "got = m.knn2(q, db); want = oracle.knn2(q, db)\n"
        "for g, w in zip(got, want):\n"
        "    assert np.array_equal(g, w), 'planted second best'\n"
        "assert (want[2] <= 2).mean() > 0.5, 'most second-best distances are the planted ones'\n"
        "desc3 = np.zeros((2, ndb, 32), np.uint8); desc3[0] = db; desc3[1, :nq] = q; counts3 = np.array([ndb, nq], np.int32)\n"
        "d_desc = hiprt.DevBuf.from_numpy(desc3); d_cnt = hiprt.DevBuf.from_numpy(counts3)\n"
        "d_bi, d_bd, d_sd = (hiprt.DevBuf(2 * ndb * 4) for _ in range(3))\n"
        "check(ex._L.orbhip_hamming_knn2_seq_device(ex.handle, d_desc.ptr, d_cnt.ptr, ndb, 2, 1, d_bi.ptr, d_bd.ptr, d_sd.ptr), ex.handle)\n"
        "ex.sync()\n"
        "assert np.array_equal(d_bi.to_numpy(np.int32, (2, ndb))[1, :nq], want[0]) and np.array_equal(d_bd.to_numpy(np.int32, (2, ndb))[1, :nq], want[1]) and np.array_equal(d_sd.to_numpy(np.int32, (2, ndb))[1, :nq], want[2])\n"
        "ex.close()\n"
        "print('knn2 ok')\n"
    ) % (os.path.join(root, "vi-orb-slam-icra2018_amd"), os.path.join(root, "oracle"), os.path.join(root, "tests"))
    out = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, ORBHIP_KNN2_MFMA=mfma), capture_output=True, text=True)
    assert out.returncode == 0 and "knn2 ok" in out.stdout, out.stdout + out.stderr
