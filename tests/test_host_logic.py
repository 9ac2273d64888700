"""CPU tests of the product's host-side logic (no GPU): the C-ABI library loads and exports every
declared symbol, the device quadtree formulation (serial emulation) equals the list-based oracle,
the device cos/sin sequence equals libm on the whole angle domain."""
import ctypes as C
import os
import re
import struct
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
NATIVE = os.path.join(ROOT, "tests", "native")


@pytest.fixture(scope="module")
def native():
    subprocess.check_call(["make", "-C", NATIVE, "all"], stdout=subprocess.DEVNULL)
    qt = C.CDLL(os.path.join(NATIVE, "libqt_emul.so"))
    qt.qt_emul_distribute.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int]
    qt.qt_emul_distribute_first_counted.argtypes = qt.qt_emul_distribute.argtypes
    tr = C.CDLL(os.path.join(NATIVE, "libtrig_host.so"))
    tr.trig_host_sweep.restype = C.c_long
    tr.trig_host_sweep.argtypes = [C.c_uint32, C.c_uint32, C.c_uint32, C.POINTER(C.c_float)]
    tr.trig_host_sincos.argtypes = [C.c_float, C.POINTER(C.c_float), C.POINTER(C.c_float)]
    return qt, tr


def test_library_exports_every_declared_symbol():
    from orbhip import capi
    if not os.path.exists(capi.LIB_PATH):
        import __graft_entry__
        __graft_entry__.build()
    header = open(os.path.join(ROOT, "include", "orbhip.h")).read()
    declared = sorted(set(re.findall(r"\b(orbhip_[a-z0-9_]+)\s*\(", header)))
    assert declared == sorted(capi.SYMBOLS)
    assert sorted(capi.exported_symbols()) == sorted(capi.SYMBOLS)


def test_no_gpu_means_loud_failure_not_fallback():
    """Without a HIP device orbhip_create must fail with a reason (never a CPU path)."""
    from orbhip import capi
    L = capi.load()
    if L.orbhip_device_count() > 0:
        pytest.skip("a GPU is present")
    assert not L.orbhip_create(0, 1000, 1.2, 8, 20, 7, 640, 480, 1)
    assert "no HIP device" in capi.last_error(None)
    from orbhip.extractor import ORBextractor
    with pytest.raises(capi.OrbHipError):
        ORBextractor(max_w=640, max_h=480)


def _pack(c):
    return (c["x"].astype(np.uint32) | (c["y"].astype(np.uint32) << 12) | (c["score"].astype(np.uint32) << 24))


def test_device_quadtree_formulation_equals_list_oracle(native, oracle):
    qt, _ = native
    rng = np.random.default_rng(1)
    checked = 0
    for trial in range(250):
        w, h = int(rng.integers(60, 1300)), int(rng.integers(60, 500))
        if round(w / h) < 1:
            continue
        n = int(rng.integers(1, 3000))
        xs, ys = rng.integers(0, w, n), rng.integers(0, h, n)
        if trial % 3 == 0:
            xs = np.clip(rng.normal(w / 2, w / 12, n).astype(int), 0, w - 1)
            ys = np.clip(rng.normal(h / 2, h / 12, n).astype(int), 0, h - 1)
        pts = sorted(set(zip(ys.tolist(), xs.tolist())))
        sc = rng.integers(7, 120, len(pts))
        c = np.array([(x, y, s) for (y, x), s in zip(pts, sc)], dtype=oracle.CAND_DTYPE)
        N = int(rng.integers(1, 900))
        want = _pack(c)[oracle.distribute_octtree(c, w, h, N)]
        out = np.zeros(N + 64, np.uint32)
        packed = np.ascontiguousarray(_pack(c))
        S = qt.qt_emul_distribute(packed.ctypes.data, len(c), w, h, N, out.ctypes.data, len(out))
        assert S == len(want) and np.array_equal(out[:S], want), (trial, w, h, len(c), N)
        # ... and with the first pass labelled and counted by the caller, as k_quadtree's batch gather does for a single root (r06)
        out[:] = 0
        S = qt.qt_emul_distribute_first_counted(packed.ctypes.data, len(c), w, h, N, out.ctypes.data, len(out))
        assert S == len(want) and np.array_equal(out[:S], want), ("first counted", trial, w, h, len(c), N)
        checked += 1
    assert checked > 150


def test_device_quadtree_on_real_candidates(native, oracle):
    from orbhip import synth
    qt, _ = native
    img = synth.make_frames(11, 752, 480, 1)[0]
    ex = oracle.Extractor(1000)
    ex(img)
    for l in range(8):
        c = ex.level_cands(l)
        lvl = ex.pyramid(l)
        N = ex.params.mnFeaturesPerLevel[l]
        regw, regh = lvl.shape[1] - 32, lvl.shape[0] - 32
        want = _pack(c)[oracle.distribute_octtree(c, regw, regh, N)]
        out = np.zeros(N + 64, np.uint32)
        packed = np.ascontiguousarray(_pack(c))
        S = qt.qt_emul_distribute(packed.ctypes.data, len(c), regw, regh, N, out.ctypes.data, len(out))
        assert S == len(want) and np.array_equal(out[:S], want)


def test_device_trig_sequence_equals_libm_exhaustively(native):
    """Every float in [0, 2*pi*1.01] (1.09e9 values, ~5 s): the double-precision sequence the
    kernel evaluates gives the same bits as this machine's cosf/sinf (what the reference calls,
    src/ORBextractor.cc:115)."""
    _, tr = native
    hi = struct.unpack("<I", struct.pack("<f", 6.2831855 * 1.01))[0]
    bad = C.c_float()
    assert tr.trig_host_sweep(0, hi, 1, C.byref(bad)) == 0, bad.value


def test_device_trig_known_values(native):
    _, tr = native
    s, c = C.c_float(), C.c_float()
    tr.trig_host_sincos(0.0, C.byref(s), C.byref(c))
    assert (s.value, c.value) == (0.0, 1.0)
    for deg in range(0, 361, 15):
        rad = np.float32(deg) * np.float32(np.pi / np.float32(180))
        tr.trig_host_sincos(rad, C.byref(s), C.byref(c))
        assert abs(s.value - np.sin(np.float64(rad))) < 1e-7 and abs(c.value - np.cos(np.float64(rad))) < 1e-7


def test_synth_frames_are_deterministic_and_textured():
    from orbhip import synth
    a = synth.make_frames(5, 320, 240, 2)
    b = synth.make_frames(5, 320, 240, 2)
    assert np.array_equal(a, b) and a.dtype == np.uint8 and a.shape == (2, 240, 320)
    assert a.std() > 20 and not np.array_equal(a[0], a[1])


def test_kernel_index_division_tricks_are_exact():
    """The kernels split flat indices with floor((n + 0.5) * (1.0f / d)) (float32) or a 16-bit reciprocal; every use must be
    exact over the whole range it can see (a 20-bit reciprocal in k_fast's fallback once was not: memory fault found by the
    soak).  Ranges: k_fast items / dword groups per row, staging chunks, fallback pixels / run width; k_resize staging chunks;
    k_fast cell-of-column and row-of-(cell,row) with 16-bit reciprocals; k_describe idx / 10."""
    f32 = np.float32

    def float_trick(maxd, maxn):
        for d in range(1, maxd + 1):
            n = np.arange(0, maxn, dtype=np.int64)
            r = ((n.astype(np.float32) + f32(0.5)) * (f32(1.0) / f32(d))).astype(np.int32)
            assert np.array_equal(r, n // d), d
    float_trick(130, 70 * 130)      # item / GPR
    float_trick(24, 72 * 24)        # i / nchunk
    float_trick(330, 66 * 330)      # px / TW
    float_trick(16, 1024)           # k_resize: i / nch
    for wcell in range(20, 100):    # cellMagic: column -> cell, columns < 400
        m = 65536 // wcell + 1
        c = np.arange(400)
        assert np.array_equal((c * m) >> 16, c // wcell)
    for dh in range(1, 70):         # dhMagic: (cell, row) index -> cell, up to 8 cells
        m = 65536 // dh + 1
        i = np.arange(8 * dh)
        assert np.array_equal((i * m) >> 16, i // dh)
    for spw in range(5, 81):        # spwMagic: score words per row
        m = (1 << 20) // spw + 1
        i = np.arange(66 * spw)
        assert np.array_equal((i * m) >> 20, i // spw)
    idx = np.arange(384)
    assert np.array_equal((idx * 6554) >> 16, idx // 10)
    for gpr in range(1, 129):       # grpMagic (k_fast_fix): thread -> row segment, tid // groups per row
        m = 65536 // gpr + 1
        t = np.arange(256)
        assert np.array_equal((t * m) >> 16, t // gpr)
    lane = np.arange(64)            # k_blur's LDS-DMA: lane // 13
    assert np.array_equal((lane * 5) >> 6, lane // 13)


def test_lerp_compass_formulas_are_exact_for_every_q_v_t():
    """csrc/k_fast.hip compass4: the bytewise compare built from v_lerp_u8 ((a + b + (c & 1)) >> 1 per byte).  Emulated
    here for every pixel pair and every threshold 1..254: bright <=> q - v > t, dark <=> q - v < -t, bit for bit."""
    q, v = np.meshgrid(np.arange(256, dtype=np.int64), np.arange(256, dtype=np.int64), indexing="ij")

    def lerp(a, b, c):
        return (a + b + (c & 1)) >> 1

    for t in range(1, 255):
        c, c2 = t & 1, (t & 1) ^ 1
        KB = ((t + c) >> 1) + 128
        KD = (253 + c2 - t) >> 1
        assert 128 <= KB <= 255 and 0 <= KD <= 127
        nv = 255 - v
        m = lerp(q, nv, c)
        assert m.min() >= 0 and m.max() <= 255
        bright = lerp(m, 255 - KB, 1) >= 128
        m2 = lerp(q, nv, c2)
        notdark = lerp(m2, 255 - (KD + 1), 1) >= 128
        assert np.array_equal(bright, q - v > t), t
        assert np.array_equal(~notdark, q - v < -t), t


def test_loose_compass_formulas_are_exact_for_bright_and_one_value_wide_for_dark():
    """csrc/k_fast.hip compass4_loose (the fixed-layout kernel): ONE halving per neighbour, exact for bright; the dark cut may
    only ADD pixels (the work list is scored exactly afterwards) and adds exactly q - v == -t."""
    q, v = np.meshgrid(np.arange(256, dtype=np.int64), np.arange(256, dtype=np.int64), indexing="ij")

    def lerp(a, b, c):
        return (a + b + (c & 1)) >> 1

    for t in range(1, 255):
        c = t & 1
        KB = ((t + c) >> 1) + 128
        KD = (254 + c - t) >> 1
        assert 128 <= KB <= 255 and 0 <= KD <= 127
        a = lerp(q, 255 - v, c)
        bright = lerp(a, 255 - KB, 1) >= 128
        notdark = lerp(a, 255 - (KD + 1), 1) >= 128
        assert np.array_equal(bright, q - v > t), t
        assert np.array_equal(~notdark, q - v <= -t), t


def test_describe_rounding_and_resize_multiply_tricks_are_exact():
    """k_describe: cvRound of a rotated pattern coordinate (|x| <= 18.4) is read from the low mantissa bits of x + 1.5 * 2^23, and
    the LDS offset 40 * row + column is one v_mad_i32_i24 of the two bit patterns (its multiplicand is the low 24 bits, sign-extended)
    minus a constant.  k_resize: (b * (r >> 4)) >> 16 is the high word of the 24-bit product (b << 12) * (r & ~15)."""
    rng = np.random.default_rng(5)
    x = np.concatenate([rng.uniform(-18.5, 18.5, 2_000_000).astype(np.float32),
                        (np.arange(-37, 38) * 0.5).astype(np.float32)])
    x = np.concatenate([x, np.nextafter(x[-75:], np.float32(100)), np.nextafter(x[-75:], np.float32(-100))])
    bits = (x + np.float32(12582912.0)).view(np.int32).astype(np.int64)
    want = np.rint(x).astype(np.int64)                     # round half to even, as cvRound / __float2int_rn
    assert np.array_equal(bits - 0x4B400000, want)
    r, c = bits[: len(bits) // 2], bits[len(bits) // 2: 2 * (len(bits) // 2)]
    low24 = ((r & 0xFFFFFF) ^ 0x800000) - 0x800000          # the operand v_mad_i32_i24 sees
    off = (low24 * 40 + c - (0x400000 * 40 + 0x4B400000)) & 0xFFFFFFFF
    off = (off ^ 0x80000000) - 0x80000000
    assert np.array_equal(off, want[: len(r)] * 40 + want[len(r): 2 * len(r)])

    b = np.arange(0, 2049, dtype=np.int64)[:, None]
    rr = np.concatenate([rng.integers(0, 255 * 2048 + 1, 4000), [0, 15, 16, 255 * 2048]]).astype(np.int64)[None, :]
    f0, f1 = b << 12, rr & 0x7FFFF0
    assert f0.max() < 1 << 24 and f1.max() < 1 << 24
    assert np.array_equal((f0 * f1) >> 32, (b * (rr >> 4)) >> 16)


def test_dropin_sources_compile_against_opencv_declarations():
    """include/orbhip/cvlite.h has two branches: its own minimal cv:: types (what both boxes use: no OpenCV installed) and
    `#ifdef ORBHIP_USE_OPENCV` -> <opencv2/core/core.hpp>, the branch a build inside the reference tree takes.  That branch is
    syntax-checked here against tests/native/opencv_stub (declarations of the few OpenCV 2.4 names used, written from the API
    and labelled as a stub -- it pins nothing about OpenCV, it catches drift such as relying on headers cvlite.h happens to
    include or on Mat::step being a size_t)."""
    import glob
    import subprocess
    srcs = sorted(glob.glob(os.path.join(ROOT, "vi-orb-slam-icra2018_amd", "host", "*.cc")))
    assert len(srcs) >= 6
    for src in srcs:
        r = subprocess.run(["g++", "-std=c++11", "-fsyntax-only", "-DORBHIP_USE_OPENCV",
                            "-I" + os.path.join(ROOT, "tests", "native", "opencv_stub"), "-I" + os.path.join(ROOT, "include"),
                            "-I" + os.path.join(ROOT, "include", "orbhip"), src], capture_output=True, text=True)
        assert r.returncode == 0, os.path.basename(src) + ":\n" + r.stderr[:3000]
