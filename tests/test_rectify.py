"""Undistortion and rectification: the oracle against independent numpy models on
CPU; the HIP kernels and the host map builder against the oracle."""
import numpy as np
import pytest

# Examples/Stereo/EuRoC.yaml (LEFT.*): the calibration the reference's stereo driver reads
K_L = np.array([458.654, 0.0, 367.215, 0.0, 457.296, 248.375, 0.0, 0.0, 1.0]).reshape(3, 3)
D_L = np.array([-0.28340811, 0.07395907, 0.00019359, 1.76187114e-05, 0.0])
R_L = np.array([0.999966347530033, -0.001422739138722922, 0.008079580483432283, 0.001365741834644127,
                0.9999741760894847, 0.007055629199258132, -0.008089410156878961, -0.007044357138835809,
                0.9999424675829176]).reshape(3, 3)
P_L = np.array([435.2046959714599, 0, 367.4517211914062, 0, 0, 435.2046959714599, 252.2008514404297, 0, 0, 0, 1,
                0]).reshape(3, 4)


def _np_undistort(xy, K, D, P):
    """cvUndistortPoints, vectorised in float64 with the same operation order."""
    K = np.asarray(K, np.float32).astype(np.float64).reshape(3, 3)
    k = np.zeros(8)
    k[:len(D)] = np.asarray(D, np.float32).astype(np.float64)
    x = (xy[:, 0].astype(np.float64) - K[0, 2]) * (1.0 / K[0, 0])
    y = (xy[:, 1].astype(np.float64) - K[1, 2]) * (1.0 / K[1, 1])
    x0, y0 = x.copy(), y.copy()
    for _ in range(5 if len(D) else 1):
        r2 = x * x + y * y
        icdist = (1 + ((k[7] * r2 + k[6]) * r2 + k[5]) * r2) / (1 + ((k[4] * r2 + k[1]) * r2 + k[0]) * r2)
        dx = 2 * k[2] * x * y + k[3] * (r2 + 2 * x * x)
        dy = k[2] * (r2 + 2 * y * y) + 2 * k[3] * x * y
        x, y = (x0 - dx) * icdist, (y0 - dy) * icdist
    RR = np.eye(3) if P is None else np.asarray(P, np.float32).astype(np.float64).reshape(3, 3)
    xx = RR[0, 0] * x + RR[0, 1] * y + RR[0, 2]
    yy = RR[1, 0] * x + RR[1, 1] * y + RR[1, 2]
    ww = 1.0 / (RR[2, 0] * x + RR[2, 1] * y + RR[2, 2])
    return np.stack([(xx * ww).astype(np.float32), (yy * ww).astype(np.float32)], 1)


def _np_remap(src, mx, my):
    """cv::remap INTER_LINEAR / BORDER_CONSTANT 0 with floating weights rounded like the 15-bit table."""
    h, w = src.shape
    sx = np.rint(mx.astype(np.float32) * np.float32(32)).astype(np.int64)
    sy = np.rint(my.astype(np.float32) * np.float32(32)).astype(np.int64)
    ix, iy, fx, fy = sx >> 5, sy >> 5, sx & 31, sy & 31
    pad = np.zeros((h + 2, w + 2), np.int64)
    pad[1:-1, 1:-1] = src

    def tap(yy, xx):
        ok = (xx >= -1) & (xx <= w) & (yy >= -1) & (yy <= h)
        return np.where(ok, pad[np.clip(yy + 1, 0, h + 1), np.clip(xx + 1, 0, w + 1)], 0)
    acc = tap(iy, ix) * (32 - fx) * (32 - fy) + tap(iy, ix + 1) * fx * (32 - fy) + tap(iy + 1, ix) * (32 - fx) * fy + \
        tap(iy + 1, ix + 1) * fx * fy
    return ((acc * 32 + (1 << 14)) >> 15).astype(np.uint8)


def _np_init_map(K, D, R, P, w, h):
    M = np.zeros((3, 3))
    for r in range(3):
        for c in range(3):
            s = 0.0
            for k in range(3):
                s += P[r, k] * R[k, c]
            M[r, c] = s
    m = M
    det = m[0, 0] * (m[1, 1] * m[2, 2] - m[1, 2] * m[2, 1]) - m[0, 1] * (m[1, 0] * m[2, 2] - m[1, 2] * m[2, 0]) + \
        m[0, 2] * (m[1, 0] * m[2, 1] - m[1, 1] * m[2, 0])
    d = 1.0 / det
    ir = np.array([(m[1, 1] * m[2, 2] - m[1, 2] * m[2, 1]) * d, (m[0, 2] * m[2, 1] - m[0, 1] * m[2, 2]) * d,
                   (m[0, 1] * m[1, 2] - m[0, 2] * m[1, 1]) * d, (m[1, 2] * m[2, 0] - m[1, 0] * m[2, 2]) * d,
                   (m[0, 0] * m[2, 2] - m[0, 2] * m[2, 0]) * d, (m[0, 2] * m[1, 0] - m[0, 0] * m[1, 2]) * d,
                   (m[1, 0] * m[2, 1] - m[1, 1] * m[2, 0]) * d, (m[0, 1] * m[2, 0] - m[0, 0] * m[2, 1]) * d,
                   (m[0, 0] * m[1, 1] - m[0, 1] * m[1, 0]) * d])
    k = np.zeros(8)
    k[:len(D)] = D
    k1, k2, p1, p2, k3, k4, k5, k6 = k
    i = np.arange(h, dtype=np.float64)[:, None]

    def run(a, b, c):                       # _x = i*ir[b] + ir[c], then w-1 sequential additions of ir[a]
        first = i * ir[b] + ir[c]
        steps = np.full((h, w - 1), ir[a])
        return np.add.accumulate(np.concatenate([first, steps], 1), axis=1)
    _x, _y, _w = run(0, 1, 2), run(3, 4, 5), run(6, 7, 8)
    ww = 1.0 / _w
    x, y = _x * ww, _y * ww
    x2, y2 = x * x, y * y
    r2, _2xy = x2 + y2, 2 * x * y
    kr = (1 + ((k3 * r2 + k2) * r2 + k1) * r2) / (1 + ((k6 * r2 + k5) * r2 + k4) * r2)
    u = K[0, 0] * (x * kr + p1 * _2xy + p2 * (r2 + 2 * x2)) + K[0, 2]
    v = K[1, 1] * (y * kr + p1 * (r2 + 2 * y2) + p2 * _2xy) + K[1, 2]
    return u.astype(np.float32), v.astype(np.float32)


def test_oracle_undistort_points_matches_numpy_model(oracle):
    rng = np.random.default_rng(1)
    xy = np.stack([rng.uniform(0, 752, 500), rng.uniform(0, 480, 500)], 1).astype(np.float32)
    for D in (D_L[:4], D_L, np.array([-0.28, 0.07, 1e-4, 2e-5, 0.01, 0.001, -0.002, 0.0005]), np.zeros(0)):
        for P in (K_L, None):
            got = oracle.undistort_points(xy, K_L, D, P)
            assert got.tobytes() == _np_undistort(xy, K_L, D, P).tobytes()
    # undistorting and re-distorting returns to the start (five iterations: converged near the centre, within
    # a fraction of a pixel at the corners -- the same residual OpenCV leaves)
    u = oracle.undistort_points(xy, K_L, D_L[:4], None).astype(np.float64)
    k1, k2, p1, p2 = D_L[:4]
    r2 = (u ** 2).sum(1)
    xd = u[:, 0] * (1 + k1 * r2 + k2 * r2 * r2) + 2 * p1 * u[:, 0] * u[:, 1] + p2 * (r2 + 2 * u[:, 0] ** 2)
    yd = u[:, 1] * (1 + k1 * r2 + k2 * r2 * r2) + p1 * (r2 + 2 * u[:, 1] ** 2) + 2 * p2 * u[:, 0] * u[:, 1]
    back = np.stack([xd * K_L[0, 0] + K_L[0, 2], yd * K_L[1, 1] + K_L[1, 2]], 1)
    err = np.abs(back - xy).max(1)
    centre = np.hypot(xy[:, 0] - 376, xy[:, 1] - 240) < 200
    assert err[centre].max() < 0.01 and err.max() < 1.0


def test_oracle_rectify_map_and_remap_match_numpy_models(oracle):
    from orbhip import synth
    mx, my = oracle.init_undistort_rectify_map(K_L, D_L, R_L, P_L, 752, 480)
    rx, ry = _np_init_map(K_L, D_L, R_L, P_L[:, :3], 752, 480)
    assert mx.tobytes() == rx.tobytes() and my.tobytes() == ry.tobytes()
    img = synth.make_frames(5, 752, 480, 1)[0]
    assert np.array_equal(oracle.remap_linear(img, mx, my), _np_remap(img, mx, my))
    # identity map: exact copy except where the +1 taps fall outside (weight 0 there, so still exact)
    yy, xx = np.mgrid[0:480, 0:752].astype(np.float32)
    assert np.array_equal(oracle.remap_linear(img, xx, yy), img)
    # half-pixel shift: average of two neighbours, rounded half up
    sh = oracle.remap_linear(img, xx + np.float32(0.5), yy)
    exp = (img[:, :-1].astype(np.int32) + img[:, 1:] + 1) >> 1
    assert np.array_equal(sh[:, :-1], exp) and np.array_equal(sh[:, -1], (img[:, -1].astype(np.int32) + 1) >> 1)
    # maps that leave the image: constant border 0
    far = oracle.remap_linear(img, xx - 2000, yy + 3000)
    assert (far == 0).all()


@pytest.mark.gpu
def test_hip_undistort_matches_oracle(oracle):
    from orbhip import rectify, synth
    from orbhip.extractor import ORBextractor
    ex = ORBextractor(1000, max_w=752, max_h=480)
    k, d = ex(synth.make_frames(6, 752, 480, 1)[0])
    xy = np.stack([k["x"], k["y"]], 1)
    for D in (D_L[:4], D_L, np.array([-0.28, 0.07, 1e-4, 2e-5, 0.01, 0.001, -0.002, 0.0005])):
        un = rectify.UndistortKeyPoints(ex, k, K_L, D)
        ref = oracle.undistort_points(xy, K_L, D, K_L)
        assert un["x"].tobytes() == ref[:, 0].tobytes() and un["y"].tobytes() == ref[:, 1].tobytes()
        for f in ("size", "angle", "response", "octave", "class_id"):
            assert np.array_equal(un[f], k[f])
        assert rectify.undistort_points(ex, xy, K_L, D, None).tobytes() == oracle.undistort_points(xy, K_L, D, None).tobytes()
    assert rectify.UndistortKeyPoints(ex, k, K_L, np.zeros(4)).tobytes() == k.tobytes()      # the reference's shortcut
    b = rectify.ComputeImageBounds(ex, 752, 480, K_L, D_L[:4])
    c = oracle.undistort_points(np.array([[0, 0], [752, 0], [0, 480], [752, 480]], np.float32), K_L, D_L[:4], K_L)
    assert b == (min(c[0, 0], c[2, 0]), max(c[1, 0], c[3, 0]), min(c[0, 1], c[1, 1]), max(c[2, 1], c[3, 1]))
    assert rectify.ComputeImageBounds(ex, 752, 480, K_L, np.zeros(4)) == (0, 752, 0, 480)
    ex.close()


@pytest.mark.gpu
def test_hip_remap_matches_oracle(oracle):
    import hiprt
    from orbhip import rectify, synth
    from orbhip.extractor import ORBextractor
    ex = ORBextractor(500, max_w=320, max_h=240)
    mx, my = rectify.initUndistortRectifyMap(K_L, D_L, R_L, P_L, 752, 480)
    rx, ry = oracle.init_undistort_rectify_map(K_L, D_L, R_L, P_L, 752, 480)
    assert mx.tobytes() == rx.tobytes() and my.tobytes() == ry.tobytes()
    imgs = synth.make_frames(7, 752, 480, 3)
    rect = rectify.Rectifier(ex, mx, my)
    for im in imgs[:2]:
        assert np.array_equal(rect(im), oracle.remap_linear(im, mx, my))
    # batched, device resident, padded strides
    host = np.zeros((3, 480, 768), np.uint8)
    host[:, :, :752] = imgs
    d_src, d_dst = hiprt.DevBuf.from_numpy(host), hiprt.DevBuf(3 * 480 * 832)
    rect.remap_device(d_src.ptr, 3, 752, 480, 768, 480 * 768, d_dst.ptr, 832, 480 * 832)
    ex.sync()
    out = d_dst.to_numpy(np.uint8, (3, 480, 832))
    for b in range(3):
        assert np.array_equal(out[b, :, :752], oracle.remap_linear(imgs[b], mx, my))
    # odd sizes, maps that run off every edge, destination size different from the source
    rng = np.random.default_rng(8)
    src = rng.integers(0, 256, (97, 131), dtype=np.uint8)
    yy, xx = np.mgrid[0:75, 0:101].astype(np.float32)
    wx = (xx * np.float32(1.7) - 20 + rng.normal(0, 0.3, xx.shape)).astype(np.float32)
    wy = (yy * np.float32(1.6) - 12 + rng.normal(0, 0.3, xx.shape)).astype(np.float32)
    wx[0, :5] = [-1.0, -0.5, 130.0, 130.5, 131.0]
    wy[1, :5] = [-1.0, -0.25, 96.0, 96.75, 97.0]
    wx[2, 0], wy[2, 0] = 1e9, -1e9
    r2 = rectify.Rectifier(ex, wx, wy)
    ref2 = oracle.remap_linear(src, wx, wy)
    assert np.array_equal(r2(src), ref2)                     # rows of 131 bytes: the direct kernel
    padded = np.zeros((97, 144), np.uint8)
    padded[:, :131] = src
    padded[:, 131:] = 0xEE                                    # row padding must never leak into the result
    assert np.array_equal(r2(padded[:, :131]), ref2)         # 16-byte aligned rows: the tiled kernel, windows of
    #                                                          ~110 x 27 source pixels, every image edge crossed
    # a map whose tiles need more than the LDS window: the tiled kernel falls back per tile
    wx3 = (xx * np.float32(4.5) - 100).astype(np.float32)
    wy3 = (yy * np.float32(3.5) - 50).astype(np.float32)
    big = np.zeros((300, 512), np.uint8)
    big[:, :500] = rng.integers(0, 256, (300, 500), dtype=np.uint8)
    r3 = rectify.Rectifier(ex, wx3, wy3)
    assert np.array_equal(r3(big[:, :500]), oracle.remap_linear(np.ascontiguousarray(big[:, :500]), wx3, wy3))
    ex.close()
    d_src.free()
    d_dst.free()
