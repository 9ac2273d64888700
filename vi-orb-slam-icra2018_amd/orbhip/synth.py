"""Seeded procedural workloads (there are no datasets on either box).

Scenes are sums of random rotated rectangles / discs over band-limited noise, so that FAST finds
far more corners than the per-level quota at iniThFAST=20 while flat regions only respond at
minThFAST=7.  Consecutive frames of a stream are the same scene under a slowly varying
similarity + shear warp, so descriptor matches exist between neighbouring frames.
"""
import numpy as np

GEOMETRY = {
    "euroc": (752, 480),   # Examples/Monocular/EuRoC.yaml (cam0 752x480)
    "tum": (640, 480),     # Examples/RGB-D/TUM1.yaml
    "kitti": (1241, 376),  # Examples/Stereo/KITTI00-02.yaml
}


def _smooth_noise(rng, h, w, cell, amp):
    gh, gw = h // cell + 3, w // cell + 3
    g = rng.standard_normal((gh, gw)).astype(np.float32)
    ys = np.arange(h, dtype=np.float32) / cell
    xs = np.arange(w, dtype=np.float32) / cell
    y0 = ys.astype(np.int32)
    x0 = xs.astype(np.int32)
    fy = (ys - y0)[:, None]
    fx = (xs - x0)[None, :]
    a = g[y0][:, x0]
    b = g[y0][:, x0 + 1]
    c = g[y0 + 1][:, x0]
    d = g[y0 + 1][:, x0 + 1]
    return amp * ((a * (1 - fx) + b * fx) * (1 - fy) + (c * (1 - fx) + d * fx) * fy)


def make_scene(seed, width, height, margin=96, nshapes=None):
    """A float32 canvas (height+2*margin, width+2*margin) in [0,255]."""
    rng = np.random.default_rng(seed)
    H, W = height + 2 * margin, width + 2 * margin
    img = np.full((H, W), 118.0, np.float32)
    img += _smooth_noise(rng, H, W, 64, 30.0)
    img += _smooth_noise(rng, H, W, 9, 6.0)
    if nshapes is None:
        nshapes = (H * W) // 1500
    for _ in range(nshapes):
        cx, cy = rng.uniform(0, W), rng.uniform(0, H)
        # leave roughly a quarter of the canvas with low contrast only
        if (int(cx) // 160 + int(cy) // 160) % 4 == 0:
            amp = rng.uniform(-14, 14)
        else:
            amp = rng.uniform(-110, 110)
        if rng.random() < 0.7:
            hw, hh = rng.uniform(4, 40), rng.uniform(4, 40)
            th = rng.uniform(0, np.pi)
            r = int(np.hypot(hw, hh)) + 2
            x0, x1 = max(0, int(cx) - r), min(W, int(cx) + r + 1)
            y0, y1 = max(0, int(cy) - r), min(H, int(cy) + r + 1)
            if x0 >= x1 or y0 >= y1:
                continue
            yy, xx = np.mgrid[y0:y1, x0:x1].astype(np.float32)
            u = (xx - cx) * np.cos(th) + (yy - cy) * np.sin(th)
            v = -(xx - cx) * np.sin(th) + (yy - cy) * np.cos(th)
            m = (np.abs(u) <= hw) & (np.abs(v) <= hh)
        else:
            rad = rng.uniform(3, 24)
            r = int(rad) + 2
            x0, x1 = max(0, int(cx) - r), min(W, int(cx) + r + 1)
            y0, y1 = max(0, int(cy) - r), min(H, int(cy) + r + 1)
            if x0 >= x1 or y0 >= y1:
                continue
            yy, xx = np.mgrid[y0:y1, x0:x1].astype(np.float32)
            m = (xx - cx) ** 2 + (yy - cy) ** 2 <= rad * rad
        img[y0:y1, x0:x1] += amp * m
    img += rng.standard_normal((H, W)).astype(np.float32) * 1.5
    return np.clip(img, 0, 255)


def warp_frame(scene, width, height, t, margin=96):
    """Frame t of a stream: bilinear sample of the scene under a small similarity + shear."""
    ang = 0.004 * t + 0.02 * np.sin(0.05 * t)
    sc = 1.0 + 0.02 * np.sin(0.031 * t)
    sh = 0.01 * np.sin(0.017 * t)
    tx = 12.0 * np.sin(0.043 * t) + 0.35 * (t % 40)
    ty = 9.0 * np.cos(0.029 * t)
    H, W = scene.shape
    cx, cy = width / 2.0, height / 2.0
    yy, xx = np.mgrid[0:height, 0:width].astype(np.float32)
    dx, dy = xx - cx, yy - cy
    c, s = np.cos(ang) * sc, np.sin(ang) * sc
    sx = c * dx - s * dy + sh * dy + cx + margin + tx
    sy = s * dx + c * dy + cy + margin + ty
    sx = np.clip(sx, 0, W - 2.001)
    sy = np.clip(sy, 0, H - 2.001)
    x0 = sx.astype(np.int32)
    y0 = sy.astype(np.int32)
    fx, fy = sx - x0, sy - y0
    a = scene[y0, x0]
    b = scene[y0, x0 + 1]
    cc = scene[y0 + 1, x0]
    d = scene[y0 + 1, x0 + 1]
    out = (a * (1 - fx) + b * fx) * (1 - fy) + (cc * (1 - fx) + d * fx) * fy
    return np.clip(np.rint(out), 0, 255).astype(np.uint8)


_POOL = None   # (scene, width, height) of make_frames_parallel's forked workers


def _pool_warp(t):
    scene, width, height = _POOL
    return warp_frame(scene, width, height, t)


def make_frames_parallel(seed, width, height, count, workers, start=0):
    """make_frames by a pool of FORKED workers (32 ms a frame on one core): the same frames.  For callers that have not touched
    the GPU yet (bench.py draws its 1024-frame batch this way before it imports torch)."""
    global _POOL
    workers = max(1, min(workers, count // 8))
    if workers == 1:
        return make_frames(seed, width, height, count, start)
    import multiprocessing as mp
    _POOL = (make_scene(seed, width, height), width, height)
    try:
        with mp.get_context("fork").Pool(workers) as pool:
            return np.stack(pool.map(_pool_warp, range(start, start + count), chunksize=max(1, count // (4 * workers))))
    finally:
        _POOL = None


def make_frames(seed, width, height, count, start=0):
    """`count` consecutive uint8 frames (count, height, width) of the stream `seed`."""
    scene = make_scene(seed, width, height)
    return np.stack([warp_frame(scene, width, height, start + t) for t in range(count)])


CONTENT_CLASSES = ("textured", "indoor_sparse", "white_noise", "low_contrast")


def load_photographs():
    """The camera photographs this image's Python packages install as sample data (scikit-learn: china.jpg, flower.jpg, 427 x 640;
    matplotlib: grace_hopper.jpg, 600 x 512), as grey uint8 arrays at their own sizes -- the only real pictures available here
    (no network, no dataset).  Read where they lie at run time, never copied; [] when the packages or PIL are absent.
    Grey = (4899 R + 9617 G + 1868 B + 8192) >> 14, the fixed-point RGB -> grey of cv::cvtColor (ref: the grey conversion every
    frame goes through before the extractor sees it, src/Tracking.cc GrabImageMonocular)."""
    import os
    out = []
    try:
        from PIL import Image
    except Exception:
        return out
    paths = []
    try:
        import sklearn.datasets as skd
        paths += [os.path.join(os.path.dirname(skd.__file__), "images", n) for n in ("china.jpg", "flower.jpg")]
    except Exception:
        pass
    try:
        import matplotlib
        paths.append(os.path.join(os.path.dirname(matplotlib.__file__), "mpl-data", "sample_data", "grace_hopper.jpg"))
    except Exception:
        pass
    for p in paths:
        if not os.path.exists(p):
            continue
        rgb = np.asarray(Image.open(p).convert("RGB")).astype(np.int32)
        out.append(((4899 * rgb[:, :, 0] + 9617 * rgb[:, :, 1] + 1868 * rgb[:, :, 2] + 8192) >> 14).astype(np.uint8))
    return out


def photograph_frames(width, height, count):
    """`count` frames of width x height cut from the photographs: each photograph mirrored out to cover the frame, then windows of
    it moved by a few pixels per frame (and flipped every other round) so that the frames differ.  None without photographs."""
    photos = load_photographs()
    if not photos:
        return None
    big = []
    for g in photos:
        ry = -(-(height + 64) // g.shape[0]) + 1
        rx = -(-(width + 64) // g.shape[1]) + 1
        row = np.concatenate([g if i % 2 == 0 else g[:, ::-1] for i in range(rx)], axis=1)
        big.append(np.concatenate([row if i % 2 == 0 else row[::-1] for i in range(ry)], axis=0))
    frames = []
    for i in range(count):
        b = big[i % len(big)]
        r = i // len(big)
        y0, x0 = (7 * r) % 64, (11 * r) % 64
        f = b[y0:y0 + height, x0:x0 + width]
        frames.append(f[:, ::-1] if (r & 1) else f)
    return np.ascontiguousarray(np.stack(frames))


def make_frames_class(kind, seed, width, height, count):
    """`count` frames of one of the content classes bench.py reports (FAST's cost depends on what it looks at):
      textured       make_frames: shapes of every contrast over band-limited noise (the headline's frames)
      indoor_sparse  large flat regions (walls, floor) of slowly varying brightness with sensor noise of 1-2 levels, a dozen
                     objects with hard edges and a few posters of fine texture: most cells find nothing at iniThFAST and many
                     find nothing at minThFAST either
      white_noise    independent uniform pixels: most pixels pass the compass test and many are corners (the dense fallbacks of
                     the FAST kernel's lists)
      low_contrast   the textured scene squeezed to a quarter of its range around mid-grey: the second FAST pass (minThFAST)
                     does most of the work"""
    if kind == "textured":
        return make_frames(seed, width, height, count)
    rng = np.random.default_rng(seed)
    if kind == "white_noise":
        return rng.integers(0, 256, (count, height, width), dtype=np.uint8)
    if kind == "low_contrast":
        f = make_frames(seed, width, height, count).astype(np.float32)
        return np.clip(np.rint((f - 118.0) * 0.25 + 118.0), 0, 255).astype(np.uint8)
    if kind == "indoor_sparse":
        margin = 96
        H, W = height + 2 * margin, width + 2 * margin
        img = np.full((H, W), 135.0, np.float32) + _smooth_noise(rng, H, W, 200, 22.0)
        yy, xx = np.mgrid[0:H, 0:W].astype(np.float32)
        img[yy > 0.62 * H + 0.05 * xx] -= 40.0                                     # the floor
        for _ in range(14):                                                         # furniture: flat boxes with hard edges
            cx, cy, hw, hh = rng.uniform(0, W), rng.uniform(0, H), rng.uniform(20, 90), rng.uniform(20, 110)
            th = rng.uniform(-0.2, 0.2)
            u = (xx - cx) * np.cos(th) + (yy - cy) * np.sin(th)
            v = -(xx - cx) * np.sin(th) + (yy - cy) * np.cos(th)
            img[(np.abs(u) <= hw) & (np.abs(v) <= hh)] = rng.uniform(40, 220)
        for _ in range(4):                                                          # posters: fine texture
            x0, y0 = int(rng.uniform(0, W - 120)), int(rng.uniform(0, H - 90))
            img[y0:y0 + 90, x0:x0 + 120] = 128 + _smooth_noise(rng, 90, 120, 4, 60.0)
        img += rng.standard_normal((H, W)).astype(np.float32) * 1.2
        scene = np.clip(img, 0, 255)
        return np.stack([warp_frame(scene, width, height, t) for t in range(count)])
    raise ValueError("unknown content class %r" % (kind,))


def make_descriptor_db(seed, n):
    """Random 256-bit descriptors (n, 32) uint8 -- SURVEY 8d config 5 database."""
    rng = np.random.default_rng(seed)
    return rng.integers(0, 256, (n, 32), dtype=np.uint8)


def make_queries(seed, db, nq, max_flips=40):
    """Queries = database rows with k in [0, max_flips] random bit flips (non-trivial
    best/second structure).  Returns (queries, source_row)."""
    rng = np.random.default_rng(seed)
    rows = rng.integers(0, len(db), nq)
    q = db[rows].copy()
    for i in range(nq):
        k = int(rng.integers(0, max_flips + 1))
        bits = rng.choice(256, size=k, replace=False)
        for b in bits:
            q[i, b >> 3] ^= np.uint8(1 << (b & 7))
    return q, rows


def make_stereo_pair(seed, width, height, disparity=17, t=0):
    """Rectified stereo pair of one scene: the right camera sees every point `disparity` pixels further
    left (constant depth plane), so row-aligned matches exist (BASELINE config 3 geometry)."""
    scene = make_scene(seed, width + disparity + 64, height)
    big = warp_frame(scene, width + disparity + 32, height, t)
    return big[:, 16:16 + width].copy(), big[:, 16 + disparity:16 + disparity + width].copy()
