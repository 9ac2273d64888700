#!/usr/bin/env python3
"""Measures the BASELINE.json configurations other than the headline one on ONE MI355X and prints a
markdown table (committed under profiles/).  Parity for these geometries is covered by
tests/test_gpu_parity.py; this script only times them.  Synthetic frames (orbhip/synth.py)."""
import ctypes as C
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "vi-orb-slam-icra2018_amd"))
import numpy as np  # noqa: E402
import torch  # noqa: E402  (device buffers only)

from orbhip import distributed as D, synth  # noqa: E402
from orbhip.extractor import ORBextractor  # noqa: E402
from orbhip.vocabulary import ORBVocabulary  # noqa: E402


def run_extract(W, H, nfeat, B, total_frames, match, uniq=16, seed=0):
    frames_u = synth.make_frames(seed, W, H, min(uniq, B))
    frames = np.concatenate([frames_u] * ((B + len(frames_u) - 1) // len(frames_u)))[:B]
    stride = (W + 15) // 16 * 16
    host = np.zeros((B, H, stride), np.uint8)
    host[:, :, :W] = frames
    d_img = torch.from_numpy(host).cuda()
    ex = ORBextractor(nfeat, 1.2, 8, 20, 7, max_w=W, max_h=H, max_batch=B)
    cap = ex.cap
    i32 = dict(dtype=torch.int32, device="cuda")
    d_kps = torch.empty((B, cap, 7), **i32)
    d_desc = torch.empty((B, cap, 32), dtype=torch.uint8, device="cuda")
    d_cnt = torch.zeros(B, **i32)
    d_a, d_b, d_c, d_d = (torch.empty((B, cap), **i32) for _ in range(4))
    d_wt = torch.empty((B, cap), dtype=torch.float32, device="cuda")
    d_nm = torch.zeros(B, **i32)
    L = ex._L
    if match == "bow":
        ORBVocabulary(ex).loadFromBinaryBlob(D.make_synthetic_vocabulary(4242, 10, 6))
    if match == "proj":
        # tracking with the motion model: frame b-1's keypoints are the projected points of frame b
        # (constant-position guess), th = 15 (mono), levels +-1 -- src/Tracking.cc TrackWithMotionModel
        from orbhip import guided
        gp = guided.grid_params(0, W, 0, H)
        d_off = torch.empty((B, 64 * 48 + 1), **i32)
        d_idx = torch.empty((B, cap), **i32)
        d_q = torch.zeros((B, cap, 8), **i32)
        d_qd = torch.zeros((B, cap, 32), dtype=torch.uint8, device="cuda")
        d_nq = torch.zeros(B, **i32)
        sf = torch.tensor([float(np.float32(1.2) ** l) for l in range(8)], dtype=torch.float32, device="cuda")

        def build_queries():
            k = d_kps.roll(1, 0)
            octv = k[:, :, 5].clamp(0, 7).long()
            d_q[:, :, 0] = k[:, :, 0]
            d_q[:, :, 1] = k[:, :, 1]
            d_q[:, :, 2] = (15.0 * sf[octv]).view(torch.int32)
            d_q[:, :, 4] = k[:, :, 5] - 1
            d_q[:, :, 5] = k[:, :, 5] + 1
            d_q[:, :, 6] = k[:, :, 3]
            d_q[:, :, 7] = 3
            d_qd.copy_(d_desc.roll(1, 0))
            d_nq.copy_(d_cnt.roll(1, 0))
            d_nq[0] = 0

    if match == "init":
        # monocular initialisation (src/Tracking.cc:1636-1671): frame b is the initial frame of pair b, frame b + 1 the
        # current one; mvbPrevMatched starts as the initial frame's keypoints and is carried from step to step
        from orbhip import guided
        gp = guided.grid_params(0, W, 0, H)
        d_off = torch.empty((B, 64 * 48 + 1), **i32)
        d_idx = torch.empty((B, cap), **i32)
        d_prev = torch.zeros((B, cap, 2), dtype=torch.float32, device="cuda")

    def step():
        ex.extract_batch_device(d_img.data_ptr(), B, W, H, stride, H * stride, d_kps.data_ptr(), d_desc.data_ptr(), cap,
                                d_cnt.data_ptr())
        if match == "init":
            assert L.orbhip_grid_build_device(ex.handle, d_kps.data_ptr() + cap * 28, d_cnt.data_ptr() + 4, cap, B - 1, gp[0], gp[1],
                                              gp[2], gp[3], d_off.data_ptr(), d_idx.data_ptr()) == 0
            assert L.orbhip_search_for_initialization_device(ex.handle, d_kps.data_ptr(), d_desc.data_ptr(), d_cnt.data_ptr(), cap,
                                                             d_kps.data_ptr() + cap * 28, d_desc.data_ptr() + cap * 32,
                                                             d_cnt.data_ptr() + 4, cap, B - 1, gp[0], gp[1], gp[2], gp[3],
                                                             d_off.data_ptr(), d_idx.data_ptr(), d_prev.data_ptr(), 100, 0.9, 1,
                                                             d_a.data_ptr(), d_nm.data_ptr()) == 0
        elif match == "bow":
            assert L.orbhip_vocab_transform_device(ex.handle, d_desc.data_ptr(), B * cap, 4, d_a.data_ptr(), d_wt.data_ptr(),
                                                   d_b.data_ptr()) == 0
            assert L.orbhip_search_by_bow_seq_device(ex.handle, d_desc.data_ptr(), d_kps.data_ptr(), d_cnt.data_ptr(),
                                                     d_b.data_ptr(), d_wt.data_ptr(), None, cap, B, 1, 0, C.c_float(0.7), 1,
                                                     d_c.data_ptr(), d_d.data_ptr(), d_nm.data_ptr()) == 0
        elif match == "proj":
            assert L.orbhip_grid_build_device(ex.handle, d_kps.data_ptr(), d_cnt.data_ptr(), cap, B, gp[0], gp[1], gp[2], gp[3],
                                              d_off.data_ptr(), d_idx.data_ptr()) == 0
            assert L.orbhip_search_by_projection_device(ex.handle, d_kps.data_ptr(), d_desc.data_ptr(), d_cnt.data_ptr(), cap, B,
                                                        None, None, gp[0], gp[1], gp[2], gp[3], d_off.data_ptr(),
                                                        d_idx.data_ptr(), d_q.data_ptr(), d_qd.data_ptr(), d_nq.data_ptr(), cap,
                                                        0, 0.9, 1, 100, d_a.data_ptr(), d_nm.data_ptr()) == 0
        elif match == "brute":
            assert L.orbhip_hamming_knn2_seq_device(ex.handle, d_desc.data_ptr(), d_cnt.data_ptr(), cap, B, 1, d_a.data_ptr(),
                                                    d_b.data_ptr(), d_c.data_ptr()) == 0
    torch.cuda.synchronize()
    if match == "proj":
        ex.extract_batch_device(d_img.data_ptr(), B, W, H, stride, H * stride, d_kps.data_ptr(), d_desc.data_ptr(), cap,
                                d_cnt.data_ptr())
        ex.sync()
        build_queries()
        torch.cuda.synchronize()
    if match == "init":
        ex.extract_batch_device(d_img.data_ptr(), B, W, H, stride, H * stride, d_kps.data_ptr(), d_desc.data_ptr(), cap,
                                d_cnt.data_ptr())
        ex.sync()
        d_prev.copy_(d_kps[:, :, 0:2].view(torch.float32))
        torch.cuda.synchronize()
    for _ in range(2):
        step()
    ex.sync()
    nsteps = max(1, (total_frames + B - 1) // B)
    t0 = time.perf_counter()
    for _ in range(nsteps):
        step()
    ex.sync()
    dt = time.perf_counter() - t0
    ms = (C.c_float * 6)()
    L.orbhip_get_stage_times(ex.handle, ms)
    kp = float(d_cnt.cpu().numpy().mean())
    if match == "proj":
        kp = float(d_nm.cpu().numpy()[1:].mean())      # matches per frame instead
    if match == "init":
        kp = float(d_nm.cpu().numpy()[:B - 1].mean())
    ex.close()
    return nsteps * B / dt, dt, kp, list(ms)


EUROC_K = [458.654, 0.0, 367.215, 0.0, 457.296, 248.375, 0.0, 0.0, 1.0]             # Examples/Stereo/EuRoC.yaml LEFT.*
EUROC_D = [-0.28340811, 0.07395907, 0.00019359, 1.76187114e-05, 0.0]
EUROC_R = [0.999966347530033, -0.001422739138722922, 0.008079580483432283, 0.001365741834644127, 0.9999741760894847,
           0.007055629199258132, -0.008089410156878961, -0.007044357138835809, 0.9999424675829176]
EUROC_P = [435.2046959714599, 0, 367.4517211914062, 0, 0, 435.2046959714599, 252.2008514404297, 0, 0, 0, 1, 0]


def run_stereo(W, H, nfeat, B, total_pairs, mb, mbf, uniq=8, rectify=False):
    """Config 3 end to end: (optionally cv::remap of both raw images,) left and right extraction on two
    contexts (two streams), then Frame::ComputeStereoMatches on the resident pyramids."""
    pairs = [synth.make_stereo_pair(300 + i, W, H, disparity=10 + 3 * i) for i in range(min(uniq, B))]
    stride = (W + 15) // 16 * 16
    host = np.zeros((2, B, H, stride), np.uint8)
    for b in range(B):
        host[0, b, :, :W], host[1, b, :, :W] = pairs[b % len(pairs)]
    d_img = torch.from_numpy(host).cuda()
    exs = [ORBextractor(nfeat, 1.2, 8, 20, 7, max_w=W, max_h=H, max_batch=B) for _ in range(2)]
    cap = exs[0].cap
    i32 = dict(dtype=torch.int32, device="cuda")
    rects = None
    if rectify:
        from orbhip import rectify as RC
        mx, my = RC.initUndistortRectifyMap(EUROC_K, EUROC_D, EUROC_R, EUROC_P, W, H)
        rects = [RC.Rectifier(e, mx, my) for e in exs]      # same maps for both eyes: timing only
        d_raw = d_img
        d_img = torch.empty_like(d_raw)
    d_kps = torch.empty((2, B, cap, 7), **i32)
    d_desc = torch.empty((2, B, cap, 32), dtype=torch.uint8, device="cuda")
    d_cnt = torch.zeros((2, B), **i32)
    d_u = torch.empty((B, cap), dtype=torch.float32, device="cuda")
    d_z = torch.empty_like(d_u)
    d_nm = torch.zeros(B, **i32)
    L = exs[0]._L

    def step():
        for s in range(2):
            if rects:
                rects[s].remap_device(d_raw[s].data_ptr(), B, W, H, stride, H * stride, d_img[s].data_ptr(), stride, H * stride)
            exs[s].extract_batch_device(d_img[s].data_ptr(), B, W, H, stride, H * stride, d_kps[s].data_ptr(),
                                        d_desc[s].data_ptr(), cap, d_cnt[s].data_ptr())
        assert L.orbhip_stereo_match_device(exs[0].handle, exs[1].handle, d_kps[0].data_ptr(), d_desc[0].data_ptr(),
                                            d_cnt[0].data_ptr(), d_kps[1].data_ptr(), d_desc[1].data_ptr(),
                                            d_cnt[1].data_ptr(), cap, B, mb, mbf, d_u.data_ptr(), d_z.data_ptr(),
                                            d_nm.data_ptr()) == 0
    torch.cuda.synchronize()
    for _ in range(2):
        step()
    exs[0].sync()
    nsteps = max(1, (total_pairs + B - 1) // B)
    t0 = time.perf_counter()
    for _ in range(nsteps):
        step()
    exs[0].sync()
    dt = time.perf_counter() - t0
    # time of the stereo stage alone
    t1 = time.perf_counter()
    for _ in range(5):
        L.orbhip_stereo_match_device(exs[0].handle, exs[1].handle, d_kps[0].data_ptr(), d_desc[0].data_ptr(),
                                     d_cnt[0].data_ptr(), d_kps[1].data_ptr(), d_desc[1].data_ptr(), d_cnt[1].data_ptr(), cap,
                                     B, mb, mbf, d_u.data_ptr(), d_z.data_ptr(), d_nm.data_ptr())
    exs[0].sync()
    st = (time.perf_counter() - t1) / 5
    good = float((d_u >= 0).sum().item()) / B
    for e in exs:
        e.close()
    return nsteps * B / dt, st * 1e3, good


def run_single_frame_latency(W, H, nfeat):
    img = synth.make_frames(5, W, H, 1)[0]
    ex = ORBextractor(nfeat, max_w=W, max_h=H)
    for _ in range(5):
        ex(img)
    t0 = time.perf_counter()
    n = 200
    for _ in range(n):
        ex(img)
    dt = (time.perf_counter() - t0) / n
    t = (ex.GetTimeOfComputePyramid(), ex.GetTimeOfComputeKeyPointsOctTree(), ex.GetTImeOfComputeDescriptor())
    ex.close()
    return dt * 1e3, t


def run_host_batch(W, H, nfeat, B):
    """orbhip_extract_batch: B frames through HOST pointers (H2D of the images, D2H of keypoints and descriptors)."""
    import ctypes as C
    from orbhip.capi import KP_DTYPE, _p
    imgs = np.ascontiguousarray(synth.make_frames(6, W, H, 8)[np.arange(B) % 8])
    ex = ORBextractor(nfeat, max_w=W, max_h=H, max_batch=B)
    ptrs = (C.c_void_p * B)(*[imgs[b].ctypes.data for b in range(B)])
    kps = np.zeros((B, ex.cap), KP_DTYPE)
    desc = np.zeros((B, ex.cap, 32), np.uint8)
    n = np.zeros(B, np.int32)

    def call():
        assert ex._L.orbhip_extract_batch(ex.handle, ptrs, B, W, H, W, _p(kps), _p(desc), ex.cap, _p(n)) == 0
    call()
    t0 = time.perf_counter()
    reps = 5
    for _ in range(reps):
        call()
    dt = (time.perf_counter() - t0) / reps
    ex.close()
    return B / dt, dt * 1e3, float(n.mean())


def run_fuse(W, H, nfeat, B, iters=30):
    """LocalMapping::SearchInNeighbors' matcher step for B key frames at once: the points seen by key frame b-1 projected into
    key frame b (constant-position guess + noise), th = 3, levels [predicted - 1, predicted], chi-square gate (mono):
    orbhip_grid_build_device + orbhip_window_best_device on resident data.  Returns (key frames/s, points/s, fused/frame)."""
    from orbhip import guided
    frames_u = synth.make_frames(5, W, H, 16)
    frames = np.concatenate([frames_u] * ((B + 15) // 16))[:B]
    stride = (W + 15) // 16 * 16
    host = np.zeros((B, H, stride), np.uint8)
    host[:, :, :W] = frames
    d_img = torch.from_numpy(host).cuda()
    ex = ORBextractor(nfeat, 1.2, 8, 20, 7, max_w=W, max_h=H, max_batch=B)
    cap = ex.cap
    i32 = dict(dtype=torch.int32, device="cuda")
    d_kps = torch.empty((B, cap, 7), **i32)
    d_desc = torch.empty((B, cap, 32), dtype=torch.uint8, device="cuda")
    d_cnt = torch.zeros(B, **i32)
    ex.extract_batch_device(d_img.data_ptr(), B, W, H, stride, H * stride, d_kps.data_ptr(), d_desc.data_ptr(), cap, d_cnt.data_ptr())
    ex.sync()
    gp = guided.grid_params(0, W, 0, H)
    d_off = torch.empty((B, 64 * 48 + 1), **i32)
    d_idx = torch.empty((B, cap), **i32)
    sf = torch.tensor([float(np.float32(1.2) ** l) for l in range(8)], dtype=torch.float32, device="cuda")
    k = d_kps.roll(1, 0)
    octv = k[:, :, 5].clamp(0, 7).long()
    d_q = torch.zeros((B, cap, 8), **i32)
    g = torch.Generator(device="cuda").manual_seed(1)
    d_q[:, :, 0] = (k[:, :, 0].view(torch.float32) + torch.randn((B, cap), device="cuda", generator=g)).view(torch.int32)
    d_q[:, :, 1] = (k[:, :, 1].view(torch.float32) + torch.randn((B, cap), device="cuda", generator=g)).view(torch.int32)
    d_q[:, :, 2] = (3.0 * sf[octv]).view(torch.int32)
    d_q[:, :, 4] = k[:, :, 5] - 1
    d_q[:, :, 5] = k[:, :, 5]
    d_q[:, :, 7] = 1
    d_qd = d_desc.roll(1, 0).contiguous()
    d_nq = d_cnt.roll(1, 0).contiguous()
    d_bi = torch.empty((B, cap), **i32)
    d_bd = torch.empty((B, cap), **i32)
    inv_s2 = (np.float32(1) / (np.float32(1.2) ** np.arange(8, dtype=np.float32)) ** 2).astype(np.float32)
    L = ex._L
    torch.cuda.synchronize()

    def step():
        assert L.orbhip_grid_build_device(ex.handle, d_kps.data_ptr(), d_cnt.data_ptr(), cap, B, gp[0], gp[1], gp[2], gp[3],
                                          d_off.data_ptr(), d_idx.data_ptr()) == 0
        assert L.orbhip_window_best_device(ex.handle, d_kps.data_ptr(), d_desc.data_ptr(), cap, B, None,
                                           inv_s2.ctypes.data_as(C.c_void_p), 8, gp[0], gp[1], gp[2], gp[3], d_off.data_ptr(),
                                           d_idx.data_ptr(), d_q.data_ptr(), d_qd.data_ptr(), d_nq.data_ptr(), cap,
                                           d_bi.data_ptr(), d_bd.data_ptr()) == 0
    for _ in range(3):
        step()
    ex.sync()
    t0 = time.perf_counter()
    for _ in range(iters):
        step()
    ex.sync()
    dt = time.perf_counter() - t0
    npts = int(d_nq.sum().item())
    fused = float((d_bd <= 50).sum().item()) / B
    ex.close()
    return iters * B / dt, iters * npts / dt, fused


def run_distinctive(P=100000, nmax=16, iters=30):
    """MapPoint::ComputeDistinctiveDescriptors for P map points with 1..nmax observations each, resident on the device."""
    rng = np.random.default_rng(12)
    cnt = rng.integers(1, nmax + 1, P)
    off = np.concatenate([[0], np.cumsum(cnt)]).astype(np.int32)
    desc = rng.integers(0, 256, (int(off[-1]), 32), dtype=np.uint8)
    ex = ORBextractor(500, 1.2, 8, 20, 7, max_w=320, max_h=240)
    d_desc, d_off = torch.from_numpy(desc).cuda(), torch.from_numpy(off).cuda()
    d_best = torch.empty(P, dtype=torch.int32, device="cuda")
    d_med = torch.empty(P, dtype=torch.int32, device="cuda")
    L = ex._L
    torch.cuda.synchronize()

    def step():
        assert L.orbhip_distinctive_descriptors_device(ex.handle, d_desc.data_ptr(), d_off.data_ptr(), P, d_best.data_ptr(),
                                                       d_med.data_ptr()) == 0
    for _ in range(3):
        step()
    ex.sync()
    t0 = time.perf_counter()
    for _ in range(iters):
        step()
    ex.sync()
    dt = (time.perf_counter() - t0) / iters
    ex.close()
    return P / dt, dt * 1e3, float(cnt.mean())


def run_big_knn(nq, ndb):
    db = torch.randint(0, 256, (ndb, 32), dtype=torch.uint8, device="cuda")
    idx = torch.randint(0, ndb, (nq,), device="cuda")
    q = db[idx].clone()
    q[:, 0] ^= 0x55
    ex = ORBextractor(1000, max_w=640, max_h=480)
    bi = torch.empty(nq, dtype=torch.int32, device="cuda")
    bd = torch.empty_like(bi)
    sd = torch.empty_like(bi)
    L = ex._L
    torch.cuda.synchronize()
    for _ in range(2):
        assert L.orbhip_hamming_knn2_device(ex.handle, q.data_ptr(), nq, db.data_ptr(), ndb, bi.data_ptr(), bd.data_ptr(),
                                            sd.data_ptr()) == 0
    ex.sync()
    t0 = time.perf_counter()
    n = 10
    for _ in range(n):
        L.orbhip_hamming_knn2_device(ex.handle, q.data_ptr(), nq, db.data_ptr(), ndb, bi.data_ptr(), bd.data_ptr(), sd.data_ptr())
    ex.sync()
    dt = (time.perf_counter() - t0) / n
    ok = bool((bi.cpu() == idx.cpu().int()).float().mean() > 0.99)
    ex.close()
    return dt, ok


def main():
    print("| config (BASELINE.json) | workload | frames/s | notes |")
    print("|---|---|---|---|")
    if len(sys.argv) > 1 and sys.argv[1] == "proj":            # profiling aid: the guided-search row only
        fps, dt, kp, ms = run_extract(640, 480, 1000, 512, 4096, "proj", seed=1)
        print("| 1b | proj | %.0f frames/s | %.1f matches per frame |" % (fps, kp))
        return
    if len(sys.argv) > 1 and sys.argv[1] == "single":
        lat, t = run_single_frame_latency(640, 480, 1000)
        print("| 1 | single frame | %.0f | %.3f ms per call; pyramid %.3f / keypoints %.3f / descriptors %.3f ms |" % (1e3 / lat, lat, t[0], t[1], t[2]))
        return
    if len(sys.argv) > 1 and sys.argv[1] == "host":
        fps, ms, kp = run_host_batch(640, 480, 1000, 256)
        print("| 1d | host batch | %.0f frames/s | %.2f ms per call, %.1f kp |" % (fps, ms, kp))
        return
    if len(sys.argv) > 1 and sys.argv[1] == "init":            # profiling aid
        fps, dt, kp, ms = run_extract(640, 480, 2000, 256, 2048, "init", seed=1)
        print("| 1c | init | %.0f frames/s | %.1f matches per pair |" % (fps, kp))
        return
    if len(sys.argv) > 1 and sys.argv[1] == "fuse":            # profiling aid
        kfs, pts, fused = run_fuse(640, 480, 1000, 256)
        print("| 1e | fuse | %.0f key frames/s | %.1f M points/s, %.1f fused per key frame |" % (kfs, pts / 1e6, fused))
        return
    if len(sys.argv) > 1 and sys.argv[1] == "distinctive":     # profiling aid
        pps, ms, avg = run_distinctive()
        print("| 1f | distinctive | %.1f M points/s | %.3f ms per 100000 points, %.1f observations per point |" % (pps / 1e6, ms, avg))
        return
    if len(sys.argv) > 1 and sys.argv[1] == "rectify":         # profiling aid
        pps, st, good = run_stereo(752, 480, 1200, 128, 1024, 0.11, 47.9, rectify=True)
        print("| 3c | rectify+stereo | %.0f pairs/s | stereo stage %.3f ms per 128 pairs, %.0f depth points per pair |" % (pps, st, good))
        return
    if len(sys.argv) > 1 and sys.argv[1] == "stereo":          # profiling aid: the stereo row only
        pps, st, good = run_stereo(1241, 376, 2000, 128, 1024, 0.53716, 386.1448)
        print("| 3b | stereo | %.0f pairs/s | stereo stage %.3f ms per 128 pairs, %.0f depth points per pair |" % (pps, st, good))
        return
    fps, dt, kp, ms = run_extract(752, 480, 1000, 512, 3682, "bow", seed=1)
    print("| 2: EuRoC MH_01 full sequence | 3682 frames 752x480, 1000 feat, batches of 512, extract + transform + SearchByBoW | %.0f | %.1f kp/frame, %.3f s for the sequence |" % (fps, kp, dt))
    fps, dt, kp, ms = run_extract(640, 480, 1000, 512, 4096, "proj", seed=1)
    print("| 1b: tracking front end (headline geometry) | 640x480, 1000 feat, batches of 512: extract + AssignFeaturesToGrid + SearchByProjection(last frame, th 15) | %.0f | %.1f matches per frame |" % (fps, kp))
    fps, dt, kp, ms = run_extract(640, 480, 2000, 256, 2048, "init", seed=1)
    print("| 1c: monocular initialisation (headline geometry) | 640x480, 2000 feat (mpIniORBextractor), batches of 256: extract + AssignFeaturesToGrid + SearchForInitialization(frame b, frame b+1, window 100) | %.0f | %.1f matches per pair |" % (fps, kp))
    fps, dt, kp, ms = run_extract(1241, 376, 2000, 256, 2048, "bow", seed=2)
    print("| 3: KITTI 00 stereo | 1241x376, 2000 feat, L+R images as 2 frames per pair, extract + transform + SearchByBoW | %.0f images/s = %.0f stereo pairs/s | %.1f kp/image |" % (fps, fps / 2, kp))
    pps, st, good = run_stereo(1241, 376, 2000, 128, 1024, 0.53716, 386.1448)
    print("| 3b: KITTI 00 stereo frame | 1241x376 pairs, 2000 feat: extract L + extract R (two contexts) + ComputeStereoMatches on the resident pyramids | %.0f stereo pairs/s | stereo stage %.3f ms per 128 pairs, %.0f depth points per pair |" % (pps, st, good))
    pps, st, good = run_stereo(752, 480, 1200, 128, 1024, 0.11, 47.9, rectify=True)
    print("| 3c: EuRoC stereo frame from raw images | 752x480 pairs, 1200 feat (Examples/Stereo/EuRoC.yaml): cv::remap L + R, extract L + R, ComputeStereoMatches | %.0f stereo pairs/s | %.0f depth points per pair |" % (pps, good))
    tot = 0.0
    frames = 0
    for i, n in enumerate([2912, 1710, 2280, 3040]):
        fps, dt, kp, ms = run_extract(752, 480, 1000, 512, n, "bow", seed=10 + i)
        tot += dt
        frames += (n + 511) // 512 * 512
    print("| 4: V101/V102/V201/MH02 streams | 4 streams 752x480 run back to back on ONE GPU (1 stream/GPU needs 4 GPUs) | %.0f | %d frames in %.3f s |" % (frames / tot, frames, tot))
    fps, dt, kp, ms = run_extract(640, 480, 4000, 256, 2048, "brute", seed=3)
    print("| 5a: TUM fr1_desk at 4000 feat | 640x480, 4000 feat, extract + brute-force match vs previous frame | %.0f | %.1f kp/frame |" % (fps, kp))
    dt, ok = run_big_knn(4000, 1000000)
    print("| 5b: 1M-descriptor relocalisation query | 4000 queries x 1 000 000 database rows, best/second | %.1f queries-batches/s | %.2f ms per query batch, %.2f T pair-evals/s, %.1f GB/s database stream, exact=%s |" % (
        1 / dt, dt * 1e3, 4000 * 1e6 / dt / 1e12, 32e6 / dt / 1e9, ok))
    dt, ok = run_big_knn(8, 1000000)
    print("| 5c: few-query regime | 8 queries x 1 000 000 rows | - | %.3f ms, %.1f GB/s database stream |" % (dt * 1e3, 32e6 / dt / 1e9))
    fps, ms, kp = run_host_batch(640, 480, 1000, 256)
    print("| 1d: batch through host pointers (incl. PCIe) | 256 frames 640x480, 1000 feat, orbhip_extract_batch from pageable host memory: H2D images, extraction, D2H keypoints + descriptors | %.0f | %.2f ms per 256-frame call, %.1f kp/frame |" % (fps, ms, kp))
    kfs, pts, fused = run_fuse(640, 480, 1000, 256)
    print("| 1e: Fuse window search (LocalMapping::SearchInNeighbors' matcher step) | 256 resident key frames 640x480, 1000 feat: AssignFeaturesToGrid + the points of key frame b-1 projected into key frame b, th 3, levels [l-1, l], chi-square gate (orbhip_window_best_device) | %.0f key frames/s | %.1f M points/s, %.1f fused per key frame |" % (kfs, pts / 1e6, fused))
    pps, ms, avg = run_distinctive()
    print("| 1f: ComputeDistinctiveDescriptors (LocalMapping, per point of a key frame) | 100000 resident map points with 1..16 observations (orbhip_distinctive_descriptors_device) | %.1f M points/s | %.3f ms per launch, %.1f observations per point |" % (pps / 1e6, ms, avg))
    lat, t = run_single_frame_latency(640, 480, 1000)
    print("| 1: single frame (host pointers, incl. PCIe) | 640x480, 1000 feat, orbhip_extract per call | %.0f | %.3f ms per call; device stage times pyramid %.3f / keypoints %.3f / descriptors %.3f ms |" % (1e3 / lat, lat, t[0], t[1], t[2]))


if __name__ == "__main__":
    main()
