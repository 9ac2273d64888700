"""Python mirror of the frame grid and the guided search:
Frame::AssignFeaturesToGrid / GetFeaturesInArea (src/Frame.cc:574-589, :671-724) and the two
ORBmatcher::SearchByProjection variants that run on it (src/ORBmatcher.cc:45-129, :1341-1498), and
ORBmatcher::SearchForInitialization (:405-520).
All arithmetic of the search runs in liborbhip; the helpers here only fill orbhip_proj_query records the way
the reference derives the window of a point (float32 throughout, as in the C++ code)."""
import ctypes as C

import numpy as np

from . import capi
from .capi import GRID_COLS, GRID_ROWS, KP_DTYPE, Q_ACTIVE, Q_OBSERVED, QUERY_DTYPE, _p, check

f32 = np.float32


def grid_params(min_x, max_x, min_y, max_y):
    """(mnMinX, mnMinY, mfGridElementWidthInv, mfGridElementHeightInv) -- src/Frame.cc:556-557."""
    return f32(min_x), f32(min_y), f32(GRID_COLS) / (f32(max_x) - f32(min_x)), f32(GRID_ROWS) / (f32(max_y) - f32(min_y))


def AssignFeaturesToGrid(ctx, kps_un, gp):
    """CSR (cell_off[3073], cell_idx) of mGrid; cell id = ix * 48 + iy."""
    kps_un = np.ascontiguousarray(kps_un, KP_DTYPE)
    off = np.zeros(GRID_COLS * GRID_ROWS + 1, np.int32)
    idx = np.zeros(max(len(kps_un), 1), np.int32)
    check(capi.load().orbhip_grid_build(ctx.handle, _p(kps_un), len(kps_un), gp[0], gp[1], gp[2], gp[3], _p(off), _p(idx)),
          ctx.handle, "orbhip_grid_build")
    return off, idx[:off[-1]].copy()


def GetFeaturesInArea(ctx, kps_un, gp, x, y, r, min_level=-1, max_level=-1):
    """Batch of windows: x, y, r, min_level, max_level broadcast against each other.  Returns (off, idx) CSR."""
    x, y, r, mn, mx = np.broadcast_arrays(np.asarray(x, f32), np.asarray(y, f32), np.asarray(r, f32),
                                          np.asarray(min_level, np.int32), np.asarray(max_level, np.int32))
    q = np.zeros(x.size, QUERY_DTYPE)
    q["u"], q["v"], q["radius"], q["min_level"], q["max_level"] = x.ravel(), y.ravel(), r.ravel(), mn.ravel(), mx.ravel()
    kps_un = np.ascontiguousarray(kps_un, KP_DTYPE)
    off = np.zeros(len(q) + 1, np.int32)
    cap = max(1, len(kps_un)) * max(1, len(q))
    cap = min(cap, 1 << 24)
    idx = np.zeros(cap, np.int32)
    check(capi.load().orbhip_features_in_area(ctx.handle, _p(kps_un), len(kps_un), gp[0], gp[1], gp[2], gp[3], _p(q), len(q),
                                              _p(off), _p(idx), cap), ctx.handle, "orbhip_features_in_area")
    return off, idx[:off[-1]].copy()


def queries_for_map_points(proj_x, proj_y, proj_xr, view_cos, level, in_view, observed, th, scale_factors):
    """Windows of SearchByProjection(Frame&, vector<MapPoint*>&, th): src/ORBmatcher.cc:51-69, :131-137."""
    n = len(proj_x)
    q = np.zeros(n, QUERY_DTYPE)
    r = np.where(np.asarray(view_cos, f32) > 0.998, f32(2.5), f32(4.0)).astype(f32)
    if float(th) != 1.0:
        r = r * f32(th)
    level = np.asarray(level, np.int32)
    q["u"], q["v"], q["proj_xr"] = proj_x, proj_y, proj_xr
    q["radius"] = r * np.asarray(scale_factors, f32)[level]
    q["min_level"], q["max_level"] = level - 1, level
    q["flags"] = np.where(in_view, Q_ACTIVE, 0) | np.where(observed, Q_OBSERVED, 0)
    return q


def queries_for_last_frame(u, v, ur, last_octave, angle, valid, observed, th, scale_factors, forward=False,
                           backward=False):
    """Windows of SearchByProjection(CurrentFrame, LastFrame, th, bMono): src/ORBmatcher.cc:1385-1407."""
    n = len(u)
    q = np.zeros(n, QUERY_DTYPE)
    o = np.asarray(last_octave, np.int32)
    q["u"], q["v"], q["proj_xr"], q["angle"] = u, v, ur, angle
    q["radius"] = f32(th) * np.asarray(scale_factors, f32)[o]
    if forward:
        q["min_level"], q["max_level"] = o, -1
    elif backward:
        q["min_level"], q["max_level"] = 0, o
    else:
        q["min_level"], q["max_level"] = o - 1, o + 1
    q["flags"] = np.where(valid, Q_ACTIVE, 0) | np.where(observed, Q_OBSERVED, 0)
    return q


def SearchByProjection(ctx, kps_un, desc, gp, queries, qdesc, u_right=None, occupied=None, use_ratio=True, nnratio=0.8,
                       check_ori=True, th_high=100):
    """(nmatches, match[feature] = query index or -1)."""
    kps_un = np.ascontiguousarray(kps_un, KP_DTYPE)
    desc = np.ascontiguousarray(desc, np.uint8).reshape(-1, 32)
    queries = np.ascontiguousarray(queries, QUERY_DTYPE)
    qdesc = np.ascontiguousarray(qdesc, np.uint8).reshape(-1, 32)
    ur = None if u_right is None else np.ascontiguousarray(u_right, f32)
    occ = None if occupied is None else np.ascontiguousarray(occupied, np.uint8)
    match = np.empty(max(len(kps_un), 1), np.int32)
    nm = C.c_int()
    check(capi.load().orbhip_search_by_projection(ctx.handle, _p(kps_un), _p(desc), len(kps_un), _p(ur), _p(occ), gp[0], gp[1],
                                                  gp[2], gp[3], _p(queries), _p(qdesc), len(queries), 1 if use_ratio else 0,
                                                  nnratio, 1 if check_ori else 0, th_high, _p(match), C.byref(nm)),
          ctx.handle, "orbhip_search_by_projection")
    return nm.value, match[:len(kps_un)].copy()


def ComputeDistinctiveDescriptors(ctx, desc, off):
    """MapPoint::ComputeDistinctiveDescriptors (src/MapPoint.cc:283-349) for the points whose observed descriptors are rows
    off[p]..off[p+1] of desc: (best row within each list or -1, its median)."""
    desc = np.ascontiguousarray(desc, np.uint8).reshape(-1, 32)
    off = np.ascontiguousarray(off, np.int32)
    P = len(off) - 1
    best = np.empty(max(P, 1), np.int32)
    med = np.empty(max(P, 1), np.int32)
    check(capi.load().orbhip_distinctive_descriptors(ctx.handle, _p(desc) if len(desc) else None, _p(off), P, _p(best), _p(med)),
          ctx.handle, "orbhip_distinctive_descriptors")
    return best[:P].copy(), med[:P].copy()


def WindowBest(ctx, kps_un, desc, gp, queries, qdesc, u_right=None, inv_level_sigma2=None):
    """Per-point best feature of a KeyFrame window, the inner loop of ORBmatcher::Fuse (src/ORBmatcher.cc:887-950; the
    chi-square gate when inv_level_sigma2 is given), Fuse(Scw) (:1044-1075) and SearchBySim3 (:1190-1224):
    (best_idx[nq], best_dist[nq]), -1 / 256 when none."""
    kps_un = np.ascontiguousarray(kps_un, KP_DTYPE)
    desc = np.ascontiguousarray(desc, np.uint8).reshape(-1, 32)
    queries = np.ascontiguousarray(queries, QUERY_DTYPE)
    qdesc = np.ascontiguousarray(qdesc, np.uint8).reshape(-1, 32)
    ur = None if u_right is None else np.ascontiguousarray(u_right, f32)
    sg = None if inv_level_sigma2 is None else np.ascontiguousarray(inv_level_sigma2, f32)
    bi = np.empty(max(len(queries), 1), np.int32)
    bd = np.empty(max(len(queries), 1), np.int32)
    check(capi.load().orbhip_window_best(ctx.handle, _p(kps_un), _p(desc), len(kps_un), _p(ur), _p(sg),
                                         0 if sg is None else len(sg), gp[0], gp[1], gp[2], gp[3], _p(queries), _p(qdesc),
                                         len(queries), _p(bi), _p(bd)),
          ctx.handle, "orbhip_window_best")
    return bi[:len(queries)].copy(), bd[:len(queries)].copy()


def WindowBestSet(ctx, key, queries, qdesc, u_right=None, inv_level_sigma2=None):
    """WindowBest into a resident set (orbhip_set_put with a grid): only the projected points travel."""
    queries = np.ascontiguousarray(queries, QUERY_DTYPE)
    qdesc = np.ascontiguousarray(qdesc, np.uint8).reshape(-1, 32)
    ur = None if u_right is None else np.ascontiguousarray(u_right, f32)
    sg = None if inv_level_sigma2 is None else np.ascontiguousarray(inv_level_sigma2, f32)
    bi = np.empty(max(len(queries), 1), np.int32)
    bd = np.empty(max(len(queries), 1), np.int32)
    check(capi.load().orbhip_window_best_set(ctx.handle, key, _p(ur), _p(sg), 0 if sg is None else len(sg), _p(queries), _p(qdesc),
                                             len(queries), _p(bi), _p(bd)), ctx.handle, "orbhip_window_best_set")
    return bi[:len(queries)].copy(), bd[:len(queries)].copy()


def SearchForInitialization(ctx, kps1_un, desc1, kps2_un, desc2, gp, prev_matched, window_size=100, nnratio=0.9,
                            check_ori=True):
    """ORBmatcher::SearchForInitialization (src/ORBmatcher.cc:405-520).  Returns (nmatches, vnMatches12,
    vbPrevMatched after the update of :512-515)."""
    kps1_un = np.ascontiguousarray(kps1_un, KP_DTYPE)
    kps2_un = np.ascontiguousarray(kps2_un, KP_DTYPE)
    desc1 = np.ascontiguousarray(desc1, np.uint8).reshape(-1, 32)
    desc2 = np.ascontiguousarray(desc2, np.uint8).reshape(-1, 32)
    prev = np.array(prev_matched, f32).reshape(-1, 2).copy()
    if len(prev) != len(kps1_un):
        raise ValueError("vbPrevMatched must hold one point per keypoint of frame 1")
    m12 = np.empty(max(len(kps1_un), 1), np.int32)
    nm = C.c_int()
    check(capi.load().orbhip_search_for_initialization(ctx.handle, _p(kps1_un), _p(desc1), len(kps1_un), _p(kps2_un), _p(desc2),
                                                       len(kps2_un), gp[0], gp[1], gp[2], gp[3], _p(prev), int(window_size),
                                                       nnratio, 1 if check_ori else 0, _p(m12), C.byref(nm)),
          ctx.handle, "orbhip_search_for_initialization")
    return nm.value, m12[:len(kps1_un)].copy(), prev


def SearchForTriangulation(ctx, kps1_un, desc1, skip1, groups1, kps2_un, desc2, skip2, groups2, F12, ex, ey, scale_factors2,
                           level_sigma2_2, u_right1=None, u_right2=None, only_stereo=False, check_ori=True):
    """ORBmatcher::SearchForTriangulation (src/ORBmatcher.cc:657-827).  groups = FeatureVector as (node ids, offsets,
    feature indices).  Returns (nmatches, vMatches12); vMatchedPairs = [(i, m) for i, m in enumerate(vMatches12) if m >= 0]."""
    kps1_un = np.ascontiguousarray(kps1_un, KP_DTYPE)
    kps2_un = np.ascontiguousarray(kps2_un, KP_DTYPE)
    desc1 = np.ascontiguousarray(desc1, np.uint8).reshape(-1, 32)
    desc2 = np.ascontiguousarray(desc2, np.uint8).reshape(-1, 32)
    skip1, skip2 = np.ascontiguousarray(skip1, np.uint8), np.ascontiguousarray(skip2, np.uint8)
    g1 = [np.ascontiguousarray(a, np.int32) for a in groups1]
    g2 = [np.ascontiguousarray(a, np.int32) for a in groups2]
    ur1 = None if u_right1 is None else np.ascontiguousarray(u_right1, f32)
    ur2 = None if u_right2 is None else np.ascontiguousarray(u_right2, f32)
    F = np.ascontiguousarray(F12, f32).reshape(9)
    sf = np.ascontiguousarray(scale_factors2, f32)
    s2 = np.ascontiguousarray(level_sigma2_2, f32)
    m12 = np.empty(max(len(kps1_un), 1), np.int32)
    nm = C.c_int()
    check(capi.load().orbhip_search_for_triangulation(ctx.handle, _p(kps1_un), _p(desc1), len(kps1_un), _p(skip1), _p(ur1),
                                                      _p(g1[0]), _p(g1[1]), _p(g1[2]), len(g1[0]), _p(kps2_un), _p(desc2),
                                                      len(kps2_un), _p(skip2), _p(ur2), _p(g2[0]), _p(g2[1]), _p(g2[2]),
                                                      len(g2[0]), _p(F), float(ex), float(ey), _p(sf), _p(s2), len(sf),
                                                      1 if only_stereo else 0, 1 if check_ori else 0, _p(m12), C.byref(nm)),
          ctx.handle, "orbhip_search_for_triangulation")
    return nm.value, m12[:len(kps1_un)].copy()
