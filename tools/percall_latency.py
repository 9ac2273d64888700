"""Per-call latency of the host-pointer entry points at the reference's call granularity (ONE frame / ONE key-frame pair per
call, as Tracking / LocalMapping issue them): liborbhip on the GPU beside the oracle on one host core, same inputs.  Through
the Python wrappers (about 10-15 us of ctypes / numpy per call on both sides); tools/native/latency_dropin.cpp has the C++
view of the first three rows.  Prints a markdown table.   usage (on the GPU box): python tools/percall_latency.py"""
import os
import sys
import time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "vi-orb-slam-icra2018_amd"), os.path.join(ROOT, "oracle")]
import numpy as np
import orb_oracle_py as O
from orbhip import distributed as D
from orbhip import guided, synth
from orbhip.capi import QUERY_DTYPE
from orbhip.extractor import ORBextractor, ORBmatcher
from orbhip.vocabulary import ORBVocabulary


def bench(f, n):
    for _ in range(max(3, n // 10)):
        f()
    t = time.perf_counter()
    for _ in range(n):
        f()
    return (time.perf_counter() - t) / n * 1e3


W, H, NF = 640, 480, 1000
fr = synth.make_frames(5, W, H, 2)
ex = ORBextractor(NF, max_w=W, max_h=H)
ref = O.Extractor(NF)
(k0, d0), (k1, d1) = ex(fr[0]), ex(fr[1])
rng = np.random.default_rng(0)
rows = [("`ORBextractor::operator()` 640x480, 1000 features", bench(lambda: ex(fr[0]), 500), bench(lambda: ref(fr[0]), 10))]

blob = D.make_synthetic_vocabulary(52, k=10, L=6)
voc = ORBVocabulary(ex)
voc.loadFromBinaryBlob(blob)
ovoc = O.Vocabulary(blob)
rows.append(("`ORBVocabulary::transform` descent, 1000 descriptors (k 10, L 6, levelsup 4)", bench(lambda: voc.transform_raw(d1, 4), 300),
             bench(lambda: ovoc.transform(d1, 4), 20)))

fv = []
for d in (d0, d1):
    w, wt, nid = ovoc.transform(d, 4)
    fv.append(O.feature_vector(nid, wt))
M = ORBmatcher(0.7, True, ctx=ex)
v0 = np.ones(len(d0), np.uint8)
rows.append(("`SearchByBoW(KF, F)` 1000 x 1000 features", bench(lambda: M.SearchByBoW(d0, v0, k0["angle"], fv[0], d1, None, k1["angle"], fv[1]), 300),
             bench(lambda: O.search_by_bow(d0, v0, k0["angle"], fv[0], d1, None, k1["angle"], fv[1], th=50, th_mode=0, nnratio=0.7, check_ori=True), 30)))

# the same call between RESIDENT sets (orbhip_set_put once per key frame / frame; what the C++ drop-in does since r03)
M.put_set(1, k0, d0, fv[0])
M.put_set(2, k1, d1, fv[1])
rows.append(("`SearchByBoW(KF, F)` between resident sets (`orbhip_search_by_bow_sets`)",
             bench(lambda: M.SearchByBoW_sets(1, v0, len(k0), 2, None, len(k1)), 300), rows[-1][2]))

# SearchByProjection(CurrentFrame, LastFrame, th 15, mono): the points of the last frame near where they were
gp = guided.grid_params(0, W, 0, H)
sf = (np.float32(1.2) ** np.arange(8, dtype=np.float32)).astype(np.float32)
u = k0["x"] + rng.normal(0, 2, len(k0)).astype(np.float32)
v = k0["y"] + rng.normal(0, 2, len(k0)).astype(np.float32)
q = guided.queries_for_last_frame(u, v, np.full(len(k0), -1, np.float32), k0["octave"], k0["angle"], np.ones(len(k0), bool),
                                  np.zeros(len(k0), bool), 15, sf)
rows.append(("`SearchByProjection(CurrentFrame, LastFrame, 15)` 1000 points, 1000 features",
             bench(lambda: guided.SearchByProjection(ex, k1, d1, gp, q, d0, use_ratio=False, th_high=100), 300),
             bench(lambda: O.search_by_projection(k1, d1, gp, q, d0, use_ratio=False, th_high=100), 30)))

prev = np.stack([k0["x"], k0["y"]], 1).astype(np.float32)
rows.append(("`SearchForInitialization` window 100", bench(lambda: guided.SearchForInitialization(ex, k0, d0, k1, d1, gp, prev, 100), 200),
             bench(lambda: O.search_for_initialization(k0, d0, k1, d1, gp, prev, 100), 10)))

sig = (1 / sf ** 2).astype(np.float32)
qf = np.zeros(len(k0), QUERY_DTYPE)
qf["u"], qf["v"] = u, v
qf["radius"] = 3 * sf[k0["octave"]]
qf["min_level"], qf["max_level"], qf["flags"] = k0["octave"] - 1, k0["octave"], 1
rows.append(("`Fuse` window search (1000 points into one key frame)", bench(lambda: guided.WindowBest(ex, k1, d1, gp, qf, d0, None, sig), 300),
             bench(lambda: O.window_best(k1, d1, gp, qf, d0, None, sig), 30)))

M.put_set(3, k1, d1, None, gp)
rows.append(("`Fuse` window search into a resident key frame (`orbhip_window_best_set`)",
             bench(lambda: guided.WindowBestSet(ex, 3, qf, d0, None, sig), 300), rows[-1][2]))

# LocalMapping::CreateNewMapPoints: SearchForTriangulation of a key-frame pair (FeatureVectors of levelsup 4 -> ~100 nodes)
F12 = np.array([[0, 0, 0], [0, 0, -1], [0, 1, 0]], np.float32)
sg2 = (sf ** 2).astype(np.float32)
sk0, sk1 = np.zeros(len(k0), np.uint8), np.zeros(len(k1), np.uint8)
rows.append(("`SearchForTriangulation` key-frame pair, 1000 x 1000 features",
             bench(lambda: guided.SearchForTriangulation(ex, k0, d0, sk0, fv[0], k1, d1, sk1, fv[1], F12, 320.0, 240.0, sf, sg2), 300),
             bench(lambda: O.search_for_triangulation(k0, d0, sk0, fv[0], k1, d1, sk1, fv[1], F12, 320.0, 240.0, sf, sg2), 30)))

# Frame::ComputeStereoMatches on the pyramids the two extractors hold (EuRoC stereo geometry, 1200 features)
from orbhip.extractor import ComputeStereoMatches
SL, SR = synth.make_stereo_pair(7, 752, 480, disparity=21)
exL, exR = ORBextractor(1200, max_w=752, max_h=480), ORBextractor(1200, max_w=752, max_h=480)
oL, oR = O.Extractor(1200), O.Extractor(1200)
(kL, dL), (kR, dR) = exL(SL), exR(SR)
oL(SL), oR(SR)
mb, mbf = 0.11, 47.9
rows.append(("`Frame::ComputeStereoMatches` 752x480 pair, 1200 features a side (pyramids resident)",
             bench(lambda: ComputeStereoMatches(exL, kL, dL, exR, kR, dR, mb, mbf), 300),
             bench(lambda: O.stereo_matches(oL, kL, dL, oR, kR, dR, mb, mbf), 20)))

# the two small steps of the Frame constructor (src/Frame.cc:574-589, :748-778)
from orbhip import rectify
Kc = np.array([[458.654, 0, 367.215], [0, 457.296, 248.375], [0, 0, 1]], np.float32)
Dc = np.array([-0.28340811, 0.07395907, 0.00019359, 1.76187114e-05], np.float32)
rows.append(("`Frame::UndistortKeyPoints` 1000 keypoints (EuRoC cam0 coefficients)", bench(lambda: rectify.UndistortKeyPoints(ex, k1, Kc, Dc), 300),
             bench(lambda: O.undistort_points(np.stack([k1["x"], k1["y"]], 1), Kc, Dc, Kc), 50)))
rows.append(("`Frame::AssignFeaturesToGrid` 1000 keypoints", bench(lambda: guided.AssignFeaturesToGrid(ex, k1, gp), 300),
             bench(lambda: O.grid_build(k1, gp), 50)))

# the Frame constructor's device work in ONE launch (orbhip_frame_build): extraction + UndistortKeyPoints + AssignFeaturesToGrid +
# the vocabulary transform; beside it the SUM of the four oracle calls, and the sum of the four separate device calls above
def four_oracle_calls():
    k, d = ref(fr[0])
    xy = O.undistort_points(np.stack([k["x"], k["y"]], 1), Kc, Dc, Kc)
    ku = k.copy()
    ku["x"], ku["y"] = xy[:, 0], xy[:, 1]
    O.grid_build(ku, gp)
    ovoc.transform(d, 4)
byname = {r[0]: r for r in rows}
four = sum(byname[n][1] for n in ("`ORBextractor::operator()` 640x480, 1000 features",
                                   "`ORBVocabulary::transform` descent, 1000 descriptors (k 10, L 6, levelsup 4)",
                                   "`Frame::UndistortKeyPoints` 1000 keypoints (EuRoC cam0 coefficients)",
                                   "`Frame::AssignFeaturesToGrid` 1000 keypoints"))
rows.append(("`Frame::Frame` device work in one launch (`orbhip_frame_build`: extract + undistort + grid + transform; the four separate calls above: %.3f)" % four,
             bench(lambda: ex.frame_build(fr[0], Kc, Dc, gp, 4), 500), bench(four_oracle_calls, 10)))
rows.append(("`orbhip_frame_build` without the transform (extract + undistort + grid)",
             bench(lambda: ex.frame_build(fr[0], Kc, Dc, gp, -1), 500), float("nan")))
# SearchByBoW(KF, F) as the drop-in issues it since r04: the frame enters the matcher context's set table from the device
# block of its frame_build (orbhip_set_put_from_frame, once per frame) -- registration + search, per frame
r = ex.frame_build(fr[1], Kc, Dc, gp, 4)
fvF = O.feature_vector(r["node_id"], r["weight"])
def register_and_search():
    M.put_set_from_frame(2, ex, fvF)
    M.SearchByBoW_sets(1, v0, len(k0), 2, None, len(r["kps"]))
rows.append(("`SearchByBoW(KF, F)` with the frame registered from its `orbhip_frame_build` block (registration + search)",
             bench(register_and_search, 300), byname["`SearchByBoW(KF, F)` 1000 x 1000 features"][2]))

# how the C++ drop-in (host/*.cc) reaches a row whose stand-alone C-ABI call does not beat a host core
HOW = {
    "`SearchByBoW(KF, F)` 1000 x 1000 features": "only with ORBHIP_NO_SETS=1; the drop-in's default is the resident-set row below",
    "`Fuse` window search (1000 points into one key frame)": "only with ORBHIP_NO_SETS=1; default: the resident key frame row below",
    "`SearchForTriangulation` key-frame pair, 1000 x 1000 features": "per call (LocalMapping: once per new key frame and neighbour)",
    "`Frame::UndistortKeyPoints` 1000 keypoints (EuRoC cam0 coefficients)": "frame-build only: the Frame constructor takes orbhip_frame_build's by-product; a stand-alone call is this row",
    "`Frame::AssignFeaturesToGrid` 1000 keypoints": "frame-build only: the by-product of orbhip_frame_build; called alone the drop-in bins on the host (r06: the host column), this entry point serves batches and resident sets",
}
print("| call (one per frame / key-frame pair) | liborbhip per call (ms) | oracle, one host core (ms) | how the drop-in reaches it |")
print("|---|---|---|---|")
for name, g, c in rows:
    print("| %s | %.3f | %.3f | %s |" % (name, g, c, HOW.get(name, "")))
# the floor under any of these calls on this box: what a launch and a synchronisation cost before anything is computed
import ctypes
from orbhip import capi
L = capi.load()
print()
print("| floor (`orbhip_debug_roundtrip`, C side, no Python in the loop) | ms |")
print("|---|---|")
for mode, what in ((0, "empty kernel + one synchronisation"), (2, "... the kernel storing its result to page-locked memory"),
                   (1, "4 KB in, empty kernel, 4 KB out, one synchronisation")):
    us = ctypes.c_double()
    capi.check(L.orbhip_debug_roundtrip(ex.handle, mode, 2000, ctypes.byref(us)), ex.handle, "orbhip_debug_roundtrip")
    print("| %s | %.4f |" % (what, us.value / 1e3))
t = time.perf_counter()
for _ in range(2000):
    L.orbhip_set_has(ex.handle, 12345, 1)
print("| one ctypes call of the Python binding (no device work) | %.4f |" % ((time.perf_counter() - t) / 2000 * 1e3))
