"""Committed golden vectors (tests/golden/*.npz, made by tools/gen_golden.py from the oracle):
the oracle must still reproduce them on CPU; the HIP path must reproduce them on the GPU."""
import glob
import os

import numpy as np
import pytest

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
EXTRACT = sorted(glob.glob(os.path.join(GOLD, "extract_*.npz")))


def _fv(node):
    ids = sorted(set(int(v) for v in node))
    lists = [np.nonzero(node == k)[0] for k in ids]
    off = np.concatenate([[0], np.cumsum([len(x) for x in lists])]).astype(np.int32)
    return np.array(ids, np.int32), off, np.concatenate(lists).astype(np.int32)


def test_golden_files_present():
    assert len(EXTRACT) == 3 and os.path.exists(os.path.join(GOLD, "matching_q250_db600.npz"))
    for name in ("vocab_k5_L3.npz", "stereo_376x241_f400_l6.npz", "guided_376x241.npz", "rectify_376x241.npz",
                 "init_search_376x241.npz", "triangulation_376x241.npz", "window_best_376x241.npz", "distinctive_p80.npz"):
        assert os.path.exists(os.path.join(GOLD, name)), name


@pytest.mark.parametrize("path", EXTRACT, ids=[os.path.basename(p) for p in EXTRACT])
def test_oracle_reproduces_golden_extract(oracle, path):
    g = np.load(path)
    nf, nl, ini, mn = (int(v) for v in g["params"])
    ex = oracle.Extractor(nf, 1.2, nl, ini, mn)
    k, d = ex(g["image"])
    assert k.tobytes() == g["keypoints"].tobytes() and np.array_equal(d, g["descriptors"])
    assert [len(ex.level_cands(l)) for l in range(nl)] == g["level_candidates"].tolist()
    assert [int(ex.pyramid(l).astype(np.uint64).sum()) for l in range(nl)] == g["level_pixel_sum"].tolist()
    assert [int(ex.blurred(l).astype(np.uint64).sum()) for l in range(nl)] == g["blurred_pixel_sum"].tolist()


def test_oracle_reproduces_golden_matching(oracle):
    g = np.load(os.path.join(GOLD, "matching_q250_db600.npz"))
    bi, bd, sd = oracle.knn2(g["q"], g["db"])
    assert np.array_equal(bi, g["best_idx"]) and np.array_equal(bd, g["best_d"]) and np.array_equal(sd, g["second_d"])
    fv1, fv2 = _fv(g["node1"]), _fv(g["node2"])
    n, m12, m21 = oracle.search_by_bow(g["q"], g["valid1"], g["angle1"], fv1, g["db"], None, g["angle2"], fv2, th=50,
                                       th_mode=0, nnratio=0.7, check_ori=True)
    assert n == int(g["bow_kf_f_n"]) and np.array_equal(m12, g["bow_kf_f_m12"]) and np.array_equal(m21, g["bow_kf_f_m21"])
    n, m12, m21 = oracle.search_by_bow(g["q"], g["valid1"], g["angle1"], fv1, g["db"], g["valid2"], g["angle2"], fv2,
                                       th=50, th_mode=1, nnratio=0.75, check_ori=True)
    assert n == int(g["bow_kf_kf_n"]) and np.array_equal(m12, g["bow_kf_kf_m12"])


@pytest.mark.gpu
@pytest.mark.parametrize("path", EXTRACT, ids=[os.path.basename(p) for p in EXTRACT])
def test_hip_reproduces_golden_extract(path):
    from orbhip.extractor import ORBextractor
    g = np.load(path)
    nf, nl, ini, mn = (int(v) for v in g["params"])
    img = g["image"]
    ex = ORBextractor(nf, 1.2, nl, ini, mn, max_w=img.shape[1], max_h=img.shape[0])
    k, d = ex(img)
    assert k.tobytes() == g["keypoints"].tobytes() and np.array_equal(d, g["descriptors"])
    assert [len(ex.level_candidates(l)) for l in range(nl)] == g["level_candidates"].tolist()
    assert [int(ex.image_pyramid(l).astype(np.uint64).sum()) for l in range(nl)] == g["level_pixel_sum"].tolist()
    assert [int(ex.blurred(l).astype(np.uint64).sum()) for l in range(nl)] == g["blurred_pixel_sum"].tolist()
    ex.close()


@pytest.mark.gpu
def test_hip_reproduces_golden_matching():
    from orbhip.extractor import ORBextractor, ORBmatcher
    g = np.load(os.path.join(GOLD, "matching_q250_db600.npz"))
    ex = ORBextractor(300, max_w=320, max_h=240)
    m = ORBmatcher(0.7, True, ctx=ex)
    bi, bd, sd = m.knn2(g["q"], g["db"])
    assert np.array_equal(bi, g["best_idx"]) and np.array_equal(bd, g["best_d"]) and np.array_equal(sd, g["second_d"])
    fv1, fv2 = _fv(g["node1"]), _fv(g["node2"])
    n, m12, m21 = m.SearchByBoW(g["q"], g["valid1"], g["angle1"], fv1, g["db"], None, g["angle2"], fv2, kf_kf=False)
    assert n == int(g["bow_kf_f_n"]) and np.array_equal(m12, g["bow_kf_f_m12"]) and np.array_equal(m21, g["bow_kf_f_m21"])
    m2 = ORBmatcher(0.75, True, ctx=ex)
    n, m12, m21 = m2.SearchByBoW(g["q"], g["valid1"], g["angle1"], fv1, g["db"], g["valid2"], g["angle2"], fv2, kf_kf=True)
    assert n == int(g["bow_kf_kf_n"]) and np.array_equal(m12, g["bow_kf_kf_m12"]) and np.array_equal(m21, g["bow_kf_kf_m21"])
    ex.close()


# ---- vocabulary, stereo, guided search, undistortion / rectification ----
def _g(name):
    return np.load(os.path.join(GOLD, name))


def test_oracle_reproduces_golden_next_rows(oracle):
    g = _g("vocab_k5_L3.npz")
    V = oracle.Vocabulary(g["blob"].tobytes())
    w, wt, nid = V.transform(g["desc"], 1)
    assert np.array_equal(w, g["word"]) and np.array_equal(wt, g["weight"]) and np.array_equal(nid, g["node"])
    bw, bv = V.bow(w, wt)
    assert np.array_equal(bw, g["bow_word"]) and np.array_equal(bv, g["bow_value"])
    g = _g("stereo_376x241_f400_l6.npz")
    exL, exR = oracle.Extractor(400, 1.2, 6), oracle.Extractor(400, 1.2, 6)
    kL, dL = exL(g["left"])
    kR, dR = exR(g["right"])
    u, z, n = oracle.stereo_matches(exL, kL, dL, exR, kR, dR, g["mb_mbf"][0], g["mb_mbf"][1])
    assert n == int(g["n_before_cut"]) and u.tobytes() == g["u_right"].tobytes() and z.tobytes() == g["depth"].tobytes()
    g = _g("guided_376x241.npz")
    gp = tuple(g["grid"])
    off, idx = oracle.grid_build(g["kps"], gp)
    assert np.array_equal(off, g["cell_off"]) and np.array_equal(idx, g["cell_idx"])
    n, m = oracle.search_by_projection(g["kps"], g["desc"], gp, g["queries"], g["qdesc"], occupied=g["occupied"],
                                       use_ratio=False, nnratio=0.9, check_ori=True)
    assert n == int(g["n_last"]) and np.array_equal(m, g["match_last"])
    n, m = oracle.search_by_projection(g["kps"], g["desc"], gp, g["queries_ratio"], g["qdesc"], occupied=g["occupied"],
                                       use_ratio=True, nnratio=0.8, check_ori=False)
    assert n == int(g["n_ratio"]) and np.array_equal(m, g["match_ratio"])
    g = _g("init_search_376x241.npz")
    n, m, p = oracle.search_for_initialization(g["kps1"], g["desc1"], g["kps2"], g["desc2"], tuple(g["grid"]), g["prev"], 30, 0.9, True)
    assert n == int(g["n"]) and np.array_equal(m, g["matches12"]) and np.array_equal(p, g["prev_out"])
    n, m, p = oracle.search_for_initialization(g["kps1"], g["desc1"], g["kps2"], g["desc2"], tuple(g["grid"]), p, 30, 0.9, True)
    assert n == int(g["n_again"]) and np.array_equal(m, g["matches12_again"]) and np.array_equal(p, g["prev_out_again"])
    g = _g("triangulation_376x241.npz")
    n, m = oracle.search_for_triangulation(g["kps1"], g["desc1"], g["skip1"], (g["node1"], g["off1"], g["idx1"]), g["kps2"],
                                           g["desc2"], g["skip2"], (g["node2"], g["off2"], g["idx2"]), g["F12"],
                                           g["epipole"][0], g["epipole"][1], g["scale_factors"], g["level_sigma2"])
    assert n == int(g["n"]) and np.array_equal(m, g["matches12"])
    g = _g("window_best_376x241.npz")
    bi, bd = oracle.window_best(g["kps"], g["desc"], tuple(g["grid"]), g["queries"], g["qdesc"])
    assert np.array_equal(bi, g["best_idx"]) and np.array_equal(bd, g["best_dist"])
    bi, bd = oracle.window_best(g["kps"], g["desc"], tuple(g["grid"]), g["queries"], g["qdesc"], g["u_right"], g["inv_level_sigma2"])
    assert np.array_equal(bi, g["gated_idx"]) and np.array_equal(bd, g["gated_dist"])
    g = _g("distinctive_p80.npz")
    best, med = oracle.distinctive_descriptors(g["desc"], g["off"])
    assert np.array_equal(best, g["best"]) and np.array_equal(med, g["median"])
    g = _g("rectify_376x241.npz")
    mx, my = oracle.init_undistort_rectify_map(g["K"], g["D"], g["R"], g["P"], 376, 241)
    assert float(mx.astype(np.float64).sum()) == float(g["map_x_sum"]) and np.array_equal(mx[120], g["map_x_row"])
    assert float(my.astype(np.float64).sum()) == float(g["map_y_sum"]) and np.array_equal(my[:, 188], g["map_y_col"])
    assert np.array_equal(oracle.remap_linear(g["image"], mx, my), g["rectified"])
    assert oracle.undistort_points(g["points"], g["K"], g["D"][:4], g["K"]).tobytes() == g["undistorted"].tobytes()


@pytest.mark.gpu
def test_hip_reproduces_golden_next_rows():
    from orbhip import guided, rectify
    from orbhip.extractor import ComputeStereoMatches, ORBextractor
    from orbhip.vocabulary import ORBVocabulary
    ex = ORBextractor(400, 1.2, 6, max_w=376, max_h=241)
    exR = ORBextractor(400, 1.2, 6, max_w=376, max_h=241)
    g = _g("vocab_k5_L3.npz")
    voc = ORBVocabulary(ex)
    voc.loadFromBinaryBlob(g["blob"].tobytes())
    w, wt, nid = voc.transform_raw(g["desc"], 1)
    assert np.array_equal(w, g["word"]) and np.array_equal(wt, g["weight"]) and np.array_equal(nid, g["node"])
    (bw, bv), _ = voc.transform(g["desc"], 1)
    assert np.array_equal(bw, g["bow_word"]) and np.array_equal(bv, g["bow_value"])
    g = _g("stereo_376x241_f400_l6.npz")
    kL, dL = ex(g["left"])
    kR, dR = exR(g["right"])
    u, z, n = ComputeStereoMatches(ex, kL, dL, exR, kR, dR, float(g["mb_mbf"][0]), float(g["mb_mbf"][1]))
    assert n == int(g["n_before_cut"]) and u.tobytes() == g["u_right"].tobytes() and z.tobytes() == g["depth"].tobytes()
    g = _g("guided_376x241.npz")
    gp = tuple(g["grid"])
    off, idx = guided.AssignFeaturesToGrid(ex, g["kps"], gp)
    assert np.array_equal(off, g["cell_off"]) and np.array_equal(idx, g["cell_idx"])
    n, m = guided.SearchByProjection(ex, g["kps"], g["desc"], gp, g["queries"], g["qdesc"], occupied=g["occupied"],
                                     use_ratio=False, nnratio=0.9, check_ori=True)
    assert n == int(g["n_last"]) and np.array_equal(m, g["match_last"])
    n, m = guided.SearchByProjection(ex, g["kps"], g["desc"], gp, g["queries_ratio"], g["qdesc"], occupied=g["occupied"],
                                     use_ratio=True, nnratio=0.8, check_ori=False)
    assert n == int(g["n_ratio"]) and np.array_equal(m, g["match_ratio"])
    g = _g("init_search_376x241.npz")
    n, m, p = guided.SearchForInitialization(ex, g["kps1"], g["desc1"], g["kps2"], g["desc2"], tuple(g["grid"]), g["prev"], 30, 0.9)
    assert n == int(g["n"]) and np.array_equal(m, g["matches12"]) and np.array_equal(p, g["prev_out"])
    n, m, p = guided.SearchForInitialization(ex, g["kps1"], g["desc1"], g["kps2"], g["desc2"], tuple(g["grid"]), p, 30, 0.9)
    assert n == int(g["n_again"]) and np.array_equal(m, g["matches12_again"]) and np.array_equal(p, g["prev_out_again"])
    g = _g("triangulation_376x241.npz")
    n, m = guided.SearchForTriangulation(ex, g["kps1"], g["desc1"], g["skip1"], (g["node1"], g["off1"], g["idx1"]), g["kps2"],
                                         g["desc2"], g["skip2"], (g["node2"], g["off2"], g["idx2"]), g["F12"],
                                         g["epipole"][0], g["epipole"][1], g["scale_factors"], g["level_sigma2"])
    assert n == int(g["n"]) and np.array_equal(m, g["matches12"])
    g = _g("window_best_376x241.npz")
    bi, bd = guided.WindowBest(ex, g["kps"], g["desc"], tuple(g["grid"]), g["queries"], g["qdesc"])
    assert np.array_equal(bi, g["best_idx"]) and np.array_equal(bd, g["best_dist"])
    bi, bd = guided.WindowBest(ex, g["kps"], g["desc"], tuple(g["grid"]), g["queries"], g["qdesc"], g["u_right"], g["inv_level_sigma2"])
    assert np.array_equal(bi, g["gated_idx"]) and np.array_equal(bd, g["gated_dist"])
    g = _g("distinctive_p80.npz")
    best, med = guided.ComputeDistinctiveDescriptors(ex, g["desc"], g["off"])
    assert np.array_equal(best, g["best"]) and np.array_equal(med, g["median"])
    g = _g("rectify_376x241.npz")
    mx, my = rectify.initUndistortRectifyMap(g["K"], g["D"], g["R"], g["P"], 376, 241)
    assert float(mx.astype(np.float64).sum()) == float(g["map_x_sum"]) and np.array_equal(mx[120], g["map_x_row"])
    assert np.array_equal(rectify.Rectifier(ex, mx, my)(g["image"]), g["rectified"])
    assert rectify.undistort_points(ex, g["points"], g["K"], g["D"][:4], g["K"]).tobytes() == g["undistorted"].tobytes()
    ex.close()
    exR.close()
