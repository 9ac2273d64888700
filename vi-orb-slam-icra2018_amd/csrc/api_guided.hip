// api_guided.hip -- C ABI, part 4: frame grid, window queries and the guided searches (SearchByProjection, the Fuse /
// SearchBySim3 window search, SearchForInitialization).
#include "api_common.h"

// ------------------------------------------------------------------------------------------------
// frame grid and guided search
// ------------------------------------------------------------------------------------------------

extern "C" int orbhip_grid_build_device(orbhip_ctx *c, const void *d_kps, const void *d_counts, int cap, int B, float min_x,
                                        float min_y, float inv_w, float inv_h, void *d_cell_off, void *d_cell_idx)
{
    if (!c || !d_kps || !d_counts || cap <= 0 || B <= 0 || !d_cell_off || !d_cell_idx || !grid_params_ok(inv_w, inv_h))
        return fail(c, ORBHIP_E_ARG, "orbhip_grid_build_device: bad argument");
    HIPCHK(c, orb_enter(c));
    launch_grid_build(c->stream, (const orbhip_keypoint *)d_kps, (const int32_t *)d_counts, cap, B, min_x, min_y, inv_w,
                      inv_h, (int32_t *)d_cell_off, (int32_t *)d_cell_idx);
    HIPCHK(c, hipGetLastError());
    return ORBHIP_OK;
}

extern "C" int orbhip_grid_build(orbhip_ctx *c, const orbhip_keypoint *kps, int n, float min_x, float min_y, float inv_w,
                                 float inv_h, int32_t *cell_off, int32_t *cell_idx)
{
    if (!c || n < 0 || (n > 0 && (!kps || !cell_idx)) || !cell_off || !grid_params_ok(inv_w, inv_h))
        return fail(c, ORBHIP_E_ARG, "orbhip_grid_build: bad argument");
    HIPCHK(c, orb_enter(c));
    const int cap = std::max(n, 1);
    Packed P(c);
    int rc;
    if ((rc = P.begin((size_t)cap * (28 + 4) + (ORBHIP_GRID_CELLS + 1) * 4 + 8 * 256))) return rc;
    const int32_t cnt[4] = {n, 0, 0, 0};
    const orbhip_keypoint *dk = (const orbhip_keypoint *)P.in(kps, (size_t)n * 28);
    P.off += (size_t)(cap - n) * 28;
    const int32_t *dc = (const int32_t *)P.in(cnt, 16);
    int32_t *doff = (int32_t *)P.out((ORBHIP_GRID_CELLS + 1) * 4), *didx = (int32_t *)P.out((size_t)cap * 4);
    if ((rc = P.upload())) return rc;
    if ((rc = orbhip_grid_build_device(c, dk, dc, cap, 1, min_x, min_y, inv_w, inv_h, doff, didx))) return rc;
    if ((rc = P.download())) return rc;   // offsets | entries in one copy (the entries are at most n)
    memcpy(cell_off, P.host(doff), (ORBHIP_GRID_CELLS + 1) * 4);
    const int total = cell_off[ORBHIP_GRID_CELLS];
    if (total) memcpy(cell_idx, P.host(didx), (size_t)total * 4);
    return ORBHIP_OK;
}

extern "C" int orbhip_features_in_area(orbhip_ctx *c, const orbhip_keypoint *kps, int n, float min_x, float min_y,
                                       float inv_w, float inv_h, const orbhip_proj_query *queries, int nq,
                                       int32_t *out_off, int32_t *out_idx, int out_cap)
{
    if (!c || n < 0 || nq < 0 || (n > 0 && !kps) || (nq > 0 && !queries) || !out_off || out_cap < 0 ||
        (out_cap > 0 && !out_idx) || !grid_params_ok(inv_w, inv_h))
        return fail(c, ORBHIP_E_ARG, "orbhip_features_in_area: bad argument");
    out_off[0] = 0;
    if (nq == 0) return ORBHIP_OK;
    HIPCHK(c, orb_enter(c));
    const int cap = std::max(n, 1);
    hipStream_t s = c->stream;
    int slots = 64;
    std::vector<int32_t> cnt(nq), idx;
    for (int attempt = 0; attempt < 2; attempt++) {
        TmpDev T(c);
        int rc;
        if ((rc = T.reserve((size_t)cap * 32 + (ORBHIP_GRID_CELLS + 1) * 4 + (size_t)nq * (32 + 4 + (size_t)slots * 4) + 8192)))
            return rc;
        orbhip_keypoint *dk = (orbhip_keypoint *)T.take((size_t)cap * 28);
        int32_t *dc = (int32_t *)T.take(16), *doff = (int32_t *)T.take((ORBHIP_GRID_CELLS + 1) * 4),
                *didx = (int32_t *)T.take((size_t)cap * 4);
        orbhip_proj_query *dq = (orbhip_proj_query *)T.take((size_t)nq * sizeof(orbhip_proj_query));
        int32_t *dcnt = (int32_t *)T.take((size_t)nq * 4), *dout = (int32_t *)T.take((size_t)nq * slots * 4);
        TMPCHK(c, T);
        if (n) HIPCHK(c, hipMemcpyAsync(dk, kps, (size_t)n * 28, hipMemcpyHostToDevice, s));
        HIPCHK(c, hipMemcpyAsync(dc, &n, 4, hipMemcpyHostToDevice, s));
        HIPCHK(c, hipMemcpyAsync(dq, queries, (size_t)nq * sizeof(orbhip_proj_query), hipMemcpyHostToDevice, s));
        if ((rc = orbhip_grid_build_device(c, dk, dc, cap, 1, min_x, min_y, inv_w, inv_h, doff, didx))) return rc;
        launch_area_list(s, dk, min_x, min_y, inv_w, inv_h, doff, didx, dq, nq, slots, dcnt, dout);
        HIPCHK(c, hipGetLastError());
        HIPCHK(c, hipMemcpyAsync(cnt.data(), dcnt, (size_t)nq * 4, hipMemcpyDeviceToHost, s));
        HIPCHK(c, hipStreamSynchronize(s));
        int mx = 0;
        for (int i = 0; i < nq; i++) mx = std::max(mx, cnt[i]);
        if (mx > slots) {   // a window with more features than the first guess: once more with room for all
            slots = mx;
            continue;
        }
        idx.resize((size_t)nq * slots);
        HIPCHK(c, hipMemcpy(idx.data(), dout, (size_t)nq * slots * 4, hipMemcpyDeviceToHost));
        break;
    }
    for (int i = 0; i < nq; i++) out_off[i + 1] = out_off[i] + cnt[i];
    if (out_off[nq] > out_cap) return fail(c, ORBHIP_E_ARG, "orbhip_features_in_area: out_cap too small");
    for (int i = 0; i < nq; i++)
        for (int k = 0; k < cnt[i]; k++) out_idx[out_off[i] + k] = idx[(size_t)i * slots + k];
    return ORBHIP_OK;
}

extern "C" int orbhip_search_by_projection_device(orbhip_ctx *c, const void *d_kps, const void *d_desc, const void *d_counts,
                                                  int cap, int B, const void *d_u_right, const void *d_occupied, float min_x,
                                                  float min_y, float inv_w, float inv_h, const void *d_cell_off,
                                                  const void *d_cell_idx, const void *d_queries, const void *d_qdesc,
                                                  const void *d_nq, int cap_q, int use_ratio, float nnratio, int check_ori,
                                                  int th_high, void *d_match, void *d_nmatches)
{
    if (!c || !d_kps || !d_desc || !d_counts || cap <= 0 || B <= 0 || !d_cell_off || !d_cell_idx || !d_queries || !d_qdesc ||
        !d_nq || cap_q <= 0 || !d_match || !d_nmatches || !grid_params_ok(inv_w, inv_h) || cap >= (1 << 19))
        return fail(c, ORBHIP_E_ARG, "orbhip_search_by_projection_device: bad argument");
    if (proj_assign_lds(cap) > 120 * 1024)
        return fail(c, ORBHIP_E_ARG, "orbhip_search_by_projection_device: cap too large for the per-frame match table in LDS");
    HIPCHK(c, orb_enter(c));
    int rc;
    if ((rc = orb_match_scratch(c, proj_scratch_bytes(B, cap_q, cap)))) return rc;
    launch_search_by_projection(c->stream, (const orbhip_keypoint *)d_kps, (const uint8_t *)d_desc, (const int32_t *)d_counts,
                                cap, B, (const float *)d_u_right, (const uint8_t *)d_occupied, min_x, min_y, inv_w, inv_h,
                                (const int32_t *)d_cell_off, (const int32_t *)d_cell_idx, (const orbhip_proj_query *)d_queries,
                                (const uint8_t *)d_qdesc, (const int32_t *)d_nq, cap_q, use_ratio, nnratio, check_ori, th_high,
                                (int32_t *)d_match, (int32_t *)d_nmatches, c->d_match);
    HIPCHK(c, hipGetLastError());
    return ORBHIP_OK;
}

extern "C" int orbhip_search_by_projection(orbhip_ctx *c, const orbhip_keypoint *kps, const uint8_t *desc, int n,
                                           const float *u_right, const uint8_t *occupied, float min_x, float min_y,
                                           float inv_w, float inv_h, const orbhip_proj_query *queries, const uint8_t *qdesc,
                                           int nq, int use_ratio, float nnratio, int check_ori, int th_high, int32_t *match,
                                           int *nmatches)
{
    if (!c || n < 0 || nq < 0 || (n > 0 && (!kps || !desc || !match)) || (nq > 0 && (!queries || !qdesc)) ||
        !grid_params_ok(inv_w, inv_h))
        return fail(c, ORBHIP_E_ARG, "orbhip_search_by_projection: bad argument");
    if (nmatches) *nmatches = 0;
    for (int i = 0; i < n; i++) match[i] = -1;
    if (n == 0 || nq == 0) return ORBHIP_OK;
    HIPCHK(c, orb_enter(c));
    Packed P(c);
    int rc;
    if ((rc = P.begin((size_t)n * (28 + 32 + 4 + 1 + 4 + 4) + (ORBHIP_GRID_CELLS + 1) * 4 + (size_t)nq * (sizeof(orbhip_proj_query) + 32) + 16 * 256)))
        return rc;
    const int32_t cnts[4] = {n, nq, 0, 0};
    const orbhip_keypoint *dk = (const orbhip_keypoint *)P.in(kps, (size_t)n * 28);
    const uint8_t *dd = (const uint8_t *)P.in(desc, (size_t)n * 32);
    const float *dur = u_right ? (const float *)P.in(u_right, (size_t)n * 4) : nullptr;
    const uint8_t *docc = occupied ? (const uint8_t *)P.in(occupied, (size_t)n) : nullptr;
    const orbhip_proj_query *dq = (const orbhip_proj_query *)P.in(queries, (size_t)nq * sizeof(orbhip_proj_query));
    const uint8_t *dqd = (const uint8_t *)P.in(qdesc, (size_t)nq * 32);
    int32_t *dc = (int32_t *)P.in(cnts, 16);                    // n | nq | number of matches (comes back with the matches)
    int32_t *dm = (int32_t *)P.out((size_t)n * 4);
    int32_t *doff = (int32_t *)P.out((ORBHIP_GRID_CELLS + 1) * 4), *didx = (int32_t *)P.out((size_t)n * 4);   // device scratch
    if ((rc = P.upload())) return rc;
    if ((rc = orbhip_grid_build_device(c, dk, dc, n, 1, min_x, min_y, inv_w, inv_h, doff, didx))) return rc;
    if ((rc = orbhip_search_by_projection_device(c, dk, dd, dc, n, 1, dur, docc, min_x, min_y, inv_w, inv_h, doff, didx, dq, dqd,
                                                 dc + 1, nq, use_ratio, nnratio, check_ori, th_high, dm, dc + 2)))
        return rc;
    // counts | matches are adjacent: one copy back
    P.off = (size_t)((uint8_t *)dm - P.d) + (size_t)n * 4;
    if ((rc = P.download(dc))) return rc;
    memcpy(match, P.host(dm), (size_t)n * 4);
    if (nmatches) *nmatches = ((const int32_t *)P.host(dc))[2];
    return ORBHIP_OK;
}

extern "C" int orbhip_window_best_device(orbhip_ctx *c, const void *d_kps, const void *d_desc, int cap, int B,
                                         const void *d_u_right, const float *inv_level_sigma2, int nlevels, float min_x,
                                         float min_y, float inv_w, float inv_h, const void *d_cell_off, const void *d_cell_idx,
                                         const void *d_queries, const void *d_qdesc, const void *d_nq, int cap_q,
                                         void *d_best_idx, void *d_best_dist)
{
    if (!c || !d_kps || !d_desc || cap <= 0 || B <= 0 || !d_cell_off || !d_cell_idx || !d_queries || !d_qdesc || !d_nq ||
        cap_q <= 0 || !d_best_idx || !d_best_dist || !grid_params_ok(inv_w, inv_h) || cap >= (1 << 23) ||
        (inv_level_sigma2 && (nlevels <= 0 || nlevels > 16)))
        return fail(c, ORBHIP_E_ARG, "orbhip_window_best_device: bad argument");
    HIPCHK(c, orb_enter(c));
    int rc;
    if ((rc = orb_match_scratch(c, window_best_scratch_bytes(B, cap)))) return rc;
    launch_window_best(c->stream, (const orbhip_keypoint *)d_kps, (const uint8_t *)d_desc, cap, B, (const float *)d_u_right,
                       inv_level_sigma2, nlevels, min_x, min_y, inv_w, inv_h, (const int32_t *)d_cell_off,
                       (const int32_t *)d_cell_idx, (const orbhip_proj_query *)d_queries, (const uint8_t *)d_qdesc,
                       (const int32_t *)d_nq, cap_q, (int32_t *)d_best_idx, (int32_t *)d_best_dist, c->d_match);
    HIPCHK(c, hipGetLastError());
    return ORBHIP_OK;
}

extern "C" int orbhip_window_best(orbhip_ctx *c, const orbhip_keypoint *kps, const uint8_t *desc, int n, const float *u_right,
                                  const float *inv_level_sigma2, int nlevels, float min_x, float min_y, float inv_w,
                                  float inv_h, const orbhip_proj_query *queries, const uint8_t *qdesc, int nq,
                                  int32_t *best_idx, int32_t *best_dist)
{
    if (!c || n < 0 || nq < 0 || (n > 0 && (!kps || !desc)) || (nq > 0 && (!queries || !qdesc || !best_idx || !best_dist)) ||
        !grid_params_ok(inv_w, inv_h) || (inv_level_sigma2 && (nlevels <= 0 || nlevels > 16)))
        return fail(c, ORBHIP_E_ARG, "orbhip_window_best: bad argument");
    for (int i = 0; i < nq; i++) {
        best_idx[i] = -1;
        best_dist[i] = 256;
    }
    if (n == 0 || nq == 0) return ORBHIP_OK;
    if (inv_level_sigma2)
        for (int i = 0; i < n; i++)
            if (kps[i].octave < 0 || kps[i].octave >= nlevels)
                return fail(c, ORBHIP_E_ARG, "orbhip_window_best: a keypoint's octave has no entry in inv_level_sigma2");
    HIPCHK(c, orb_enter(c));
    Packed P(c);
    int rc;
    if ((rc = P.begin((size_t)n * (28 + 32 + 4 + 4) + (ORBHIP_GRID_CELLS + 1) * 4 + (size_t)nq * (sizeof(orbhip_proj_query) + 32 + 8) + 16 * 256)))
        return rc;
    // one page-locked staging block, one copy in and one copy out: a call moves ~100 KB and is latency-bound
    const int32_t cnts[4] = {n, nq, 0, 0};
    const orbhip_keypoint *dk = (const orbhip_keypoint *)P.in(kps, (size_t)n * 28);
    const uint8_t *dd = (const uint8_t *)P.in(desc, (size_t)n * 32);
    const float *dur = u_right ? (const float *)P.in(u_right, (size_t)n * 4) : nullptr;
    const int32_t *dc = (const int32_t *)P.in(cnts, 16);
    const orbhip_proj_query *dq = (const orbhip_proj_query *)P.in(queries, (size_t)nq * sizeof(orbhip_proj_query));
    const uint8_t *dqd = (const uint8_t *)P.in(qdesc, (size_t)nq * 32);
    int32_t *dout = (int32_t *)P.out((size_t)nq * 8);
    int32_t *dbi = dout, *dbd = dout + nq;
    const size_t backEnd = (size_t)((uint8_t *)dout - P.d) + (size_t)nq * 8;
    int32_t *doff = (int32_t *)P.out((ORBHIP_GRID_CELLS + 1) * 4), *didx = (int32_t *)P.out((size_t)n * 4);   // device scratch
    if ((rc = P.upload())) return rc;
    if ((rc = orbhip_grid_build_device(c, dk, dc, n, 1, min_x, min_y, inv_w, inv_h, doff, didx))) return rc;
    if ((rc = orbhip_window_best_device(c, dk, dd, n, 1, dur, inv_level_sigma2, nlevels, min_x, min_y, inv_w, inv_h, doff, didx,
                                        dq, dqd, dc + 1, nq, dbi, dbd)))
        return rc;
    P.off = backEnd;
    if ((rc = P.download(dout))) return rc;
    memcpy(best_idx, P.host(dbi), (size_t)nq * 4);
    memcpy(best_dist, P.host(dbd), (size_t)nq * 4);
    return ORBHIP_OK;
}

extern "C" int orbhip_search_for_initialization_device(orbhip_ctx *c, const void *d_kps1, const void *d_desc1,
                                                       const void *d_counts1, int cap1, const void *d_kps2,
                                                       const void *d_desc2, const void *d_counts2, int cap2, int B,
                                                       float min_x, float min_y, float inv_w, float inv_h,
                                                       const void *d_cell_off2, const void *d_cell_idx2, void *d_prev_matched,
                                                       int window_size, float nnratio, int check_ori, void *d_matches12,
                                                       void *d_nmatches)
{
    if (!c || !d_kps1 || !d_desc1 || !d_counts1 || !d_kps2 || !d_desc2 || !d_counts2 || cap1 <= 0 || cap2 <= 0 || B <= 0 ||
        !d_cell_off2 || !d_cell_idx2 || !d_prev_matched || !d_matches12 || !d_nmatches || window_size < 0 ||
        !grid_params_ok(inv_w, inv_h) || cap2 >= (1 << 23) || cap1 >= (1 << 23))
        return fail(c, ORBHIP_E_ARG, "orbhip_search_for_initialization_device: bad argument");
    if (init_assign_lds(cap1, cap2) > 112 * 1024)
        return fail(c, ORBHIP_E_ARG, "orbhip_search_for_initialization_device: cap too large for the per-pair match tables in LDS");
    HIPCHK(c, orb_enter(c));
    int rc;
    if ((rc = orb_match_scratch(c, init_scratch_bytes(B, cap1, cap2)))) return rc;
    launch_search_for_initialization(c->stream, (const orbhip_keypoint *)d_kps1, (const uint8_t *)d_desc1,
                                     (const int32_t *)d_counts1, cap1, (const orbhip_keypoint *)d_kps2, (const uint8_t *)d_desc2,
                                     (const int32_t *)d_counts2, cap2, B, min_x, min_y, inv_w, inv_h, (const int32_t *)d_cell_off2,
                                     (const int32_t *)d_cell_idx2, (float *)d_prev_matched, window_size, nnratio, check_ori,
                                     /*TH_LOW*/ 50, (int32_t *)d_matches12, (int32_t *)d_nmatches, c->d_match);
    HIPCHK(c, hipGetLastError());
    return ORBHIP_OK;
}

extern "C" int orbhip_search_for_initialization(orbhip_ctx *c, const orbhip_keypoint *kps1, const uint8_t *desc1, int n1,
                                                const orbhip_keypoint *kps2, const uint8_t *desc2, int n2, float min_x,
                                                float min_y, float inv_w, float inv_h, float *prev_matched, int window_size,
                                                float nnratio, int check_ori, int32_t *matches12, int *nmatches)
{
    if (!c || n1 < 0 || n2 < 0 || (n1 > 0 && (!kps1 || !desc1 || !matches12 || !prev_matched)) || (n2 > 0 && (!kps2 || !desc2)) ||
        window_size < 0 || !grid_params_ok(inv_w, inv_h))
        return fail(c, ORBHIP_E_ARG, "orbhip_search_for_initialization: bad argument");
    if (nmatches) *nmatches = 0;
    for (int i = 0; i < n1; i++) matches12[i] = -1;
    if (n1 == 0 || n2 == 0) return ORBHIP_OK;
    HIPCHK(c, orb_enter(c));
    Packed P(c);
    int rc;
    if ((rc = P.begin((size_t)n1 * (28 + 32 + 8 + 4) + (size_t)n2 * (28 + 32 + 4) + (ORBHIP_GRID_CELLS + 1) * 4 + 16 * 256))) return rc;
    const int32_t cnts[4] = {n1, n2, 0, 0};
    const orbhip_keypoint *dk1 = (const orbhip_keypoint *)P.in(kps1, (size_t)n1 * 28), *dk2 = (const orbhip_keypoint *)P.in(kps2, (size_t)n2 * 28);
    const uint8_t *dd1 = (const uint8_t *)P.in(desc1, (size_t)n1 * 32), *dd2 = (const uint8_t *)P.in(desc2, (size_t)n2 * 32);
    // the matches and their count are stored by the kernel straight into the page-locked block (no copy command behind it: that
    // node started ~8 us after the kernel ended); vbPrevMatched is an input on the device and is brought up to date on the host
    // from the matches -- the reference's last loop, vbPrevMatched[i1] = F2.mvKeysUn[vnMatches12[i1]].pt (:511-515)
    int32_t *dc = (int32_t *)P.in(cnts, 16);
    float *dpm = (float *)P.in(prev_matched, (size_t)n1 * 8);
    int32_t *doff = (int32_t *)P.out((ORBHIP_GRID_CELLS + 1) * 4), *didx = (int32_t *)P.out((size_t)n2 * 4);   // device scratch
    int32_t *hm = (int32_t *)P.out_host((size_t)n1 * 4), *hn = (int32_t *)P.out_host(16);
    if ((rc = P.upload())) return rc;
    if ((rc = orbhip_grid_build_device(c, dk2, dc + 1, n2, 1, min_x, min_y, inv_w, inv_h, doff, didx))) return rc;
    if ((rc = orbhip_search_for_initialization_device(c, dk1, dd1, dc, n1, dk2, dd2, dc + 1, n2, 1, min_x, min_y, inv_w, inv_h, doff,
                                                      didx, dpm, window_size, nnratio, check_ori, hm, hn)))
        return rc;
    HIPCHK(c, hipStreamSynchronize(c->stream));
    memcpy(matches12, hm, (size_t)n1 * 4);
    for (int i = 0; i < n1; i++)
        if (hm[i] >= 0 && hm[i] < n2) {
            prev_matched[2 * i] = kps2[hm[i]].x;
            prev_matched[2 * i + 1] = kps2[hm[i]].y;
        }
    if (nmatches) *nmatches = hn[0];
    return ORBHIP_OK;
}

