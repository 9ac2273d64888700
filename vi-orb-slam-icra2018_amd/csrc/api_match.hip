// api_match.hip -- C ABI, part 3: Hamming best/second search, SearchByBoW (merge walk over the two FeatureVectors and the
// rotation histogram on the host), the vocabulary, the distinctive-descriptor choice and SearchForTriangulation.
#include "api_common.h"
#include <atomic>

// ------------------------------------------------------------------------------------------------
// matching
// ------------------------------------------------------------------------------------------------
int orb_match_scratch(orbhip_ctx *c, size_t bytes)
{
    if (bytes <= c->d_match_bytes && c->d_match) return ORBHIP_OK;
    if (c->d_match) HIPCHK(c, hipFree(c->d_match));
    c->d_match = nullptr;
    c->d_match_bytes = 0;
    HIPCHK(c, hipMalloc(&c->d_match, bytes));
    c->d_match_bytes = bytes;
    return ORBHIP_OK;
}

extern "C" int orbhip_hamming_knn2_device(orbhip_ctx *c, const void *d_q, int nq, const void *d_db, int ndb,
                                          void *d_best_idx, void *d_best_d, void *d_second_d)
{
    if (!c || nq < 0 || ndb < 0 || (nq > 0 && (!d_q || !d_best_idx || !d_best_d || !d_second_d)) || (ndb > 0 && !d_db))
        return fail(c, ORBHIP_E_ARG, "orbhip_hamming_knn2_device: bad argument");
    if (nq == 0) return ORBHIP_OK;
    HIPCHK(c, orb_enter(c));
    int rc;
    const size_t need = knn2_scratch_bytes(nq, ndb);
    if ((rc = orb_match_scratch(c, need))) return rc;
    if (c->stageTiming >= 2) HIPCHK(c, hipEventRecord(c->ev[6], c->stream));
    launch_knn2(c->stream, (const uint8_t *)d_q, nq, (const uint8_t *)d_db, ndb, (int32_t *)d_best_idx,
                (int32_t *)d_best_d, (int32_t *)d_second_d, c->d_match, need);
    if (c->stageTiming >= 2) HIPCHK(c, hipEventRecord(c->ev[7], c->stream));
    HIPCHK(c, hipGetLastError());
    c->haveMatchEvents = c->stageTiming >= 2;
    return ORBHIP_OK;
}

extern "C" int orbhip_hamming_knn2_seq_device(orbhip_ctx *c, const void *d_desc, const void *d_counts, int cap,
                                              int B, int lag, void *d_best_idx, void *d_best_d, void *d_second_d)
{
    if (!c || !d_desc || !d_counts || cap <= 0 || B <= 0 || lag < 0 || !d_best_idx || !d_best_d || !d_second_d)
        return fail(c, ORBHIP_E_ARG, "orbhip_hamming_knn2_seq_device: bad argument");
    HIPCHK(c, orb_enter(c));
    if (c->stageTiming >= 2) HIPCHK(c, hipEventRecord(c->ev[6], c->stream));
    launch_knn2_seq(c->stream, (const uint8_t *)d_desc, (const int32_t *)d_counts, cap, B, lag,
                    (int32_t *)d_best_idx, (int32_t *)d_best_d, (int32_t *)d_second_d);
    if (c->stageTiming >= 2) HIPCHK(c, hipEventRecord(c->ev[7], c->stream));
    HIPCHK(c, hipGetLastError());
    c->haveMatchEvents = c->stageTiming >= 2;
    return ORBHIP_OK;
}

extern "C" int orbhip_hamming_knn2(orbhip_ctx *c, const uint8_t *q, int nq, const uint8_t *db, int ndb,
                                   int32_t *best_idx, int32_t *best_d, int32_t *second_d)
{
    if (!c || nq < 0 || ndb < 0 || (nq > 0 && (!q || !best_idx || !best_d || !second_d)) || (ndb > 0 && !db))
        return fail(c, ORBHIP_E_ARG, "orbhip_hamming_knn2: bad argument");
    if (nq == 0) return ORBHIP_OK;
    HIPCHK(c, orb_enter(c));
    TmpDev T(c);
    int rc;
    if ((rc = T.reserve((size_t)nq * 32 + (size_t)ndb * 32 + (size_t)nq * 12 + 2048))) return rc;
    uint8_t *dq = (uint8_t *)T.take((size_t)nq * 32), *ddb = (uint8_t *)T.take((size_t)ndb * 32 + 32);
    int32_t *dbi = (int32_t *)T.take((size_t)nq * 4), *dbd = (int32_t *)T.take((size_t)nq * 4),
            *dsd = (int32_t *)T.take((size_t)nq * 4);
    TMPCHK(c, T);
    HIPCHK(c, hipMemcpyAsync(dq, q, (size_t)nq * 32, hipMemcpyHostToDevice, c->stream));
    if (ndb) HIPCHK(c, hipMemcpyAsync(ddb, db, (size_t)ndb * 32, hipMemcpyHostToDevice, c->stream));
    if ((rc = orbhip_hamming_knn2_device(c, dq, nq, ddb, ndb, dbi, dbd, dsd))) return rc;
    HIPCHK(c, hipMemcpyAsync(best_idx, dbi, (size_t)nq * 4, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipMemcpyAsync(best_d, dbd, (size_t)nq * 4, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipMemcpyAsync(second_d, dsd, (size_t)nq * 4, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return ORBHIP_OK;
}

extern "C" int orbhip_hamming_knn2_lists(orbhip_ctx *c, const uint8_t *q, int nq, const uint8_t *db, int ndb,
                                         const int32_t *off, const int32_t *cand, int32_t *best_idx,
                                         int32_t *best_d, int32_t *second_d)
{
    if (!c || nq < 0 || ndb < 0 || (nq > 0 && (!q || !off || !best_idx || !best_d || !second_d)))
        return fail(c, ORBHIP_E_ARG, "orbhip_hamming_knn2_lists: bad argument");
    if (nq == 0) return ORBHIP_OK;
    const int ncand = off[nq];
    for (int i = 0; i < nq; i++)
        if (off[i] > off[i + 1] || off[i] < 0) return fail(c, ORBHIP_E_ARG, "offsets must be non-decreasing");
    for (int t = 0; t < ncand; t++)
        if (cand[t] < 0 || cand[t] >= ndb) return fail(c, ORBHIP_E_ARG, "candidate index out of range");
    HIPCHK(c, orb_enter(c));
    TmpDev T(c);
    int rc;
    if ((rc = T.reserve((size_t)nq * 32 + (size_t)ndb * 32 + (size_t)nq * 16 + (size_t)ncand * 4 + 4096))) return rc;
    uint8_t *dq = (uint8_t *)T.take((size_t)nq * 32), *ddb = (uint8_t *)T.take((size_t)ndb * 32 + 32);
    int32_t *doff = (int32_t *)T.take((size_t)(nq + 1) * 4), *dcand = (int32_t *)T.take((size_t)ncand * 4 + 4);
    int32_t *dbi = (int32_t *)T.take((size_t)nq * 4), *dbd = (int32_t *)T.take((size_t)nq * 4),
            *dsd = (int32_t *)T.take((size_t)nq * 4);
    TMPCHK(c, T);
    HIPCHK(c, hipMemcpyAsync(dq, q, (size_t)nq * 32, hipMemcpyHostToDevice, c->stream));
    if (ndb) HIPCHK(c, hipMemcpyAsync(ddb, db, (size_t)ndb * 32, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync(doff, off, (size_t)(nq + 1) * 4, hipMemcpyHostToDevice, c->stream));
    if (ncand) HIPCHK(c, hipMemcpyAsync(dcand, cand, (size_t)ncand * 4, hipMemcpyHostToDevice, c->stream));
    launch_knn2_lists(c->stream, dq, nq, ddb, doff, dcand, dbi, dbd, dsd);
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, hipMemcpyAsync(best_idx, dbi, (size_t)nq * 4, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipMemcpyAsync(best_d, dbd, (size_t)nq * 4, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipMemcpyAsync(second_d, dsd, (size_t)nq * 4, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return ORBHIP_OK;
}

// ORBmatcher::ComputeThreeMaxima, ref: src/ORBmatcher.cc:1629-1670
void orb_three_maxima(const std::vector<int> *histo, int L, int &ind1, int &ind2, int &ind3)
{
    int max1 = 0, max2 = 0, max3 = 0;
    ind1 = ind2 = ind3 = -1;
    for (int i = 0; i < L; i++) {
        const int s = (int)histo[i].size();
        if (s > max1) {
            max3 = max2; max2 = max1; max1 = s;
            ind3 = ind2; ind2 = ind1; ind1 = i;
        } else if (s > max2) {
            max3 = max2; max2 = s;
            ind3 = ind2; ind2 = i;
        } else if (s > max3) {
            max3 = s;
            ind3 = i;
        }
    }
    if (max2 < 0.1f * (float)max1) {
        ind2 = -1;
        ind3 = -1;
    } else if (max3 < 0.1f * (float)max1) {
        ind3 = -1;
    }
}

extern "C" int orbhip_search_by_bow(orbhip_ctx *c, const uint8_t *desc1, int n1, const uint8_t *valid1,
                                    const float *angle1, const int32_t *node1, const int32_t *off1,
                                    const int32_t *idx1, int ng1, const uint8_t *desc2, int n2,
                                    const uint8_t *valid2, const float *angle2, const int32_t *node2,
                                    const int32_t *off2, const int32_t *idx2, int ng2, int th, int th_mode,
                                    float nnratio, int check_ori, int32_t *match12, int32_t *match21,
                                    int *nmatches)
{
    if (!c || n1 < 0 || n2 < 0 || ng1 < 0 || ng2 < 0 || !match12 || !match21 || !nmatches ||
        (n1 > 0 && (!desc1 || !valid1)) || (n2 > 0 && !desc2) || (check_ori && (!angle1 || !angle2)) ||
        (ng1 > 0 && (!node1 || !off1 || !idx1)) || (ng2 > 0 && (!node2 || !off2 || !idx2)))
        return fail(c, ORBHIP_E_ARG, "orbhip_search_by_bow: bad argument");
    for (int i = 0; i < n1; i++) match12[i] = -1;
    for (int i = 0; i < n2; i++) match21[i] = -1;
    *nmatches = 0;
    if (n1 == 0 || n2 == 0 || ng1 == 0 || ng2 == 0) return ORBHIP_OK;
    // merge walk over the two FeatureVectors (ref: :180-264): pairs of equal node ids
    std::vector<int32_t> pairs;
    {
        int g1 = 0, g2 = 0;
        while (g1 < ng1 && g2 < ng2) {
            if (node1[g1] == node2[g2]) {
                pairs.push_back(g1);
                pairs.push_back(g2);
                g1++;
                g2++;
            } else if (node1[g1] < node2[g2])
                g1++;
            else
                g2++;
        }
    }
    const int npairs = (int)pairs.size() / 2;
    if (npairs == 0) return ORBHIP_OK;
    const int m1 = off1[ng1], m2 = off2[ng2];
    // k_bow_match reduces (distance << 20 | position in the node's side-2 list) over the wave: the position needs 20 bits
    if (m2 >= (1 << 20)) return fail(c, ORBHIP_E_SIZE, "orbhip_search_by_bow: more than 2^20 - 1 entries in the second FeatureVector");
    for (int t = 0; t < m1; t++)
        if (idx1[t] < 0 || idx1[t] >= n1) return fail(c, ORBHIP_E_ARG, "idx1 out of range");
    for (int t = 0; t < m2; t++)
        if (idx2[t] < 0 || idx2[t] >= n2) return fail(c, ORBHIP_E_ARG, "idx2 out of range");
    HIPCHK(c, orb_enter(c));
    Packed P(c);
    int rc;
    const size_t total = (size_t)n1 * 32 + (size_t)n2 * 32 + (size_t)n1 + (size_t)n2 + (size_t)(ng1 + ng2 + 2) * 4 +
                         (size_t)(m1 + m2 + 2) * 4 + pairs.size() * 4 + (size_t)(n1 + n2) * 4 + 16 * 256;
    if ((rc = P.begin(total))) return rc;
    const uint8_t *dd1 = (const uint8_t *)P.in(desc1, (size_t)n1 * 32), *dd2 = (const uint8_t *)P.in(desc2, (size_t)n2 * 32);
    const uint8_t *dv1 = (const uint8_t *)P.in(valid1, (size_t)n1);
    const uint8_t *dv2 = valid2 ? (const uint8_t *)P.in(valid2, (size_t)n2) : nullptr;
    const int32_t *do1 = (const int32_t *)P.in(off1, (size_t)(ng1 + 1) * 4), *do2 = (const int32_t *)P.in(off2, (size_t)(ng2 + 1) * 4);
    const int32_t *di1 = (const int32_t *)P.in(idx1, (size_t)m1 * 4), *di2 = (const int32_t *)P.in(idx2, (size_t)m2 * 4);
    const int32_t *dp = (const int32_t *)P.in(pairs.data(), pairs.size() * 4);
    // match12 | match21 start as -1 (part of the upload) and come back together
    // match12 | match21 start as -1; the kernel only stores the matches.  Nodes of up to 128 features (the register path of
    // k_bow_match) never read them back, so they live in the page-locked block and no copy follows; a frame with a larger node
    // keeps them on the device (that path polls match21)
    bool hostOut = true;
    for (int p = 0; p < npairs && hostOut; p++)
        if (off2[pairs[2 * p + 1] + 1] - off2[pairs[2 * p + 1]] > 128) hostOut = false;
    int32_t *dm12, *dm21;
    if (hostOut) {
        dm12 = (int32_t *)P.out_host_fill(0xFF, (size_t)n1 * 4);
        dm21 = (int32_t *)P.out_host_fill(0xFF, (size_t)n2 * 4);
    } else {
        dm12 = (int32_t *)P.in_fill(0xFF, (size_t)n1 * 4);
        dm21 = (int32_t *)P.in_fill(0xFF, (size_t)n2 * 4);
    }
    if ((rc = P.upload())) return rc;
    launch_bow_match(c->stream, dd1, dv1, do1, di1, dd2, dv2, do2, di2, dp, npairs, th, th_mode, nnratio, dm12, dm21);
    HIPCHK(c, hipGetLastError());
    if (hostOut) {
        HIPCHK(c, hipStreamSynchronize(c->stream));
        memcpy(match12, dm12, (size_t)n1 * 4);
        memcpy(match21, dm21, (size_t)n2 * 4);
    } else {
        if ((rc = P.download(dm12))) return rc;
        memcpy(match12, P.host(dm12), (size_t)n1 * 4);
        memcpy(match21, P.host(dm21), (size_t)n2 * 4);
    }
    *nmatches = orb_bow_rotation_check(pairs.data(), npairs, off1, idx1, angle1, angle2, check_ori, match12, match21);
    return ORBHIP_OK;
}

// Rotation consistency of SearchByBoW (ref: src/ORBmatcher.cc:236-246, 267-285): histogram in the reference's visiting order
// (shared nodes in order, side-1 features in list order), all bins but the three largest are cleared.  Returns the matches left.
int orb_bow_rotation_check(const int32_t *pairs, int npairs, const int32_t *off1, const int32_t *idx1, const float *angle1,
                           const float *angle2, int check_ori, int32_t *match12, int32_t *match21)
{
    int nm = 0;
    std::vector<int> hist[30];
    const float factor = 1.0f / 30;
    for (int p = 0; p < npairs; p++) {
        const int g1 = pairs[2 * p];
        for (int a = off1[g1]; a < off1[g1 + 1]; a++) {
            const int i1 = idx1[a];
            const int i2 = match12[i1];
            if (i2 < 0) continue;
            nm++;
            if (check_ori) {
                float rot = angle1[i1] - angle2[i2];
                if (rot < 0.0) rot += 360.0f;
                int bin = (int)roundf(rot * factor);
                if (bin == 30) bin = 0;
                if (bin >= 0 && bin < 30) hist[bin].push_back(i1);
            }
        }
    }
    if (check_ori) {
        int i1, i2, i3;
        orb_three_maxima(hist, 30, i1, i2, i3);
        for (int i = 0; i < 30; i++) {
            if (i == i1 || i == i2 || i == i3) continue;
            for (int a : hist[i]) {
                match21[match12[a]] = -1;
                match12[a] = -1;
                nm--;
            }
        }
    }
    return nm;
}

// ------------------------------------------------------------------------------------------------
// vocabulary
// ------------------------------------------------------------------------------------------------
extern "C" int orbhip_vocab_load(orbhip_ctx *c, const void *blob, size_t nbytes)
{
    if (!c || !blob) return fail(c, ORBHIP_E_ARG, "orbhip_vocab_load: bad argument");
    OrbVocabHost H;
    std::string err;
    int rc = orb_vocab_parse((const uint8_t *)blob, nbytes, H, err);
    if (rc != ORBHIP_OK) return fail(c, rc, "orbhip_vocab_load: " + err);
    HIPCHK(c, orb_enter(c));
    const size_t ne = (size_t)H.nnodes - 1;
    // one device block, 256-byte aligned sections (tables by edge, see OrbVocabDev)
    size_t off[5], total = 0;
    const size_t sizes[5] = {ne * 32, ne * 8, ne * 4, ne * 4, ne * 4};
    for (int i = 0; i < 5; i++) {
        off[i] = total;
        total += align_up(sizes[i], 256);
    }
    // The block is reference-counted: a context that borrowed the previous tables (orbhip_vocab_share) keeps them alive and
    // keeps reading the OLD vocabulary until it shares again -- orbhip_vocab_generation tells it to.
    // The new tables are complete before the context sees them; the swap happens under vocMutex, so that a borrower's
    // orbhip_vocab_share / orbhip_vocab_generation on another thread reads either the old pair or the new one.
    HIPCHK(c, hipStreamSynchronize(c->stream));
    void *blk = nullptr;
    HIPCHK(c, hipMalloc(&blk, total));
    std::shared_ptr<void> hold(blk, [](void *p) { (void)hipFree(p); });
    uint8_t *base = (uint8_t *)blk;
    const void *src[5] = {H.edesc.data(), H.erange.data(), H.child.data(), H.eword.data(), H.eweight.data()};
    for (int i = 0; i < 5; i++) HIPCHK(c, hipMemcpyAsync(base + off[i], src[i], sizes[i], hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    OrbVocabDev V;
    V.k = H.k; V.L = H.L; V.scoring = H.scoring; V.weighting = H.weighting; V.nnodes = H.nnodes; V.nwords = H.nwords;
    V.rootFirst = H.childOff[0];
    V.rootLast = H.childOff[1];
    V.desc = base + off[0];
    V.erange = (int32_t *)(base + off[1]);
    V.eid = (int32_t *)(base + off[2]);
    V.eword = (int32_t *)(base + off[3]);
    V.eweight = (float *)(base + off[4]);
    static std::atomic<unsigned long long> loads{0};
    V.gen = ++loads;
    {
        std::lock_guard<std::mutex> g(c->vocMutex);
        c->vocHold = std::move(hold);   // (the old block lives on while another context borrows it)
        c->voc = V;
    }
    return ORBHIP_OK;
}

// Which load the context's tables come from: a process-wide counter, 0 without a vocabulary.  A borrower compares it with the value
// it saw when it shared and shares again when the lender has loaded since (host/ORBextractor.cc).
extern "C" unsigned long long orbhip_vocab_generation(const orbhip_ctx *c)
{
    if (!c) return 0ull;
    std::lock_guard<std::mutex> g(c->vocMutex);
    return c->voc.desc ? c->voc.gen : 0ull;
}

// The tables of `src` serve `dst` as well (same device): what lets an extractor's context run the transform inside
// orbhip_frame_build on the vocabulary that ORBVocabulary loaded into its own context.  Borrowed, not copied: 58 MB for the
// stock tree.  The block is reference-counted (OrbCtx::vocHold): `src` may load another vocabulary or be destroyed while `dst` still
// runs on the tables it borrowed; a vocabulary loaded into dst later replaces the borrowed one.
extern "C" int orbhip_vocab_share(orbhip_ctx *dst, const orbhip_ctx *src)
{
    if (!dst || !src) return fail(dst, ORBHIP_E_ARG, "orbhip_vocab_share: no vocabulary in the source context");
    if (dst == src) return src->voc.desc ? ORBHIP_OK : fail(dst, ORBHIP_E_ARG, "orbhip_vocab_share: no vocabulary in the source context");
    std::shared_ptr<void> hold;
    OrbVocabDev V;
    {
        std::lock_guard<std::mutex> g(src->vocMutex);   // a consistent (block, tables) pair while the lender may be loading
        hold = src->vocHold;
        V = src->voc;
    }
    if (!V.desc) return fail(dst, ORBHIP_E_ARG, "orbhip_vocab_share: no vocabulary in the source context");
    if (dst->device != src->device) return fail(dst, ORBHIP_E_ARG, "orbhip_vocab_share: the two contexts are on different devices");
    HIPCHK(dst, hipSetDevice(dst->device));
    HIPCHK(dst, hipStreamSynchronize(dst->stream));
    std::lock_guard<std::mutex> g(dst->vocMutex);
    dst->vocHold = std::move(hold);
    dst->voc = V;
    return ORBHIP_OK;
}

extern "C" int orbhip_vocab_load_device(orbhip_ctx *c, const void *d_blob, size_t nbytes)
{
    if (!c || !d_blob || nbytes < 24) return fail(c, ORBHIP_E_ARG, "orbhip_vocab_load_device: bad argument");
    HIPCHK(c, orb_enter(c));
    std::vector<uint8_t> host(nbytes);
    HIPCHK(c, hipMemcpy(host.data(), d_blob, nbytes, hipMemcpyDeviceToHost));
    return orbhip_vocab_load(c, host.data(), nbytes);
}

extern "C" int orbhip_vocab_info(const orbhip_ctx *c, int *k, int *L, int *scoring, int *weighting, int *nnodes,
                                 int *nwords)
{
    if (!c || !c->voc.desc) return ORBHIP_E_ARG;
    if (k) *k = c->voc.k;
    if (L) *L = c->voc.L;
    if (scoring) *scoring = c->voc.scoring;
    if (weighting) *weighting = c->voc.weighting;
    if (nnodes) *nnodes = c->voc.nnodes;
    if (nwords) *nwords = c->voc.nwords;
    return ORBHIP_OK;
}

extern "C" int orbhip_vocab_transform_device(orbhip_ctx *c, const void *d_desc, int n, int levelsup, void *d_word,
                                             void *d_weight, void *d_node)
{
    if (!c || n < 0 || (n > 0 && (!d_desc || !d_word || !d_weight || !d_node)))
        return fail(c, ORBHIP_E_ARG, "orbhip_vocab_transform_device: bad argument");
    if (!c->voc.desc) return fail(c, ORBHIP_E_ARG, "orbhip_vocab_transform: no vocabulary loaded");
    if (n == 0) return ORBHIP_OK;
    HIPCHK(c, orb_enter(c));
    launch_vocab_transform(c->stream, c->voc, (const uint8_t *)d_desc, n, levelsup, (int32_t *)d_word, (float *)d_weight,
                           (int32_t *)d_node);
    HIPCHK(c, hipGetLastError());
    return ORBHIP_OK;
}

extern "C" int orbhip_vocab_transform(orbhip_ctx *c, const uint8_t *desc, int n, int levelsup, int32_t *word_id,
                                      float *weight, int32_t *node_id)
{
    if (!c || n < 0 || (n > 0 && (!desc || !word_id || !weight || !node_id)))
        return fail(c, ORBHIP_E_ARG, "orbhip_vocab_transform: bad argument");
    if (n == 0) return ORBHIP_OK;
    HIPCHK(c, orb_enter(c));
    Packed P(c);
    int rc;
    if ((rc = P.begin((size_t)n * 44 + 4 * 256))) return rc;
    const uint8_t *dd = (const uint8_t *)P.in(desc, (size_t)n * 32);
    int32_t *dw = (int32_t *)P.out_host((size_t)n * 4), *dn = (int32_t *)P.out_host((size_t)n * 4);   // written by the kernel over PCIe
    float *dwt = (float *)P.out_host((size_t)n * 4);
    if ((rc = P.upload())) return rc;
    if ((rc = orbhip_vocab_transform_device(c, dd, n, levelsup, dw, dwt, dn))) return rc;
    HIPCHK(c, hipStreamSynchronize(c->stream));
    memcpy(word_id, dw, (size_t)n * 4);
    memcpy(weight, dwt, (size_t)n * 4);
    memcpy(node_id, dn, (size_t)n * 4);
    return ORBHIP_OK;
}

extern "C" int orbhip_search_by_bow_seq_device(orbhip_ctx *c, const void *d_desc, const void *d_kps,
                                               const void *d_counts, const void *d_node, const void *d_weight,
                                               const void *d_valid, int cap, int B, int lag, int th_mode, float nnratio,
                                               int check_ori, void *d_match12, void *d_match21, void *d_nmatches)
{
    if (!c || !d_desc || !d_kps || !d_counts || !d_node || !d_weight || cap <= 0 || B <= 0 || lag < 0 ||
        !d_match12 || !d_match21 || !d_nmatches)
        return fail(c, ORBHIP_E_ARG, "orbhip_search_by_bow_seq_device: bad argument");
    if (cap > 4096)   // the sorted keys, match table and work items of a frame pair live in LDS: 36 bytes per slot
        return fail(c, ORBHIP_E_SIZE, "orbhip_search_by_bow_seq_device: more than 4096 feature slots per frame (the per-pair "
                                      "tables exceed the 160 KB of LDS); use orbhip_search_by_bow per pair");
    HIPCHK(c, orb_enter(c));
    if (c->stageTiming >= 2) HIPCHK(c, hipEventRecord(c->ev[6], c->stream));
    HIPCHK(c, launch_bow_seq(c->stream, (const uint8_t *)d_desc, (const orbhip_keypoint *)d_kps, (const int32_t *)d_counts,
                             (const int32_t *)d_node, (const float *)d_weight, (const uint8_t *)d_valid, cap, B, lag, 50, th_mode,
                             nnratio, check_ori, (int32_t *)d_match12, (int32_t *)d_match21, (int32_t *)d_nmatches));
    if (c->stageTiming >= 2) HIPCHK(c, hipEventRecord(c->ev[7], c->stream));
    c->haveMatchEvents = c->stageTiming >= 2;
    return ORBHIP_OK;
}

extern "C" int orbhip_distinctive_descriptors_device(orbhip_ctx *c, const void *d_desc, const void *d_off, int P, void *d_best,
                                                    void *d_best_median)
{
    if (!c || P < 0 || (P > 0 && (!d_desc || !d_off || !d_best || !d_best_median)))
        return fail(c, ORBHIP_E_ARG, "orbhip_distinctive_descriptors_device: bad argument");
    if (P == 0) return ORBHIP_OK;
    HIPCHK(c, orb_enter(c));
    launch_distinctive(c->stream, (const uint8_t *)d_desc, (const int32_t *)d_off, P, (int32_t *)d_best, (int32_t *)d_best_median);
    HIPCHK(c, hipGetLastError());
    return ORBHIP_OK;
}

extern "C" int orbhip_distinctive_descriptors(orbhip_ctx *c, const uint8_t *desc, const int32_t *off, int P, int32_t *best,
                                             int32_t *best_median)
{
    if (!c || P < 0 || (P > 0 && (!off || !best)))
        return fail(c, ORBHIP_E_ARG, "orbhip_distinctive_descriptors: bad argument");
    if (P == 0) return ORBHIP_OK;
    if (off[0] != 0) return fail(c, ORBHIP_E_ARG, "orbhip_distinctive_descriptors: off[0] must be 0");
    for (int p = 0; p < P; p++)
        if (off[p + 1] < off[p] || off[p + 1] - off[p] >= (1 << 20))
            return fail(c, ORBHIP_E_ARG, "orbhip_distinctive_descriptors: offsets must ascend, a list holds fewer than 2^20 rows");
    const int total = off[P];
    if (total > 0 && !desc) return fail(c, ORBHIP_E_ARG, "orbhip_distinctive_descriptors: bad argument");
    HIPCHK(c, orb_enter(c));
    TmpDev T(c);
    int rc;
    if ((rc = T.reserve((size_t)total * 32 + (size_t)(P + 1) * 4 + (size_t)P * 8 + 4096))) return rc;
    uint8_t *dd = (uint8_t *)T.take((size_t)total * 32 + 32);
    int32_t *doff = (int32_t *)T.take((size_t)(P + 1) * 4), *db = (int32_t *)T.take((size_t)P * 4),
            *dm = (int32_t *)T.take((size_t)P * 4);
    TMPCHK(c, T);
    hipStream_t s = c->stream;
    if (total > 0) HIPCHK(c, hipMemcpyAsync(dd, desc, (size_t)total * 32, hipMemcpyHostToDevice, s));
    HIPCHK(c, hipMemcpyAsync(doff, off, (size_t)(P + 1) * 4, hipMemcpyHostToDevice, s));
    if ((rc = orbhip_distinctive_descriptors_device(c, dd, doff, P, db, dm))) return rc;
    HIPCHK(c, hipMemcpyAsync(best, db, (size_t)P * 4, hipMemcpyDeviceToHost, s));
    std::vector<int32_t> med;
    if (!best_median) {
        med.resize(P);
        best_median = med.data();
    }
    HIPCHK(c, hipMemcpyAsync(best_median, dm, (size_t)P * 4, hipMemcpyDeviceToHost, s));
    HIPCHK(c, hipStreamSynchronize(s));
    return ORBHIP_OK;
}

extern "C" int orbhip_search_for_triangulation(orbhip_ctx *c, const orbhip_keypoint *kps1, const uint8_t *desc1, int n1,
                                               const uint8_t *skip1, const float *u_right1, const int32_t *node1,
                                               const int32_t *off1, const int32_t *idx1, int ng1,
                                               const orbhip_keypoint *kps2, const uint8_t *desc2, int n2,
                                               const uint8_t *skip2, const float *u_right2, const int32_t *node2,
                                               const int32_t *off2, const int32_t *idx2, int ng2, const float F12[9],
                                               float ex, float ey, const float *scale_factors2,
                                               const float *level_sigma2_2, int nlevels2, int only_stereo, int check_ori,
                                               int32_t *matches12, int *nmatches)
{
    if (!c || n1 < 0 || n2 < 0 || ng1 < 0 || ng2 < 0 || !matches12 || !nmatches || !F12 || !scale_factors2 ||
        !level_sigma2_2 || nlevels2 < 1 || nlevels2 > 64 || n2 > 65535 || (n1 > 0 && (!kps1 || !desc1 || !skip1)) ||
        (n2 > 0 && (!kps2 || !desc2 || !skip2)) || (ng1 > 0 && (!node1 || !off1 || !idx1)) ||
        (ng2 > 0 && (!node2 || !off2 || !idx2)))
        return fail(c, ORBHIP_E_ARG, "orbhip_search_for_triangulation: bad argument");
    for (int i = 0; i < n1; i++) matches12[i] = -1;
    *nmatches = 0;
    if (n1 == 0 || n2 == 0 || ng1 == 0 || ng2 == 0) return ORBHIP_OK;
    std::vector<int32_t> pairs;   // merge walk over the two FeatureVectors (ref: :690-765)
    for (int g1 = 0, g2 = 0; g1 < ng1 && g2 < ng2;) {
        if (node1[g1] == node2[g2]) {
            pairs.push_back(g1++);
            pairs.push_back(g2++);
        } else if (node1[g1] < node2[g2])
            g1++;
        else
            g2++;
    }
    const int npairs = (int)pairs.size() / 2;
    if (npairs == 0) return ORBHIP_OK;
    const int m1 = off1[ng1], m2 = off2[ng2];
    for (int t = 0; t < m1; t++)
        if (idx1[t] < 0 || idx1[t] >= n1) return fail(c, ORBHIP_E_ARG, "idx1 out of range");
    for (int t = 0; t < m2; t++)
        if (idx2[t] < 0 || idx2[t] >= n2) return fail(c, ORBHIP_E_ARG, "idx2 out of range");
    for (int i = 0; i < n2; i++)
        if (kps2[i].octave < 0 || kps2[i].octave >= nlevels2) return fail(c, ORBHIP_E_ARG, "octave of key frame 2 out of range");
    HIPCHK(c, orb_enter(c));
    Packed P(c);
    int rc;
    const size_t total = (size_t)(n1 + n2) * (28 + 32 + 1 + 4) + (size_t)(ng1 + ng2 + 2) * 4 + (size_t)(m1 + m2 + 2) * 4 +
                         pairs.size() * 4 + (size_t)n1 * 4 + 512 + 20 * 256;
    if ((rc = P.begin(total))) return rc;
    const orbhip_keypoint *dk1 = (const orbhip_keypoint *)P.in(kps1, (size_t)n1 * 28), *dk2 = (const orbhip_keypoint *)P.in(kps2, (size_t)n2 * 28);
    const uint8_t *dd1 = (const uint8_t *)P.in(desc1, (size_t)n1 * 32), *dd2 = (const uint8_t *)P.in(desc2, (size_t)n2 * 32);
    const uint8_t *ds1 = (const uint8_t *)P.in(skip1, (size_t)n1), *ds2 = (const uint8_t *)P.in(skip2, (size_t)n2);
    const float *du1 = u_right1 ? (const float *)P.in(u_right1, (size_t)n1 * 4) : nullptr;
    const float *du2 = u_right2 ? (const float *)P.in(u_right2, (size_t)n2 * 4) : nullptr;
    const int32_t *do1 = (const int32_t *)P.in(off1, (size_t)(ng1 + 1) * 4), *do2 = (const int32_t *)P.in(off2, (size_t)(ng2 + 1) * 4);
    const int32_t *di1 = (const int32_t *)P.in(idx1, (size_t)m1 * 4), *di2 = (const int32_t *)P.in(idx2, (size_t)m2 * 4);
    const int32_t *dp = (const int32_t *)P.in(pairs.data(), pairs.size() * 4);
    const float *dsf = (const float *)P.in(scale_factors2, (size_t)nlevels2 * 4), *dsg = (const float *)P.in(level_sigma2_2, (size_t)nlevels2 * 4);
    int32_t *dm = (int32_t *)P.in_fill(0xFF, (size_t)n1 * 4);
    if ((rc = P.upload())) return rc;
    launch_tri_match(c->stream, dk1, dd1, ds1, du1, do1, di1, dk2, dd2, ds2, du2, do2, di2, dp, npairs, F12, ex, ey,
                     only_stereo ? 1 : 0, /*TH_LOW*/ 50, dsf, dsg, dm);
    HIPCHK(c, hipGetLastError());
    if ((rc = P.download(dm))) return rc;
    memcpy(matches12, P.host(dm), (size_t)n1 * 4);
    // rotation consistency (ref: :745-755, :775-794)
    int nm = 0;
    std::vector<int> hist[30];
    const float factor = 1.0f / 30;
    for (int i1 = 0; i1 < n1; i1++) {
        const int i2 = matches12[i1];
        if (i2 < 0) continue;
        nm++;
        if (check_ori) {
            float rot = kps1[i1].angle - kps2[i2].angle;
            if (rot < 0.0) rot += 360.0f;
            int bin = (int)roundf(rot * factor);
            if (bin == 30) bin = 0;
            if (bin >= 0 && bin < 30) hist[bin].push_back(i1);
        }
    }
    if (check_ori) {
        int a, b, d;
        orb_three_maxima(hist, 30, a, b, d);
        for (int i = 0; i < 30; i++) {
            if (i == a || i == b || i == d) continue;
            for (int i1 : hist[i]) {
                matches12[i1] = -1;
                nm--;
            }
        }
    }
    *nmatches = nm;
    return ORBHIP_OK;
}

