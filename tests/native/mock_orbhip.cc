// mock_orbhip.cc -- a host-only stand-in for liborbhip.so's C ABI (include/orbhip.h), test infrastructure for the ThreadSanitizer
// build of the drop-in's host side (tests/test_host_tsan.py; ref: src/System.cc:365-375 -- Tracking,
// LocalMapping and LoopClosing run matchers at once).  No GPU, no arithmetic of the product: every search answers with a
// deterministic function of ITS INPUTS (hashes of the descriptors it was handed), so that
//   * a result computed single-threaded equals the same call made from a thread -- unless the host code handed the library
//     the wrong data (a stale resident set, another frame's key, a buffer another thread is writing);
//   * the resident-set entry points answer from the descriptors STORED under the key: a set that was evicted, replaced or
//     mixed up shows as a different answer, not as a crash;
//   * a context entered by two threads at once aborts (the library's contexts are not re-entrant: include/orbhip.h).
// The table of resident sets mirrors csrc/api_sets.hip: least recently used out beyond the limit (4 .. 96), find = use.
#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "orbhip.h"

namespace {
struct Set {
    uint64_t key = 0, fp = 0;
    int n = 0, ng = 0;
    unsigned long stamp = 0;
    std::vector<uint8_t> desc;
    std::vector<orbhip_keypoint> kps;
};
}  // namespace

struct orbhip_ctx {
    std::atomic<int> inside{0};
    std::string err;
    int limit = 96;
    unsigned long clock = 0;
    std::vector<Set> sets;
};

namespace {
std::atomic<long> g_calls{0}, g_evictions{0};
struct Guard {
    orbhip_ctx *c;
    explicit Guard(orbhip_ctx *c_) : c(c_)
    {
        g_calls.fetch_add(1, std::memory_order_relaxed);
        if (c && c->inside.fetch_add(1) != 0) {
            fprintf(stderr, "mock_orbhip: a context was entered by two threads at once\n");
            abort();
        }
    }
    ~Guard() { if (c) c->inside.fetch_sub(1); }
};
uint64_t mix(uint64_t h, uint64_t v) { h ^= v + 0x9E3779B97F4A7C15ull + (h << 6) + (h >> 2); return h * 0x100000001B3ull; }
uint64_t hash32(const uint8_t *d)
{
    uint64_t w[4];
    memcpy(w, d, 32);
    return mix(mix(mix(mix(1469598103934665603ull, w[0]), w[1]), w[2]), w[3]);
}
Set *find(orbhip_ctx *c, uint64_t key)
{
    for (Set &s : c->sets)
        if (s.key == key) {
            s.stamp = ++c->clock;
            return &s;
        }
    return nullptr;
}
void put(orbhip_ctx *c, uint64_t key, const orbhip_keypoint *kps, const uint8_t *desc, int n, int ng)
{
    Set *s = find(c, key);
    if (!s) {
        while ((int)c->sets.size() >= c->limit) {   // least recently used out
            size_t lru = 0;
            for (size_t i = 1; i < c->sets.size(); i++)
                if (c->sets[i].stamp < c->sets[lru].stamp) lru = i;
            c->sets.erase(c->sets.begin() + lru);
            g_evictions.fetch_add(1, std::memory_order_relaxed);
        }
        c->sets.push_back(Set());
        s = &c->sets.back();
        s->key = key;
    }
    s->n = n;
    s->ng = ng;
    s->desc.assign(desc, desc + (size_t)n * 32);
    s->kps.assign(kps, kps + n);
    s->fp = orbhip_set_fingerprint_rows(kps, desc, desc + (size_t)(n - 1) * 32, n);
    s->stamp = ++c->clock;
}
// the canned SearchByBoW: side-1 feature i1 (valid) proposes side-2 feature h % n2 when bit 8 of its hash is set; first come,
// first served (match21 decides); threshold mode, ratio and the rotation check enter the hash so that matchers differ
int canned_bow(const uint8_t *d1, int n1, const uint8_t *v1, const uint8_t *d2, int n2, const uint8_t *v2, int th_mode, float ratio,
               int check_ori, int32_t *m12, int32_t *m21)
{
    for (int i = 0; i < n1; i++) m12[i] = -1;
    for (int i = 0; i < n2; i++) m21[i] = -1;
    int nm = 0;
    const uint64_t salt = (uint64_t)(th_mode * 2 + check_ori) + (uint64_t)(ratio * 1000.f);
    for (int i1 = 0; i1 < n1; i1++) {
        if (v1 && !v1[i1]) continue;
        const uint64_t h = mix(hash32(d1 + (size_t)i1 * 32), salt);
        if (!(h & 256)) continue;
        const int i2 = (int)((h >> 16) % (uint64_t)n2);
        if (m21[i2] >= 0 || (v2 && !v2[i2])) continue;
        if ((hash32(d2 + (size_t)i2 * 32) ^ h) & 1) continue;   // (depends on side 2's content too)
        m12[i1] = i2;
        m21[i2] = i1;
        nm++;
    }
    return nm;
}
void canned_window(const uint8_t *d, int n, const orbhip_proj_query *q, const uint8_t *qdesc, int nq, int32_t *bestIdx, int32_t *bestDist)
{
    for (int k = 0; k < nq; k++) {
        bestIdx[k] = -1;
        bestDist[k] = 256;
        if (!(q[k].flags & ORBHIP_Q_ACTIVE) || n == 0) continue;
        const uint64_t h = hash32(qdesc + (size_t)k * 32);
        const int i = (int)((h >> 20) % (uint64_t)n);
        bestIdx[k] = i;
        bestDist[k] = (int)((h ^ hash32(d + (size_t)i * 32)) % 120);
    }
}
int bad(orbhip_ctx *c, const char *what)
{
    if (c) c->err = what;
    return ORBHIP_E_ARG;
}
}  // namespace

extern "C" {
long mock_orbhip_calls() { return g_calls.load(); }
long mock_orbhip_evictions() { return g_evictions.load(); }

orbhip_ctx *orbhip_create(int, int, float, int, int, int, int, int, int) { return new orbhip_ctx(); }
void orbhip_destroy(orbhip_ctx *c) { delete c; }
const char *orbhip_last_error(const orbhip_ctx *c) { return c ? c->err.c_str() : "mock_orbhip: no context"; }

uint64_t orbhip_set_fingerprint_rows(const orbhip_keypoint *kps, const uint8_t *first, const uint8_t *last, int n)
{
    if (n <= 0) return 0;
    uint64_t k0[2] = {0, 0};
    memcpy(k0, kps, 16);
    return mix(mix(mix(mix((uint64_t)n, k0[0]), k0[1]), hash32(first)), hash32(last));
}
int orbhip_set_put(orbhip_ctx *c, uint64_t key, const orbhip_keypoint *kps, const uint8_t *desc, int n, const int32_t *, const int32_t *,
                   const int32_t *, int ng, float, float, float, float)
{
    if (!c || n <= 0 || !kps || !desc) return bad(c, "orbhip_set_put");
    Guard g(c);
    put(c, key, kps, desc, n, ng);
    return ORBHIP_OK;
}
int orbhip_set_put_from_frame(orbhip_ctx *c, uint64_t, orbhip_ctx *, const int32_t *, const int32_t *, const int32_t *, int)
{
    return bad(c, "mock: no frame builds");
}
uint64_t orbhip_frame_fingerprint(const orbhip_ctx *) { return 0; }
int orbhip_set_has(orbhip_ctx *c, uint64_t key, int n)
{
    if (!c) return 0;
    Guard g(c);
    Set *s = find(c, key);
    return s && s->n == n;
}
int orbhip_set_info(orbhip_ctx *c, uint64_t key, int *n, int *ng, uint64_t *fp)
{
    if (!c) return 0;
    Guard g(c);
    Set *s = find(c, key);
    if (!s) return 0;
    if (n) *n = s->n;
    if (ng) *ng = s->ng;
    if (fp) *fp = s->fp;
    return 1;
}
int orbhip_set_limit(orbhip_ctx *c, int m)
{
    if (!c) return 0;
    Guard g(c);
    c->limit = m < 4 ? 4 : m > 96 ? 96 : m;
    return c->limit;
}
int orbhip_set_drop(orbhip_ctx *c, uint64_t key)
{
    if (!c) return ORBHIP_E_ARG;
    Guard g(c);
    for (size_t i = 0; i < c->sets.size();)
        if (key == 0 || c->sets[i].key == key) c->sets.erase(c->sets.begin() + i);
        else i++;
    return ORBHIP_OK;
}
int orbhip_search_by_bow(orbhip_ctx *c, const uint8_t *desc1, int n1, const uint8_t *valid1, const float *, const int32_t *, const int32_t *,
                         const int32_t *, int, const uint8_t *desc2, int n2, const uint8_t *valid2, const float *, const int32_t *,
                         const int32_t *, const int32_t *, int, int, int th_mode, float nnratio, int check_ori, int32_t *match12,
                         int32_t *match21, int *nmatches)
{
    if (!c || !match12 || !match21 || !nmatches) return bad(c, "orbhip_search_by_bow");
    Guard g(c);
    *nmatches = (n1 && n2) ? canned_bow(desc1, n1, valid1, desc2, n2, valid2, th_mode, nnratio, check_ori, match12, match21) : 0;
    return ORBHIP_OK;
}
int orbhip_search_by_bow_sets(orbhip_ctx *c, uint64_t key1, const uint8_t *valid1, uint64_t key2, const uint8_t *valid2, int, int th_mode,
                              float nnratio, int check_ori, int32_t *match12, int32_t *match21, int *nmatches)
{
    if (!c) return ORBHIP_E_ARG;
    Guard g(c);
    Set *a = find(c, key1), *b = find(c, key2);
    if (!a || !b) return bad(c, "orbhip_search_by_bow_sets: unknown set");
    *nmatches = canned_bow(a->desc.data(), a->n, valid1, b->desc.data(), b->n, valid2, th_mode, nnratio, check_ori, match12, match21);
    return ORBHIP_OK;
}
int orbhip_window_best(orbhip_ctx *c, const orbhip_keypoint *, const uint8_t *desc, int n, const float *, const float *, int, float, float,
                       float, float, const orbhip_proj_query *q, const uint8_t *qdesc, int nq, int32_t *best_idx, int32_t *best_dist)
{
    if (!c) return ORBHIP_E_ARG;
    Guard g(c);
    canned_window(desc, n, q, qdesc, nq, best_idx, best_dist);
    return ORBHIP_OK;
}
int orbhip_window_best_set(orbhip_ctx *c, uint64_t key, const float *, const float *, int, const orbhip_proj_query *q, const uint8_t *qdesc,
                           int nq, int32_t *best_idx, int32_t *best_dist)
{
    if (!c) return ORBHIP_E_ARG;
    Guard g(c);
    Set *s = find(c, key);
    if (!s) return bad(c, "orbhip_window_best_set: unknown set");
    canned_window(s->desc.data(), s->n, q, qdesc, nq, best_idx, best_dist);
    return ORBHIP_OK;
}
int orbhip_search_by_projection(orbhip_ctx *c, const orbhip_keypoint *, const uint8_t *desc, int n, const float *, const uint8_t *occupied,
                                float, float, float, float, const orbhip_proj_query *q, const uint8_t *qdesc, int nq, int, float, int, int,
                                int32_t *match, int *nmatches)
{
    if (!c || !match || !nmatches) return bad(c, "orbhip_search_by_projection");
    Guard g(c);
    std::vector<int32_t> bi(nq), bd(nq);
    canned_window(desc, n, q, qdesc, nq, bi.data(), bd.data());
    for (int i = 0; i < n; i++) match[i] = -1;
    int nm = 0;
    for (int k = 0; k < nq; k++) {
        const int i = bi[k];
        if (i < 0 || bd[k] > 60 || match[i] >= 0 || (occupied && occupied[i])) continue;
        match[i] = k;
        nm++;
    }
    *nmatches = nm;
    return ORBHIP_OK;
}
int orbhip_search_for_triangulation(orbhip_ctx *c, const orbhip_keypoint *, const uint8_t *desc1, int n1, const uint8_t *skip1, const float *,
                                    const int32_t *, const int32_t *, const int32_t *, int, const orbhip_keypoint *, const uint8_t *desc2, int n2,
                                    const uint8_t *skip2, const float *, const int32_t *, const int32_t *, const int32_t *, int, const float *,
                                    float, float, const float *, const float *, int, int only_stereo, int check_ori, int32_t *match12,
                                    int *nmatches)
{
    if (!c || !match12 || !nmatches) return bad(c, "orbhip_search_for_triangulation");
    Guard g(c);
    std::vector<int32_t> m21(n2 > 0 ? n2 : 1);
    std::vector<uint8_t> v1(n1 > 0 ? n1 : 1), v2(n2 > 0 ? n2 : 1);
    for (int i = 0; i < n1; i++) v1[i] = !skip1[i];
    for (int i = 0; i < n2; i++) v2[i] = !skip2[i];
    *nmatches = (n1 && n2) ? canned_bow(desc1, n1, v1.data(), desc2, n2, v2.data(), 2 + only_stereo, 0.6f, check_ori, match12, m21.data()) : 0;
    return ORBHIP_OK;
}

// ---- what the host library references but the matcher's schedule never reaches: clear refusals ----
int orbhip_tables(int, float, int, int, int, float *, float *, float *, float *, int32_t *, int32_t *) { return ORBHIP_E_ARG; }
int orbhip_max_keypoints(const orbhip_ctx *) { return 0; }
int orbhip_extract(orbhip_ctx *c, const uint8_t *, int, int, int, orbhip_keypoint *, uint8_t *, int, int *, float *) { return bad(c, "mock: no extraction"); }
int orbhip_frame_build(orbhip_ctx *c, const uint8_t *, int, int, int, const orbhip_frame_params *, orbhip_keypoint *, orbhip_keypoint *, uint8_t *,
                       int, int *, int32_t *, int32_t *, int32_t *, float *, int32_t *) { return bad(c, "mock: no frame builds"); }
int orbhip_get_pyramid_level(orbhip_ctx *c, int, int, uint8_t *, int, int *, int *) { return bad(c, "mock"); }
int orbhip_get_stage_times(orbhip_ctx *c, float *) { return bad(c, "mock"); }
int orbhip_set_host_pyramid(orbhip_ctx *c, int) { return bad(c, "mock"); }
int orbhip_host_pyramid_level(orbhip_ctx *c, int, int, const uint8_t **, int *, int *, int *) { return bad(c, "mock"); }
int orbhip_features_in_area(orbhip_ctx *c, const orbhip_keypoint *, int, float, float, float, float, const orbhip_proj_query *, int, int32_t *,
                            int32_t *, int) { return bad(c, "mock"); }
int orbhip_search_for_initialization(orbhip_ctx *c, const orbhip_keypoint *, const uint8_t *, int, const orbhip_keypoint *, const uint8_t *, int,
                                     float, float, float, float, float *, int, float, int, int32_t *, int *) { return bad(c, "mock"); }
int orbhip_distinctive_descriptors(orbhip_ctx *c, const uint8_t *, const int32_t *, int, int32_t *, int32_t *) { return bad(c, "mock"); }
int orbhip_stereo_match(orbhip_ctx *c, orbhip_ctx *, const orbhip_keypoint *, const uint8_t *, int, const orbhip_keypoint *, const uint8_t *, int,
                        float, float, float *, float *, int *) { return bad(c, "mock"); }
int orbhip_undistort_keypoints(orbhip_ctx *c, const orbhip_keypoint *, int, const float *, const float *, int, const float *, orbhip_keypoint *)
{ return bad(c, "mock"); }
int orbhip_vocab_load(orbhip_ctx *c, const void *, size_t) { return bad(c, "mock"); }
int orbhip_vocab_share(orbhip_ctx *c, const orbhip_ctx *) { return bad(c, "mock"); }
unsigned long long orbhip_vocab_generation(const orbhip_ctx *) { return 0; }
int orbhip_vocab_info(const orbhip_ctx *, int *, int *, int *, int *, int *, int *) { return ORBHIP_E_ARG; }
int orbhip_vocab_text_to_binary(const char *, size_t, void *, size_t, size_t *, double *, size_t) { return ORBHIP_E_ARG; }
int orbhip_vocab_transform(orbhip_ctx *c, const uint8_t *, int, int, int32_t *, float *, int32_t *) { return bad(c, "mock"); }
}
