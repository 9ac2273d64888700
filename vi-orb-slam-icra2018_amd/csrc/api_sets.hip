// api_sets.hip -- C ABI, part 7: resident feature sets.  A key frame's matching data -- descriptors, keypoint records, the
// FeatureVector as CSR, the 64 x 48 feature grid -- does not change after the key frame is created, yet every per-call
// matcher entry point of Tracking / LocalMapping (SearchByBoW(KF, F), Fuse, SearchBySim3, ...; ref: src/ORBmatcher.cc:159-288,
// 825-975) used to upload it again: a call moved ~100 KB around a 10-25 us kernel and lost to one host core.  A set keeps
// that data on the device under a caller-chosen 64-bit key (the drop-in classes use KeyFrame::mnId / Frame::mnId); the
// *_sets entry points then upload only what changes from call to call (validity masks, the node pairs, the projected
// points) and get their results through page-locked memory.  At most ORB_MAX_SETS sets per context, least recently used out.
#include "api_common.h"

#define ORB_MAX_SETS 96

// A set's device block: keypoints | descriptors | count | grid offsets | grid entries | CSR offsets | CSR indices.  The first
// five parts are laid out like the result block of orbhip_frame_build (api_frame.hip), so that a frame becomes a set with
// one device-to-device copy.  Blocks and their page-locked CSR staging are pooled: a frame per camera image enters and
// leaves the table, and a hipMalloc / hipHostMalloc pair per frame would cost more than the search it serves.
struct OrbSetMem {
    uint8_t *block = nullptr;          // device
    uint8_t *h_csr = nullptr;          // page-locked: CSR offsets | CSR indices on their way to the device
    int capN = 0;
    size_t bytes = 0, oK = 0, oD = 0, oC = 0, oG = 0, oE = 0, oO = 0, oI = 0, csrBytes = 0;
};
struct OrbSet {
    uint64_t key = 0, fingerprint = 0;
    int n = 0, ng = 0;
    bool grid = false;
    float minX = 0, minY = 0, invW = 0, invH = 0;
    OrbSetMem mem;
    orbhip_keypoint *d_kps = nullptr;
    uint8_t *d_desc = nullptr;
    int32_t *d_off = nullptr, *d_idx = nullptr, *d_cellOff = nullptr, *d_cellIdx = nullptr, *d_cnt = nullptr;
    // host copies of what the host side of a search walks (merge of the node lists, rotation histogram)
    std::vector<float> angle;
    std::vector<int32_t> node, off, idx;
    unsigned long stamp = 0;
};

struct OrbSetTable {
    int limit = ORB_MAX_SETS;          // orbhip_set_limit
    std::vector<OrbSet *> sets;
    std::vector<OrbSetMem> pool;       // blocks of evicted / replaced sets
    unsigned long clock = 0;
};

static OrbSetTable *table(orbhip_ctx *c)
{
    if (!c->setTable) c->setTable = new OrbSetTable();
    return static_cast<OrbSetTable *>(c->setTable);
}

static void mem_free(OrbSetMem &m)
{
    if (m.block) (void)hipFree(m.block);
    if (m.h_csr) (void)hipHostFree(m.h_csr);
    m = OrbSetMem();
}

static void mem_layout(OrbSetMem &m, int capN)
{
    size_t o = 0;
    auto carve = [&](size_t bytes) { const size_t at = o; o = align_up(o + bytes, 256); return at; };
    m.capN = capN;
    m.oK = carve((size_t)capN * sizeof(orbhip_keypoint));
    m.oD = carve((size_t)capN * 32 + 32);
    m.oC = carve(16);
    m.oG = carve((ORBHIP_GRID_CELLS + 1) * 4);
    m.oE = carve((size_t)capN * 4);
    m.oO = carve((size_t)(capN + 1) * 4);
    m.oI = carve((size_t)capN * 4 + 4);
    m.bytes = o;
    m.csrBytes = m.bytes - m.oO;
}

// a set leaves the table: its memory goes to the pool (the caller has made sure no queued kernel reads it)
static void set_retire(OrbSetTable *T, OrbSet *s)
{
    if (s->mem.block) {
        if (T->pool.size() < 8)
            T->pool.push_back(s->mem);
        else
            mem_free(s->mem);
    }
    delete s;
}

void orb_sets_release(orbhip_ctx *c)
{
    if (!c->setTable) return;
    OrbSetTable *T = static_cast<OrbSetTable *>(c->setTable);
    for (OrbSet *s : T->sets) {
        mem_free(s->mem);
        delete s;
    }
    for (OrbSetMem &m : T->pool) mem_free(m);
    delete T;
    c->setTable = nullptr;
}

static OrbSet *find_set(orbhip_ctx *c, uint64_t key)
{
    if (!c->setTable) return nullptr;
    OrbSetTable *T = static_cast<OrbSetTable *>(c->setTable);
    for (OrbSet *s : T->sets)
        if (s->key == key) {
            s->stamp = ++T->clock;
            return s;
        }
    return nullptr;
}

extern "C" int orbhip_set_has(orbhip_ctx *c, uint64_t key, int n)
{
    if (!c) return 0;
    OrbSet *s = find_set(c, key);
    return s && s->n == n ? 1 : 0;
}

extern "C" int orbhip_set_info(orbhip_ctx *c, uint64_t key, int *n, int *ng, uint64_t *fingerprint)
{
    if (!c) return 0;
    OrbSet *s = find_set(c, key);
    if (!s) return 0;
    if (n) *n = s->n;
    if (ng) *ng = s->ng;
    if (fingerprint) *fingerprint = s->fingerprint;
    return 1;
}

// At most `max_sets` resident sets in this context from now on (clamped to [4, ORB_MAX_SETS]; a search holds two at a time); the
// least recently used ones beyond it leave at the next orbhip_set_put.  Returns the limit in force.
extern "C" int orbhip_set_limit(orbhip_ctx *c, int max_sets)
{
    if (!c) return 0;
    OrbSetTable *T = table(c);
    T->limit = max_sets < 4 ? 4 : max_sets > ORB_MAX_SETS ? ORB_MAX_SETS : max_sets;
    return T->limit;
}

extern "C" int orbhip_set_drop(orbhip_ctx *c, uint64_t key)
{
    if (!c) return ORBHIP_E_ARG;
    if (!c->setTable) return ORBHIP_OK;
    HIPCHK(c, orb_enter(c));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    OrbSetTable *T = static_cast<OrbSetTable *>(c->setTable);
    for (size_t i = 0; i < T->sets.size();)
        if (key == 0 || T->sets[i]->key == key) {
            set_retire(T, T->sets[i]);
            T->sets.erase(T->sets.begin() + i);
        } else
            i++;
    return ORBHIP_OK;
}

// Makes room for `key` with `n` features: a set of the same key and the least recently used one of a full table are retired
// (after the stream has drained: a queued kernel may still read them), memory comes from the pool when a block there is
// large enough.  Returns the new, empty set or nullptr (c->err set).
static OrbSet *set_acquire(orbhip_ctx *c, uint64_t key, int n, int capWant = 0)
{
    OrbSetTable *T = table(c);
    bool drained = false;
    auto drain = [&]() {
        if (!drained && hipStreamSynchronize(c->stream) != hipSuccess) return false;
        drained = true;
        return true;
    };
    for (size_t i = 0; i < T->sets.size(); i++)
        if (T->sets[i]->key == key) {
            if (!drain()) return fail(c, ORBHIP_E_HIP, "orbhip_set_put: stream synchronisation failed"), nullptr;
            set_retire(T, T->sets[i]);
            T->sets.erase(T->sets.begin() + i);
            break;
        }
    while (T->sets.size() >= (size_t)T->limit) {
        size_t lru = 0;
        for (size_t i = 1; i < T->sets.size(); i++)
            if (T->sets[i]->stamp < T->sets[lru]->stamp) lru = i;
        if (!drain()) return fail(c, ORBHIP_E_HIP, "orbhip_set_put: stream synchronisation failed"), nullptr;
        set_retire(T, T->sets[lru]);
        T->sets.erase(T->sets.begin() + lru);
    }
    OrbSet *s = new OrbSet();
    // (capWant: a block of exactly that capacity -- the layout of a frame's result block, one copy fills it)
    int best = -1;
    for (size_t i = 0; i < T->pool.size(); i++)
        if (capWant > 0 ? T->pool[i].capN == capWant
                        : (T->pool[i].capN >= n && (best < 0 || T->pool[i].capN < T->pool[best].capN)))
            best = (int)i;
    if (best >= 0) {
        s->mem = T->pool[best];
        T->pool.erase(T->pool.begin() + best);
    } else {
        // capacity in steps of 256 features, so that the frames of a sequence (whose counts differ by a few) share blocks
        mem_layout(s->mem, capWant > 0 ? capWant : (int)align_up((size_t)n + 8, 256));
        void *p = nullptr;
        if (hipMalloc(&p, s->mem.bytes) != hipSuccess) {
            (void)hipGetLastError();
            delete s;
            return fail(c, ORBHIP_E_HIP, "orbhip_set_put: out of device memory"), nullptr;
        }
        s->mem.block = (uint8_t *)p;
        if (hipHostMalloc(&p, s->mem.csrBytes, hipHostMallocDefault) != hipSuccess) {
            (void)hipGetLastError();
            mem_free(s->mem);
            delete s;
            return fail(c, ORBHIP_E_HIP, "orbhip_set_put: out of page-locked memory"), nullptr;
        }
        s->mem.h_csr = (uint8_t *)p;
    }
    const OrbSetMem &m = s->mem;
    s->key = key;
    s->n = n;
    s->d_kps = (orbhip_keypoint *)(m.block + m.oK);
    s->d_desc = m.block + m.oD;
    s->d_cnt = (int32_t *)(m.block + m.oC);
    s->d_cellOff = (int32_t *)(m.block + m.oG);
    s->d_cellIdx = (int32_t *)(m.block + m.oE);
    s->d_off = (int32_t *)(m.block + m.oO);
    s->d_idx = (int32_t *)(m.block + m.oI);
    return s;
}

static int csr_check(orbhip_ctx *c, const char *who, const int32_t *node, const int32_t *off, const int32_t *idx, int ng, int n)
{
    const int m = ng > 0 ? off[ng] : 0;
    if (ng > 0 && (off[0] != 0 || m > n)) return fail(c, ORBHIP_E_ARG, std::string(who) + ": the FeatureVector holds more entries than the set has features");
    for (int g = 0; g < ng; g++)
        if (off[g] > off[g + 1] || off[g] < 0 || (g > 0 && node[g] <= node[g - 1]))
            return fail(c, ORBHIP_E_ARG, std::string(who) + ": the FeatureVector must be a CSR over ascending node ids");
    for (int t = 0; t < m; t++)
        if (idx[t] < 0 || idx[t] >= n) return fail(c, ORBHIP_E_ARG, std::string(who) + ": feature index out of range");
    return ORBHIP_OK;
}

// the FeatureVector of a new set: host copies for the merge walk, device copy through the set's own page-locked staging
// (asynchronous: every later use of the set is behind it on the context's stream)
static hipError_t csr_upload(orbhip_ctx *c, OrbSet *s, const int32_t *node, const int32_t *off, const int32_t *idx, int ng)
{
    const OrbSetMem &m = s->mem;
    const int cnt = ng > 0 ? off[ng] : 0;
    s->ng = ng;
    if (ng > 0) {
        s->node.assign(node, node + ng);
        s->off.assign(off, off + ng + 1);
        s->idx.assign(idx, idx + cnt);
        memcpy(m.h_csr, off, (size_t)(ng + 1) * 4);
        memcpy(m.h_csr + (m.oI - m.oO), idx, (size_t)cnt * 4);
    } else {
        memset(m.h_csr, 0, 4);
    }
    return hipMemcpyAsync(m.block + m.oO, m.h_csr, (m.oI - m.oO) + (size_t)cnt * 4 + 4, hipMemcpyHostToDevice, c->stream);
}

extern "C" int orbhip_set_put(orbhip_ctx *c, uint64_t key, const orbhip_keypoint *kps, const uint8_t *desc, int n,
                              const int32_t *node, const int32_t *off, const int32_t *idx, int ng, float min_x, float min_y,
                              float inv_w, float inv_h)
{
    if (!c || key == 0 || n <= 0 || !kps || !desc || ng < 0 || ng > n || (ng > 0 && (!node || !off || !idx)))
        return fail(c, ORBHIP_E_ARG, "orbhip_set_put: bad argument");
    int rc;
    if ((rc = csr_check(c, "orbhip_set_put", node, off, idx, ng, n))) return rc;
    HIPCHK(c, orb_enter(c));
    OrbSet *s = set_acquire(c, key, n);
    if (!s) return ORBHIP_E_HIP;
    OrbSetTable *T = table(c);
    s->fingerprint = orbhip_set_fingerprint(kps, desc, n);
    s->grid = inv_w > 0.f && inv_h > 0.f;
    s->minX = min_x; s->minY = min_y; s->invW = inv_w; s->invH = inv_h;
    const OrbSetMem &m = s->mem;
    // keypoints | descriptors | count: one packed upload
    Packed P(c);
    if ((rc = P.begin(m.oG + 4096))) {
        set_retire(T, s);
        return rc;
    }
    memcpy(P.h + m.oK, kps, (size_t)n * sizeof(orbhip_keypoint));
    memcpy(P.h + m.oD, desc, (size_t)n * 32);
    const int32_t cnt[4] = {n, 0, 0, 0};
    memcpy(P.h + m.oC, cnt, 16);
    hipError_t e = hipMemcpyAsync(m.block, P.h, m.oC + 16, hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess) e = csr_upload(c, s, node, off, idx, ng);
    if (e == hipSuccess && s->grid) {
        launch_grid_build(c->stream, s->d_kps, s->d_cnt, m.capN, 1, min_x, min_y, inv_w, inv_h, s->d_cellOff, s->d_cellIdx);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);   // (P's page-locked block is reused by the next call)
    if (e != hipSuccess) {
        (void)hipStreamSynchronize(c->stream);
        set_retire(T, s);
        return fail(c, ORBHIP_E_HIP, std::string("orbhip_set_put: ") + hipGetErrorString(e));
    }
    s->angle.resize(n);
    for (int i = 0; i < n; i++) s->angle[i] = kps[i].angle;
    s->stamp = ++T->clock;
    T->sets.push_back(s);
    return ORBHIP_OK;
}

// api_frame.hip
int orb_frame_block(orbhip_ctx *src, const uint8_t **d_blk, size_t *setBytes, size_t off[5], int *n, int *dcap, bool *grid,
                    float gp[4], const orbhip_keypoint **h_kps_un, const uint8_t **h_desc);
int orb_frame_mark_busy(orbhip_ctx *src, hipStream_t copier);

// The frame that `src` built last (orbhip_frame_build) becomes the resident set `key` of `c`: undistorted keypoints,
// descriptors, count and grid go from device block to device block; only the FeatureVector (built on the host from the
// transform's node ids: DBoW2::FeatureVector is an ordered map) travels.  No synchronisation: every later use of the set is
// behind the copies on c's stream, and src's next orbhip_frame_build waits for them.
extern "C" int orbhip_set_put_from_frame(orbhip_ctx *c, uint64_t key, orbhip_ctx *src, const int32_t *node, const int32_t *off,
                                         const int32_t *idx, int ng)
{
    if (!c || !src || key == 0 || ng < 0 || (ng > 0 && (!node || !off || !idx)))
        return fail(c, ORBHIP_E_ARG, "orbhip_set_put_from_frame: bad argument");
    if (c->device != src->device) return fail(c, ORBHIP_E_ARG, "orbhip_set_put_from_frame: the two contexts are on different devices");
    const uint8_t *blk = nullptr, *hdesc = nullptr;
    const orbhip_keypoint *hkps = nullptr;
    size_t setBytes = 0, so[5];
    int n = 0, dcap = 0;
    bool grid = false;
    float gp[4];
    if (orb_frame_block(src, &blk, &setBytes, so, &n, &dcap, &grid, gp, &hkps, &hdesc) != ORBHIP_OK)
        return fail(c, ORBHIP_E_ARG, "orbhip_set_put_from_frame: the source context holds no frame (orbhip_frame_build)");
    if (ng > n) return fail(c, ORBHIP_E_ARG, "orbhip_set_put_from_frame: bad argument");
    int rc;
    if ((rc = csr_check(c, "orbhip_set_put_from_frame", node, off, idx, ng, n))) return rc;
    HIPCHK(c, orb_enter(c));
    OrbSet *s = set_acquire(c, key, n, dcap);
    if (!s) return ORBHIP_E_HIP;
    OrbSetTable *T = table(c);
    const OrbSetMem &m = s->mem;
    s->fingerprint = orbhip_set_fingerprint(hkps, hdesc, n);
    s->grid = grid;
    s->minX = gp[0]; s->minY = gp[1]; s->invW = gp[2]; s->invH = gp[3];
    // the two blocks carve the same parts in the same order, each for its own capacity: part by part
    const size_t part[5] = {(size_t)n * sizeof(orbhip_keypoint), (size_t)n * 32, 16, grid ? (size_t)(ORBHIP_GRID_CELLS + 1) * 4 : 0,
                            grid ? (size_t)n * 4 : 0};
    const size_t to[5] = {m.oK, m.oD, m.oC, m.oG, m.oE};
    hipError_t e = hipSuccess;
    if (m.capN == dcap) {
        e = hipMemcpyAsync(m.block, blk, grid ? so[4] + part[4] : so[2] + 16, hipMemcpyDeviceToDevice, c->stream);   // identical layouts: one copy
    } else {
        for (int k = 0; k < 5 && e == hipSuccess; k++)
            if (part[k]) e = hipMemcpyAsync(m.block + to[k], blk + so[k], part[k], hipMemcpyDeviceToDevice, c->stream);
    }
    if (e == hipSuccess) e = csr_upload(c, s, node, off, idx, ng);
    if (e == hipSuccess && orb_frame_mark_busy(src, c->stream) != ORBHIP_OK) e = hipErrorUnknown;
    if (e != hipSuccess) {
        (void)hipStreamSynchronize(c->stream);
        set_retire(T, s);
        return fail(c, ORBHIP_E_HIP, std::string("orbhip_set_put_from_frame: ") + hipGetErrorString(e));
    }
    s->angle.resize(n);
    for (int i = 0; i < n; i++) s->angle[i] = hkps[i].angle;
    s->stamp = ++T->clock;
    T->sets.push_back(s);
    return ORBHIP_OK;
}

// SearchByBoW between two resident sets (ref: src/ORBmatcher.cc:159-288 with th_mode 0, :522-655 with th_mode 1): what travels
// per call is the validity masks and the list of shared nodes; the matches come back through page-locked memory.
extern "C" int orbhip_search_by_bow_sets(orbhip_ctx *c, uint64_t key1, const uint8_t *valid1, uint64_t key2,
                                         const uint8_t *valid2, int th, int th_mode, float nnratio, int check_ori,
                                         int32_t *match12, int32_t *match21, int *nmatches)
{
    if (!c || !valid1 || !match12 || !match21 || !nmatches) return fail(c, ORBHIP_E_ARG, "orbhip_search_by_bow_sets: bad argument");
    OrbSet *s1 = find_set(c, key1), *s2 = find_set(c, key2);
    if (!s1 || !s2) return fail(c, ORBHIP_E_ARG, "orbhip_search_by_bow_sets: unknown set (orbhip_set_put)");
    const int n1 = s1->n, n2 = s2->n;
    for (int i = 0; i < n1; i++) match12[i] = -1;
    for (int i = 0; i < n2; i++) match21[i] = -1;
    *nmatches = 0;
    std::vector<int32_t> pairs;
    for (int g1 = 0, g2 = 0; g1 < s1->ng && g2 < s2->ng;) {     // merge walk over the two FeatureVectors (ref: :180-264)
        if (s1->node[g1] == s2->node[g2]) {
            pairs.push_back(g1++);
            pairs.push_back(g2++);
        } else if (s1->node[g1] < s2->node[g2])
            g1++;
        else
            g2++;
    }
    const int npairs = (int)pairs.size() / 2;
    if (npairs == 0) return ORBHIP_OK;
    // (k_bow_match keeps the position inside a node's side-2 list in 20 bits, as in orbhip_search_by_bow)
    if (s2->ng > 0 && s2->off[s2->ng] >= (1 << 20))
        return fail(c, ORBHIP_E_SIZE, "orbhip_search_by_bow_sets: more than 2^20 - 1 entries in the second FeatureVector");
    HIPCHK(c, orb_enter(c));
    Packed P(c);
    int rc;
    if ((rc = P.begin((size_t)n1 + (size_t)n2 + pairs.size() * 4 + (size_t)(n1 + n2) * 4 + 12 * 256))) return rc;
    bool hostOut = true;      // (as in orbhip_search_by_bow: nodes of more than 128 side-2 features poll match21 on the device)
    for (int p = 0; p < npairs && hostOut; p++)
        if (s2->off[pairs[2 * p + 1] + 1] - s2->off[pairs[2 * p + 1]] > 128) hostOut = false;
    // the masks and the node pairs: read by the kernel straight from the page-locked block (no copy command) when the
    // matches go there too; a frame with a node of more than 128 features keeps everything on the device
    static const bool zeroCopy = ORB_TUNE("SETS_COPY", 0) == 0;
    const bool hostIn = hostOut && zeroCopy;
    const uint8_t *dv1 = (const uint8_t *)(hostIn ? P.in_host(valid1, (size_t)n1) : P.in(valid1, (size_t)n1));
    const uint8_t *dv2 = !valid2 ? nullptr : (const uint8_t *)(hostIn ? P.in_host(valid2, (size_t)n2) : P.in(valid2, (size_t)n2));
    const int32_t *dp = (const int32_t *)(hostIn ? P.in_host(pairs.data(), pairs.size() * 4) : P.in(pairs.data(), pairs.size() * 4));
    int32_t *dm12, *dm21;
    if (hostOut) {
        dm12 = (int32_t *)P.out_host_fill(0xFF, (size_t)n1 * 4);
        dm21 = (int32_t *)P.out_host_fill(0xFF, (size_t)n2 * 4);
    } else {
        dm12 = (int32_t *)P.in_fill(0xFF, (size_t)n1 * 4);
        dm21 = (int32_t *)P.in_fill(0xFF, (size_t)n2 * 4);
    }
    if ((rc = P.upload())) return rc;
    launch_bow_match(c->stream, s1->d_desc, dv1, s1->d_off, s1->d_idx, s2->d_desc, dv2, s2->d_off, s2->d_idx, dp, npairs, th, th_mode,
                     nnratio, dm12, dm21);
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
    *nmatches = orb_bow_rotation_check(pairs.data(), npairs, s1->off.data(), s1->idx.data(), s1->angle.data(), s2->angle.data(),
                                       check_ori, match12, match21);
    return ORBHIP_OK;
}

// The per-point window search of Fuse / SearchBySim3 (ref: src/ORBmatcher.cc:825-975, 1102-1326) into a resident key frame:
// its keypoints, descriptors and grid stay on the device, the projected points travel.
extern "C" int orbhip_window_best_set(orbhip_ctx *c, uint64_t key, const float *u_right, const float *inv_level_sigma2, int nlevels,
                                      const orbhip_proj_query *queries, const uint8_t *qdesc, int nq, int32_t *best_idx,
                                      int32_t *best_dist)
{
    if (!c || nq < 0 || (nq > 0 && (!queries || !qdesc || !best_idx || !best_dist)) ||
        (inv_level_sigma2 && (nlevels <= 0 || nlevels > 16)))
        return fail(c, ORBHIP_E_ARG, "orbhip_window_best_set: bad argument");
    OrbSet *s = find_set(c, key);
    if (!s || !s->grid) return fail(c, ORBHIP_E_ARG, "orbhip_window_best_set: unknown set, or a set without a grid (orbhip_set_put)");
    for (int i = 0; i < nq; i++) {
        best_idx[i] = -1;
        best_dist[i] = 256;
    }
    if (nq == 0) return ORBHIP_OK;
    HIPCHK(c, orb_enter(c));
    Packed P(c);
    int rc;
    const int n = s->n;
    if ((rc = P.begin((size_t)n * 4 + (size_t)nq * (sizeof(orbhip_proj_query) + 32 + 8) + 12 * 256))) return rc;
    const int32_t cnts[4] = {nq, 0, 0, 0};
    const float *dur = u_right ? (const float *)P.in(u_right, (size_t)n * 4) : nullptr;
    const int32_t *dc = (const int32_t *)P.in(cnts, 16);
    const orbhip_proj_query *dq = (const orbhip_proj_query *)P.in(queries, (size_t)nq * sizeof(orbhip_proj_query));
    const uint8_t *dqd = (const uint8_t *)P.in(qdesc, (size_t)nq * 32);
    int32_t *dbi = (int32_t *)P.out_host((size_t)nq * 4), *dbd = (int32_t *)P.out_host((size_t)nq * 4);   // written over PCIe, no copy back
    if ((rc = P.upload())) return rc;
    if ((rc = orbhip_window_best_device(c, s->d_kps, s->d_desc, n, 1, dur, inv_level_sigma2, nlevels, s->minX, s->minY, s->invW, s->invH,
                                        s->d_cellOff, s->d_cellIdx, dq, dqd, dc, nq, dbi, dbd)))
        return rc;
    HIPCHK(c, hipStreamSynchronize(c->stream));
    memcpy(best_idx, dbi, (size_t)nq * 4);
    memcpy(best_dist, dbd, (size_t)nq * 4);
    return ORBHIP_OK;
}
