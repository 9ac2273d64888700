// orbhip_api.hip -- the C ABI of liborbhip.so (declared in include/orbhip.h): context and buffer
// management, the per-batch launch sequence, host staging, parity/debug read-back, the host side of
// SearchByBoW (merge walk over the two FeatureVectors, rotation histogram) and the RCCL
// broadcast.  No CPU fallback anywhere: every compute step is a HIP kernel.
#include "orbhip_internal.h"

#include <dlfcn.h>
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <mutex>

static std::string g_last_error;
static std::mutex g_err_mutex;

static void comm_release(orbhip_ctx *c);
static void pipe_release(orbhip_ctx *c);
static void graph_release(orbhip_ctx *c);

static int fail(orbhip_ctx *c, int code, const std::string &msg)
{
    if (c)
        c->err = msg;
    else {
        std::lock_guard<std::mutex> lk(g_err_mutex);
        g_last_error = msg;
    }
    return code;
}

#define HIPCHK(c, expr)                                                                              \
    do {                                                                                             \
        hipError_t e_ = (expr);                                                                      \
        if (e_ != hipSuccess)                                                                        \
            return fail((c), ORBHIP_E_HIP, std::string(#expr) + ": " + hipGetErrorString(e_));       \
    } while (0)

static inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

template <class T>
static int ensure(orbhip_ctx *c, T *&ptr, size_t &cap, size_t need)
{
    if (need <= cap && ptr) return ORBHIP_OK;
    if (ptr) HIPCHK(c, hipFree(ptr));
    ptr = nullptr;
    cap = 0;
    void *p = nullptr;
    HIPCHK(c, hipMalloc(&p, need ? need : 16));
    ptr = reinterpret_cast<T *>(p);
    cap = need;
    return ORBHIP_OK;
}

extern "C" int orbhip_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

extern "C" const char *orbhip_last_error(const orbhip_ctx *ctx)
{
    if (ctx) return ctx->err.c_str();
    return g_last_error.c_str();
}

// (Re)configure for a w x h image with level-0 row stride `stride0`, batch B: geometry, tables and
// device buffers.  Cheap when nothing changed.
static int configure(orbhip_ctx *c, int w, int h, int stride0, int B)
{
    if (w <= 0 || h <= 0 || B <= 0) return fail(c, ORBHIP_E_ARG, "bad image size or batch");
    if (w > c->max_w || h > c->max_h || B > c->max_batch)
        return fail(c, ORBHIP_E_SIZE, "image or batch larger than the context was created for");
    const bool geomChanged = (w != c->cur_w || h != c->cur_h);
    if (geomChanged) {
        int rc = orb_build_geometry(c, w, h, stride0);
        if (rc != ORBHIP_OK) {
            c->cur_w = c->cur_h = 0;
            return fail(c, rc, "image too small for the 30-px cell grid / quadtree roots of some level, "
                               "or larger than the supported tile bounds");
        }
        // resize tables
        std::vector<int32_t> all;
        bool chainOk[ORBHIP_MAX_LEVELS] = {};
        for (int l = 1; l < c->nlevels; l++) {
            std::vector<int32_t> xt, yt;
            orb_build_resize_tables(c->G.lv[l - 1].w, c->G.lv[l - 1].h, c->G.lv[l].w, c->G.lv[l].h, xt, yt);
            while (all.size() % 4) all.push_back(0);
            c->resizeTabOff[l][0] = all.size();
            all.insert(all.end(), xt.begin(), xt.end());
            while (all.size() % 4) all.push_back(0);
            c->resizeTabOff[l][1] = all.size();
            all.insert(all.end(), yt.begin(), yt.end());
            std::vector<int32_t> gt;
            c->resizeGroups[l] = orb_build_resize_groups(xt, yt, c->G.lv[l].w, gt);
            c->resizeHint[l][0] = resize_hint_fits(xt.data(), yt.data(), c->G.lv[l - 1].w, c->G.lv[l - 1].h, c->G.lv[l].w, c->G.lv[l].h, 32);
            c->resizeHint[l][1] = resize_hint_fits(xt.data(), yt.data(), c->G.lv[l - 1].w, c->G.lv[l - 1].h, c->G.lv[l].w, c->G.lv[l].h, 8);
            while (all.size() % 4) all.push_back(0);
            c->resizeTabOff[l][2] = all.size();
            if (c->resizeGroups[l]) all.insert(all.end(), gt.begin(), gt.end());
            // chained pyramid of the single-frame path (k_pyramid_chain)
            chainOk[l] = resize_hint_pointwise(xt.data(), yt.data(), c->G.lv[l - 1].w, c->G.lv[l - 1].h, c->G.lv[l].w, c->G.lv[l].h);
            c->chainLevels.winx[l] = (float)c->G.lv[l - 1].w / (float)c->G.lv[l].w;
            c->chainLevels.winy[l] = (float)c->G.lv[l - 1].h / (float)c->G.lv[l].h;
            c->chainLevels.xoff[l] = (uint32_t)c->resizeTabOff[l][0];
            c->chainLevels.yoff[l] = (uint32_t)c->resizeTabOff[l][1];
        }
        if (c->nlevels < 2 || !chain_plan(c->G, chainOk, c->chainLevels, c->chainTiles, c->chainGroups)) {
            c->chainTiles.clear();
            c->chainGroups.clear();
        }
        int rc2;
        if ((rc2 = ensure(c, c->d_resizeTab, c->cap_resize, all.size() * 4 + 16))) return rc2;
        if ((rc2 = ensure(c, c->d_fastTiles, c->cap_fastTiles, c->fastTiles.size() * sizeof(FastTile)))) return rc2;
        if ((rc2 = ensure(c, c->d_blurTiles, c->cap_blurTiles, c->blurTiles.size() * sizeof(BlurTile)))) return rc2;
        if ((rc2 = ensure(c, c->d_chainTiles, c->cap_chainTiles, c->chainTiles.size() * sizeof(ChainTile) + 16))) return rc2;
        if (!c->chainTiles.empty())
            HIPCHK(c, hipMemcpyAsync(c->d_chainTiles, c->chainTiles.data(), c->chainTiles.size() * sizeof(ChainTile),
                                     hipMemcpyHostToDevice, c->stream));
        HIPCHK(c, hipMemcpyAsync(c->d_resizeTab, all.data(), all.size() * 4, hipMemcpyHostToDevice, c->stream));
        HIPCHK(c, hipMemcpyAsync(c->d_fastTiles, c->fastTiles.data(), c->fastTiles.size() * sizeof(FastTile),
                                 hipMemcpyHostToDevice, c->stream));
        HIPCHK(c, hipMemcpyAsync(c->d_blurTiles, c->blurTiles.data(), c->blurTiles.size() * sizeof(BlurTile),
                                 hipMemcpyHostToDevice, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));  // the host vectors above go out of scope
    }
    c->G.lv[0].stride = stride0;
    const OrbLevels &G = c->G;
    const size_t Bm = (size_t)c->max_batch;  // buffers are sized for the context's batch once
    int rc;
    if ((rc = ensure(c, c->d_pyr, c->cap_pyr, Bm * c->pyrFrameBytes))) return rc;
    if ((rc = ensure(c, c->d_blur, c->cap_blur, Bm * (c->lvl0FrameBytes + c->pyrFrameBytes)))) return rc;
    if ((rc = ensure(c, c->d_cand, c->cap_cand, Bm * (size_t)G.totalCands * 4))) return rc;
    if ((rc = ensure(c, c->d_cellCnt, c->cap_cells, Bm * (size_t)G.totalCells * 2 + 64))) return rc;
    if ((rc = ensure(c, c->d_pts, c->cap_pts, Bm * (size_t)G.totalPts * 4))) return rc;
    if ((rc = ensure(c, c->d_pnode, c->cap_pnode, Bm * (size_t)G.totalPts * 4))) return rc;
    if ((rc = ensure(c, c->d_lvlKp, c->cap_kps, Bm * (size_t)G.totalKps * 4))) return rc;
    if ((rc = ensure(c, c->d_lvlAngle, c->cap_angle, Bm * (size_t)G.totalKps * 4))) return rc;
    if ((rc = ensure(c, c->d_lvlCandCnt, c->cap_cnt1, Bm * ORBHIP_MAX_LEVELS * 4))) return rc;
    if ((rc = ensure(c, c->d_lvlKpCnt, c->cap_cnt2, Bm * ORBHIP_MAX_LEVELS * 4))) return rc;
    if ((rc = ensure(c, c->d_lvl0, c->cap_lvl0, Bm * c->lvl0FrameBytes))) return rc;
    // quadtree node tables of levels whose feature quota exceeds what the LDS holds (about 2000 features on one level)
    if (const size_t qt = quadtree_table_scratch_bytes(G, (int)Bm))
        if ((rc = ensure(c, c->d_qtTables, c->cap_qtTables, qt))) return rc;
    if ((size_t)G.outCap > c->cap_out) {
        size_t d1 = 0, d2 = 0;
        if (c->d_kps) HIPCHK(c, hipFree(c->d_kps));
        if (c->d_desc) HIPCHK(c, hipFree(c->d_desc));
        c->d_kps = nullptr;
        c->d_desc = nullptr;
        // one block for keypoints | descriptors | counts of a call (carved per call for its B, orbhip_extract_batch):
        // the results of the host-pointer API come back in ONE device-to-host copy
        if ((rc = ensure(c, c->d_kps, d1, Bm * (size_t)G.outCap * (sizeof(orbhip_keypoint) + 32) + Bm * 4 + 1024))) return rc;
        (void)d2;
        c->cap_out = (size_t)G.outCap;
    }
    return ORBHIP_OK;
}

extern "C" orbhip_ctx *orbhip_create(int device, int nfeatures, float scaleFactor, int nlevels,
                                     int iniThFAST, int minThFAST, int max_w, int max_h, int max_batch)
{
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) {
        fail(nullptr, ORBHIP_E_NODEVICE, "no HIP device visible (liborbhip has no CPU fallback)");
        return nullptr;
    }
    if (device < 0 || device >= ndev || max_w <= 0 || max_h <= 0 || max_batch <= 0) {
        fail(nullptr, ORBHIP_E_ARG, "orbhip_create: bad device index or sizes");
        return nullptr;
    }
    orbhip_ctx *c = new orbhip_ctx();
    c->device = device;
    c->max_w = max_w;
    c->max_h = max_h;
    c->max_batch = max_batch;
    if (orb_init_tables(c, nfeatures, scaleFactor, nlevels, iniThFAST, minThFAST) != ORBHIP_OK) {
        fail(nullptr, ORBHIP_E_ARG, "orbhip_create: bad ORB parameters");
        delete c;
        return nullptr;
    }
    auto bail = [&](const char *what, hipError_t e) {
        fail(nullptr, ORBHIP_E_HIP, std::string(what) + ": " + hipGetErrorString(e));
        orbhip_destroy(c);
        return (orbhip_ctx *)nullptr;
    };
    hipError_t e;
    if ((e = hipSetDevice(device)) != hipSuccess) return bail("hipSetDevice", e);
    hipDeviceProp_t prop;
    if ((e = hipGetDeviceProperties(&prop, device)) != hipSuccess) return bail("hipGetDeviceProperties", e);
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
        fail(nullptr, ORBHIP_E_NODEVICE, std::string("device is ") + prop.gcnArchName + ", liborbhip is built for gfx950");
        orbhip_destroy(c);
        return nullptr;
    }
    if ((e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking)) != hipSuccess) return bail("hipStreamCreate", e);
    if ((e = hipStreamCreateWithFlags(&c->stream2, hipStreamNonBlocking)) != hipSuccess) return bail("hipStreamCreate", e);
    for (int i = 0; i < 3; i++)
        if ((e = hipEventCreate(&c->evx[i])) != hipSuccess) return bail("hipEventCreate", e);
    for (int i = 0; i < 2; i++)
        if ((e = hipEventCreateWithFlags(&c->evp[i], hipEventDisableTiming)) != hipSuccess) return bail("hipEventCreate", e);
    for (int i = 0; i < 8; i++)
        if ((e = hipEventCreate(&c->ev[i])) != hipSuccess) return bail("hipEventCreate", e);
    // size everything for the largest image now, so per-frame calls never allocate
    const int stride0 = (int)align_up((size_t)max_w, 64);
    int rc = configure(c, max_w, max_h, stride0, max_batch);
    if (rc != ORBHIP_OK) {
        fail(nullptr, rc, "orbhip_create: " + c->err);
        orbhip_destroy(c);
        return nullptr;
    }
    return c;
}

extern "C" void orbhip_destroy(orbhip_ctx *c)
{
    if (!c) return;
    (void)hipSetDevice(c->device);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    comm_release(c);
    pipe_release(c);
    graph_release(c);
    if (c->h_in) (void)hipHostFree(c->h_in);
    if (c->h_pyr) (void)hipHostFree(c->h_pyr);
    void *bufs[] = {c->d_lvl0, c->d_pyr, c->d_blur, c->d_cand, c->d_cellCnt, c->d_pts, c->d_pnode,
                    c->d_lvlCandCnt, c->d_lvlKp, c->d_lvlKpCnt, c->d_lvlAngle, c->d_kps, c->d_desc,
                    c->d_counts, c->d_qtTables, c->d_fastTiles, c->d_blurTiles, c->d_chainTiles, c->d_resizeTab, c->d_match, c->d_vocBlock, c->d_maps, c->d_tmp};
    for (void *b : bufs)
        if (b) (void)hipFree(b);
    if (c->h_stage) (void)hipHostFree(c->h_stage);
    if (c->h_pack) (void)hipHostFree(c->h_pack);
    for (int i = 0; i < 8; i++)
        if (c->ev[i]) (void)hipEventDestroy(c->ev[i]);
    for (int i = 0; i < 3; i++)
        if (c->evx[i]) (void)hipEventDestroy(c->evx[i]);
    for (int i = 0; i < 2; i++)
        if (c->evp[i]) (void)hipEventDestroy(c->evp[i]);
    if (c->stream2) (void)hipStreamDestroy(c->stream2);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
}

extern "C" int orbhip_sync(orbhip_ctx *c)
{
    if (!c) return ORBHIP_E_ARG;
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return ORBHIP_OK;
}

extern "C" void *orbhip_stream(orbhip_ctx *c) { return c ? (void *)c->stream : nullptr; }

extern "C" int orbhip_get_tables(const orbhip_ctx *c, int *nlevels, double *scaleFactor, float *sf,
                                 float *isf, float *s2, float *is2, int *perLevel, int *umax)
{
    if (!c) return ORBHIP_E_ARG;
    if (nlevels) *nlevels = c->nlevels;
    if (scaleFactor) *scaleFactor = c->scaleFactor;
    for (int i = 0; i < c->nlevels; i++) {
        if (sf) sf[i] = c->mvScaleFactor[i];
        if (isf) isf[i] = c->mvInvScaleFactor[i];
        if (s2) s2[i] = c->mvLevelSigma2[i];
        if (is2) is2[i] = c->mvInvLevelSigma2[i];
        if (perLevel) perLevel[i] = c->mnFeaturesPerLevel[i];
    }
    if (umax) memcpy(umax, c->umax, sizeof(int) * 16);
    return ORBHIP_OK;
}

extern "C" int orbhip_tables(int nfeatures, float scaleFactor, int nlevels, int iniThFAST, int minThFAST,
                             float *sf, float *isf, float *s2, float *is2, int *perLevel, int *umax)
{
    orbhip_ctx tmp;
    const int rc = orb_init_tables(&tmp, nfeatures, scaleFactor, nlevels, iniThFAST, minThFAST);
    if (rc != ORBHIP_OK) return fail(nullptr, rc, "orbhip_tables: bad ORB parameters");
    return orbhip_get_tables(&tmp, nullptr, nullptr, sf, isf, s2, is2, perLevel, umax);
}

extern "C" int orbhip_max_keypoints(const orbhip_ctx *c)
{
    if (!c) return ORBHIP_E_ARG;
    // independent of the image size except through nIni (<= a handful); use the create-time value
    return (int)c->cap_out;
}

extern "C" int orbhip_level_size(const orbhip_ctx *c, int w, int h, int level, int *lw, int *lh)
{
    if (!c || level < 0 || level >= c->nlevels || !lw || !lh) return ORBHIP_E_ARG;
    orb_level_size(c, w, h, level, lw, lh);
    return ORBHIP_OK;
}

// The launch sequence of one batch.  lvl0: device pointer of frame 0 / level 0.
// Stage boundaries are marked with HIP events on the context stream (ev[0..5]).
static int run_pipeline(orbhip_ctx *c, const uint8_t *lvl0, int stride0, size_t frame0, int B,
                        orbhip_keypoint *d_kps, uint8_t *d_desc, int32_t *d_counts, int cap, uint8_t *h_pyr_dst = nullptr)
{
    const OrbLevels &G = c->G;
    hipStream_t s = c->stream;
    // (timing events are not recorded into a graph capture: events recorded by a graph node cannot be read back with
    // hipEventElapsedTime on this runtime; the graph path refreshes the stage times with an eager run now and then)
    const bool ev = !c->capturing;
    if (ev) HIPCHK(c, hipEventRecord(c->ev[0], s));
    // E2 pyramid.  A frame or two: several levels per launch (k_pyramid_chain; ORBHIP_NO_CHAIN=1 keeps one launch per level)
    static const bool noChain = getenv("ORBHIP_NO_CHAIN") && atoi(getenv("ORBHIP_NO_CHAIN")) != 0;
    const bool chained = B < 8 && !noChain && !c->chainGroups.empty();
    if (chained)
        for (const ChainGroup &grp : c->chainGroups)
            launch_pyramid_chain(s, G, c->chainLevels, grp, c->d_chainTiles, lvl0, stride0, frame0, c->d_pyr, c->pyrFrameBytes,
                                 c->d_resizeTab, B, h_pyr_dst);   // (the host copy of the pyramid is written by the kernel itself)
    // batches: level l from level l-1 (sequential dependency), all frames per launch
    for (int l = 1; l < G.nlevels && !chained; l++) {
        const OrbLevel &S = G.lv[l - 1], &D = G.lv[l];
        const uint8_t *src = (l == 1) ? lvl0 : c->d_pyr + S.imgOff;
        const int sstride = (l == 1) ? stride0 : S.stride;
        const size_t sframe = (l == 1) ? frame0 : c->pyrFrameBytes;
        launch_resize(s, src, S.w, S.h, sstride, sframe, c->d_pyr + D.imgOff, D.w, D.h, D.stride,
                      c->pyrFrameBytes, c->d_resizeTab + c->resizeTabOff[l][0],
                      c->d_resizeTab + c->resizeTabOff[l][1],
                      c->resizeGroups[l] ? c->d_resizeTab + c->resizeTabOff[l][2] : nullptr, c->resizeHint[l][B >= 8 ? 0 : 1], B);
    }
    if (ev) HIPCHK(c, hipEventRecord(c->ev[1], s));
    // host copy of levels 1.. (orbhip_set_host_pyramid): one copy of the B frames' pyramid block into pinned memory.  A
    // frame or two: on the second stream, beside FAST / quadtree / blur / describe (those do not use it then); the main
    // stream joins it at the end.  Batches (the second stream carries the blur): behind the describe kernel.
    const bool pyrFork = h_pyr_dst && B < 8 && G.nlevels > 1 && !chained;
    if (pyrFork) {
        HIPCHK(c, hipEventRecord(c->evp[0], s));
        HIPCHK(c, hipStreamWaitEvent(c->stream2, c->evp[0], 0));
        HIPCHK(c, hipMemcpyAsync(h_pyr_dst, c->d_pyr, (size_t)B * c->pyrFrameBytes, hipMemcpyDeviceToHost, c->stream2));
        HIPCHK(c, hipEventRecord(c->evp[1], c->stream2));
    }
    // E3 FAST alone on the device (it is the kernel whose roofline is reported), then the quadtree
    // (latency-bound, a few thousand workgroups) and the blur (streaming) run CONCURRENTLY on two
    // streams: both only depend on the pyramid / the FAST output; the describe kernel joins them.
    // batches: runs of up to 5 cells per workgroup; a frame or two: one cell per workgroup (four times the workgroups,
    // each a shorter chain -- the single-frame FAST time is one workgroup's latency)
    if (B >= 8)
        launch_fast(s, G, lvl0, stride0, frame0, c->d_pyr, c->pyrFrameBytes, c->d_fastTiles, c->nFastTilesBatch, c->d_cand,
                    c->d_cellCnt, B);
    else
        launch_fast(s, G, lvl0, stride0, frame0, c->d_pyr, c->pyrFrameBytes, c->d_fastTiles + c->nFastTilesBatch,
                    (int)c->fastTiles.size() - c->nFastTilesBatch, c->d_cand, c->d_cellCnt, B);
    if (ev) HIPCHK(c, hipEventRecord(c->ev[2], s));
    if (B >= 8) {
        HIPCHK(c, hipStreamWaitEvent(c->stream2, c->ev[2], 0));
        HIPCHK(c, hipEventRecord(c->evx[1], c->stream2));
        launch_blur(c->stream2, G, lvl0, stride0, frame0, c->d_pyr, c->pyrFrameBytes, c->d_blur,
                    c->lvl0FrameBytes + c->pyrFrameBytes, c->d_blurTiles, (int)c->blurTiles.size(), B);
        HIPCHK(c, hipEventRecord(c->evx[2], c->stream2));
        launch_quadtree(s, G, c->d_cand, c->d_cellCnt, c->d_pts, c->d_pnode, c->d_lvlCandCnt, c->d_lvlKp,
                        c->d_lvlKpCnt, B, c->d_qtTables);
        if (ev) HIPCHK(c, hipEventRecord(c->ev[3], s));
        HIPCHK(c, hipStreamWaitEvent(s, c->evx[2], 0));
    } else {
        // a frame or two: the blur takes a few microseconds, a cross-stream hand-over costs more than it hides
        launch_quadtree(s, G, c->d_cand, c->d_cellCnt, c->d_pts, c->d_pnode, c->d_lvlCandCnt, c->d_lvlKp,
                        c->d_lvlKpCnt, B, c->d_qtTables);
        if (ev) HIPCHK(c, hipEventRecord(c->ev[3], s));
        if (ev) HIPCHK(c, hipEventRecord(c->evx[1], s));
        launch_blur(s, G, lvl0, stride0, frame0, c->d_pyr, c->pyrFrameBytes, c->d_blur,
                    c->lvl0FrameBytes + c->pyrFrameBytes, c->d_blurTiles, (int)c->blurTiles.size(), B);
        if (ev) HIPCHK(c, hipEventRecord(c->evx[2], s));
    }
    if (ev) HIPCHK(c, hipEventRecord(c->ev[4], s));
    // E5+E7+E8 describe
    launch_describe(s, G, lvl0, stride0, frame0, c->d_pyr, c->pyrFrameBytes, c->d_blur,
                    c->lvl0FrameBytes + c->pyrFrameBytes, c->d_lvlKp, c->d_lvlKpCnt, c->d_lvlAngle, d_kps,
                    d_desc, d_counts, cap, B);
    if (ev) HIPCHK(c, hipEventRecord(c->ev[5], s));
    if (pyrFork)
        HIPCHK(c, hipStreamWaitEvent(s, c->evp[1], 0));
    else if (h_pyr_dst && G.nlevels > 1 && !chained)
        HIPCHK(c, hipMemcpyAsync(h_pyr_dst, c->d_pyr, (size_t)B * c->pyrFrameBytes, hipMemcpyDeviceToHost, s));
    HIPCHK(c, hipGetLastError());
    if (ev) c->haveStageEvents = true;
    c->last_lvl0 = lvl0;
    c->last_stride0 = stride0;
    c->last_frame0 = frame0;
    c->last_B = B;
    return ORBHIP_OK;
}

extern "C" int orbhip_get_stage_times(orbhip_ctx *c, float ms[6])
{
    if (!c || !ms) return fail(c, ORBHIP_E_ARG, "orbhip_get_stage_times: bad argument");
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    for (int i = 0; i < 6; i++) ms[i] = 0.f;
    if (c->haveStageEvents) {
        HIPCHK(c, hipStreamSynchronize(c->stream2));
        HIPCHK(c, hipEventElapsedTime(&ms[0], c->ev[0], c->ev[1]));    // pyramid
        HIPCHK(c, hipEventElapsedTime(&ms[1], c->ev[1], c->ev[2]));    // FAST
        HIPCHK(c, hipEventElapsedTime(&ms[2], c->ev[2], c->ev[3]));    // quadtree (overlaps the blur)
        HIPCHK(c, hipEventElapsedTime(&ms[3], c->evx[1], c->evx[2]));  // blur (second stream)
        HIPCHK(c, hipEventElapsedTime(&ms[4], c->ev[4], c->ev[5]));    // describe
    }
    if (c->haveMatchEvents) HIPCHK(c, hipEventElapsedTime(&ms[5], c->ev[6], c->ev[7]));
    return ORBHIP_OK;
}

extern "C" int orbhip_extract_batch_device(orbhip_ctx *c, const void *d_imgs, int B, int w, int h, int stride,
                                           size_t frame_stride, void *d_kps, void *d_desc, int cap,
                                           void *d_counts)
{
    if (!c || !d_imgs || !d_kps || !d_desc || !d_counts || cap <= 0 || stride < w)
        return fail(c, ORBHIP_E_ARG, "orbhip_extract_batch_device: bad argument");
    HIPCHK(c, hipSetDevice(c->device));
    const bool aliasOk = (stride % 16 == 0) && (((uintptr_t)d_imgs) % 16 == 0) && (frame_stride % 16 == 0);
    int rc;
    if (aliasOk) {
        if ((rc = configure(c, w, h, stride, B))) return rc;
        return run_pipeline(c, (const uint8_t *)d_imgs, stride, frame_stride, B, (orbhip_keypoint *)d_kps,
                            (uint8_t *)d_desc, (int32_t *)d_counts, cap);
    }
    // unaligned input: repack into the context's level-0 buffer first
    const int s0 = (int)align_up((size_t)w, 64);
    if ((rc = configure(c, w, h, s0, B))) return rc;
    for (int b = 0; b < B; b++)
        HIPCHK(c, hipMemcpy2DAsync(c->d_lvl0 + (size_t)b * c->lvl0FrameBytes, s0,
                                   (const uint8_t *)d_imgs + (size_t)b * frame_stride, stride, w, h,
                                   hipMemcpyDeviceToDevice, c->stream));
    return run_pipeline(c, c->d_lvl0, s0, c->lvl0FrameBytes, B, (orbhip_keypoint *)d_kps, (uint8_t *)d_desc,
                        (int32_t *)d_counts, cap);
}

static int host_stage(orbhip_ctx *c, size_t bytes)
{
    if (bytes <= c->h_stage_bytes) return ORBHIP_OK;
    if (c->h_stage) HIPCHK(c, hipHostFree(c->h_stage));
    c->h_stage = nullptr;
    c->h_stage_bytes = 0;
    void *p = nullptr;
    HIPCHK(c, hipHostMalloc(&p, bytes, hipHostMallocDefault));
    c->h_stage = (uint8_t *)p;
    c->h_stage_bytes = bytes;
    return ORBHIP_OK;
}

// pinned block for the host copy of the pyramid (levels 1..) of B frames; nullptr when the copy is not asked for
static int host_pyr_stage(orbhip_ctx *c, int B, uint8_t **dst)
{
    *dst = nullptr;
    c->h_pyr_B = 0;
    if (!c->hostPyr || c->G.nlevels < 2) return ORBHIP_OK;
    const size_t bytes = (size_t)B * c->pyrFrameBytes;
    if (bytes > c->h_pyr_bytes) {
        if (c->h_pyr) HIPCHK(c, hipHostFree(c->h_pyr));
        c->h_pyr = nullptr;
        c->h_pyr_bytes = 0;
        void *p = nullptr;
        HIPCHK(c, hipHostMalloc(&p, bytes, hipHostMallocDefault));
        c->h_pyr = (uint8_t *)p;
        c->h_pyr_bytes = bytes;
    }
    *dst = c->h_pyr;
    return ORBHIP_OK;
}

static void graph_release(orbhip_ctx *c)
{
    if (c->g_exec) (void)hipGraphExecDestroy(c->g_exec);
    if (c->g_graph) (void)hipGraphDestroy(c->g_graph);
    c->g_exec = nullptr;
    c->g_graph = nullptr;
    c->g_w = c->g_h = c->g_B = 0;
}

// A frame or two through host pointers (how Tracking.cc calls the extractor, src/Frame.cc:591-597): the whole chain -- copy
// in, seven resize launches, FAST, quadtree, blur, describe, copy out -- is ONE hipGraph launch.  Issued one by one the twelve
// launches cost the host ~3.5 us each and the device waits for them; the graph is captured from the very same call sequence
// (run_pipeline) at the first call of a geometry and replayed afterwards.  ORBHIP_NO_GRAPH=1 keeps the eager sequence.
static int extract_small_graph(orbhip_ctx *c, const uint8_t *const *imgs, int B, int w, int h, int stride, int s0, size_t kbytes,
                               size_t dbytes, size_t cbytes, size_t koff, size_t doff, size_t coff, int dcap)
{
    const size_t inBytes = (size_t)B * c->lvl0FrameBytes;
    if (inBytes > c->h_in_bytes) {
        graph_release(c);
        if (c->h_in) HIPCHK(c, hipHostFree(c->h_in));
        c->h_in = nullptr;
        c->h_in_bytes = 0;
        void *p = nullptr;
        HIPCHK(c, hipHostMalloc(&p, inBytes, hipHostMallocDefault));
        c->h_in = (uint8_t *)p;
        c->h_in_bytes = inBytes;
    }
    int rc;
    if ((rc = host_stage(c, coff + align_up(cbytes, 256)))) return rc;
    uint8_t *hpyr = nullptr;
    if ((rc = host_pyr_stage(c, B, &hpyr))) return rc;
    c->h_in_valid = false;
    for (int b = 0; b < B; b++) {
        if (!imgs[b]) return fail(c, ORBHIP_E_ARG, "orbhip_extract_batch: null image");
        uint8_t *dst = c->h_in + (size_t)b * c->lvl0FrameBytes;
        if (stride == s0)
            memcpy(dst, imgs[b], (size_t)s0 * (h - 1) + w);
        else
            for (int y = 0; y < h; y++) memcpy(dst + (size_t)y * s0, imgs[b] + (size_t)y * stride, (size_t)w);
    }
    uint8_t *blk = reinterpret_cast<uint8_t *>(c->d_kps);
    static const bool directOut = !(getenv("ORBHIP_COPY_OUT") && atoi(getenv("ORBHIP_COPY_OUT")) != 0);   // A/B: 1 = result copy node
    const void *key[5] = {c->d_lvl0, blk, c->h_in, c->h_stage, hpyr};
    const bool same = c->g_exec && c->g_w == w && c->g_h == h && c->g_B == B && memcmp(key, c->g_key, sizeof(key)) == 0;
    if (!same) {
        graph_release(c);
        HIPCHK(c, hipStreamSynchronize(c->stream));
        c->capturing = true;
        HIPCHK(c, hipStreamBeginCapture(c->stream, hipStreamCaptureModeThreadLocal));
        // the describe kernel writes keypoints, descriptors and counts straight into the page-locked result block (posted PCIe
        // writes of a few dozen KB that overlap the kernel): no copy node behind it -- that node started 8 us after describe ended
        uint8_t *out = directOut ? c->h_stage : blk;
        hipError_t e = hipMemcpyAsync(c->d_lvl0, c->h_in, inBytes, hipMemcpyHostToDevice, c->stream);
        rc = e == hipSuccess ? run_pipeline(c, c->d_lvl0, s0, c->lvl0FrameBytes, B, (orbhip_keypoint *)(out + koff), out + doff,
                                            (int32_t *)(out + coff), dcap, hpyr)
                             : ORBHIP_E_HIP;
        if (rc == ORBHIP_OK && !directOut) e = hipMemcpyAsync(c->h_stage, blk, coff + cbytes, hipMemcpyDeviceToHost, c->stream);
        hipGraph_t g = nullptr;
        const hipError_t e2 = hipStreamEndCapture(c->stream, &g);
        c->capturing = false;
        if (rc != ORBHIP_OK || e != hipSuccess || e2 != hipSuccess || !g) {
            if (g) (void)hipGraphDestroy(g);
            return fail(c, ORBHIP_E_HIP, std::string("graph capture of the single-frame chain failed: ") +
                                             hipGetErrorString(e != hipSuccess ? e : e2));
        }
        c->g_graph = g;
        HIPCHK(c, hipGraphInstantiate(&c->g_exec, g, nullptr, nullptr, 0));
        c->g_w = w; c->g_h = h; c->g_B = B;
        memcpy(c->g_key, key, sizeof(key));
        c->g_calls = 0;
    }
    if ((c->g_calls++ & 255u) == 0) {
        // the first call of a geometry and every 256th one run the same chain eagerly: that refreshes the stage times behind
        // GetTimeOfComputePyramid / ...KeyPointsOctTree / ...Descriptor (include/ORBextractor.h:51-53)
        uint8_t *out = directOut ? c->h_stage : blk;
        HIPCHK(c, hipMemcpyAsync(c->d_lvl0, c->h_in, inBytes, hipMemcpyHostToDevice, c->stream));
        if ((rc = run_pipeline(c, c->d_lvl0, s0, c->lvl0FrameBytes, B, (orbhip_keypoint *)(out + koff), out + doff,
                               (int32_t *)(out + coff), dcap, hpyr)))
            return rc;
        if (!directOut) HIPCHK(c, hipMemcpyAsync(c->h_stage, blk, coff + cbytes, hipMemcpyDeviceToHost, c->stream));
    } else {
        HIPCHK(c, hipGraphLaunch(c->g_exec, c->stream));
    }
    HIPCHK(c, hipStreamSynchronize(c->stream));
    c->h_in_valid = true;
    c->h_pyr_B = hpyr ? B : 0;
    (void)kbytes; (void)dbytes;
    return ORBHIP_OK;
}

extern "C" int orbhip_extract_batch(orbhip_ctx *c, const uint8_t *const *imgs, int B, int w, int h, int stride,
                                    orbhip_keypoint *kps, uint8_t *desc, int cap, int *n_out)
{
    if (!c || !imgs || !kps || !desc || !n_out || cap <= 0 || stride < w)
        return fail(c, ORBHIP_E_ARG, "orbhip_extract_batch: bad argument");
    HIPCHK(c, hipSetDevice(c->device));
    const int s0 = (int)align_up((size_t)w, 64);
    int rc;
    if ((rc = configure(c, w, h, s0, B))) return rc;
    const int dcap = (int)c->cap_out;
    // results: keypoints | descriptors | counts of the B frames are one device block that goes to pinned staging in ONE
    // asynchronous copy behind the kernels and ONE synchronisation (every extra copy or wait is a device round trip --
    // most of a single frame's overhead); the n[b] valid entries are then copied out on the host
    const size_t kbytes = (size_t)B * dcap * sizeof(orbhip_keypoint), dbytes = (size_t)B * dcap * 32, cbytes = (size_t)B * 4;
    const size_t koff = 0, doff = align_up(kbytes, 256), coff = doff + align_up(dbytes, 256);
    uint8_t *blk = reinterpret_cast<uint8_t *>(c->d_kps);
    static const bool noGraph = getenv("ORBHIP_NO_GRAPH") && atoi(getenv("ORBHIP_NO_GRAPH")) != 0;
    if (B < 8 && !noGraph) {
        if ((rc = extract_small_graph(c, imgs, B, w, h, stride, s0, kbytes, dbytes, cbytes, koff, doff, coff, dcap))) return rc;
    } else {
        for (int b = 0; b < B; b++) {
            if (!imgs[b]) return fail(c, ORBHIP_E_ARG, "orbhip_extract_batch: null image");
            HIPCHK(c, hipMemcpy2DAsync(c->d_lvl0 + (size_t)b * c->lvl0FrameBytes, s0, imgs[b], stride, w, h,
                                       hipMemcpyHostToDevice, c->stream));
        }
        uint8_t *hpyr = nullptr;
        if ((rc = host_pyr_stage(c, B, &hpyr))) return rc;
        c->h_in_valid = false;
        if ((rc = run_pipeline(c, c->d_lvl0, s0, c->lvl0FrameBytes, B, (orbhip_keypoint *)(blk + koff), blk + doff,
                               (int32_t *)(blk + coff), dcap, hpyr)))
            return rc;
        if ((rc = host_stage(c, coff + align_up(cbytes, 256)))) return rc;
        HIPCHK(c, hipMemcpyAsync(c->h_stage, blk, coff + cbytes, hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));
        c->h_pyr_B = hpyr ? B : 0;
    }
    memcpy(n_out, c->h_stage + coff, cbytes);
    for (int b = 0; b < B; b++) {
        const int n = n_out[b];
        if (n > cap || n > dcap) return fail(c, ORBHIP_E_CAPACITY, "orbhip_extract_batch: output capacity too small");
        memcpy(kps + (size_t)b * cap, c->h_stage + koff + (size_t)b * dcap * sizeof(orbhip_keypoint), (size_t)n * sizeof(orbhip_keypoint));
        memcpy(desc + (size_t)b * cap * 32, c->h_stage + doff + (size_t)b * dcap * 32, (size_t)n * 32);
    }
    return ORBHIP_OK;
}

extern "C" int orbhip_extract(orbhip_ctx *c, const uint8_t *img, int w, int h, int stride, orbhip_keypoint *kps,
                              uint8_t *desc, int cap, int *n_out, float timings_ms[3])
{
    if (!c || !img || !n_out) return fail(c, ORBHIP_E_ARG, "orbhip_extract: bad argument");
    const uint8_t *imgs[1] = {img};
    int rc = orbhip_extract_batch(c, imgs, 1, w, h, stride, kps, desc, cap, n_out);
    if (rc == ORBHIP_OK && timings_ms) {
        // the reference's three timers: pyramid | FAST + quadtree (+ orientation) | blur + BRIEF
        float ms[6];
        if ((rc = orbhip_get_stage_times(c, ms))) return rc;
        timings_ms[0] = ms[0];
        timings_ms[1] = ms[1] + ms[2];
        timings_ms[2] = ms[3] + ms[4];
    }
    return rc;
}

// ------------------------------------------------------------------------------------------------
// host-fed pipeline: frames arrive from host memory (the reference's frames always do: cv::imread,
// Examples/Monocular/mono_euroc.cc:73), batch n + 1 is copied in and batch n - 1 copied out while batch n computes
// ------------------------------------------------------------------------------------------------
static void pipe_release(orbhip_ctx *c)
{
    OrbPipe *P = c->pipe;
    if (!P) return;
    if (P->sIn) (void)hipStreamSynchronize(P->sIn);
    if (P->sOut) (void)hipStreamSynchronize(P->sOut);
    for (uint8_t *p : P->d_in)
        if (p) (void)hipFree(p);
    for (uint8_t *p : P->d_out)
        if (p) (void)hipFree(p);
    for (uint8_t *p : P->h_out)
        if (p) (void)hipHostFree(p);
    if (P->d_bowScratch) (void)hipFree(P->d_bowScratch);
    for (auto *v : {&P->evIn, &P->evK, &P->evOut})
        for (hipEvent_t e : *v)
            if (e) (void)hipEventDestroy(e);
    if (P->sIn) (void)hipStreamDestroy(P->sIn);
    if (P->sOut) (void)hipStreamDestroy(P->sOut);
    delete P;
    c->pipe = nullptr;
}

extern "C" void *orbhip_host_alloc(size_t nbytes)
{
    void *p = nullptr;
    if (hipHostMalloc(&p, nbytes ? nbytes : 16, hipHostMallocDefault) != hipSuccess) return nullptr;
    return p;
}

extern "C" void orbhip_host_free(void *p)
{
    if (p) (void)hipHostFree(p);
}

extern "C" int orbhip_pipe_create(orbhip_ctx *c, int depth, int B, int w, int h)
{
    if (!c || depth < 2 || depth > 8 || B < 1 || w < 1 || h < 1) return fail(c, ORBHIP_E_ARG, "orbhip_pipe_create: bad argument");
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    pipe_release(c);
    // device rows are the image rows when they are 16-byte multiples (level 0 is then read in place), else padded to 64
    const int stride = (w % 16 == 0) ? w : (int)align_up((size_t)w, 64);
    int rc;
    if ((rc = configure(c, w, h, stride, B))) return rc;
    OrbPipe *P = new OrbPipe();
    c->pipe = P;
    P->depth = depth; P->B = B; P->w = w; P->h = h; P->stride = stride;
    P->dcap = (int)c->cap_out;
    P->frameBytes = align_up((size_t)stride * h, 256);
    P->inBytes = P->frameBytes * B;
    const size_t kbytes = (size_t)B * P->dcap * sizeof(orbhip_keypoint), dbytes = (size_t)B * P->dcap * 32;
    P->koff = 0;
    P->doff = align_up(kbytes, 256);
    P->coff = P->doff + align_up(dbytes, 256);
    P->m12off = P->coff + align_up((size_t)B * 4, 256);
    P->m21off = P->m12off + align_up((size_t)B * P->dcap * 4, 256);
    P->nmoff = P->m21off + align_up((size_t)B * P->dcap * 4, 256);
    P->outBytes = P->nmoff + align_up((size_t)B * 4, 256);
    auto bail = [&](const char *what, hipError_t e) {
        const int r = fail(c, ORBHIP_E_HIP, std::string(what) + ": " + hipGetErrorString(e));
        const std::string keep = c->err;
        pipe_release(c);
        c->err = keep;
        return r;
    };
    hipError_t e;
    if ((e = hipStreamCreateWithFlags(&P->sIn, hipStreamNonBlocking)) != hipSuccess) return bail("hipStreamCreate", e);
    if ((e = hipStreamCreateWithFlags(&P->sOut, hipStreamNonBlocking)) != hipSuccess) return bail("hipStreamCreate", e);
    P->d_in.assign(depth, nullptr); P->d_out.assign(depth, nullptr); P->h_out.assign(depth, nullptr);
    P->evIn.assign(depth, nullptr); P->evK.assign(depth, nullptr); P->evOut.assign(depth, nullptr);
    P->slotB.assign(depth, 0);
    for (int i = 0; i < depth; i++) {
        void *p = nullptr;
        if ((e = hipMalloc(&p, P->inBytes)) != hipSuccess) return bail("hipMalloc (input slot)", e);
        P->d_in[i] = (uint8_t *)p;
        if ((e = hipMalloc(&p, P->outBytes)) != hipSuccess) return bail("hipMalloc (output slot)", e);
        P->d_out[i] = (uint8_t *)p;
        if ((e = hipHostMalloc(&p, P->outBytes, hipHostMallocDefault)) != hipSuccess) return bail("hipHostMalloc (result slot)", e);
        P->h_out[i] = (uint8_t *)p;
        for (hipEvent_t *ev : {&P->evIn[i], &P->evK[i], &P->evOut[i]})
            if ((e = hipEventCreateWithFlags(ev, hipEventDisableTiming)) != hipSuccess) return bail("hipEventCreate", e);
    }
    return ORBHIP_OK;
}

extern "C" int orbhip_pipe_destroy(orbhip_ctx *c)
{
    if (!c) return ORBHIP_E_ARG;
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    pipe_release(c);
    return ORBHIP_OK;
}

extern "C" int orbhip_pipe_submit(orbhip_ctx *c, const uint8_t *frames, int B, int stride, size_t frame_stride)
{
    if (!c || !c->pipe) return fail(c, ORBHIP_E_ARG, "orbhip_pipe_submit: no pipeline (orbhip_pipe_create)");
    OrbPipe *P = c->pipe;
    if (!frames || B < 1 || B > P->B || stride < P->w || frame_stride < (size_t)stride * (P->h - 1) + P->w)
        return fail(c, ORBHIP_E_ARG, "orbhip_pipe_submit: bad argument");
    if (P->submitted - P->waited >= P->depth)
        return fail(c, ORBHIP_E_CAPACITY, "orbhip_pipe_submit: every slot holds results that were not collected (orbhip_pipe_wait)");
    HIPCHK(c, hipSetDevice(c->device));
    const int s = (int)(P->submitted % P->depth);
    int rc;
    if ((rc = configure(c, P->w, P->h, P->stride, B))) return rc;
    // copy in: the slot's previous kernels must have read it (evK of the batch `depth` submissions ago)
    if (P->submitted >= P->depth) HIPCHK(c, hipStreamWaitEvent(P->sIn, P->evK[s], 0));
    if (stride == P->stride && (size_t)stride * P->h == P->frameBytes && frame_stride == P->frameBytes) {
        HIPCHK(c, hipMemcpyAsync(P->d_in[s], frames, P->frameBytes * B, hipMemcpyHostToDevice, P->sIn));      // one contiguous block
    } else if (stride == P->stride) {
        // whole frames are contiguous on both sides: a 2-D copy with one "row" per frame
        HIPCHK(c, hipMemcpy2DAsync(P->d_in[s], P->frameBytes, frames, frame_stride, (size_t)stride * (P->h - 1) + P->w, B,
                                   hipMemcpyHostToDevice, P->sIn));
    } else {
        hipMemcpy3DParms q = {};
        q.srcPtr = make_hipPitchedPtr(const_cast<uint8_t *>(frames), stride, P->w, frame_stride / stride);
        q.dstPtr = make_hipPitchedPtr(P->d_in[s], P->stride, P->w, P->frameBytes / P->stride);
        q.extent = make_hipExtent(P->w, P->h, B);
        q.kind = hipMemcpyHostToDevice;
        if (frame_stride % stride != 0 || P->frameBytes % P->stride != 0) {
            for (int b = 0; b < B; b++)   // pitches that are not whole rows: frame by frame
                HIPCHK(c, hipMemcpy2DAsync(P->d_in[s] + (size_t)b * P->frameBytes, P->stride, frames + (size_t)b * frame_stride, stride,
                                           P->w, P->h, hipMemcpyHostToDevice, P->sIn));
        } else {
            HIPCHK(c, hipMemcpy3DAsync(&q, P->sIn));
        }
    }
    HIPCHK(c, hipEventRecord(P->evIn[s], P->sIn));
    // compute: after the copy, and after the previous results of this output slot have left the device
    HIPCHK(c, hipStreamWaitEvent(c->stream, P->evIn[s], 0));
    if (P->submitted >= P->depth) HIPCHK(c, hipStreamWaitEvent(c->stream, P->evOut[s], 0));
    uint8_t *blk = P->d_out[s];
    if ((rc = run_pipeline(c, P->d_in[s], P->stride, P->frameBytes, B, (orbhip_keypoint *)(blk + P->koff), blk + P->doff,
                           (int32_t *)(blk + P->coff), P->dcap)))
        return rc;
    size_t outBytes = P->coff + (size_t)B * 4;
    if (P->bow) {
        // Frame::ComputeBoW (src/Frame.cc:739-746) + ORBmatcher::SearchByBoW of every frame against its predecessor in the batch
        // (Tracking::TrackReferenceKeyFrame, src/Tracking.cc:1881-1885), on the slot's device-resident outputs
        const size_t n = (size_t)B * P->dcap;
        int32_t *word = (int32_t *)P->d_bowScratch, *node = word + 2 * (size_t)P->B * P->dcap;
        float *wt = (float *)(word + (size_t)P->B * P->dcap);
        if ((rc = orbhip_vocab_transform_device(c, blk + P->doff, (int)n, P->levelsup, word, wt, node))) return rc;
        if ((rc = orbhip_search_by_bow_seq_device(c, blk + P->doff, blk + P->koff, blk + P->coff, node, wt, nullptr, P->dcap, B, 1, 0,
                                                  P->nnratio, P->check_ori, blk + P->m12off, blk + P->m21off, blk + P->nmoff)))
            return rc;
        outBytes = P->nmoff + (size_t)B * 4;
    }
    HIPCHK(c, hipEventRecord(P->evK[s], c->stream));
    // copy out
    HIPCHK(c, hipStreamWaitEvent(P->sOut, P->evK[s], 0));
    HIPCHK(c, hipMemcpyAsync(P->h_out[s], blk, outBytes, hipMemcpyDeviceToHost, P->sOut));
    HIPCHK(c, hipEventRecord(P->evOut[s], P->sOut));
    P->slotB[s] = B;
    P->submitted++;
    return ORBHIP_OK;
}

extern "C" int orbhip_pipe_wait(orbhip_ctx *c, const orbhip_keypoint **kps, const uint8_t **desc, const int32_t **n_out,
                                int *B, int *cap)
{
    if (!c || !c->pipe) return fail(c, ORBHIP_E_ARG, "orbhip_pipe_wait: no pipeline (orbhip_pipe_create)");
    OrbPipe *P = c->pipe;
    if (P->waited >= P->submitted) return fail(c, ORBHIP_E_ARG, "orbhip_pipe_wait: nothing submitted");
    HIPCHK(c, hipSetDevice(c->device));
    const int s = (int)(P->waited % P->depth);
    HIPCHK(c, hipEventSynchronize(P->evOut[s]));
    if (kps) *kps = (const orbhip_keypoint *)(P->h_out[s] + P->koff);
    if (desc) *desc = P->h_out[s] + P->doff;
    if (n_out) *n_out = (const int32_t *)(P->h_out[s] + P->coff);
    if (B) *B = P->slotB[s];
    if (cap) *cap = P->dcap;
    P->lastWaited = s;
    P->waited++;
    return ORBHIP_OK;
}

extern "C" int orbhip_pipe_enable_bow(orbhip_ctx *c, int levelsup, float nnratio, int check_ori)
{
    if (!c || !c->pipe) return fail(c, ORBHIP_E_ARG, "orbhip_pipe_enable_bow: no pipeline (orbhip_pipe_create)");
    if (!c->voc.desc) return fail(c, ORBHIP_E_ARG, "orbhip_pipe_enable_bow: no vocabulary loaded");
    OrbPipe *P = c->pipe;
    if (P->dcap > 4096) return fail(c, ORBHIP_E_SIZE, "orbhip_pipe_enable_bow: more than 4096 feature slots per frame");
    HIPCHK(c, hipSetDevice(c->device));
    if (!P->d_bowScratch) {
        void *p = nullptr;
        HIPCHK(c, hipMalloc(&p, (size_t)P->B * P->dcap * 12 + 256));
        P->d_bowScratch = (uint8_t *)p;
    }
    P->bow = true;
    P->levelsup = levelsup;
    P->nnratio = nnratio;
    P->check_ori = check_ori;
    return ORBHIP_OK;
}

extern "C" int orbhip_pipe_matches(orbhip_ctx *c, const int32_t **match12, const int32_t **match21, const int32_t **nmatches)
{
    if (!c || !c->pipe || !c->pipe->bow || c->pipe->lastWaited < 0)
        return fail(c, ORBHIP_E_ARG, "orbhip_pipe_matches: no batch with matches has been collected");
    OrbPipe *P = c->pipe;
    const uint8_t *h = P->h_out[P->lastWaited];
    if (match12) *match12 = (const int32_t *)(h + P->m12off);
    if (match21) *match21 = (const int32_t *)(h + P->m21off);
    if (nmatches) *nmatches = (const int32_t *)(h + P->nmoff);
    return ORBHIP_OK;
}

static int copy_level(orbhip_ctx *c, const uint8_t *src, int sstride, int w, int h, uint8_t *dst, int dst_stride)
{
    HIPCHK(c, hipMemcpy2DAsync(dst, dst_stride, src, sstride, w, h, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return ORBHIP_OK;
}

extern "C" int orbhip_get_pyramid_level(orbhip_ctx *c, int frame, int level, uint8_t *dst, int dst_stride,
                                        int *w, int *h)
{
    if (!c || !c->last_lvl0 || frame < 0 || frame >= c->last_B || level < 0 || level >= c->nlevels)
        return fail(c, ORBHIP_E_ARG, "orbhip_get_pyramid_level: bad argument or no frame extracted yet");
    const OrbLevel &L = c->G.lv[level];
    if (w) *w = L.w;
    if (h) *h = L.h;
    if (!dst) return ORBHIP_OK;
    if (dst_stride < L.w) return fail(c, ORBHIP_E_ARG, "dst_stride too small");
    HIPCHK(c, hipSetDevice(c->device));
    if (level == 0)
        return copy_level(c, c->last_lvl0 + (size_t)frame * c->last_frame0, c->last_stride0, L.w, L.h, dst, dst_stride);
    return copy_level(c, c->d_pyr + (size_t)frame * c->pyrFrameBytes + L.imgOff, L.stride, L.w, L.h, dst, dst_stride);
}

extern "C" int orbhip_set_host_pyramid(orbhip_ctx *c, int on)
{
    if (!c) return fail(c, ORBHIP_E_ARG, "orbhip_set_host_pyramid: null context");
    c->hostPyr = on != 0;
    if (!c->hostPyr) c->h_pyr_B = 0;
    return ORBHIP_OK;
}

extern "C" int orbhip_host_pyramid_level(orbhip_ctx *c, int frame, int level, const uint8_t **ptr, int *stride, int *w, int *h)
{
    if (!c || !ptr || !stride || frame < 0 || level < 0 || level >= c->nlevels)
        return fail(c, ORBHIP_E_ARG, "orbhip_host_pyramid_level: bad argument");
    const OrbLevel &L = c->G.lv[level];
    if (w) *w = L.w;
    if (h) *h = L.h;
    if (level == 0) {
        // the pinned copy of the caller's frame that the single-frame path uploads from (rows last_stride0 apart)
        if (!c->h_in_valid || frame >= c->last_B || c->last_lvl0 != c->d_lvl0)
            return fail(c, ORBHIP_E_ARG, "orbhip_host_pyramid_level: level 0 is not staged on the host for this call (use the caller's image)");
        *ptr = c->h_in + (size_t)frame * c->lvl0FrameBytes;
        *stride = c->last_stride0;
        return ORBHIP_OK;
    }
    if (frame >= c->h_pyr_B)
        return fail(c, ORBHIP_E_ARG, "orbhip_host_pyramid_level: no host pyramid for this frame (orbhip_set_host_pyramid before orbhip_extract*)");
    *ptr = c->h_pyr + (size_t)frame * c->pyrFrameBytes + L.imgOff;
    *stride = L.stride;
    return ORBHIP_OK;
}

extern "C" int orbhip_debug_get_blurred_level(orbhip_ctx *c, int frame, int level, uint8_t *dst, int dst_stride,
                                              int *w, int *h)
{
    if (!c || !c->last_lvl0 || frame < 0 || frame >= c->last_B || level < 0 || level >= c->nlevels)
        return fail(c, ORBHIP_E_ARG, "orbhip_debug_get_blurred_level: bad argument");
    const OrbLevel &L = c->G.lv[level];
    if (w) *w = L.w;
    if (h) *h = L.h;
    if (!dst) return ORBHIP_OK;
    HIPCHK(c, hipSetDevice(c->device));
    const size_t bf = c->lvl0FrameBytes + c->pyrFrameBytes;
    const uint8_t *src = c->d_blur + (size_t)frame * bf + (level == 0 ? 0 : c->G.boff1 + L.imgOff);
    return copy_level(c, src, level == 0 ? c->G.bstride0 : L.stride, L.w, L.h, dst, dst_stride);
}

extern "C" int orbhip_debug_get_candidates(orbhip_ctx *c, int frame, int level, orbhip_cand *out, int cap,
                                           int *n_out)
{
    if (!c || !c->last_lvl0 || frame < 0 || frame >= c->last_B || level < 0 || level >= c->nlevels || !n_out)
        return fail(c, ORBHIP_E_ARG, "orbhip_debug_get_candidates: bad argument");
    HIPCHK(c, hipSetDevice(c->device));
    const OrbLevels &G = c->G;
    const OrbLevel &L = G.lv[level];
    const int ncells = L.nCols * L.nRows;
    std::vector<uint16_t> cnt(ncells);
    std::vector<uint32_t> slots((size_t)L.ptCap);
    HIPCHK(c, hipMemcpyAsync(cnt.data(), c->d_cellCnt + (size_t)frame * G.totalCells + L.cellBase, (size_t)ncells * 2,
                             hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipMemcpyAsync(slots.data(), c->d_cand + (size_t)frame * G.totalCands + L.candBase, (size_t)L.ptCap * 4,
                             hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    int n = 0;
    for (int cell = 0; cell < ncells; cell++)
        for (int k = 0; k < cnt[cell]; k++) {
            if (out && n < cap) {
                const uint32_t p = slots[(size_t)cell * L.cellCap + k];
                out[n].x = (int)(p & 0xFFF);
                out[n].y = (int)((p >> 12) & 0xFFF);
                out[n].score = (int)(p >> 24);
            }
            n++;
        }
    *n_out = n;
    if (out && n > cap) return fail(c, ORBHIP_E_CAPACITY, "orbhip_debug_get_candidates: capacity");
    return ORBHIP_OK;
}

extern "C" int orbhip_debug_get_level_keypoints(orbhip_ctx *c, int frame, int level, orbhip_keypoint *out, int cap,
                                                int *n_out)
{
    if (!c || !c->last_lvl0 || frame < 0 || frame >= c->last_B || level < 0 || level >= c->nlevels || !n_out)
        return fail(c, ORBHIP_E_ARG, "orbhip_debug_get_level_keypoints: bad argument");
    HIPCHK(c, hipSetDevice(c->device));
    const OrbLevels &G = c->G;
    const OrbLevel &L = G.lv[level];
    int32_t cnts[ORBHIP_MAX_LEVELS];
    HIPCHK(c, hipMemcpyAsync(cnts, c->d_lvlKpCnt + (size_t)frame * ORBHIP_MAX_LEVELS, sizeof(cnts), hipMemcpyDeviceToHost,
                             c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    const int n = cnts[level];
    *n_out = n;
    if (!out) return ORBHIP_OK;
    if (n > cap) return fail(c, ORBHIP_E_CAPACITY, "orbhip_debug_get_level_keypoints: capacity");
    std::vector<uint32_t> pk(n);
    std::vector<float> ang(n);
    HIPCHK(c, hipMemcpyAsync(pk.data(), c->d_lvlKp + (size_t)frame * G.totalKps + L.kpBase, (size_t)n * 4,
                             hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipMemcpyAsync(ang.data(), c->d_lvlAngle + (size_t)frame * G.totalKps + L.kpBase, (size_t)n * 4,
                             hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    for (int i = 0; i < n; i++) {
        out[i].x = (float)((int)(pk[i] & 0xFFF) + ORB_MIN_BORDER);
        out[i].y = (float)((int)((pk[i] >> 12) & 0xFFF) + ORB_MIN_BORDER);
        out[i].size = L.kpSize;
        out[i].angle = ang[i];
        out[i].response = (float)(pk[i] >> 24);
        out[i].octave = level;
        out[i].class_id = -1;
    }
    return ORBHIP_OK;
}

// ------------------------------------------------------------------------------------------------
// matching
// ------------------------------------------------------------------------------------------------
static int match_scratch(orbhip_ctx *c, size_t bytes)
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
    HIPCHK(c, hipSetDevice(c->device));
    int rc;
    const size_t need = knn2_scratch_bytes(nq, ndb);
    if ((rc = match_scratch(c, need))) return rc;
    HIPCHK(c, hipEventRecord(c->ev[6], c->stream));
    launch_knn2(c->stream, (const uint8_t *)d_q, nq, (const uint8_t *)d_db, ndb, (int32_t *)d_best_idx,
                (int32_t *)d_best_d, (int32_t *)d_second_d, c->d_match, need);
    HIPCHK(c, hipEventRecord(c->ev[7], c->stream));
    HIPCHK(c, hipGetLastError());
    c->haveMatchEvents = true;
    return ORBHIP_OK;
}

extern "C" int orbhip_hamming_knn2_seq_device(orbhip_ctx *c, const void *d_desc, const void *d_counts, int cap,
                                              int B, int lag, void *d_best_idx, void *d_best_d, void *d_second_d)
{
    if (!c || !d_desc || !d_counts || cap <= 0 || B <= 0 || lag < 0 || !d_best_idx || !d_best_d || !d_second_d)
        return fail(c, ORBHIP_E_ARG, "orbhip_hamming_knn2_seq_device: bad argument");
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipEventRecord(c->ev[6], c->stream));
    launch_knn2_seq(c->stream, (const uint8_t *)d_desc, (const int32_t *)d_counts, cap, B, lag,
                    (int32_t *)d_best_idx, (int32_t *)d_best_d, (int32_t *)d_second_d);
    HIPCHK(c, hipEventRecord(c->ev[7], c->stream));
    HIPCHK(c, hipGetLastError());
    c->haveMatchEvents = true;
    return ORBHIP_OK;
}

// Bump allocator over one temporary device block (host-pointer matching entry points).
// Device staging for the host-pointer entry points: one grow-only block per context.  Calls on a context are serialised
// on its stream and every such entry point synchronises before it returns, so the block is free again at the next call.
struct TmpDev {
    orbhip_ctx *c;
    uint8_t *base = nullptr;
    size_t used = 0, cap = 0;
    std::vector<void *> extra;   // blocks taken beyond the reserved size (a reserve() total that undercounts must not
                                 // become a null pointer handed to a copy: ADVICE r01)
    explicit TmpDev(orbhip_ctx *ctx) : c(ctx) {}
    ~TmpDev()
    {
        // Every entry point synchronises before its normal return; an early error return may leave asynchronous copies from
        // the caller's (or this frame's stack) memory in flight -- drain them before that memory goes away.
        if (c && c->stream && hipStreamQuery(c->stream) != hipSuccess) (void)hipStreamSynchronize(c->stream);
        for (void *p : extra) (void)hipFree(p);
    }
    int reserve(size_t bytes)
    {
        bytes += 4096;
        if (bytes > c->d_tmp_bytes || !c->d_tmp) {
            if (c->d_tmp) {
                HIPCHK(c, hipStreamSynchronize(c->stream));
                HIPCHK(c, hipFree(c->d_tmp));
            }
            c->d_tmp = nullptr;
            c->d_tmp_bytes = 0;
            const size_t want = bytes + bytes / 2;
            HIPCHK(c, hipMalloc(&c->d_tmp, want));
            c->d_tmp_bytes = want;
        }
        base = (uint8_t *)c->d_tmp;
        cap = c->d_tmp_bytes;
        return ORBHIP_OK;
    }
    void *take(size_t bytes)
    {
        used = align_up(used, 256);
        if (used + bytes <= cap) {
            void *p = base + used;
            used += bytes;
            return p;
        }
        // beyond the reservation: a block of its own (slow, but never a null or overlapping pointer)
        void *p = nullptr;
        if (hipMalloc(&p, bytes ? bytes : 16) != hipSuccess) {
            fprintf(stderr, "[orbhip] staging allocation of %zu bytes failed\n", bytes);
            abort();   // an entry point that continued would copy through a null pointer
        }
        extra.push_back(p);
        return p;
    }
};

// The small host-pointer calls (one frame's descriptors, index lists, a few KB of results -- the per-frame calls of
// Tracking) used to issue one copy per argument: nine pageable host-to-device copies and two back for a SearchByBoW, each
// ~8 us of runtime work, around a 25 us kernel.  Packed mirrors ONE device block in ONE page-locked host block: inputs are
// memcpy'd to the offsets of their device twins and travel in one copy, outputs come back in one copy.
struct Packed {
    orbhip_ctx *c;
    TmpDev T;
    uint8_t *h = nullptr, *d = nullptr;
    size_t off = 0, inEnd = 0, outBeg = 0, cap = 0;
    explicit Packed(orbhip_ctx *ctx) : c(ctx), T(ctx) {}
    int begin(size_t total)
    {
        total += 8192;
        int rc;
        if ((rc = T.reserve(total))) return rc;
        if (total > c->h_pack_bytes) {
            if (c->h_pack) HIPCHK(c, hipHostFree(c->h_pack));
            c->h_pack = nullptr;
            c->h_pack_bytes = 0;
            void *p = nullptr;
            HIPCHK(c, hipHostMalloc(&p, total + total / 2, hipHostMallocDefault));
            c->h_pack = (uint8_t *)p;
            c->h_pack_bytes = total + total / 2;
        }
        d = (uint8_t *)T.take(total);
        h = c->h_pack;
        cap = total;
        return ORBHIP_OK;
    }
    // device twin of `bytes` bytes copied from src (inputs first, outputs after)
    void *in(const void *src, size_t bytes)
    {
        off = align_up(off, 256);
        if (bytes) memcpy(h + off, src, bytes);
        void *p = d + off;
        off += bytes;
        inEnd = off;
        return p;
    }
    void *in_fill(int byte, size_t bytes)
    {
        off = align_up(off, 256);
        memset(h + off, byte, bytes);
        void *p = d + off;
        off += bytes;
        inEnd = off;
        return p;
    }
    void *out(size_t bytes)
    {
        off = align_up(off, 256);
        if (!outBeg) outBeg = off;
        void *p = d + off;
        off += bytes;
        return p;
    }
    // Outputs that the kernels only WRITE (plain stores, a few KB) can live in the page-locked block itself: the device
    // stores to it over PCIe while the kernel runs and no copy follows -- the pointer is valid on both sides.
    void *out_host(size_t bytes)
    {
        off = align_up(off, 256);
        void *p = h + off;
        off += bytes;
        return p;
    }
    void *out_host_fill(int byte, size_t bytes)
    {
        void *p = out_host(bytes);
        memset(p, byte, bytes);
        return p;
    }
    const void *host(const void *dev) const { return h + ((const uint8_t *)dev - d); }
    int upload()
    {
        if (off > cap) return fail(c, ORBHIP_E_SIZE, "internal: packed staging block undersized");
        HIPCHK(c, hipMemcpyAsync(d, h, inEnd, hipMemcpyHostToDevice, c->stream));
        return ORBHIP_OK;
    }
    // everything from the first output (or `from`, for in/out regions) to the end of the block, then the stream is idle
    int download(const void *from = nullptr)
    {
        const size_t b = from ? (size_t)((const uint8_t *)from - d) : outBeg;
        if (off > b) HIPCHK(c, hipMemcpyAsync(h + b, d + b, off - b, hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));
        return ORBHIP_OK;
    }
};

extern "C" int orbhip_hamming_knn2(orbhip_ctx *c, const uint8_t *q, int nq, const uint8_t *db, int ndb,
                                   int32_t *best_idx, int32_t *best_d, int32_t *second_d)
{
    if (!c || nq < 0 || ndb < 0 || (nq > 0 && (!q || !best_idx || !best_d || !second_d)) || (ndb > 0 && !db))
        return fail(c, ORBHIP_E_ARG, "orbhip_hamming_knn2: bad argument");
    if (nq == 0) return ORBHIP_OK;
    HIPCHK(c, hipSetDevice(c->device));
    TmpDev T(c);
    int rc;
    if ((rc = T.reserve((size_t)nq * 32 + (size_t)ndb * 32 + (size_t)nq * 12 + 2048))) return rc;
    uint8_t *dq = (uint8_t *)T.take((size_t)nq * 32), *ddb = (uint8_t *)T.take((size_t)ndb * 32 + 32);
    int32_t *dbi = (int32_t *)T.take((size_t)nq * 4), *dbd = (int32_t *)T.take((size_t)nq * 4),
            *dsd = (int32_t *)T.take((size_t)nq * 4);
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
    HIPCHK(c, hipSetDevice(c->device));
    TmpDev T(c);
    int rc;
    if ((rc = T.reserve((size_t)nq * 32 + (size_t)ndb * 32 + (size_t)nq * 16 + (size_t)ncand * 4 + 4096))) return rc;
    uint8_t *dq = (uint8_t *)T.take((size_t)nq * 32), *ddb = (uint8_t *)T.take((size_t)ndb * 32 + 32);
    int32_t *doff = (int32_t *)T.take((size_t)(nq + 1) * 4), *dcand = (int32_t *)T.take((size_t)ncand * 4 + 4);
    int32_t *dbi = (int32_t *)T.take((size_t)nq * 4), *dbd = (int32_t *)T.take((size_t)nq * 4),
            *dsd = (int32_t *)T.take((size_t)nq * 4);
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
static void three_maxima(const std::vector<int> *histo, int L, int &ind1, int &ind2, int &ind3)
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
    for (int t = 0; t < m1; t++)
        if (idx1[t] < 0 || idx1[t] >= n1) return fail(c, ORBHIP_E_ARG, "idx1 out of range");
    for (int t = 0; t < m2; t++)
        if (idx2[t] < 0 || idx2[t] >= n2) return fail(c, ORBHIP_E_ARG, "idx2 out of range");
    HIPCHK(c, hipSetDevice(c->device));
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
    // rotation consistency (ref: :236-246, :267-285): histogram in the reference's visiting order
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
        three_maxima(hist, 30, i1, i2, i3);
        for (int i = 0; i < 30; i++) {
            if (i == i1 || i == i2 || i == i3) continue;
            for (int a : hist[i]) {
                match21[match12[a]] = -1;
                match12[a] = -1;
                nm--;
            }
        }
    }
    *nmatches = nm;
    return ORBHIP_OK;
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
    HIPCHK(c, hipSetDevice(c->device));
    const size_t ne = (size_t)H.nnodes - 1;
    // one device block, 256-byte aligned sections (tables by edge, see OrbVocabDev)
    size_t off[5], total = 0;
    const size_t sizes[5] = {ne * 32, ne * 8, ne * 4, ne * 4, ne * 4};
    for (int i = 0; i < 5; i++) {
        off[i] = total;
        total += align_up(sizes[i], 256);
    }
    if (c->d_vocBlock) HIPCHK(c, hipFree(c->d_vocBlock));
    c->d_vocBlock = nullptr;
    c->voc = OrbVocabDev();
    HIPCHK(c, hipMalloc(&c->d_vocBlock, total));
    uint8_t *base = (uint8_t *)c->d_vocBlock;
    const void *src[5] = {H.edesc.data(), H.erange.data(), H.child.data(), H.eword.data(), H.eweight.data()};
    for (int i = 0; i < 5; i++) HIPCHK(c, hipMemcpyAsync(base + off[i], src[i], sizes[i], hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    OrbVocabDev &V = c->voc;
    V.k = H.k; V.L = H.L; V.scoring = H.scoring; V.weighting = H.weighting; V.nnodes = H.nnodes; V.nwords = H.nwords;
    V.rootFirst = H.childOff[0];
    V.rootLast = H.childOff[1];
    V.desc = base + off[0];
    V.erange = (int32_t *)(base + off[1]);
    V.eid = (int32_t *)(base + off[2]);
    V.eword = (int32_t *)(base + off[3]);
    V.eweight = (float *)(base + off[4]);
    return ORBHIP_OK;
}

extern "C" int orbhip_vocab_load_device(orbhip_ctx *c, const void *d_blob, size_t nbytes)
{
    if (!c || !d_blob || nbytes < 24) return fail(c, ORBHIP_E_ARG, "orbhip_vocab_load_device: bad argument");
    HIPCHK(c, hipSetDevice(c->device));
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
    HIPCHK(c, hipSetDevice(c->device));
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
    HIPCHK(c, hipSetDevice(c->device));
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
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipEventRecord(c->ev[6], c->stream));
    launch_bow_seq(c->stream, (const uint8_t *)d_desc, (const orbhip_keypoint *)d_kps, (const int32_t *)d_counts,
                   (const int32_t *)d_node, (const float *)d_weight, (const uint8_t *)d_valid, cap, B, lag, 50, th_mode,
                   nnratio, check_ori, (int32_t *)d_match12, (int32_t *)d_match21, (int32_t *)d_nmatches);
    HIPCHK(c, hipEventRecord(c->ev[7], c->stream));
    HIPCHK(c, hipGetLastError());
    c->haveMatchEvents = true;
    return ORBHIP_OK;
}

// ------------------------------------------------------------------------------------------------
// stereo
// ------------------------------------------------------------------------------------------------
extern "C" int orbhip_stereo_match_device(orbhip_ctx *L, orbhip_ctx *R, const void *d_kpsL, const void *d_descL,
                                          const void *d_cntL, const void *d_kpsR, const void *d_descR,
                                          const void *d_cntR, int cap, int B, float mb, float mbf, void *d_uRight,
                                          void *d_depth, void *d_nmatch)
{
    if (!L || !R || !d_kpsL || !d_descL || !d_cntL || !d_kpsR || !d_descR || !d_cntR || cap <= 0 || B <= 0 ||
        !d_uRight || !d_depth || !d_nmatch || !(mb > 0.f) || !(mbf > 0.f))
        return fail(L, ORBHIP_E_ARG, "orbhip_stereo_match_device: bad argument");
    if (!L->last_lvl0 || !R->last_lvl0 || L->cur_w != R->cur_w || L->cur_h != R->cur_h || L->nlevels != R->nlevels ||
        B > L->last_B || B > R->last_B || L->device != R->device || L->cur_h > 4095 || cap > 65535)
        return fail(L, ORBHIP_E_ARG, "orbhip_stereo_match_device: both contexts must have just extracted images of the "
                                     "same size (at most 4095 rows, at most 65535 keypoints per image) on the same device");
    HIPCHK(L, hipSetDevice(L->device));
    int rc;
    if ((rc = match_scratch(L, stereo_scratch_bytes(B, cap)))) return rc;
    // the right pyramid / keypoints are produced on the right context's stream
    HIPCHK(L, hipEventRecord(R->evx[0], R->stream));
    HIPCHK(L, hipStreamWaitEvent(L->stream, R->evx[0], 0));
    launch_stereo(L, R, (const orbhip_keypoint *)d_kpsL, (const uint8_t *)d_descL, (const int32_t *)d_cntL,
                  (const orbhip_keypoint *)d_kpsR, (const uint8_t *)d_descR, (const int32_t *)d_cntR, cap, B, mb, mbf,
                  (float *)d_uRight, (float *)d_depth, (int32_t *)L->d_match, (int32_t *)d_nmatch);
    HIPCHK(L, hipGetLastError());
    // ... and the right context must not overwrite its pyramid before these kernels have read it
    HIPCHK(L, hipEventRecord(L->evx[0], L->stream));
    HIPCHK(L, hipStreamWaitEvent(R->stream, L->evx[0], 0));
    return ORBHIP_OK;
}

extern "C" int orbhip_stereo_match(orbhip_ctx *L, orbhip_ctx *R, const orbhip_keypoint *kpsL, const uint8_t *descL, int nL,
                                   const orbhip_keypoint *kpsR, const uint8_t *descR, int nR, float mb, float mbf,
                                   float *mvuRight, float *mvDepth, int *nmatch)
{
    if (!L || !R || nL < 0 || nR < 0 || (nL > 0 && (!kpsL || !descL || !mvuRight || !mvDepth)) || (nR > 0 && (!kpsR || !descR)))
        return fail(L, ORBHIP_E_ARG, "orbhip_stereo_match: bad argument");
    if (nmatch) *nmatch = 0;
    for (int i = 0; i < nL; i++) mvuRight[i] = mvDepth[i] = -1.0f;
    if (nL == 0 || nR == 0) return ORBHIP_OK;
    HIPCHK(L, hipSetDevice(L->device));
    const int cap = std::max(nL, nR);
    Packed P(L);
    int rc;
    if ((rc = P.begin((size_t)cap * (28 + 32) * 2 + (size_t)cap * 8 + 16 * 256))) return rc;
    // (device arrays of `cap` slots each; only the first nL / nR entries travel)
    const int32_t cnts[4] = {nL, nR, 0, 0};
    const orbhip_keypoint *dkL = (const orbhip_keypoint *)P.in(kpsL, (size_t)nL * 28);
    P.off += (size_t)(cap - nL) * 28;
    const orbhip_keypoint *dkR = (const orbhip_keypoint *)P.in(kpsR, (size_t)nR * 28);
    P.off += (size_t)(cap - nR) * 28;
    const uint8_t *ddL = (const uint8_t *)P.in(descL, (size_t)nL * 32);
    P.off += (size_t)(cap - nL) * 32;
    const uint8_t *ddR = (const uint8_t *)P.in(descR, (size_t)nR * 32);
    P.off += (size_t)(cap - nR) * 32;
    int32_t *dc = (int32_t *)P.in(cnts, 16);                     // nL | nR | matches before the median cut (comes back)
    float *du = (float *)P.out((size_t)cap * 4), *dz = (float *)P.out((size_t)cap * 4);
    P.inEnd = (size_t)((uint8_t *)dc - P.d) + 16;
    if ((rc = P.upload())) return rc;
    if ((rc = orbhip_stereo_match_device(L, R, dkL, ddL, dc, dkR, ddR, dc + 1, cap, 1, mb, mbf, du, dz, dc + 2))) return rc;
    if ((rc = P.download(dc))) return rc;
    memcpy(mvuRight, P.host(du), (size_t)nL * 4);
    memcpy(mvDepth, P.host(dz), (size_t)nL * 4);
    if (nmatch) *nmatch = ((const int32_t *)P.host(dc))[2];
    return ORBHIP_OK;
}

// ------------------------------------------------------------------------------------------------
// frame grid and guided search
// ------------------------------------------------------------------------------------------------
static bool grid_params_ok(float inv_w, float inv_h) { return inv_w > 0.f && inv_h > 0.f; }

extern "C" int orbhip_grid_build_device(orbhip_ctx *c, const void *d_kps, const void *d_counts, int cap, int B, float min_x,
                                        float min_y, float inv_w, float inv_h, void *d_cell_off, void *d_cell_idx)
{
    if (!c || !d_kps || !d_counts || cap <= 0 || B <= 0 || !d_cell_off || !d_cell_idx || !grid_params_ok(inv_w, inv_h))
        return fail(c, ORBHIP_E_ARG, "orbhip_grid_build_device: bad argument");
    HIPCHK(c, hipSetDevice(c->device));
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
    HIPCHK(c, hipSetDevice(c->device));
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
    HIPCHK(c, hipSetDevice(c->device));
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
    HIPCHK(c, hipSetDevice(c->device));
    int rc;
    if ((rc = match_scratch(c, proj_scratch_bytes(B, cap_q, cap)))) return rc;
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
    HIPCHK(c, hipSetDevice(c->device));
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

extern "C" int orbhip_distinctive_descriptors_device(orbhip_ctx *c, const void *d_desc, const void *d_off, int P, void *d_best,
                                                    void *d_best_median)
{
    if (!c || P < 0 || (P > 0 && (!d_desc || !d_off || !d_best || !d_best_median)))
        return fail(c, ORBHIP_E_ARG, "orbhip_distinctive_descriptors_device: bad argument");
    if (P == 0) return ORBHIP_OK;
    HIPCHK(c, hipSetDevice(c->device));
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
    HIPCHK(c, hipSetDevice(c->device));
    TmpDev T(c);
    int rc;
    if ((rc = T.reserve((size_t)total * 32 + (size_t)(P + 1) * 4 + (size_t)P * 8 + 4096))) return rc;
    uint8_t *dd = (uint8_t *)T.take((size_t)total * 32 + 32);
    int32_t *doff = (int32_t *)T.take((size_t)(P + 1) * 4), *db = (int32_t *)T.take((size_t)P * 4),
            *dm = (int32_t *)T.take((size_t)P * 4);
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
    HIPCHK(c, hipSetDevice(c->device));
    int rc;
    if ((rc = match_scratch(c, window_best_scratch_bytes(B, cap)))) return rc;
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
    HIPCHK(c, hipSetDevice(c->device));
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
    HIPCHK(c, hipSetDevice(c->device));
    int rc;
    if ((rc = match_scratch(c, init_scratch_bytes(B, cap1, cap2)))) return rc;
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
    HIPCHK(c, hipSetDevice(c->device));
    Packed P(c);
    int rc;
    if ((rc = P.begin((size_t)n1 * (28 + 32 + 8 + 4) + (size_t)n2 * (28 + 32 + 4) + (ORBHIP_GRID_CELLS + 1) * 4 + 16 * 256))) return rc;
    const int32_t cnts[4] = {n1, n2, 0, 0};
    const orbhip_keypoint *dk1 = (const orbhip_keypoint *)P.in(kps1, (size_t)n1 * 28), *dk2 = (const orbhip_keypoint *)P.in(kps2, (size_t)n2 * 28);
    const uint8_t *dd1 = (const uint8_t *)P.in(desc1, (size_t)n1 * 32), *dd2 = (const uint8_t *)P.in(desc2, (size_t)n2 * 32);
    // counts (incl. the number of matches) | vbPrevMatched (in and out) | matches: adjacent, one copy back
    int32_t *dc = (int32_t *)P.in(cnts, 16);
    float *dpm = (float *)P.in(prev_matched, (size_t)n1 * 8);
    int32_t *dm = (int32_t *)P.out((size_t)n1 * 4);
    const size_t backEnd = (size_t)((uint8_t *)dm - P.d) + (size_t)n1 * 4;
    int32_t *doff = (int32_t *)P.out((ORBHIP_GRID_CELLS + 1) * 4), *didx = (int32_t *)P.out((size_t)n2 * 4);   // device scratch
    if ((rc = P.upload())) return rc;
    if ((rc = orbhip_grid_build_device(c, dk2, dc + 1, n2, 1, min_x, min_y, inv_w, inv_h, doff, didx))) return rc;
    if ((rc = orbhip_search_for_initialization_device(c, dk1, dd1, dc, n1, dk2, dd2, dc + 1, n2, 1, min_x, min_y, inv_w, inv_h, doff,
                                                      didx, dpm, window_size, nnratio, check_ori, dm, dc + 2)))
        return rc;
    P.off = backEnd;
    if ((rc = P.download(dc))) return rc;
    memcpy(matches12, P.host(dm), (size_t)n1 * 4);
    memcpy(prev_matched, P.host(dpm), (size_t)n1 * 8);
    if (nmatches) *nmatches = ((const int32_t *)P.host(dc))[2];
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
    HIPCHK(c, hipSetDevice(c->device));
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
        three_maxima(hist, 30, a, b, d);
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

// ------------------------------------------------------------------------------------------------
// undistortion / rectification
// ------------------------------------------------------------------------------------------------
static bool dist_ok(const float *K, const float *dist, int ndist)
{
    return K && K[0] != 0.f && K[4] != 0.f && (ndist == 0 || ((ndist == 4 || ndist == 5 || ndist == 8) && dist));
}

extern "C" int orbhip_undistort_keypoints_device(orbhip_ctx *c, const void *d_kps, const void *d_counts, int cap, int B,
                                                 const float K[9], const float *dist, int ndist, const float *P,
                                                 void *d_kps_un)
{
    if (!c || !d_kps || !d_kps_un || cap <= 0 || B <= 0 || !dist_ok(K, dist, ndist))
        return fail(c, ORBHIP_E_ARG, "orbhip_undistort_keypoints_device: bad argument");
    HIPCHK(c, hipSetDevice(c->device));
    launch_undistort(c->stream, (const orbhip_keypoint *)d_kps, (const int32_t *)d_counts, cap, B, K, dist, ndist, P,
                     (orbhip_keypoint *)d_kps_un);
    HIPCHK(c, hipGetLastError());
    return ORBHIP_OK;
}

extern "C" int orbhip_undistort_keypoints(orbhip_ctx *c, const orbhip_keypoint *kps, int n, const float K[9],
                                          const float *dist, int ndist, const float *P, orbhip_keypoint *kps_un)
{
    if (!c || n < 0 || (n > 0 && (!kps || !kps_un)) || !dist_ok(K, dist, ndist))
        return fail(c, ORBHIP_E_ARG, "orbhip_undistort_keypoints: bad argument");
    if (n == 0) return ORBHIP_OK;
    HIPCHK(c, hipSetDevice(c->device));
    TmpDev T(c);
    int rc;
    if ((rc = T.reserve((size_t)n * 56 + 1024))) return rc;
    orbhip_keypoint *di = (orbhip_keypoint *)T.take((size_t)n * 28), *dout = (orbhip_keypoint *)T.take((size_t)n * 28);
    HIPCHK(c, hipMemcpyAsync(di, kps, (size_t)n * 28, hipMemcpyHostToDevice, c->stream));
    if ((rc = orbhip_undistort_keypoints_device(c, di, nullptr, n, 1, K, dist, ndist, P, dout))) return rc;
    HIPCHK(c, hipMemcpyAsync(kps_un, dout, (size_t)n * 28, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return ORBHIP_OK;
}

extern "C" int orbhip_init_undistort_rectify_map(const double K[9], const double *dist, int ndist, const double R[9],
                                                 const double P[9], int w, int h, float *map_x, float *map_y)
{
    if (!K || !R || !P || w <= 0 || h <= 0 || !map_x || !map_y || ndist < 0 || (ndist > 0 && !dist)) return ORBHIP_E_ARG;
    orb_init_undistort_rectify_map(K, dist, ndist, R, P, w, h, map_x, map_y);
    return ORBHIP_OK;
}

extern "C" int orbhip_remap_set_maps(orbhip_ctx *c, const float *map_x, const float *map_y, int w, int h)
{
    if (!c || !map_x || !map_y || w <= 0 || h <= 0) return fail(c, ORBHIP_E_ARG, "orbhip_remap_set_maps: bad argument");
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    if (c->d_maps) HIPCHK(c, hipFree(c->d_maps));
    c->d_maps = nullptr;
    c->map_w = c->map_h = 0;
    const size_t n = (size_t)w * h;
    HIPCHK(c, hipMalloc((void **)&c->d_maps, 2 * n * sizeof(float) + 64));
    HIPCHK(c, hipMemcpy(c->d_maps, map_x, n * sizeof(float), hipMemcpyHostToDevice));
    HIPCHK(c, hipMemcpy(c->d_maps + n, map_y, n * sizeof(float), hipMemcpyHostToDevice));
    c->map_w = w;
    c->map_h = h;
    return ORBHIP_OK;
}

extern "C" int orbhip_remap_device(orbhip_ctx *c, const void *d_src, int B, int src_w, int src_h, int src_stride,
                                   size_t src_frame_stride, void *d_dst, int dst_stride, size_t dst_frame_stride)
{
    if (!c || !d_src || !d_dst || B <= 0 || src_w <= 0 || src_h <= 0 || src_stride < src_w || src_w > 32767 || src_h > 32767)
        return fail(c, ORBHIP_E_ARG, "orbhip_remap_device: bad argument");
    if (!c->d_maps) return fail(c, ORBHIP_E_ARG, "orbhip_remap_device: no maps (orbhip_remap_set_maps)");
    if (dst_stride < c->map_w) return fail(c, ORBHIP_E_ARG, "orbhip_remap_device: dst_stride smaller than the map width");
    HIPCHK(c, hipSetDevice(c->device));
    launch_remap(c->stream, (const uint8_t *)d_src, B, src_w, src_h, src_stride, src_frame_stride, c->d_maps,
                 c->d_maps + (size_t)c->map_w * c->map_h, c->map_w, c->map_h, (uint8_t *)d_dst, dst_stride, dst_frame_stride);
    HIPCHK(c, hipGetLastError());
    return ORBHIP_OK;
}

extern "C" int orbhip_remap(orbhip_ctx *c, const uint8_t *src, int src_w, int src_h, int src_stride, uint8_t *dst,
                            int dst_stride)
{
    if (!c || !src || !dst || src_w <= 0 || src_h <= 0 || src_stride < src_w)
        return fail(c, ORBHIP_E_ARG, "orbhip_remap: bad argument");
    if (!c->d_maps) return fail(c, ORBHIP_E_ARG, "orbhip_remap: no maps (orbhip_remap_set_maps)");
    if (dst_stride < c->map_w) return fail(c, ORBHIP_E_ARG, "orbhip_remap: dst_stride smaller than the map width");
    HIPCHK(c, hipSetDevice(c->device));
    TmpDev T(c);
    int rc;
    const size_t sbytes = (size_t)src_stride * src_h, dpitch = align_up((size_t)c->map_w, 64), dbytes = dpitch * c->map_h;
    if ((rc = T.reserve(sbytes + dbytes + 1024))) return rc;
    uint8_t *ds = (uint8_t *)T.take(sbytes), *dd = (uint8_t *)T.take(dbytes);
    HIPCHK(c, hipMemcpyAsync(ds, src, sbytes, hipMemcpyHostToDevice, c->stream));
    if ((rc = orbhip_remap_device(c, ds, 1, src_w, src_h, src_stride, sbytes, dd, (int)dpitch, dbytes))) return rc;
    HIPCHK(c, hipMemcpy2DAsync(dst, dst_stride, dd, dpitch, c->map_w, c->map_h, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return ORBHIP_OK;
}

// ------------------------------------------------------------------------------------------------
// RCCL (loaded lazily so that the library has no hard link-time dependency on it)
// ------------------------------------------------------------------------------------------------
typedef struct { char internal[128]; } rccl_uid_t;
typedef int (*fn_getuid)(rccl_uid_t *);
typedef int (*fn_initrank)(void **, int, rccl_uid_t, int);
typedef int (*fn_bcast)(const void *, void *, size_t, int, int, void *, hipStream_t);
typedef int (*fn_allgather)(const void *, void *, size_t, int, void *, hipStream_t);
typedef int (*fn_destroy)(void *);
typedef const char *(*fn_errstr)(int);
static struct {
    void *h = nullptr;
    bool tried = false, ok = false;
    fn_getuid getuid = nullptr;
    fn_initrank initrank = nullptr;
    fn_bcast bcast = nullptr;
    fn_allgather allgather = nullptr;
    fn_destroy destroy = nullptr;
    fn_errstr errstr = nullptr;
} g_rccl;
static std::mutex g_rccl_mutex;

// One attempt per process; the outcome (every required symbol resolved) is what later calls see.
static bool rccl_load()
{
    std::lock_guard<std::mutex> lk(g_rccl_mutex);
    if (g_rccl.tried) return g_rccl.ok;
    g_rccl.tried = true;
    const char *names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    for (const char *n : names) {
        g_rccl.h = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
        if (g_rccl.h) break;
    }
    if (!g_rccl.h) return false;
    g_rccl.getuid = (fn_getuid)dlsym(g_rccl.h, "ncclGetUniqueId");
    g_rccl.initrank = (fn_initrank)dlsym(g_rccl.h, "ncclCommInitRank");
    g_rccl.bcast = (fn_bcast)dlsym(g_rccl.h, "ncclBroadcast");
    g_rccl.allgather = (fn_allgather)dlsym(g_rccl.h, "ncclAllGather");
    g_rccl.destroy = (fn_destroy)dlsym(g_rccl.h, "ncclCommDestroy");
    g_rccl.errstr = (fn_errstr)dlsym(g_rccl.h, "ncclGetErrorString");
    g_rccl.ok = g_rccl.getuid && g_rccl.initrank && g_rccl.bcast && g_rccl.allgather && g_rccl.destroy;
    return g_rccl.ok;
}

static std::string rccl_err(const char *what, int rc)
{
    return std::string(what) + ": " + (g_rccl.errstr ? g_rccl.errstr(rc) : "error");
}

// called by orbhip_destroy (above): the communicator belongs to the context
static void comm_release(orbhip_ctx *c)
{
    if (c->comm && g_rccl.ok) (void)g_rccl.destroy(c->comm);
    c->comm = nullptr;
    c->nranks = 1;
    c->rank = 0;
}

extern "C" int orbhip_comm_unique_id(uint8_t uid[128])
{
    if (!uid) return ORBHIP_E_ARG;
    if (!rccl_load()) return fail(nullptr, ORBHIP_E_COMM, "cannot load librccl (or it lacks a required symbol)");
    rccl_uid_t u;
    int rc = g_rccl.getuid(&u);
    if (rc != 0) return fail(nullptr, ORBHIP_E_COMM, rccl_err("ncclGetUniqueId", rc));
    memcpy(uid, u.internal, 128);
    return ORBHIP_OK;
}

extern "C" int orbhip_comm_init(orbhip_ctx *c, int rank, int nranks, const uint8_t uid[128])
{
    if (!c || !uid || nranks < 1 || rank < 0 || rank >= nranks) return fail(c, ORBHIP_E_ARG, "orbhip_comm_init: bad argument");
    if (!rccl_load()) return fail(c, ORBHIP_E_COMM, "cannot load librccl (or it lacks a required symbol)");
    HIPCHK(c, hipSetDevice(c->device));
    comm_release(c);
    rccl_uid_t u;
    memcpy(u.internal, uid, 128);
    int rc = g_rccl.initrank(&c->comm, nranks, u, rank);
    if (rc != 0) {
        c->comm = nullptr;
        return fail(c, ORBHIP_E_COMM, rccl_err("ncclCommInitRank", rc));
    }
    c->rank = rank;
    c->nranks = nranks;
    return ORBHIP_OK;
}

extern "C" int orbhip_comm_destroy(orbhip_ctx *c)
{
    if (!c) return ORBHIP_E_ARG;
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    comm_release(c);
    return ORBHIP_OK;
}

extern "C" int orbhip_bcast_blob_device(orbhip_ctx *c, void *d_buf, size_t nbytes, int root)
{
    if (!c || !d_buf) return fail(c, ORBHIP_E_ARG, "orbhip_bcast_blob_device: bad argument");
    if (!c->comm) {
        if (c->nranks == 1) return ORBHIP_OK;   // no communicator and a single rank: nothing to exchange
        return fail(c, ORBHIP_E_COMM, "orbhip_comm_init was not called");
    }
    if (root < 0 || root >= c->nranks) return fail(c, ORBHIP_E_ARG, "orbhip_bcast_blob_device: bad root");
    HIPCHK(c, hipSetDevice(c->device));
    // ncclChar = 0; with a communicator the collective runs also for one rank (an in-place no-op that exercises the RCCL path)
    int rc = g_rccl.bcast(d_buf, d_buf, nbytes, 0, root, c->comm, c->stream);
    if (rc != 0) return fail(c, ORBHIP_E_COMM, rccl_err("ncclBroadcast", rc));
    return ORBHIP_OK;
}

extern "C" int orbhip_knn2_merge_device(orbhip_ctx *c, const void *d_parts, int nshards, int nq, void *d_best_idx,
                                        void *d_best_d, void *d_second_d)
{
    if (!c || nshards < 1 || nq < 0 || (nq > 0 && (!d_parts || !d_best_idx || !d_best_d || !d_second_d)))
        return fail(c, ORBHIP_E_ARG, "orbhip_knn2_merge_device: bad argument");
    if (nq == 0) return ORBHIP_OK;
    HIPCHK(c, hipSetDevice(c->device));
    launch_knn2_merge(c->stream, (const int32_t *)d_parts, nshards, nq, (int32_t *)d_best_idx, (int32_t *)d_best_d,
                      (int32_t *)d_second_d);
    HIPCHK(c, hipGetLastError());
    return ORBHIP_OK;
}

extern "C" int orbhip_knn2_allgather_merge_device(orbhip_ctx *c, const void *d_best_idx_local, const void *d_best_d_local,
                                                  const void *d_second_d_local, int nq, int shard_offset, void *d_best_idx,
                                                  void *d_best_d, void *d_second_d)
{
    if (!c || nq < 0 || (nq > 0 && (!d_best_idx_local || !d_best_d_local || !d_second_d_local || !d_best_idx || !d_best_d ||
                                    !d_second_d)))
        return fail(c, ORBHIP_E_ARG, "orbhip_knn2_allgather_merge_device: bad argument");
    if (nq == 0) return ORBHIP_OK;
    if (c->nranks > 1 && !c->comm) return fail(c, ORBHIP_E_COMM, "orbhip_comm_init was not called");
    HIPCHK(c, hipSetDevice(c->device));
    // scratch: my part [3 * nq + 1] then the gathered parts [nranks][3 * nq + 1]; part = best_idx | best_d | second_d | offset
    const size_t part = (size_t)3 * nq + 1;
    int rc;
    if ((rc = match_scratch(c, (part * (size_t)(c->nranks + 1)) * 4 + 256))) return rc;
    int32_t *mine = (int32_t *)c->d_match, *all = mine + part;
    hipStream_t s = c->stream;
    HIPCHK(c, hipMemcpyAsync(mine, d_best_idx_local, (size_t)nq * 4, hipMemcpyDeviceToDevice, s));
    HIPCHK(c, hipMemcpyAsync(mine + nq, d_best_d_local, (size_t)nq * 4, hipMemcpyDeviceToDevice, s));
    HIPCHK(c, hipMemcpyAsync(mine + 2 * (size_t)nq, d_second_d_local, (size_t)nq * 4, hipMemcpyDeviceToDevice, s));
    launch_fill_i32(s, mine + 3 * (size_t)nq, shard_offset, 1);
    if (c->comm) {
        // ncclInt32 = 2: the one exchange step of database-sharded brute force, Q x 12 bytes per rank
        int nrc = g_rccl.allgather(mine, all, part, 2, c->comm, s);
        if (nrc != 0) return fail(c, ORBHIP_E_COMM, rccl_err("ncclAllGather", nrc));
    } else {
        HIPCHK(c, hipMemcpyAsync(all, mine, part * 4, hipMemcpyDeviceToDevice, s));
    }
    launch_knn2_merge(s, all, c->nranks, nq, (int32_t *)d_best_idx, (int32_t *)d_best_d, (int32_t *)d_second_d);
    HIPCHK(c, hipGetLastError());
    return ORBHIP_OK;
}
