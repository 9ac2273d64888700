// api_extract.hip -- C ABI (include/orbhip.h), part 1: context and buffer management, the per-batch launch sequence,
// host staging and the single-frame hipGraph, pyramid access and the parity/debug read-back.  No CPU fallback
// anywhere: every compute step is a HIP kernel.
#include "api_common.h"

static std::string g_last_error;
static std::mutex g_err_mutex;

int orb_fail(orbhip_ctx *c, int code, const std::string &msg)
{
    if (c)
        c->err = msg;
    else {
        std::lock_guard<std::mutex> lk(g_err_mutex);
        g_last_error = msg;
    }
    return code;
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
int orb_configure(orbhip_ctx *c, int w, int h, int stride0, int B)
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
        c->fuseBlurOk = c->nlevels > 1;
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
            c->resizeFit[l] = ResizeFit();
            if (c->resizeGroups[l] && !resize_fit_plan(xt.data(), yt.data(), c->G.lv[l - 1].w, c->G.lv[l - 1].h, c->G.lv[l].w, c->G.lv[l].h, c->resizeFit[l]))
                c->resizeFit[l] = ResizeFit();
            c->fuseBlurOk = c->fuseBlurOk && c->resizeGroups[l] &&
                            resize_blur_fits(xt.data(), yt.data(), c->G.lv[l - 1].w, c->G.lv[l - 1].h, c->G.lv[l].w, c->G.lv[l].h);
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
        c->allocGen++;
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
    c->blurPlace = std::min(2, std::max(0, ORB_TUNE("BLUR_PLACE", 0)));
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
    {
        uint32_t bands[6 * 64 * 4];
        blur_band_table(bands);
        if ((e = hipMalloc((void **)&c->d_blurBands, sizeof(bands))) != hipSuccess) return bail("hipMalloc", e);
        if ((e = hipMemcpy(c->d_blurBands, bands, sizeof(bands), hipMemcpyHostToDevice)) != hipSuccess) return bail("hipMemcpy", e);
    }
    // size everything for the largest image now, so per-frame calls never allocate
    const int stride0 = (int)align_up((size_t)max_w, 64);
    int rc = orb_configure(c, max_w, max_h, stride0, max_batch);
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
    static const int dbg = ORB_TUNE("DEBUG_DESTROY", 0);   // (ablation build: which step leaves a sticky HIP error behind?)
    auto step = [&](const char *what) {
        if (!dbg) return;
        const hipError_t e = hipGetLastError();
        if (e != hipSuccess) fprintf(stderr, "orbhip_destroy: sticky HIP error after %s: %s\n", what, hipGetErrorString(e));
    };
    step("entry");
    (void)hipSetDevice(c->device);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    step("sync");
    orb_comm_release(c);
    orb_pipe_release(c);
    step("comm/pipe");
    orb_sets_release(c);
    step("sets");
    orb_graph_release(c);
    step("graph");
    orb_frame_release(c);
    step("frame");
    if (c->h_in) (void)hipHostFree(c->h_in);
    if (c->h_pyr) (void)hipHostFree(c->h_pyr);
    void *bufs[] = {c->d_lvl0, c->d_pyr, c->d_blur, c->d_cand, c->d_cellCnt, c->d_pts, c->d_pnode,
                    c->d_lvlCandCnt, c->d_lvlKp, c->d_lvlKpCnt, c->d_lvlAngle, c->d_kps, c->d_desc,
                    c->d_counts, c->d_qtTables, c->d_fastTiles, c->d_blurTiles, c->d_blurBands, c->d_chainTiles, c->d_resizeTab, c->d_match, c->d_maps, c->d_tmp};
    for (void *b : bufs)
        if (b) (void)hipFree(b);
    c->vocHold.reset();   // (the block itself lives on while another context borrows it)
    if (c->h_stage) (void)hipHostFree(c->h_stage);
    if (c->h_pack) (void)hipHostFree(c->h_pack);
    for (int i = 0; i < 8; i++)
        if (c->ev[i]) (void)hipEventDestroy(c->ev[i]);
    for (int i = 0; i < 3; i++)
        if (c->evx[i]) (void)hipEventDestroy(c->evx[i]);
    for (int i = 0; i < 2; i++)
        if (c->evp[i]) (void)hipEventDestroy(c->evp[i]);
    step("events");
    if (c->stream2) (void)hipStreamDestroy(c->stream2);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    step("streams");
    // Best-effort clean-up must not leave the runtime's sticky last error behind for whatever the thread calls next: e.g.
    // hipEventSynchronize on a busy event whose recording stream -- another context's, destroyed before this one -- is gone can
    // report "operation not permitted on an event last recorded in a capturing stream" (seen twice in 4440 soak configurations, r06;
    // the next context's graph capture then failed on it).
    (void)hipGetLastError();
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
int orb_run_pipeline(orbhip_ctx *c, const uint8_t *lvl0, int stride0, size_t frame0, int B,
                        orbhip_keypoint *d_kps, uint8_t *d_desc, int32_t *d_counts, int cap, uint8_t *h_pyr_dst)
{
    const OrbLevels &G = c->G;
    hipStream_t s = c->stream;
    // The chain ends with hipGetLastError() (a launch reports its failure there): start from a clean slate, so that a sticky error
    // some EARLIER call of this thread left behind -- the caller's own HIP code, a best-effort clean-up -- is not taken for ours.
    (void)hipGetLastError();
    // (timing events are not recorded into a graph capture: events recorded by a graph node cannot be read back with
    // hipEventElapsedTime on this runtime; the graph path refreshes the stage times with an eager run now and then)
    // (an event record between two kernels of a stream costs ~4 us of device time: orbhip_set_stage_timing narrows the set)
    const bool ev = !c->capturing && c->stageTiming >= 2;
    const bool evFast = !c->capturing && c->stageTiming >= 1;
    if (ev) HIPCHK(c, hipEventRecord(c->ev[0], s));
    // E2 pyramid.  A frame or two: several levels per launch (k_pyramid_chain; ORBHIP_NO_CHAIN=1 keeps one launch per level)
    static const bool noChain = ORB_SWITCH("NO_CHAIN", 0) != 0;
    const bool chained = B < 8 && !noChain && !c->chainGroups.empty();
    if (chained)
        for (const ChainGroup &grp : c->chainGroups)
            launch_pyramid_chain(s, G, c->chainLevels, grp, c->d_chainTiles, lvl0, stride0, frame0, c->d_pyr, c->pyrFrameBytes,
                                 c->d_resizeTab, B, h_pyr_dst);   // (the host copy of the pyramid is written by the kernel itself)
    // batches: level l from level l-1 (sequential dependency), all frames per launch
    // ORBHIP_FUSE_BLUR=1 (r05 experiment): levels 1.. and their blurred twins from one kernel per level; k_blur keeps level 0
    static const bool fuseSwitch = ORB_TUNE("FUSE_BLUR", 0) != 0;   // (liborbhip_ablation.so only: measured slower)
    const bool fuseBlur = fuseSwitch && B >= 8 && !chained && c->fuseBlurOk;
    static const bool fitTiles = ORB_TUNE("RESIZE_FIT", 1) != 0;   // batches: tiles fitted to the level (k_resize_fit)
    for (int l = 1; l < G.nlevels && !chained; l++) {
        const OrbLevel &S = G.lv[l - 1], &D = G.lv[l];
        const uint8_t *src = (l == 1) ? lvl0 : c->d_pyr + S.imgOff;
        const int sstride = (l == 1) ? stride0 : S.stride;
        const size_t sframe = (l == 1) ? frame0 : c->pyrFrameBytes;
        if (fuseBlur) {
            launch_resize_blur(s, src, S.w, S.h, sstride, sframe, c->d_pyr + D.imgOff, D.w, D.h, D.stride, c->pyrFrameBytes,
                               c->d_blur + G.boff1 + D.imgOff, D.stride, c->lvl0FrameBytes + c->pyrFrameBytes,
                               c->d_resizeTab + c->resizeTabOff[l][1], c->d_resizeTab + c->resizeTabOff[l][2], c->d_blurBands, B);
            continue;
        }
        if (fitTiles && B >= 8 && c->resizeFit[l].ntx > 0) {
            launch_resize_fit(s, src, S.w, S.h, sstride, sframe, c->d_pyr + D.imgOff, D.w, D.h, D.stride, c->pyrFrameBytes,
                              c->d_resizeTab + c->resizeTabOff[l][1], c->d_resizeTab + c->resizeTabOff[l][2], c->resizeFit[l], B);
            continue;
        }
        launch_resize(s, src, S.w, S.h, sstride, sframe, c->d_pyr + D.imgOff, D.w, D.h, D.stride,
                      c->pyrFrameBytes, c->d_resizeTab + c->resizeTabOff[l][0],
                      c->d_resizeTab + c->resizeTabOff[l][1],
                      c->resizeGroups[l] ? c->d_resizeTab + c->resizeTabOff[l][2] : nullptr, c->resizeHint[l][B >= 8 ? 0 : 1], B);
    }
    if (evFast) HIPCHK(c, hipEventRecord(c->ev[1], s));
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
    // E3 FAST, E4 quadtree, E6 blur, E5 + E7 describe.  FAST runs ALONE on the device (it is the kernel whose roofline is
    // reported).  Batches, default schedule (blur placement 0): the quadtree is a latency-bound kernel of few long workgroups,
    // so it is cut in two half-batches that hide behind kernels which do not depend on them -- the first half beside the blur
    // (second stream), the second half beside the describe kernel of the FIRST half:
    //     main stream  : FAST(all) | quadtree(A) | describe(A)            | describe(B)
    //     second stream:           | blur(all)   | quadtree(B)            |
    // Other placements of the blur (orbhip_set_blur_placement): 1 = one launch from the end of the pyramid, beside FAST and
    // the quadtree; 2 = alone on the main stream between FAST and the quadtree (every kernel owns the device: counters).
    // A frame or two: one cell per FAST workgroup (four times the workgroups, each a shorter chain -- the single-frame FAST
    // time is one workgroup's latency), everything on one stream (a cross-stream hand-over costs more than it hides).
    const int blurPlace = c->blurPlace;
    const size_t blurFrame = c->lvl0FrameBytes + c->pyrFrameBytes;
    // r06, batches: the blur inside the describe kernel (k_describe_blur) -- no blurred pyramid, no k_blur.  Schedule:
    //     main stream  : FAST(all) | quadtree(A) | describe+blur(A)       | describe+blur(B)
    //     second stream:                         | quadtree(B)            |
    // (orbhip_set_blur_placement 1 / 2 are measurements of k_blur: they keep the two-kernel path)
    const bool fusedDescribe = blurPlace == 0 && !fuseBlur && describe_blur_available(G, B);
    auto blur_all = [&](hipStream_t st) {
        launch_blur(st, G, lvl0, stride0, frame0, c->d_pyr, c->pyrFrameBytes, c->d_blur, blurFrame, c->d_blurTiles,
                    fuseBlur ? c->blurLevelFirst[1] : (int)c->blurTiles.size(), c->d_blurBands, B);
    };
    // quadtree / describe of the frames [b0, b0 + nb)
    const size_t qtPerFrame = B > 0 ? quadtree_table_scratch_bytes(G, 1) : 0;
    auto quadtree_part = [&](hipStream_t st, int b0, int nb) {
        launch_quadtree(st, G, c->d_cand + (size_t)b0 * G.totalCands, c->d_cellCnt + (size_t)b0 * G.totalCells,
                        c->d_pts + (size_t)b0 * G.totalPts, c->d_pnode + (size_t)b0 * G.totalPts,
                        c->d_lvlCandCnt + (size_t)b0 * ORBHIP_MAX_LEVELS, c->d_lvlKp + (size_t)b0 * G.totalKps,
                        c->d_lvlKpCnt + (size_t)b0 * ORBHIP_MAX_LEVELS, nb, c->d_qtTables ? c->d_qtTables + (size_t)b0 * qtPerFrame : nullptr);
    };
    auto describe_part = [&](hipStream_t st, int b0, int nb) {
        if (fusedDescribe) {
            launch_describe_blur(st, G, lvl0 + (size_t)b0 * frame0, stride0, frame0, c->d_pyr + (size_t)b0 * c->pyrFrameBytes,
                                 c->pyrFrameBytes, c->d_lvlKp + (size_t)b0 * G.totalKps, c->d_lvlKpCnt + (size_t)b0 * ORBHIP_MAX_LEVELS,
                                 c->d_lvlAngle + (size_t)b0 * G.totalKps, d_kps + (size_t)b0 * cap, d_desc + (size_t)b0 * cap * 32,
                                 d_counts + b0, cap, nb);
            return;
        }
        launch_describe(st, G, lvl0 + (size_t)b0 * frame0, stride0, frame0, c->d_pyr + (size_t)b0 * c->pyrFrameBytes, c->pyrFrameBytes,
                        c->d_blur + (size_t)b0 * blurFrame, blurFrame, c->d_lvlKp + (size_t)b0 * G.totalKps,
                        c->d_lvlKpCnt + (size_t)b0 * ORBHIP_MAX_LEVELS, c->d_lvlAngle + (size_t)b0 * G.totalKps, d_kps + (size_t)b0 * cap,
                        d_desc + (size_t)b0 * cap * 32, d_counts + b0, cap, nb, B < 8 ? c->describeMirror : 0);
    };
    if (B >= 8 && blurPlace == 1) {
        HIPCHK(c, hipEventRecord(c->evx[0], s));
        HIPCHK(c, hipStreamWaitEvent(c->stream2, c->evx[0], 0));
        if (ev) HIPCHK(c, hipEventRecord(c->evx[1], c->stream2));
        blur_all(c->stream2);
        HIPCHK(c, hipEventRecord(c->evx[2], c->stream2));
    }
    if (B >= 8)
        launch_fast(s, G, lvl0, stride0, frame0, c->d_pyr, c->pyrFrameBytes, c->d_fastTiles, c->nFastTilesBatch, c->d_cand,
                    c->d_cellCnt, B, c->nFastTilesTall);
    else
        launch_fast(s, G, lvl0, stride0, frame0, c->d_pyr, c->pyrFrameBytes, c->d_fastTiles + c->nFastTilesBatch,
                    (int)c->fastTiles.size() - c->nFastTilesBatch, c->d_cand, c->d_cellCnt, B);
    if (evFast) HIPCHK(c, hipEventRecord(c->ev[2], s));
#ifndef DF_SCHED
#define DF_SCHED 1
#endif
#ifndef DF_SPLIT
#define DF_SPLIT 4
#endif
    static const bool noSplit = ORB_TUNE("NO_SPLIT", 0) != 0;   // A/B: r02 schedule
    static const int fusedSched = ORB_TUNE("DESCRIBE_FUSED_SCHED", DF_SCHED);   // 1: quadtree(B) beside describe(A); 0: one quadtree launch
    static const int fusedSplit = ORB_TUNE("DESCRIBE_FUSED_SPLIT", DF_SPLIT);   // eighths of the batch in the first part
    if (fusedDescribe && B >= 16 && fusedSched >= 1) {
        const int nA = std::max(8, B * fusedSplit / 8), nB = B - nA;
        if (fusedSched == 2) {                                  // (A/B: both quadtree halves from the end of FAST, side by side)
            if (!evFast) HIPCHK(c, hipEventRecord(c->ev[2], s));
            HIPCHK(c, hipStreamWaitEvent(c->stream2, c->ev[2], 0));
        }
        quadtree_part(s, 0, nA);
        HIPCHK(c, hipEventRecord(c->ev[3], s));                 // quadtree(A) is done: the second stream may start quadtree(B)
        if (fusedSched == 1) HIPCHK(c, hipStreamWaitEvent(c->stream2, c->ev[3], 0));
        quadtree_part(c->stream2, nA, nB);
        HIPCHK(c, hipEventRecord(c->evx[0], c->stream2));
        if (ev) HIPCHK(c, hipEventRecord(c->ev[4], s));
        describe_part(s, 0, nA);
        HIPCHK(c, hipStreamWaitEvent(s, c->evx[0], 0));
        describe_part(s, nA, nB);
    } else if (fusedDescribe) {
        quadtree_part(s, 0, B);
        if (ev) HIPCHK(c, hipEventRecord(c->ev[3], s));
        if (ev) HIPCHK(c, hipEventRecord(c->ev[4], s));
        describe_part(s, 0, B);
    } else if (B >= 16 && blurPlace == 0 && !noSplit) {
        const int nA = B / 2, nB = B - nA;
        if (!evFast) HIPCHK(c, hipEventRecord(c->ev[2], s));   // (the hand-over event; the timing path has recorded it)
        HIPCHK(c, hipStreamWaitEvent(c->stream2, c->ev[2], 0));
        if (ev) HIPCHK(c, hipEventRecord(c->evx[1], c->stream2));
        blur_all(c->stream2);
        HIPCHK(c, hipEventRecord(c->evx[2], c->stream2));
        quadtree_part(c->stream2, nA, nB);                      // behind the blur, i.e. beside describe(A)
        HIPCHK(c, hipEventRecord(c->evx[0], c->stream2));
        quadtree_part(s, 0, nA);
        if (ev) HIPCHK(c, hipEventRecord(c->ev[3], s));
        HIPCHK(c, hipStreamWaitEvent(s, c->evx[2], 0));         // the blur is done
        if (ev) HIPCHK(c, hipEventRecord(c->ev[4], s));
        describe_part(s, 0, nA);
        HIPCHK(c, hipStreamWaitEvent(s, c->evx[0], 0));         // quadtree(B) is done
        describe_part(s, nA, nB);
    } else if (B >= 8) {
        if (blurPlace == 0) {
            if (!evFast) HIPCHK(c, hipEventRecord(c->ev[2], s));
            HIPCHK(c, hipStreamWaitEvent(c->stream2, c->ev[2], 0));
            if (ev) HIPCHK(c, hipEventRecord(c->evx[1], c->stream2));
            blur_all(c->stream2);
            HIPCHK(c, hipEventRecord(c->evx[2], c->stream2));
        } else if (blurPlace == 2) {
            if (ev) HIPCHK(c, hipEventRecord(c->evx[1], s));
            blur_all(s);
            HIPCHK(c, hipEventRecord(c->evx[2], s));
        }
        quadtree_part(s, 0, B);
        if (ev) HIPCHK(c, hipEventRecord(c->ev[3], s));
        if (blurPlace == 0 || blurPlace == 1) HIPCHK(c, hipStreamWaitEvent(s, c->evx[2], 0));
        if (ev) HIPCHK(c, hipEventRecord(c->ev[4], s));
        describe_part(s, 0, B);
    } else {
        quadtree_part(s, 0, B);
        if (ev) HIPCHK(c, hipEventRecord(c->ev[3], s));
        if (ev) HIPCHK(c, hipEventRecord(c->evx[1], s));
        blur_all(s);
        if (ev) HIPCHK(c, hipEventRecord(c->evx[2], s));
        if (ev) HIPCHK(c, hipEventRecord(c->ev[4], s));
        describe_part(s, 0, B);
    }
    if (ev) HIPCHK(c, hipEventRecord(c->ev[5], s));
    if (pyrFork)
        HIPCHK(c, hipStreamWaitEvent(s, c->evp[1], 0));
    else if (h_pyr_dst && G.nlevels > 1 && !chained)
        HIPCHK(c, hipMemcpyAsync(h_pyr_dst, c->d_pyr, (size_t)B * c->pyrFrameBytes, hipMemcpyDeviceToHost, s));
    HIPCHK(c, hipGetLastError());
    if (!c->capturing) {
        c->haveStageEvents = ev;
        c->haveFastEvents = evFast;
        c->haveMatchEvents = false;
    }
    c->last_lvl0 = lvl0;
    c->last_stride0 = stride0;
    c->last_frame0 = frame0;
    c->last_B = B;
    c->blurValid = !fusedDescribe;   // (orbhip_debug_get_blurred_level builds the blurred pyramid on request)
    return ORBHIP_OK;
}

extern "C" int orbhip_set_blur_placement(orbhip_ctx *c, int place)
{
    if (!c || place < 0 || place > 2) return fail(c, ORBHIP_E_ARG, "orbhip_set_blur_placement: 0, 1 or 2");
    c->blurPlace = place;
    return ORBHIP_OK;
}

extern "C" int orbhip_set_stage_timing(orbhip_ctx *c, int mode)
{
    if (!c || mode < 0 || mode > 2) return fail(c, ORBHIP_E_ARG, "orbhip_set_stage_timing: 0, 1 or 2");
    c->stageTiming = mode;
    return ORBHIP_OK;
}

extern "C" int orbhip_get_stage_times(orbhip_ctx *c, float ms[6])
{
    if (!c || !ms) return fail(c, ORBHIP_E_ARG, "orbhip_get_stage_times: bad argument");
    HIPCHK(c, orb_enter(c));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    for (int i = 0; i < 6; i++) ms[i] = 0.f;
    if (c->haveFastEvents && !c->haveStageEvents) HIPCHK(c, hipEventElapsedTime(&ms[1], c->ev[1], c->ev[2]));    // FAST
    if (c->haveStageEvents) {
        HIPCHK(c, hipStreamSynchronize(c->stream2));
        HIPCHK(c, hipEventElapsedTime(&ms[0], c->ev[0], c->ev[1]));    // pyramid
        HIPCHK(c, hipEventElapsedTime(&ms[1], c->ev[1], c->ev[2]));    // FAST
        HIPCHK(c, hipEventElapsedTime(&ms[2], c->ev[2], c->ev[3]));    // quadtree (overlaps the blur)
        if (c->blurValid) HIPCHK(c, hipEventElapsedTime(&ms[3], c->evx[1], c->evx[2]));  // blur (second stream; none with k_describe_blur)
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
    HIPCHK(c, orb_enter(c));
    c->h_pyr_B = 0;          // no host copy of this call's pyramid / level 0 (orbhip_host_pyramid_level reports that)
    c->h_in_valid = false;
    const bool aliasOk = (stride % 16 == 0) && (((uintptr_t)d_imgs) % 16 == 0) && (frame_stride % 16 == 0);
    int rc;
    if (aliasOk) {
        if ((rc = orb_configure(c, w, h, stride, B))) return rc;
        return orb_run_pipeline(c, (const uint8_t *)d_imgs, stride, frame_stride, B, (orbhip_keypoint *)d_kps,
                            (uint8_t *)d_desc, (int32_t *)d_counts, cap);
    }
    // unaligned input: repack into the context's level-0 buffer first
    const int s0 = (int)align_up((size_t)w, 64);
    if ((rc = orb_configure(c, w, h, s0, B))) return rc;
    for (int b = 0; b < B; b++)
        HIPCHK(c, hipMemcpy2DAsync(c->d_lvl0 + (size_t)b * c->lvl0FrameBytes, s0,
                                   (const uint8_t *)d_imgs + (size_t)b * frame_stride, stride, w, h,
                                   hipMemcpyDeviceToDevice, c->stream));
    return orb_run_pipeline(c, c->d_lvl0, s0, c->lvl0FrameBytes, B, (orbhip_keypoint *)d_kps, (uint8_t *)d_desc,
                        (int32_t *)d_counts, cap);
}

int orb_host_stage(orbhip_ctx *c, size_t bytes)
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
int orb_host_pyr_stage(orbhip_ctx *c, int B, uint8_t **dst)
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

void orb_graph_release(orbhip_ctx *c)
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
        orb_graph_release(c);
        if (c->h_in) HIPCHK(c, hipHostFree(c->h_in));
        c->h_in = nullptr;
        c->h_in_bytes = 0;
        void *p = nullptr;
        HIPCHK(c, hipHostMalloc(&p, inBytes, hipHostMallocDefault));
        c->h_in = (uint8_t *)p;
        c->h_in_bytes = inBytes;
    }
    int rc;
    if ((rc = orb_host_stage(c, coff + align_up(cbytes, 256)))) return rc;
    uint8_t *hpyr = nullptr;
    if ((rc = orb_host_pyr_stage(c, B, &hpyr))) return rc;
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
    static const bool directOut = ORB_TUNE("COPY_OUT", 0) == 0;   // A/B: 1 = result copy node
    const void *key[5] = {c->d_lvl0, blk, c->h_in, c->h_stage, hpyr};
    const bool same = c->g_exec && c->g_w == w && c->g_h == h && c->g_B == B && c->g_gen == c->allocGen &&
                      memcmp(key, c->g_key, sizeof(key)) == 0;
    if (!same) {
        orb_graph_release(c);
        HIPCHK(c, hipStreamSynchronize(c->stream));
        c->capturing = true;
        HIPCHK(c, hipStreamBeginCapture(c->stream, hipStreamCaptureModeThreadLocal));
        // the describe kernel writes keypoints, descriptors and counts straight into the page-locked result block (posted PCIe
        // writes of a few dozen KB that overlap the kernel): no copy node behind it -- that node started 8 us after describe ended
        uint8_t *out = directOut ? c->h_stage : blk;
        hipError_t e = hipMemcpyAsync(c->d_lvl0, c->h_in, inBytes, hipMemcpyHostToDevice, c->stream);
        rc = e == hipSuccess ? orb_run_pipeline(c, c->d_lvl0, s0, c->lvl0FrameBytes, B, (orbhip_keypoint *)(out + koff), out + doff,
                                            (int32_t *)(out + coff), dcap, hpyr)
                             : ORBHIP_E_HIP;
        if (rc == ORBHIP_OK && !directOut) e = hipMemcpyAsync(c->h_stage, blk, coff + cbytes, hipMemcpyDeviceToHost, c->stream);
        hipGraph_t g = nullptr;
        const hipError_t e2 = hipStreamEndCapture(c->stream, &g);
        c->capturing = false;
        if (rc != ORBHIP_OK || e != hipSuccess || e2 != hipSuccess || !g) {
            if (g) (void)hipGraphDestroy(g);
            const std::string inner = rc != ORBHIP_OK ? c->err : std::string();   // (what a call inside the captured chain reported)
            return fail(c, ORBHIP_E_HIP, std::string("graph capture of the single-frame chain failed: ") +
                                             hipGetErrorString(e != hipSuccess ? e : e2) + (inner.empty() ? "" : " [" + inner + "]"));
        }
        c->g_graph = g;
        HIPCHK(c, hipGraphInstantiate(&c->g_exec, g, nullptr, nullptr, 0));
        c->g_w = w; c->g_h = h; c->g_B = B;
        memcpy(c->g_key, key, sizeof(key));
        c->g_gen = c->allocGen;
        c->g_calls = 0;
    }
    if ((c->g_calls++ & 255u) == 0) {
        // the first call of a geometry and every 256th one run the same chain eagerly: that refreshes the stage times behind
        // GetTimeOfComputePyramid / ...KeyPointsOctTree / ...Descriptor (include/ORBextractor.h:51-53)
        uint8_t *out = directOut ? c->h_stage : blk;
        HIPCHK(c, hipMemcpyAsync(c->d_lvl0, c->h_in, inBytes, hipMemcpyHostToDevice, c->stream));
        if ((rc = orb_run_pipeline(c, c->d_lvl0, s0, c->lvl0FrameBytes, B, (orbhip_keypoint *)(out + koff), out + doff,
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
    HIPCHK(c, orb_enter(c));
    const int s0 = (int)align_up((size_t)w, 64);
    int rc;
    if ((rc = orb_configure(c, w, h, s0, B))) return rc;
    const int dcap = (int)c->cap_out;
    // results: keypoints | descriptors | counts of the B frames are one device block that goes to pinned staging in ONE
    // asynchronous copy behind the kernels and ONE synchronisation (every extra copy or wait is a device round trip --
    // most of a single frame's overhead); the n[b] valid entries are then copied out on the host
    const size_t kbytes = (size_t)B * dcap * sizeof(orbhip_keypoint), dbytes = (size_t)B * dcap * 32, cbytes = (size_t)B * 4;
    const size_t koff = 0, doff = align_up(kbytes, 256), coff = doff + align_up(dbytes, 256);
    uint8_t *blk = reinterpret_cast<uint8_t *>(c->d_kps);
    static const bool noGraph = ORB_SWITCH("NO_GRAPH", 0) != 0;
    if (B < 8 && !noGraph) {
        if ((rc = extract_small_graph(c, imgs, B, w, h, stride, s0, kbytes, dbytes, cbytes, koff, doff, coff, dcap))) return rc;
    } else {
        for (int b = 0; b < B; b++) {
            if (!imgs[b]) return fail(c, ORBHIP_E_ARG, "orbhip_extract_batch: null image");
            HIPCHK(c, hipMemcpy2DAsync(c->d_lvl0 + (size_t)b * c->lvl0FrameBytes, s0, imgs[b], stride, w, h,
                                       hipMemcpyHostToDevice, c->stream));
        }
        uint8_t *hpyr = nullptr;
        if ((rc = orb_host_pyr_stage(c, B, &hpyr))) return rc;
        c->h_in_valid = false;
        if ((rc = orb_run_pipeline(c, c->d_lvl0, s0, c->lvl0FrameBytes, B, (orbhip_keypoint *)(blk + koff), blk + doff,
                               (int32_t *)(blk + coff), dcap, hpyr)))
            return rc;
        if ((rc = orb_host_stage(c, coff + align_up(cbytes, 256)))) return rc;
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
    HIPCHK(c, orb_enter(c));
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
    HIPCHK(c, orb_enter(c));
    const size_t bf = c->lvl0FrameBytes + c->pyrFrameBytes;
    if (!c->blurValid) {
        // the last batch ran k_describe_blur and never built the blurred pyramid: k_blur on the same levels, now
        launch_blur(c->stream, c->G, c->last_lvl0, c->last_stride0, c->last_frame0, c->d_pyr, c->pyrFrameBytes, c->d_blur, bf,
                    c->d_blurTiles, (int)c->blurTiles.size(), c->d_blurBands, c->last_B);
        HIPCHK(c, hipGetLastError());
        c->blurValid = true;
    }
    const uint8_t *src = c->d_blur + (size_t)frame * bf + (level == 0 ? 0 : c->G.boff1 + L.imgOff);
    return copy_level(c, src, level == 0 ? c->G.bstride0 : L.stride, L.w, L.h, dst, dst_stride);
}

extern "C" int orbhip_debug_get_candidates(orbhip_ctx *c, int frame, int level, orbhip_cand *out, int cap,
                                           int *n_out)
{
    if (!c || !c->last_lvl0 || frame < 0 || frame >= c->last_B || level < 0 || level >= c->nlevels || !n_out)
        return fail(c, ORBHIP_E_ARG, "orbhip_debug_get_candidates: bad argument");
    HIPCHK(c, orb_enter(c));
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
    HIPCHK(c, orb_enter(c));
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

