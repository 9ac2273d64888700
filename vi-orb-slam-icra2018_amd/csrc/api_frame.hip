// api_frame.hip -- C ABI, part 8: the device work of the Frame constructor as ONE captured graph.
//
// Frame::Frame (ref: src/Frame.cc:518-572) runs ExtractORB (:591-597) -> UndistortKeyPoints (:748-778) ->
// AssignFeaturesToGrid (:574-589), and Tracking asks for ComputeBoW (:739-746) before the first SearchByBoW of the frame.
// As four entry points that is four launch + synchronise round trips for one dependency chain (0.121 + 0.038 + 0.048 +
// 0.040 ms in profiles/r03/percall_table.md; two of the four lost to one host core).  Here the chain is one graph:
//
//   image -> pinned staging -> [copy in] -> pyramid, FAST, quadtree, blur, describe (orb_run_pipeline)
//         -> k_undistort -> k_grid_build                          (main stream)
//         -> k_vocab_transform                                    (second stream, beside the two above)
//         -> one synchronisation; every kernel stores its results into the device block AND its page-locked twin
//
// and the block the results were written to stays on the device: orbhip_set_put_from_frame (api_sets.hip) turns it into a
// resident set of any context of the same device without the frame's keypoints / descriptors travelling again.
// The kernels are the ones behind orbhip_extract, orbhip_undistort_keypoints_device, orbhip_grid_build_device and
// orbhip_vocab_transform_device: same results by construction, checked against the four oracle calls in
// tests/test_frame_build.py.
#include "api_common.h"

#include <time.h>
#include <mutex>
#include <vector>

struct OrbFrameBuild {
    uint8_t *d_blk = nullptr;   // packed results, device
    uint8_t *h_blk = nullptr;   // page-locked twin
    size_t bytes = 0;
    int dcap = 0;
    // offsets inside the block; [oU, setEnd) has the layout of a resident set's block (api_sets.hip)
    size_t oU = 0, oD = 0, oC = 0, oG = 0, oE = 0, setEnd = 0, oK = 0, oW = 0, oT = 0, oN = 0;
    hipEvent_t evFork = nullptr, evJoin = nullptr;
    // Contexts that are copying the block (orbhip_set_put_from_frame) each leave an event of their own here, on any thread; the
    // owner's next orbhip_frame_build waits for all of them and hands the events back.  (One shared event was overwritten by a
    // second copier: the build then waited for the last copy only.)
    std::mutex busyMu;
    std::vector<hipEvent_t> busyEvents, freeEvents;
    hipGraph_t graph = nullptr;
    hipGraphExec_t exec = nullptr;
    // replay key
    int w = 0, h = 0;
    orbhip_frame_params fp;
    const void *key[7] = {};
    unsigned long gen = 0;
    unsigned calls = 0;
    // the frame whose results the block holds
    bool valid = false, undist = false, grid = false, bow = false;
    int n = 0;
    uint64_t fingerprint = 0;
};

static OrbFrameBuild *fb_of(orbhip_ctx *c) { return static_cast<OrbFrameBuild *>(c->frameBuild); }

static void fb_graph_release(OrbFrameBuild *F)
{
    if (F->exec) (void)hipGraphExecDestroy(F->exec);
    if (F->graph) (void)hipGraphDestroy(F->graph);
    F->exec = nullptr;
    F->graph = nullptr;
}

void orb_frame_release(orbhip_ctx *c)
{
    OrbFrameBuild *F = fb_of(c);
    if (!F) return;
    fb_graph_release(F);
    if (F->d_blk) (void)hipFree(F->d_blk);
    if (F->h_blk) (void)hipHostFree(F->h_blk);
    for (hipEvent_t e : F->busyEvents) (void)hipEventSynchronize(e);
    for (hipEvent_t e : F->busyEvents) (void)hipEventDestroy(e);
    for (hipEvent_t e : F->freeEvents) (void)hipEventDestroy(e);
    for (hipEvent_t e : {F->evFork, F->evJoin})
        if (e) (void)hipEventDestroy(e);
    delete F;
    c->frameBuild = nullptr;
}

// FNV-1a over the count, the first keypoint and the first and last descriptor: what tells two frames / key frames with the
// same id and the same number of features apart (Tracking::Reset restarts both id counters, ref: src/Tracking.cc:2758-2759).
extern "C" uint64_t orbhip_set_fingerprint_rows(const orbhip_keypoint *first_kp, const uint8_t *first_desc, const uint8_t *last_desc,
                                                int n)
{
    uint64_t h = 1469598103934665603ull;
    auto mix = [&](const void *p, size_t bytes) {
        const uint8_t *b = static_cast<const uint8_t *>(p);
        for (size_t i = 0; i < bytes; i++) h = (h ^ b[i]) * 1099511628211ull;
    };
    mix(&n, sizeof(n));
    if (n > 0 && first_kp) mix(first_kp, 8);           // pt.x, pt.y of the first keypoint
    if (n > 0 && first_desc && last_desc) {
        mix(first_desc, 32);
        mix(last_desc, 32);
    }
    return h ? h : 1;
}
extern "C" uint64_t orbhip_set_fingerprint(const orbhip_keypoint *kps, const uint8_t *desc, int n)
{
    return orbhip_set_fingerprint_rows(kps, desc, desc ? desc + (size_t)(n > 0 ? n - 1 : 0) * 32 : nullptr, n);
}

static bool fp_ok(const orbhip_frame_params *p)
{
    if (!p) return false;
    if (!(p->ndist == 0 || p->ndist == 4 || p->ndist == 5 || p->ndist == 8)) return false;
    if (p->ndist > 0 && p->dist[0] != 0.0f && !(p->K[0] != 0.0f && p->K[4] != 0.0f)) return false;
    return true;
}

// the chain behind the copy-in; recorded into the graph and, every 256th call, run eagerly (stage times)
static int fb_enqueue(orbhip_ctx *c, OrbFrameBuild *F, const orbhip_frame_params &fp, int s0, bool undist, bool grid, bool bow)
{
    // Every kernel stores its results twice: into the device block (read by the kernels behind it, and by
    // orbhip_set_put_from_frame later) and into its page-locked twin -- posted PCIe writes of a few dozen KB per kernel that
    // overlap the kernels; no copy command follows (a copy node behind the last kernel started ~8 us after it ended).
    uint8_t *B = F->d_blk, *H = F->h_blk;
    orbhip_keypoint *dK = (orbhip_keypoint *)(B + (undist ? F->oK : F->oU));
    orbhip_keypoint *dU = (orbhip_keypoint *)(B + F->oU);
    uint8_t *dD = B + F->oD;
    int32_t *dC = (int32_t *)(B + F->oC);
    uint8_t *hpyr = nullptr;
    int rc;
    if ((rc = orb_host_pyr_stage(c, 1, &hpyr))) return rc;
    c->describeMirror = (long long)(H - B);
    rc = orb_run_pipeline(c, c->d_lvl0, s0, c->lvl0FrameBytes, 1, dK, dD, dC, F->dcap, hpyr);
    c->describeMirror = 0;
    if (rc) return rc;
    if (bow) {
        // the transform on the second stream, beside undistortion and grid; its three outputs go to the twin only
        HIPCHK(c, hipEventRecord(F->evFork, c->stream));
        HIPCHK(c, hipStreamWaitEvent(c->stream2, F->evFork, 0));
        launch_vocab_transform(c->stream2, c->voc, dD, F->dcap, fp.levelsup, (int32_t *)(H + F->oW), (float *)(H + F->oT),
                               (int32_t *)(H + F->oN), dC);
        HIPCHK(c, hipGetLastError());
        HIPCHK(c, hipEventRecord(F->evJoin, c->stream2));
    }
    if (undist) {
        launch_undistort(c->stream, dK, dC, F->dcap, 1, fp.K, fp.dist, fp.ndist, fp.K, dU, (orbhip_keypoint *)(H + F->oU));
        HIPCHK(c, hipGetLastError());
    }
    if (grid) {
        if ((rc = launch_grid_build(c->stream, dU, dC, F->dcap, 1, fp.min_x, fp.min_y, fp.inv_w, fp.inv_h, (int32_t *)(B + F->oG),
                                    (int32_t *)(B + F->oE), (int32_t *)(H + F->oG), (int32_t *)(H + F->oE))))
            return fail(c, rc, "orbhip_frame_build: more features than the grid kernel sorts in LDS (12288)");
        HIPCHK(c, hipGetLastError());
    }
    if (bow) HIPCHK(c, hipStreamWaitEvent(c->stream, F->evJoin, 0));
    return ORBHIP_OK;
}

extern "C" int orbhip_frame_build(orbhip_ctx *c, const uint8_t *img, int w, int h, int stride, const orbhip_frame_params *fp,
                                  orbhip_keypoint *kps, orbhip_keypoint *kps_un, uint8_t *desc, int cap, int *n_out,
                                  int32_t *cell_off, int32_t *cell_idx, int32_t *word_id, float *weight, int32_t *node_id)
{
    if (!c || !img || !kps || !kps_un || !desc || !n_out || cap <= 0 || stride < w || !fp_ok(fp))
        return fail(c, ORBHIP_E_ARG, "orbhip_frame_build: bad argument");
    const bool undist = fp->ndist > 0 && fp->dist[0] != 0.0f;                  // ref: src/Frame.cc:750-754
    const bool grid = grid_params_ok(fp->inv_w, fp->inv_h);
    const bool bow = fp->levelsup >= 0;
    if (grid && (!cell_off || !cell_idx)) return fail(c, ORBHIP_E_ARG, "orbhip_frame_build: grid parameters without grid outputs");
    if (bow && (!word_id || !weight || !node_id)) return fail(c, ORBHIP_E_ARG, "orbhip_frame_build: levelsup >= 0 without transform outputs");
    if (bow && !c->voc.desc) return fail(c, ORBHIP_E_ARG, "orbhip_frame_build: no vocabulary loaded (orbhip_vocab_load)");
    HIPCHK(c, orb_enter(c));
    const int s0 = (int)align_up((size_t)w, 64);
    int rc;
    if ((rc = orb_configure(c, w, h, s0, 1))) return rc;
    OrbFrameBuild *F = fb_of(c);
    if (!F) {
        F = new OrbFrameBuild();
        c->frameBuild = F;
        HIPCHK(c, hipEventCreateWithFlags(&F->evFork, hipEventDisableTiming));
        HIPCHK(c, hipEventCreateWithFlags(&F->evJoin, hipEventDisableTiming));
    }
    const int dcap = (int)c->cap_out;
    if (F->dcap != dcap || !F->d_blk) {
        HIPCHK(c, hipStreamSynchronize(c->stream));
        fb_graph_release(F);
        if (F->d_blk) HIPCHK(c, hipFree(F->d_blk));
        if (F->h_blk) HIPCHK(c, hipHostFree(F->h_blk));
        F->d_blk = F->h_blk = nullptr;
        size_t o = 0;
        auto carve = [&](size_t bytes) { const size_t at = o; o = align_up(o + bytes, 256); return at; };
        F->oU = carve((size_t)dcap * sizeof(orbhip_keypoint));
        F->oD = carve((size_t)dcap * 32 + 32);
        F->oC = carve(16);
        F->oG = carve((ORBHIP_GRID_CELLS + 1) * 4);
        F->oE = carve((size_t)dcap * 4);
        F->setEnd = o;
        F->oK = carve((size_t)dcap * sizeof(orbhip_keypoint));
        F->oW = carve((size_t)dcap * 4);
        F->oT = carve((size_t)dcap * 4);
        F->oN = carve((size_t)dcap * 4);
        F->bytes = o;
        void *p = nullptr;
        HIPCHK(c, hipMalloc(&p, o));
        F->d_blk = (uint8_t *)p;
        HIPCHK(c, hipHostMalloc(&p, o, hipHostMallocDefault));
        F->h_blk = (uint8_t *)p;
        F->dcap = dcap;
        F->valid = false;
    }
    // page-locked input staging (shared with orbhip_extract's graph)
    if (c->lvl0FrameBytes > c->h_in_bytes) {
        orb_graph_release(c);
        fb_graph_release(F);
        if (c->h_in) HIPCHK(c, hipHostFree(c->h_in));
        c->h_in = nullptr;
        c->h_in_bytes = 0;
        void *p = nullptr;
        HIPCHK(c, hipHostMalloc(&p, c->lvl0FrameBytes, hipHostMallocDefault));
        c->h_in = (uint8_t *)p;
        c->h_in_bytes = c->lvl0FrameBytes;
    }
    uint8_t *hpyr = nullptr;
    if ((rc = orb_host_pyr_stage(c, 1, &hpyr))) return rc;
    c->h_in_valid = false;
    F->valid = false;
    if (stride == s0)
        memcpy(c->h_in, img, (size_t)s0 * (h - 1) + w);
    else
        for (int y = 0; y < h; y++) memcpy(c->h_in + (size_t)y * s0, img + (size_t)y * stride, (size_t)w);
    // a context that is still copying the last frame's block into a resident set (orbhip_set_put_from_frame) goes first
    {
        std::vector<hipEvent_t> pending;
        {
            std::lock_guard<std::mutex> g(F->busyMu);
            pending.swap(F->busyEvents);
        }
        hipError_t e = hipSuccess;
        for (hipEvent_t ev : pending)
            if (e == hipSuccess) e = hipStreamWaitEvent(c->stream, ev, 0);
        {
            // (an enqueued wait refers to the record it saw: the event may be recorded again by the next copier)
            std::lock_guard<std::mutex> g(F->busyMu);
            F->freeEvents.insert(F->freeEvents.end(), pending.begin(), pending.end());
        }
        HIPCHK(c, e);
    }
    const void *key[7] = {c->d_lvl0, F->d_blk, c->h_in, F->h_blk, hpyr, c->voc.desc, (const void *)(uintptr_t)c->voc.gen};
    const bool same = F->exec && F->w == w && F->h == h && F->gen == c->allocGen && memcmp(key, F->key, sizeof(key)) == 0 &&
                      memcmp(&F->fp, fp, sizeof(*fp)) == 0;
    static const bool noGraph = ORB_SWITCH("NO_GRAPH", 0) != 0;
    if (!same && !noGraph) {
        fb_graph_release(F);
        HIPCHK(c, hipStreamSynchronize(c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream2));
        c->capturing = true;
        HIPCHK(c, hipStreamBeginCapture(c->stream, hipStreamCaptureModeThreadLocal));
        hipError_t e = hipMemcpyAsync(c->d_lvl0, c->h_in, c->lvl0FrameBytes, hipMemcpyHostToDevice, c->stream);
        rc = e == hipSuccess ? fb_enqueue(c, F, *fp, s0, undist, grid, bow) : ORBHIP_E_HIP;
        hipGraph_t g = nullptr;
        const hipError_t e2 = hipStreamEndCapture(c->stream, &g);
        c->capturing = false;
        if (rc != ORBHIP_OK || e != hipSuccess || e2 != hipSuccess || !g) {
            if (g) (void)hipGraphDestroy(g);
            (void)hipGetLastError();
            // (a refusal of fb_enqueue itself -- ORBHIP_E_SIZE from the grid kernel's limit -- keeps its own code)
            return fail(c, rc != ORBHIP_OK ? rc : ORBHIP_E_HIP, std::string("orbhip_frame_build: graph capture failed: ") +
                                             (rc != ORBHIP_OK ? c->err : std::string(hipGetErrorString(e != hipSuccess ? e : e2))));
        }
        F->graph = g;
        HIPCHK(c, hipGraphInstantiate(&F->exec, g, nullptr, nullptr, 0));
        F->w = w;
        F->h = h;
        F->fp = *fp;
        memcpy(F->key, key, sizeof(key));
        F->gen = c->allocGen;
        F->calls = 0;
    }
    if (noGraph || (F->calls++ & 255u) == 0) {
        // eagerly: the first call of a geometry and every 256th one refresh the stage times behind GetTimeOfComputePyramid / ...
        HIPCHK(c, hipMemcpyAsync(c->d_lvl0, c->h_in, c->lvl0FrameBytes, hipMemcpyHostToDevice, c->stream));
        if ((rc = fb_enqueue(c, F, *fp, s0, undist, grid, bow))) return rc;
    } else {
        HIPCHK(c, hipGraphLaunch(F->exec, c->stream));
    }
    HIPCHK(c, hipStreamSynchronize(c->stream));
    c->h_in_valid = true;
    c->h_pyr_B = hpyr ? 1 : 0;
    // results: the packed block -> the caller's arrays
    const uint8_t *H = F->h_blk;
    const int n = *(const int32_t *)(H + F->oC);
    *n_out = n;
    if (n < 0 || n > cap || n > dcap) return fail(c, ORBHIP_E_CAPACITY, "orbhip_frame_build: output capacity too small");
    memcpy(kps_un, H + F->oU, (size_t)n * sizeof(orbhip_keypoint));
    memcpy(kps, H + (undist ? F->oK : F->oU), (size_t)n * sizeof(orbhip_keypoint));
    memcpy(desc, H + F->oD, (size_t)n * 32);
    if (grid) {
        memcpy(cell_off, H + F->oG, (ORBHIP_GRID_CELLS + 1) * 4);
        memcpy(cell_idx, H + F->oE, (size_t)n * 4);
    }
    if (bow) {
        memcpy(word_id, H + F->oW, (size_t)n * 4);
        memcpy(weight, H + F->oT, (size_t)n * 4);
        memcpy(node_id, H + F->oN, (size_t)n * 4);
    }
    F->valid = true;
    F->undist = undist;
    F->grid = grid;
    F->bow = bow;
    F->n = n;
    F->fingerprint = orbhip_set_fingerprint(kps_un, desc, n);
    return ORBHIP_OK;
}

extern "C" uint64_t orbhip_frame_fingerprint(const orbhip_ctx *c)
{
    if (!c || !c->frameBuild) return 0;
    const OrbFrameBuild *F = static_cast<const OrbFrameBuild *>(c->frameBuild);
    return F->valid && F->n > 0 ? F->fingerprint : 0;
}

// what orbhip_set_put_from_frame (api_sets.hip) needs of the block
int orb_frame_block(orbhip_ctx *src, const uint8_t **d_blk, size_t *setBytes, size_t off[5], int *n, int *dcap, bool *grid,
                    float gp[4], const orbhip_keypoint **h_kps_un, const uint8_t **h_desc)
{
    OrbFrameBuild *F = src ? fb_of(src) : nullptr;
    if (!F || !F->valid || F->n <= 0) return ORBHIP_E_ARG;
    *d_blk = F->d_blk + F->oU;
    *setBytes = F->setEnd - F->oU;
    off[0] = 0;
    off[1] = F->oD - F->oU;
    off[2] = F->oC - F->oU;
    off[3] = F->oG - F->oU;
    off[4] = F->oE - F->oU;
    *n = F->n;
    *dcap = F->dcap;
    *grid = F->grid;
    gp[0] = F->fp.min_x; gp[1] = F->fp.min_y; gp[2] = F->fp.inv_w; gp[3] = F->fp.inv_h;
    *h_kps_un = (const orbhip_keypoint *)(F->h_blk + F->oU);
    *h_desc = F->h_blk + F->oD;
    return ORBHIP_OK;
}

// the copying context's stream has the copy queued: the next orbhip_frame_build of `src` waits for it
int orb_frame_mark_busy(orbhip_ctx *src, hipStream_t copier)
{
    OrbFrameBuild *F = fb_of(src);
    if (!F) return ORBHIP_E_ARG;
    std::lock_guard<std::mutex> g(F->busyMu);
    hipEvent_t ev = nullptr;
    if (!F->freeEvents.empty()) {
        ev = F->freeEvents.back();
        F->freeEvents.pop_back();
    } else if (hipEventCreateWithFlags(&ev, hipEventDisableTiming) != hipSuccess)
        return ORBHIP_E_HIP;
    if (hipEventRecord(ev, copier) != hipSuccess) {
        F->freeEvents.push_back(ev);
        return ORBHIP_E_HIP;
    }
    F->busyEvents.push_back(ev);
    return ORBHIP_OK;
}

// ---- the floor under every per-call entry point ----
// What a host-pointer call costs before it computes anything, measured on the context's own stream: mode 0 = one empty kernel
// + one synchronisation; mode 1 = a 4 KB page-locked block copied in, the empty kernel, 4 KB copied out, one synchronisation
// (the shape of struct Packed); mode 2 = as 0 with the kernel storing one word to page-locked memory (the zero-copy result
// path).  tools/percall_latency.py prints them beside the per-call table: a row that sits at its floor cannot beat a host
// core by arithmetic.
std::atomic<unsigned> g_orbPathMask{0};
extern "C" unsigned orbhip_debug_path_mask(int reset)
{
    return reset ? g_orbPathMask.exchange(0u, std::memory_order_relaxed) : g_orbPathMask.load(std::memory_order_relaxed);
}

__global__ void k_floor(int32_t *out)
{
    if (out && threadIdx.x == 0) out[0] = 1;
}
extern "C" int orbhip_debug_roundtrip(orbhip_ctx *c, int mode, int iters, double *us_per_call)
{
    if (!c || !us_per_call || iters <= 0 || mode < 0 || mode > 2) return fail(c, ORBHIP_E_ARG, "orbhip_debug_roundtrip: bad argument");
    HIPCHK(c, orb_enter(c));
    Packed P(c);
    int rc;
    if ((rc = P.begin(16384))) return rc;
    uint8_t *din = (uint8_t *)P.in_fill(0, 4096);
    int32_t *hout = (int32_t *)P.out_host(64);
    HIPCHK(c, hipStreamSynchronize(c->stream));
    auto once = [&]() -> hipError_t {
        hipError_t e = hipSuccess;
        if (mode == 1) e = hipMemcpyAsync(din, P.h, 4096, hipMemcpyHostToDevice, c->stream);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL(k_floor, dim3(1), dim3(64), 0, c->stream, mode == 2 ? hout : (int32_t *)nullptr);
        if (mode == 1) e = hipMemcpyAsync(P.h + 8192, din, 4096, hipMemcpyDeviceToHost, c->stream);
        if (e != hipSuccess) return e;
        return hipStreamSynchronize(c->stream);
    };
    for (int i = 0; i < 20; i++) HIPCHK(c, once());
    timespec t0, t1;
    clock_gettime(CLOCK_MONOTONIC, &t0);
    for (int i = 0; i < iters; i++) HIPCHK(c, once());
    clock_gettime(CLOCK_MONOTONIC, &t1);
    *us_per_call = ((t1.tv_sec - t0.tv_sec) * 1e9 + (t1.tv_nsec - t0.tv_nsec)) / 1e3 / iters;
    return ORBHIP_OK;
}
