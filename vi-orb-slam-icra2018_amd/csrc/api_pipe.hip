// api_pipe.hip -- C ABI, part 2: the host-fed pipeline (orbhip_pipe_*).
#include "api_common.h"

// ------------------------------------------------------------------------------------------------
// host-fed pipeline: frames arrive from host memory (the reference's frames always do: cv::imread,
// Examples/Monocular/mono_euroc.cc:73), batch n + 1 is copied in and batch n - 1 copied out while batch n computes
// ------------------------------------------------------------------------------------------------
void orb_pipe_release(orbhip_ctx *c)
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
    HIPCHK(c, orb_enter(c));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    orb_pipe_release(c);
    // device rows are the image rows when they are 16-byte multiples (level 0 is then read in place), else padded to 64
    const int stride = (w % 16 == 0) ? w : (int)align_up((size_t)w, 64);
    int rc;
    if ((rc = orb_configure(c, w, h, stride, B))) return rc;
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
        orb_pipe_release(c);
        c->err = keep;
        return r;
    };
    hipError_t e;
    if ((e = hipStreamCreateWithFlags(&P->sIn, hipStreamNonBlocking)) != hipSuccess) return bail("hipStreamCreate", e);
    if ((e = hipStreamCreateWithFlags(&P->sOut, hipStreamNonBlocking)) != hipSuccess) return bail("hipStreamCreate", e);
    P->d_in.assign(depth, nullptr); P->d_out.assign(depth, nullptr); P->h_out.assign(depth + 1, nullptr);
    P->evIn.assign(depth, nullptr); P->evK.assign(depth, nullptr); P->evOut.assign(depth, nullptr);
    P->slotB.assign(depth, 0);
    for (int i = 0; i < depth; i++) {
        void *p = nullptr;
        if ((e = hipMalloc(&p, P->inBytes)) != hipSuccess) return bail("hipMalloc (input slot)", e);
        P->d_in[i] = (uint8_t *)p;
        if ((e = hipMalloc(&p, P->outBytes)) != hipSuccess) return bail("hipMalloc (output slot)", e);
        P->d_out[i] = (uint8_t *)p;
        for (hipEvent_t *ev : {&P->evIn[i], &P->evK[i], &P->evOut[i]})
            if ((e = hipEventCreateWithFlags(ev, hipEventDisableTiming)) != hipSuccess) return bail("hipEventCreate", e);
    }
    for (int i = 0; i <= depth; i++) {
        void *p = nullptr;
        if ((e = hipHostMalloc(&p, P->outBytes, hipHostMallocDefault)) != hipSuccess) return bail("hipHostMalloc (result block)", e);
        P->h_out[i] = (uint8_t *)p;
    }
    return ORBHIP_OK;
}

extern "C" int orbhip_pipe_destroy(orbhip_ctx *c)
{
    if (!c) return ORBHIP_E_ARG;
    HIPCHK(c, orb_enter(c));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    orb_pipe_release(c);
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
    HIPCHK(c, orb_enter(c));
    c->h_pyr_B = 0;          // no host copy of the pyramid in this mode (orbhip_host_pyramid_level reports that)
    c->h_in_valid = false;
    const int s = (int)(P->submitted % P->depth);
    const int hb = (int)(P->submitted % (P->depth + 1));   // never the block the last wait handed out (see OrbPipe::h_out)
    int rc;
    if ((rc = orb_configure(c, P->w, P->h, P->stride, B))) return rc;
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
    if ((rc = orb_run_pipeline(c, P->d_in[s], P->stride, P->frameBytes, B, (orbhip_keypoint *)(blk + P->koff), blk + P->doff,
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
    HIPCHK(c, hipMemcpyAsync(P->h_out[hb], blk, outBytes, hipMemcpyDeviceToHost, P->sOut));
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
    HIPCHK(c, orb_enter(c));
    const int s = (int)(P->waited % P->depth), hb = (int)(P->waited % (P->depth + 1));
    HIPCHK(c, hipEventSynchronize(P->evOut[s]));
    if (kps) *kps = (const orbhip_keypoint *)(P->h_out[hb] + P->koff);
    if (desc) *desc = P->h_out[hb] + P->doff;
    if (n_out) *n_out = (const int32_t *)(P->h_out[hb] + P->coff);
    if (B) *B = P->slotB[s];
    if (cap) *cap = P->dcap;
    P->lastWaited = hb;
    P->waited++;
    return ORBHIP_OK;
}

extern "C" int orbhip_pipe_enable_bow(orbhip_ctx *c, int levelsup, float nnratio, int check_ori)
{
    if (!c || !c->pipe) return fail(c, ORBHIP_E_ARG, "orbhip_pipe_enable_bow: no pipeline (orbhip_pipe_create)");
    if (!c->voc.desc) return fail(c, ORBHIP_E_ARG, "orbhip_pipe_enable_bow: no vocabulary loaded");
    OrbPipe *P = c->pipe;
    if (P->dcap > 4096) return fail(c, ORBHIP_E_SIZE, "orbhip_pipe_enable_bow: more than 4096 feature slots per frame");
    HIPCHK(c, orb_enter(c));
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

