// api_common.h -- shared by the api_*.hip files that implement the C ABI of liborbhip.so (include/orbhip.h): error
// reporting, grow-only buffers and the two staging helpers of the host-pointer entry points.
#ifndef ORBHIP_API_COMMON_H
#define ORBHIP_API_COMMON_H
#include "orbhip_internal.h"

#include <dlfcn.h>
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <mutex>

// records the message (in the context, or process-wide when there is none) and returns `code` (api_extract.hip)
int orb_fail(orbhip_ctx *c, int code, const std::string &msg);
static inline int fail(orbhip_ctx *c, int code, const std::string &msg) { return orb_fail(c, code, msg); }

#define HIPCHK(c, expr)                                                                              \
    do {                                                                                             \
        hipError_t e_ = (expr);                                                                      \
        if (e_ != hipSuccess)                                                                        \
            return fail((c), ORBHIP_E_HIP, std::string(#expr) + ": " + hipGetErrorString(e_));       \
    } while (0)

// Entry of a C-ABI call that launches kernels: the context's device becomes current, and the runtime's sticky last error -- whatever
// an earlier call of this thread left behind without reading it: the caller's own HIP code, a best-effort clean-up -- is cleared,
// so that the hipGetLastError() behind the launches reports THESE launches (r06: a stale error once failed a graph capture).
static inline hipError_t orb_enter(const orbhip_ctx *c)
{
    const hipError_t e = hipSetDevice(c->device);
    (void)hipGetLastError();
    return e;
}

static inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

template <class T>
static int ensure(orbhip_ctx *c, T *&ptr, size_t &cap, size_t need)
{
    if (need <= cap && ptr) return ORBHIP_OK;
    if (ptr) HIPCHK(c, hipFree(ptr));
    ptr = nullptr;
    cap = 0;
    c->allocGen++;   // a captured single-frame graph holds the old pointer: its replay key includes the generation
    void *p = nullptr;
    HIPCHK(c, hipMalloc(&p, need ? need : 16));
    ptr = reinterpret_cast<T *>(p);
    cap = need;
    return ORBHIP_OK;
}

// api_extract.hip
int orb_configure(orbhip_ctx *c, int w, int h, int stride0, int B);
int orb_run_pipeline(orbhip_ctx *c, const uint8_t *lvl0, int stride0, size_t frame0, int B, orbhip_keypoint *d_kps,
                     uint8_t *d_desc, int32_t *d_counts, int cap, uint8_t *h_pyr_dst = nullptr);
void orb_graph_release(orbhip_ctx *c);
int orb_host_stage(orbhip_ctx *c, size_t bytes);                  // page-locked result block c->h_stage of at least `bytes`
int orb_host_pyr_stage(orbhip_ctx *c, int B, uint8_t **dst);      // page-locked pyramid copy (or nullptr when not asked for)
// api_frame.hip
void orb_frame_release(orbhip_ctx *c);
// api_pipe.hip / api_comm.hip: what orbhip_destroy releases
void orb_pipe_release(orbhip_ctx *c);
void orb_comm_release(orbhip_ctx *c);
// api_match.hip
int orb_match_scratch(orbhip_ctx *c, size_t bytes);
void orb_three_maxima(const std::vector<int> *histo, int L, int &ind1, int &ind2, int &ind3);
int orb_bow_rotation_check(const int32_t *pairs, int npairs, const int32_t *off1, const int32_t *idx1, const float *angle1,
                           const float *angle2, int check_ori, int32_t *match12, int32_t *match21);
// api_sets.hip
void orb_sets_release(orbhip_ctx *c);
static inline bool grid_params_ok(float inv_w, float inv_h) { return inv_w > 0.f && inv_h > 0.f; }

// Bump allocator over one temporary device block (host-pointer matching entry points).
// Device staging for the host-pointer entry points: one grow-only block per context.  Calls on a context are serialised
// on its stream and every such entry point synchronises before it returns, so the block is free again at the next call.
struct TmpDev {
    orbhip_ctx *c;
    uint8_t *base = nullptr;
    size_t used = 0, cap = 0;
    std::vector<void *> extra;   // blocks taken beyond the reserved size (a reserve() total that undercounts must not
                                 // become a null pointer handed to a copy: ADVICE r01)
    bool bad = false;            // an overflow block could not be allocated: take() returned null, the entry point must
                                 // return ORBHIP_E_HIP before it copies anything (TMPCHK) -- the drop-in never terminates
                                 // the SLAM process (include/orbhip/hiperror.h)
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
            (void)hipGetLastError();
            bad = true;
            return nullptr;
        }
        extra.push_back(p);
        return p;
    }
};

#define TMPCHK(c, T)                                                                                 \
    do {                                                                                             \
        if ((T).bad) return fail((c), ORBHIP_E_HIP, "device staging allocation failed (out of memory)"); \
    } while (0)

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
        TMPCHK(c, T);
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
    // A small input that the kernel reads a few times (a byte mask, a list of node pairs): left in the page-locked block and
    // read over PCIe, which costs a kernel about what one small copy command costs the host -- and there is no copy to wait for.
    void *in_host(const void *src, size_t bytes)
    {
        off = align_up(off, 256);
        if (bytes) memcpy(h + off, src, bytes);
        void *p = h + off;
        off += bytes;
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
        if (inEnd == 0) return ORBHIP_OK;   // nothing travels by copy
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

#endif
