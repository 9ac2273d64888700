// k_stereo.hip -- SURVEY.md section 8f row 2: Frame::ComputeStereoMatches on the device
// (ref: src/Frame.cc:810-984), reading the two extractors' pyramids where they already are (HBM) instead
// of downloading mvImagePyramid.
//   k_stereo_rows    per stereo pair: the right keypoints binned by the rows their band kpY +- 2*scale covers
//                    (:822-836), CSR over 4-row bins built with LDS counters;
//   k_stereo_best    per left keypoint: minimum descriptor distance over the right keypoints of its row whose
//                    octave is within +-1 and u in [uL - maxD, uL] (:848-893); the reference's strict '<' over
//                    ascending right index = minimum of (distance, index);
//   k_stereo_refine  one wave per left keypoint: 11 x (11x11) centre-subtracted L1 patch distances on the
//                    keypoint's pyramid level (:896-936; all terms are integers), parabola fit, disparity,
//                    depth (:938-966); float operations are individually rounded (no contraction);
//   k_stereo_cut     per stereo pair: median of the patch distances of the accepted matches by a two-level
//                    histogram select, matches with distance >= 1.5f*1.4f*median are removed (:970-983).
#include "orbhip_internal.h"

struct StereoGeom {
    int nlevels, nRows, stride0L, stride0R;
    unsigned long long frame0L, frame0R, pyrFrameL, pyrFrameR;
    float sf[ORBHIP_MAX_LEVELS], isf[ORBHIP_MAX_LEVELS];
    int lw[ORBHIP_MAX_LEVELS], lstride[ORBHIP_MAX_LEVELS];
    unsigned long long imgOff[ORBHIP_MAX_LEVELS];
};

// Row bins: bin q holds the right keypoints whose row band touches rows 4q..4q+3 (the reference's
// vRowIndices, :822-836, at 4-row granularity plus a 4-bit mask of the rows actually covered).
// Per stereo pair: off[0..nbins] (CSR), off[STEREO_BINS_MAX + 1] = 1 when the entry capacity was exceeded.
#define STEREO_BINS_MAX 1024   // rows <= 4095
#define STEREO_OFF_STRIDE (STEREO_BINS_MAX + 8)

__device__ __forceinline__ bool stereo_band(const StereoGeom &G, const orbhip_keypoint &k, int &minr, int &maxr)
{
    const float r = __fmul_rn(2.0f, G.sf[k.octave]);            // :830
    maxr = min((int)ceilf(__fadd_rn(k.y, r)), G.nRows - 1);     // rows outside the image can hold no left keypoint
    minr = max((int)floorf(__fsub_rn(k.y, r)), 0);
    return minr <= maxr;
}

__global__ __launch_bounds__(256) void k_stereo_rows(const StereoGeom G, const orbhip_keypoint *__restrict__ kpsR,
                                                     const int32_t *__restrict__ cntR, int cap, int entCap,
                                                     int32_t *__restrict__ off, uint2 *__restrict__ ent)
{
    __shared__ int s_cnt[STEREO_BINS_MAX + 1];
    __shared__ int s_part[256];
    const int b = blockIdx.x, tid = threadIdx.x;
    const int nR = min(cntR[b], cap);
    const orbhip_keypoint *kR = kpsR + (size_t)b * cap;
    int32_t *O = off + (size_t)b * STEREO_OFF_STRIDE;
    uint2 *E = ent + (size_t)b * entCap;
    for (int q = tid; q <= STEREO_BINS_MAX; q += 256) s_cnt[q] = 0;
    __syncthreads();
    for (int j = tid; j < nR; j += 256) {
        int minr, maxr;
        if (!stereo_band(G, kR[j], minr, maxr)) continue;
        for (int q = minr >> 2; q <= (maxr >> 2); q++) atomicAdd(&s_cnt[q], 1);
    }
    __syncthreads();
    // exclusive scan of the bin counts: 4 bins per thread + a scan of the 256 partial sums
    int c[4], sum = 0;
#pragma unroll
    for (int k = 0; k < 4; k++) {
        c[k] = s_cnt[tid * 4 + k];
        sum += c[k];
    }
    s_part[tid] = sum;
    __syncthreads();
    for (int d = 1; d < 256; d <<= 1) {
        const int v = tid >= d ? s_part[tid - d] : 0;
        __syncthreads();
        s_part[tid] += v;
        __syncthreads();
    }
    int run = s_part[tid] - sum;
    const int total = s_part[255];
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 4; k++) {
        s_cnt[tid * 4 + k] = run;       // becomes the fill cursor
        O[tid * 4 + k] = run;
        run += c[k];
    }
    if (tid == 0) {
        O[STEREO_BINS_MAX] = total;
        O[STEREO_BINS_MAX + 1] = total > entCap ? 1 : 0;
    }
    if (total > entCap) return;         // k_stereo_best scans all right keypoints of this pair instead
    __syncthreads();
    for (int j = tid; j < nR; j += 256) {
        const orbhip_keypoint k = kR[j];
        int minr, maxr;
        if (!stereo_band(G, k, minr, maxr)) continue;
        for (int q = minr >> 2; q <= (maxr >> 2); q++) {
            unsigned mask = 0;
#pragma unroll
            for (int r = 0; r < 4; r++) mask |= (unsigned)(4 * q + r >= minr && 4 * q + r <= maxr) << r;
            const int pos = atomicAdd(&s_cnt[q], 1);
            E[pos] = make_uint2(__float_as_uint(k.x), (unsigned)j | ((unsigned)k.octave << 16) | (mask << 24));
        }
    }
}

__device__ __forceinline__ int hamming256(const uint4 a0, const uint4 a1, const uint4 r0, const uint4 r1)
{
    return __popc(a0.x ^ r0.x) + __popc(a0.y ^ r0.y) + __popc(a0.z ^ r0.z) + __popc(a0.w ^ r0.w) + __popc(a1.x ^ r1.x) +
           __popc(a1.y ^ r1.y) + __popc(a1.z ^ r1.z) + __popc(a1.w ^ r1.w);
}

__global__ __launch_bounds__(256) void k_stereo_best(const StereoGeom G, const orbhip_keypoint *__restrict__ kpsL,
                                                     const uint8_t *__restrict__ descL, const int32_t *__restrict__ cntL,
                                                     const orbhip_keypoint *__restrict__ kpsR,
                                                     const uint8_t *__restrict__ descR, const int32_t *__restrict__ cntR,
                                                     int cap, int entCap, const int32_t *__restrict__ off,
                                                     const uint2 *__restrict__ ent, float maxD,
                                                     float *__restrict__ bestUR, int32_t *__restrict__ bestDist)
{
    const int b = blockIdx.y;
    const int iL = blockIdx.x * 256 + threadIdx.x;
    const int nL = min(cntL[b], cap), nR = min(cntR[b], cap);
    if (iL >= nL) return;
    const orbhip_keypoint kl = kpsL[(size_t)b * cap + iL];
    const int rowi = (int)kl.y;
    const float minU = __fsub_rn(kl.x, maxD), maxU = kl.x;   // minD = 0, :841-846
    const uint4 a0 = reinterpret_cast<const uint4 *>(descL + ((size_t)b * cap + iL) * 32)[0];
    const uint4 a1 = reinterpret_cast<const uint4 *>(descL + ((size_t)b * cap + iL) * 32)[1];
    // key = distance << 16 | right index: the minimum key is the first minimum of the reference's scan in
    // ascending right index (strict '<', :881), whatever order the bin is stored in.  TH_HIGH = 100.
    unsigned bestKey = 100u << 16;
    const bool usable = rowi >= 0 && rowi < G.nRows && !(maxU < 0);
    const orbhip_keypoint *kR = kpsR + (size_t)b * cap;
    const uint4 *dR = reinterpret_cast<const uint4 *>(descR + (size_t)b * cap * 32);
    const int octLo = kl.octave - 1, octHi = kl.octave + 1;
    const int32_t *O = off + (size_t)b * STEREO_OFF_STRIDE;
    if (usable && O[STEREO_BINS_MAX + 1] == 0) {
        const uint2 *E = ent + (size_t)b * entCap;
        const int q = rowi >> 2, bit = 24 + (rowi & 3);
        int e = O[q];
        const int end = O[q + 1];
        for (; e < end; e += 2) {
            const uint2 e0 = E[e], e1 = E[min(e + 1, end - 1)];
#pragma unroll
            for (int u = 0; u < 2; u++) {
                const uint2 en = u ? e1 : e0;
                const float xr = __uint_as_float(en.x);
                const int octR = (en.y >> 16) & 255;
                const bool cand = (u == 0 || e + 1 < end) && ((en.y >> bit) & 1) && octR >= octLo && octR <= octHi &&
                                  xr >= minU && xr <= maxU;
                if (cand) {
                    const int j = en.y & 0xFFFF;
                    const unsigned key = ((unsigned)hamming256(a0, a1, dR[2 * j], dR[2 * j + 1]) << 16) | (unsigned)j;
                    bestKey = min(bestKey, key);
                }
            }
        }
    } else if (usable) {
        for (int j = 0; j < nR; j++) {
            const orbhip_keypoint k = kR[j];
            int minr, maxr;
            if (!stereo_band(G, k, minr, maxr)) continue;
            if (rowi >= minr && rowi <= maxr && k.octave >= octLo && k.octave <= octHi && k.x >= minU && k.x <= maxU) {
                const unsigned key = ((unsigned)hamming256(a0, a1, dR[2 * j], dR[2 * j + 1]) << 16) | (unsigned)j;
                bestKey = min(bestKey, key);
            }
        }
    }
    const bool found = bestKey < (100u << 16);
    bestUR[(size_t)b * cap + iL] = found ? kR[bestKey & 0xFFFF].x : -1.0f;      // uR0 of :897 (kpR.pt.x >= 0 always)
    bestDist[(size_t)b * cap + iL] = found ? (int)(bestKey >> 16) : 100;
}

__device__ __forceinline__ int wave_sum_i(int v)
{
    v += __builtin_amdgcn_update_dpp(0, v, 0xB1, 0xF, 0xF, true);
    v += __builtin_amdgcn_update_dpp(0, v, 0x4E, 0xF, 0xF, true);
    v += __builtin_amdgcn_update_dpp(0, v, 0x141, 0xF, 0xF, true);
    v += __builtin_amdgcn_update_dpp(0, v, 0x140, 0xF, 0xF, true);
    return __builtin_amdgcn_readlane(v, 0) + __builtin_amdgcn_readlane(v, 16) + __builtin_amdgcn_readlane(v, 32) +
           __builtin_amdgcn_readlane(v, 48);
}

__device__ __forceinline__ const uint8_t *stereo_level(const uint8_t *lvl0, int stride0, unsigned long long frame0,
                                                       const uint8_t *pyr, unsigned long long pyrFrame,
                                                       const StereoGeom &G, int l, int frame, int &stride)
{
    if (l == 0) {
        stride = stride0;
        return lvl0 + (size_t)frame * frame0;
    }
    stride = G.lstride[l];
    return pyr + (size_t)frame * pyrFrame + G.imgOff[l];
}

__global__ __launch_bounds__(256) void k_stereo_refine(const StereoGeom G, const uint8_t *__restrict__ lvl0L,
                                                       const uint8_t *__restrict__ pyrL,
                                                       const uint8_t *__restrict__ lvl0R,
                                                       const uint8_t *__restrict__ pyrR,
                                                       const orbhip_keypoint *__restrict__ kpsL,
                                                       const int32_t *__restrict__ cntL,
                                                       const float *__restrict__ bestUR,
                                                       const int32_t *__restrict__ bestDist, int cap, float maxD,
                                                       float mbf, float *__restrict__ uRight, float *__restrict__ depth,
                                                       int32_t *__restrict__ sad)
{
    // per wave: left patch 11 rows x 12 bytes, right strip 11 rows x 24 bytes (columns cxR-10 .. cxR+13)
    __shared__ uint32_t s_Lw[4][36];
    __shared__ uint2 s_Rw[4][34];
    const int b = blockIdx.y;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int iL = blockIdx.x * 4 + wave;
    const int nL = min(cntL[b], cap);
    if (iL >= nL) return;
    const size_t o = (size_t)b * cap + iL;
    float outU = -1.0f, outZ = -1.0f;
    int outSad = -1;
    const int bd = bestDist[o];
    const float uR0 = bestUR[o];
    const orbhip_keypoint kl = kpsL[o];
    const uint8_t *s_L = reinterpret_cast<const uint8_t *>(s_Lw[wave]);
    const uint8_t *s_R = reinterpret_cast<const uint8_t *>(s_Rw[wave]);
    if (bd < 75) {   // thOrbDist = (TH_HIGH + TH_LOW) / 2, :815 (no candidate: bd = 100)
        const int lev = kl.octave;
        const float isf = G.isf[lev];
        const float scaleduL = roundf(__fmul_rn(kl.x, isf));
        const float scaledvL = roundf(__fmul_rn(kl.y, isf));
        const float scaleduR0 = roundf(__fmul_rn(uR0, isf));
        const int w = 5, Ls = 5;
        const float iniu = __fadd_rn(scaleduR0, (float)(Ls - w)), endu = __fadd_rn(scaleduR0, (float)(Ls + w + 1));
        if (!(iniu < 0 || endu >= (float)G.lw[lev])) {
            int strideL, strideR;
            const uint8_t *imL = stereo_level(lvl0L, G.stride0L, G.frame0L, pyrL, G.pyrFrameL, G, lev, b, strideL);
            const uint8_t *imR = stereo_level(lvl0R, G.stride0R, G.frame0R, pyrR, G.pyrFrameR, G, lev, b, strideR);
            const int cy = (int)scaledvL, cxL = (int)scaleduL, cxR = (int)scaleduR0;
            // patches -> LDS with one (unaligned) 4-byte and one 8-byte load per lane; keypoints lie >= 16 pixels
            // inside the level, so the <= 3 bytes read beyond a patch row stay inside the image buffer
            if (lane < 33) {
                const int row = lane / 3, part = lane - row * 3;
                uint32_t v;
                uint2 v2;
                __builtin_memcpy(&v, imL + (size_t)(cy - w + row) * strideL + cxL - w + 4 * part, 4);
                __builtin_memcpy(&v2, imR + (size_t)(cy - w + row) * strideR + cxR - 10 + 8 * part, 8);
                s_Lw[wave][lane] = v;
                s_Rw[wave][lane] = v2;
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            const int cLv = s_L[5 * 12 + 5];
            // lane owns patch elements p0 = lane and p1 = lane + 64 (< 121 for lanes 0..56); two shifts share one
            // 32-bit accumulator (a whole patch distance is at most 121 * 510 < 2^16)
            const int dy0 = lane / 11, dx0 = lane - dy0 * 11;
            const int p1 = lane + 64, dy1 = p1 / 11, dx1 = p1 - dy1 * 11;
            const bool has1 = p1 < 121;
            const int a0 = (int)s_L[dy0 * 12 + dx0] - cLv;
            const int a1 = has1 ? (int)s_L[dy1 * 12 + dx1] - cLv : 0;
            const uint8_t *r0 = &s_R[dy0 * 24 + dx0];
            const uint8_t *r1 = &s_R[has1 ? dy1 * 24 + dx1 : 0];
            int dists[11];
#pragma unroll
            for (int s = 0; s < 11; s += 2) {       // incR = s - 5
                const int cRa = s_R[5 * 24 + s + 5];
                int acc = abs(a0 - ((int)r0[s] - cRa)) + (has1 ? abs(a1 - ((int)r1[s] - cRa)) : 0);
                if (s + 1 < 11) {
                    const int cRb = s_R[5 * 24 + s + 6];
                    acc += (abs(a0 - ((int)r0[s + 1] - cRb)) + (has1 ? abs(a1 - ((int)r1[s + 1] - cRb)) : 0)) << 16;
                }
                const int tot = wave_sum_i(acc);
                dists[s] = tot & 0xFFFF;
                if (s + 1 < 11) dists[s + 1] = (unsigned)tot >> 16;
            }
            int sadBest = 2147483647, bestinc = 0;
#pragma unroll
            for (int s = 0; s < 11; s++)
                if ((float)dists[s] < (float)sadBest) {
                    sadBest = dists[s];
                    bestinc = s - Ls;
                }
            if (!(bestinc == -Ls || bestinc == Ls)) {
                float d1 = 0.f, d2 = 0.f, d3 = 0.f;
#pragma unroll
                for (int s = 1; s < 10; s++)
                    if (s - Ls == bestinc) {
                        d1 = (float)dists[s - 1];
                        d2 = (float)dists[s];
                        d3 = (float)dists[s + 1];
                    }
                const float num = __fsub_rn(d1, d3);
                const float den = __fmul_rn(2.0f, __fsub_rn(__fadd_rn(d1, d3), __fmul_rn(2.0f, d2)));
                const float deltaR = __fdiv_rn(num, den);
                if (!(deltaR < -1 || deltaR > 1)) {
                    float bestuR = __fmul_rn(G.sf[lev], __fadd_rn(__fadd_rn(scaleduR0, (float)bestinc), deltaR));
                    float disparity = __fsub_rn(kl.x, bestuR);
                    if (disparity >= 0.f && disparity < maxD) {
                        if (disparity <= 0) {
                            disparity = 0.01f;
                            bestuR = (float)__dsub_rn((double)kl.x, 0.01);
                        }
                        outZ = __fdiv_rn(mbf, disparity);
                        outU = bestuR;
                        outSad = sadBest;
                    }
                }
            }
        }
    }
    if (lane == 0) {
        uRight[o] = outU;
        depth[o] = outZ;
        sad[o] = outSad;
    }
}

// rank n/2 of the accepted patch distances (all < 2^16: 121 terms of at most 510) by a two-level
// 256-bin histogram select in LDS
__device__ __forceinline__ int select_bin(int *hist, int tid, int target, int *s_bin, int *s_below)
{
    // thread t: exclusive prefix of bin t
    int below = 0;
    for (int j = 0; j < tid; j++) below += hist[j];
    const int mine = hist[tid];
    if (target >= below && target < below + mine) {
        *s_bin = tid;
        *s_below = below;
    }
    __syncthreads();
    return *s_bin;
}

__global__ __launch_bounds__(256) void k_stereo_cut(const int32_t *__restrict__ cntL, int cap, int32_t *__restrict__ sad,
                                                    float *__restrict__ uRight, float *__restrict__ depth,
                                                    int32_t *__restrict__ nmatch)
{
    __shared__ int s_hist[256];
    __shared__ int s_n, s_bin, s_below;
    const int b = blockIdx.x, tid = threadIdx.x;
    const int nL = min(cntL[b], cap);
    const int32_t *S = sad + (size_t)b * cap;
    s_hist[tid] = 0;
    if (tid == 0) s_n = 0;
    __syncthreads();
    int local = 0;
    for (int i = tid; i < nL; i += 256) {
        const int d = S[i];
        if (d >= 0) {
            local++;
            atomicAdd(&s_hist[(d >> 8) & 255], 1);
        }
    }
    atomicAdd(&s_n, local);
    __syncthreads();
    const int n = s_n;
    if (tid == 0) nmatch[b] = n;
    if (n == 0) return;
    const int hi = select_bin(s_hist, tid, n / 2, &s_bin, &s_below);
    const int target2 = n / 2 - s_below;
    __syncthreads();
    s_hist[tid] = 0;
    __syncthreads();
    for (int i = tid; i < nL; i += 256) {
        const int d = S[i];
        if (d >= 0 && ((d >> 8) & 255) == hi) atomicAdd(&s_hist[d & 255], 1);
    }
    __syncthreads();
    const int lo = select_bin(s_hist, tid, target2, &s_bin, &s_below);
    const int median = (hi << 8) | lo;
    const float thDist = __fmul_rn(1.5f * 1.4f, (float)median);
    for (int i = tid; i < nL; i += 256) {
        const int d = S[i];
        if (d >= 0 && !((float)d < thDist)) {
            uRight[(size_t)b * cap + i] = -1.0f;
            depth[(size_t)b * cap + i] = -1.0f;
        }
    }
}

static int stereo_ent_per_kp()
{
    static int v = -1;
    if (v < 0) {
        v = ORB_TUNE("STEREO_ENT_PER_KP", 8);   // tests force the overflow path with 1
        if (v < 1) v = 1;
    }
    return v;
}

size_t stereo_scratch_bytes(int B, int cap)
{
    return (size_t)B * ((size_t)3 * cap * 4 + (size_t)STEREO_OFF_STRIDE * 4 + (size_t)cap * stereo_ent_per_kp() * 8);
}

int launch_stereo(orbhip_ctx *L, orbhip_ctx *R, const orbhip_keypoint *kpsL, const uint8_t *descL, const int32_t *cntL,
                  const orbhip_keypoint *kpsR, const uint8_t *descR, const int32_t *cntR, int cap, int B, float mb,
                  float mbf, float *uRight, float *depth, int32_t *scratch /* stereo_scratch_bytes */, int32_t *nmatch)
{
    StereoGeom G;
    G.nlevels = L->nlevels;
    G.nRows = L->G.lv[0].h;
    G.stride0L = L->last_stride0;
    G.stride0R = R->last_stride0;
    G.frame0L = L->last_frame0;
    G.frame0R = R->last_frame0;
    G.pyrFrameL = L->pyrFrameBytes;
    G.pyrFrameR = R->pyrFrameBytes;
    for (int l = 0; l < L->nlevels; l++) {
        G.sf[l] = L->mvScaleFactor[l];
        G.isf[l] = L->mvInvScaleFactor[l];
        G.lw[l] = L->G.lv[l].w;
        G.lstride[l] = L->G.lv[l].stride;
        G.imgOff[l] = L->G.lv[l].imgOff;
    }
    const float maxD = mbf / mb;   // :842
    const int entCap = cap * stereo_ent_per_kp();
    uint2 *ent = reinterpret_cast<uint2 *>(scratch);                                    // scratch base is 256-byte aligned
    int32_t *off = scratch + 2 * (size_t)B * entCap;
    float *bestUR = reinterpret_cast<float *>(off + (size_t)B * STEREO_OFF_STRIDE);
    int32_t *bestDist = off + (size_t)B * STEREO_OFF_STRIDE + (size_t)B * cap, *sad = bestDist + (size_t)B * cap;
    hipStream_t s = L->stream;
    hipLaunchKernelGGL(k_stereo_rows, dim3(B, 1, 1), dim3(256, 1, 1), 0, s, G, kpsR, cntR, cap, entCap, off, ent);
    hipLaunchKernelGGL(k_stereo_best, dim3((cap + 255) / 256, B, 1), dim3(256, 1, 1), 0, s, G, kpsL, descL, cntL, kpsR, descR,
                       cntR, cap, entCap, off, ent, maxD, bestUR, bestDist);
    hipLaunchKernelGGL(k_stereo_refine, dim3((cap + 3) / 4, B, 1), dim3(256, 1, 1), 0, s, G, L->last_lvl0, L->d_pyr, R->last_lvl0,
                       R->d_pyr, kpsL, cntL, bestUR, bestDist, cap, maxD, mbf, uRight, depth, sad);
    hipLaunchKernelGGL(k_stereo_cut, dim3(B, 1, 1), dim3(256, 1, 1), 0, s, cntL, cap, sad, uRight, depth, nmatch);
    return ORBHIP_OK;
}
