// k_stereo.hip -- SURVEY.md section 8f row 2: Frame::ComputeStereoMatches on the device
// (ref: src/Frame.cc:810-984), reading the two extractors' pyramids where they already are (HBM) instead
// of downloading mvImagePyramid.
//   k_stereo_best    per left keypoint: minimum descriptor distance over the right keypoints whose row band
//                    (kpY +- 2*scale, :826-836) contains the left row, octave within +-1 and
//                    u in [uL - maxD, uL] (:848-893); candidates are visited in right-keypoint index order,
//                    which is the order of the reference's per-row lists, strict '<' keeps the first;
//   k_stereo_refine  one wave per left keypoint: 11 x (11x11) centre-subtracted L1 patch distances on the
//                    keypoint's pyramid level (:896-936; all terms are integers), parabola fit, disparity,
//                    depth (:938-966); float operations are individually rounded (no contraction);
//   k_stereo_cut     per stereo pair: median of the patch distances of the accepted matches by rank
//                    counting, matches with distance >= 1.5f*1.4f*median are removed (:970-983).
#include "orbhip_internal.h"

struct StereoGeom {
    int nlevels, nRows, stride0L, stride0R;
    unsigned long long frame0L, frame0R, pyrFrameL, pyrFrameR;
    float sf[ORBHIP_MAX_LEVELS], isf[ORBHIP_MAX_LEVELS];
    int lw[ORBHIP_MAX_LEVELS], lstride[ORBHIP_MAX_LEVELS];
    unsigned long long imgOff[ORBHIP_MAX_LEVELS];
};

__global__ __launch_bounds__(256) void k_stereo_best(const StereoGeom G, const orbhip_keypoint *__restrict__ kpsL,
                                                     const uint8_t *__restrict__ descL, const int32_t *__restrict__ cntL,
                                                     const orbhip_keypoint *__restrict__ kpsR,
                                                     const uint8_t *__restrict__ descR, const int32_t *__restrict__ cntR,
                                                     int cap, float maxD, int32_t *__restrict__ bestIdx,
                                                     int32_t *__restrict__ bestDist)
{
    const int b = blockIdx.y;
    const int iL = blockIdx.x * 256 + threadIdx.x;
    const int nL = min(cntL[b], cap), nR = min(cntR[b], cap);
    if (blockIdx.x * 256 >= nL) return;
    const bool live = iL < nL;
    const orbhip_keypoint kl = kpsL[(size_t)b * cap + (live ? iL : 0)];
    const int rowi = (int)kl.y;
    const float minU = __fsub_rn(kl.x, maxD), maxU = kl.x;   // minD = 0
    const uint4 a0 = reinterpret_cast<const uint4 *>(descL + ((size_t)b * cap + (live ? iL : 0)) * 32)[0];
    const uint4 a1 = reinterpret_cast<const uint4 *>(descL + ((size_t)b * cap + (live ? iL : 0)) * 32)[1];
    int best = 100, bidx = -1;   // TH_HIGH; -1 = no candidate passed
    const bool usable = live && rowi >= 0 && rowi < G.nRows && !(maxU < 0);
    const orbhip_keypoint *kR = kpsR + (size_t)b * cap;
    const uint8_t *dR = descR + (size_t)b * cap * 32;
    for (int iR = 0; iR < nR; iR++) {
        // wave-uniform loads (scalar path)
        const float xr = kR[iR].x, yr = kR[iR].y;
        const int octR = kR[iR].octave;
        const float r = __fmul_rn(2.0f, G.sf[octR]);
        const int maxr = (int)ceilf(__fadd_rn(yr, r)), minr = (int)floorf(__fsub_rn(yr, r));
        const uint32_t *row = reinterpret_cast<const uint32_t *>(dR + (size_t)iR * 32);
        const bool cand = usable && rowi >= minr && rowi <= maxr && octR >= kl.octave - 1 && octR <= kl.octave + 1 &&
                          xr >= minU && xr <= maxU;
        if (cand) {
            const int d = __popc(a0.x ^ row[0]) + __popc(a0.y ^ row[1]) + __popc(a0.z ^ row[2]) + __popc(a0.w ^ row[3]) +
                          __popc(a1.x ^ row[4]) + __popc(a1.y ^ row[5]) + __popc(a1.z ^ row[6]) + __popc(a1.w ^ row[7]);
            if (d < best) {
                best = d;
                bidx = iR;
            }
        }
    }
    if (live) {
        bestIdx[(size_t)b * cap + iL] = bidx;
        bestDist[(size_t)b * cap + iL] = best;
    }
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
                                                       const orbhip_keypoint *__restrict__ kpsR,
                                                       const int32_t *__restrict__ bestIdx,
                                                       const int32_t *__restrict__ bestDist, int cap, float maxD,
                                                       float mbf, float *__restrict__ uRight, float *__restrict__ depth,
                                                       int32_t *__restrict__ sad)
{
    __shared__ uint8_t s_L[4][128];
    __shared__ uint8_t s_R[4][256];
    const int b = blockIdx.y;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int iL = blockIdx.x * 4 + wave;
    const int nL = min(cntL[b], cap);
    if (iL >= nL) return;
    const size_t o = (size_t)b * cap + iL;
    float outU = -1.0f, outZ = -1.0f;
    int outSad = -1;
    const int bd = bestDist[o], bi = bestIdx[o];
    if (bi >= 0 && bd < 75) {   // thOrbDist = (TH_HIGH + TH_LOW) / 2, :815
        const orbhip_keypoint kl = kpsL[o];
        const float uR0 = kpsR[(size_t)b * cap + bi].x;
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
            // patches -> LDS: left 11x11, right 11x21 (columns cxR-10 .. cxR+10)
            for (int p = lane; p < 121; p += 64) {
                const int dy = p / 11, dx = p - dy * 11;
                s_L[wave][p] = imL[(size_t)(cy - w + dy) * strideL + cxL - w + dx];
            }
            for (int p = lane; p < 231; p += 64) {
                const int dy = p / 21, dx = p - dy * 21;
                s_R[wave][p] = imR[(size_t)(cy - w + dy) * strideR + cxR - 10 + dx];
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            const int cLv = s_L[wave][5 * 11 + 5];
            int dists[11];
#pragma unroll
            for (int s = 0; s < 11; s++) {       // incR = s - 5
                const int cRv = s_R[wave][5 * 21 + s + 5];
                int acc = 0;
                for (int p = lane; p < 121; p += 64) {
                    const int dy = p / 11, dx = p - dy * 11;
                    const int a = (int)s_L[wave][p] - cLv;
                    const int bb = (int)s_R[wave][dy * 21 + dx + s] - cRv;
                    acc += abs(a - bb);
                }
                dists[s] = wave_sum_i(acc);
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

__global__ __launch_bounds__(256) void k_stereo_cut(const int32_t *__restrict__ cntL, int cap, int32_t *__restrict__ sad,
                                                    float *__restrict__ uRight, float *__restrict__ depth,
                                                    int32_t *__restrict__ nmatch)
{
    __shared__ int s_n, s_median;
    const int b = blockIdx.x, tid = threadIdx.x;
    const int nL = min(cntL[b], cap);
    int32_t *S = sad + (size_t)b * cap;
    if (tid == 0) {
        s_n = 0;
        s_median = -1;
    }
    __syncthreads();
    int local = 0;
    for (int i = tid; i < nL; i += 256) local += S[i] >= 0;
    atomicAdd(&s_n, local);
    __syncthreads();
    const int n = s_n;
    if (tid == 0) nmatch[b] = n;
    if (n == 0) return;
    // the element of rank n/2 in the order (distance, index): rank by counting
    const int target = n / 2;
    for (int i = tid; i < nL; i += 256) {
        const int d = S[i];
        if (d < 0) continue;
        int rank = 0;
        for (int j = 0; j < nL; j++) {
            const int e = S[j];
            rank += (e >= 0) && (e < d || (e == d && j < i));
        }
        if (rank == target) s_median = d;
    }
    __syncthreads();
    const float thDist = __fmul_rn(1.5f * 1.4f, (float)s_median);
    for (int i = tid; i < nL; i += 256) {
        const int d = S[i];
        if (d >= 0 && !((float)d < thDist)) {
            uRight[(size_t)b * cap + i] = -1.0f;
            depth[(size_t)b * cap + i] = -1.0f;
        }
    }
}

int launch_stereo(orbhip_ctx *L, orbhip_ctx *R, const orbhip_keypoint *kpsL, const uint8_t *descL, const int32_t *cntL,
                  const orbhip_keypoint *kpsR, const uint8_t *descR, const int32_t *cntR, int cap, int B, float mb,
                  float mbf, float *uRight, float *depth, int32_t *scratch /* 3 * B * cap */, int32_t *nmatch)
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
    int32_t *bestIdx = scratch, *bestDist = scratch + (size_t)B * cap, *sad = scratch + 2 * (size_t)B * cap;
    hipStream_t s = L->stream;
    hipLaunchKernelGGL(k_stereo_best, dim3((cap + 255) / 256, B, 1), dim3(256, 1, 1), 0, s, G, kpsL, descL, cntL, kpsR, descR,
                       cntR, cap, maxD, bestIdx, bestDist);
    hipLaunchKernelGGL(k_stereo_refine, dim3((cap + 3) / 4, B, 1), dim3(256, 1, 1), 0, s, G, L->last_lvl0, L->d_pyr, R->last_lvl0,
                       R->d_pyr, kpsL, cntL, kpsR, bestIdx, bestDist, cap, maxD, mbf, uRight, depth, sad);
    hipLaunchKernelGGL(k_stereo_cut, dim3(B, 1, 1), dim3(256, 1, 1), 0, s, cntL, cap, sad, uRight, depth, nmatch);
    return ORBHIP_OK;
}
