// k_bowseq.hip -- batched ORBmatcher::SearchByBoW for a frame sequence that lives on the device
// (ref: src/ORBmatcher.cc:159-288 (KeyFrame, Frame) and :522-655 (KeyFrame, KeyFrame); called per
// frame by Tracking::TrackReferenceKeyFrame src/Tracking.cc:1881-1885).  One 256-thread workgroup
// per frame pair (side 1 = frame b - lag acting as the key frame, side 2 = frame b):
//   1. both FeatureVectors are built in LDS: a bitonic sort of 64-bit keys (node id << 32 | feature
//      index) groups the features by vocabulary node with ascending feature index inside a node --
//      the canonical FeatureVector order; stopped features (weight <= 0,
//      TemplatedVocabulary.h:1334) sort to the end and are ignored;
//   2. every node present on both sides is a work item (the reference's merge walk, :180-264);
//   3. a wave takes a work item: side-1 features are visited serially (the greedy claiming of the
//      reference is order dependent), the 64 lanes scan the node's unclaimed side-2 features and
//      reduce (best, position, second) with the lowest position winning ties, then the acceptance test
//      best <= / < TH and best < ratio * second (:228-230, :598-600);
//   4. rotation histogram of 30 bins, ComputeThreeMaxima (:1629-1670) and removal of the matches
//      outside the three dominant bins (:267-285), all in the workgroup.
// Integer/bitwise path (XOR + popcount); no MFMA.
#include "orbhip_internal.h"

#include <cstdlib>

#define BS_HISTO 30

static int bs_threads()
{
    static const int n = getenv("ORBHIP_BOW_THREADS") ? atoi(getenv("ORBHIP_BOW_THREADS")) : 512;
    return n;
}

struct BsBest {
    int b1, pos, b2;
};

__device__ __forceinline__ void bs_merge(BsBest &A, int ob1, int opos, int ob2)
{
    const bool mine = (A.b1 < ob1) || (A.b1 == ob1 && A.pos < opos);
    const int nb2 = mine ? min(A.b2, ob1) : min(ob2, A.b1);
    A.b1 = mine ? A.b1 : ob1;
    A.pos = mine ? A.pos : opos;
    A.b2 = nb2;
}

// first position in sorted keys[0..n) whose node (high 32 bits) is >= / > node
__device__ __forceinline__ int bs_bound(const unsigned long long *keys, int n, unsigned node, bool upper)
{
    int lo = 0, hi = n;
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        const unsigned v = (unsigned)(keys[mid] >> 32);
        if (upper ? (v <= node) : (v < node))
            lo = mid + 1;
        else
            hi = mid;
    }
    return lo;
}

__device__ void bs_sort(unsigned long long *keys, int NP, int tid)
{
    for (int k = 2; k <= NP; k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int t = tid; t < NP / 2; t += blockDim.x) {
                // t-th compare-exchange pair of this (k, j) phase
                const int i = ((t / j) * 2 * j) + (t % j);
                const int p = i + j;
                const bool up = ((i & k) == 0);
                const unsigned long long a = keys[i], b = keys[p];
                if ((a > b) == up) {
                    keys[i] = b;
                    keys[p] = a;
                }
            }
            __syncthreads();
        }
    }
}

__global__ __launch_bounds__(1024) void k_bow_seq(const uint8_t *__restrict__ desc,
                                                 const orbhip_keypoint *__restrict__ kps,
                                                 const int32_t *__restrict__ counts,
                                                 const int32_t *__restrict__ node, const float *__restrict__ weight,
                                                 const uint8_t *__restrict__ valid, int cap, int NP, int lag, int th,
                                                 int th_mode, float nnratio, int check_ori,
                                                 int32_t *__restrict__ match12, int32_t *__restrict__ match21,
                                                 int32_t *__restrict__ nmatches)
{
    extern __shared__ __align__(16) uint8_t smem[];
    unsigned long long *key1 = reinterpret_cast<unsigned long long *>(smem);
    unsigned long long *key2 = key1 + NP;
    int *m12 = reinterpret_cast<int *>(key2 + NP);                  // [NP] side-1 feature -> side-2 feature or -1
    uint2 *items = reinterpret_cast<uint2 *>(m12 + NP);             // [NP] (s1 | e1 << 16, s2 | e2 << 16)
    unsigned *claim = reinterpret_cast<unsigned *>(items + NP);     // [NP / 32] side-2 feature claimed
    __shared__ int s_n1v, s_n2v, s_nitems, s_next, s_nm;
    __shared__ int s_hist[BS_HISTO];
    __shared__ int s_keep[3];

    const int b = blockIdx.x;
    const int tid = threadIdx.x, lane = tid & 63;
    const int n2 = min(counts[b], cap);
    int32_t *o12 = match12 + (size_t)b * cap;
    int32_t *o21 = match21 + (size_t)b * cap;
    if (b < lag) {
        for (int i = tid; i < cap; i += blockDim.x) {
            o12[i] = -1;
            o21[i] = -1;
        }
        if (tid == 0) nmatches[b] = 0;
        return;
    }
    const int b1 = b - lag;
    const int n1 = min(counts[b1], cap);
    const uint8_t *d1 = desc + (size_t)b1 * cap * 32, *d2 = desc + (size_t)b * cap * 32;
    const int32_t *nd1 = node + (size_t)b1 * cap, *nd2 = node + (size_t)b * cap;
    const float *w1 = weight + (size_t)b1 * cap, *w2 = weight + (size_t)b * cap;

    // ---- 1. keys: (node << 32 | index); absent / stopped features sort last ----
    for (int i = tid; i < NP; i += blockDim.x) {
        key1[i] = (i < n1 && w1[i] > 0.f) ? (((unsigned long long)(unsigned)nd1[i] << 32) | (unsigned)i) : ~0ull;
        key2[i] = (i < n2 && w2[i] > 0.f) ? (((unsigned long long)(unsigned)nd2[i] << 32) | (unsigned)i) : ~0ull;
        m12[i] = -1;
    }
    for (int i = tid; i < NP / 32; i += blockDim.x) claim[i] = 0;
    if (tid < BS_HISTO) s_hist[tid] = 0;
    if (tid == 0) {
        s_n1v = 0;
        s_n2v = 0;
        s_nitems = 0;
        s_next = 0;
        s_nm = 0;
    }
    __syncthreads();
    bs_sort(key1, NP, tid);
    bs_sort(key2, NP, tid);
    // number of live entries per side
    for (int i = tid; i < NP; i += blockDim.x) {
        if (key1[i] != ~0ull && (i + 1 == NP || key1[i + 1] == ~0ull)) s_n1v = i + 1;
        if (key2[i] != ~0ull && (i + 1 == NP || key2[i + 1] == ~0ull)) s_n2v = i + 1;
    }
    __syncthreads();
    const int n1v = s_n1v, n2v = s_n2v;

    // ---- 2. work items: nodes present on both sides ----
    for (int p = tid; p < n1v; p += blockDim.x) {
        const unsigned nodeId = (unsigned)(key1[p] >> 32);
        if (p == 0 || (unsigned)(key1[p - 1] >> 32) != nodeId) {
            const int s2 = bs_bound(key2, n2v, nodeId, false), e2 = bs_bound(key2, n2v, nodeId, true);
            if (e2 > s2) {
                const int e1 = bs_bound(key1, n1v, nodeId, true);
                const int slot = atomicAdd(&s_nitems, 1);
                items[slot] = make_uint2((unsigned)p | ((unsigned)e1 << 16), (unsigned)s2 | ((unsigned)e2 << 16));
            }
        }
    }
    __syncthreads();

    // ---- 3. greedy matching, one wave per work item ----
    const int nitems = s_nitems;
    for (;;) {
        int it = 0;
        if (lane == 0) it = atomicAdd(&s_next, 1);
        it = __shfl(it, 0);
        if (it >= nitems) break;
        const uint2 item = items[it];
        const int s1 = item.x & 0xFFFF, e1 = item.x >> 16, s2 = item.y & 0xFFFF, e2 = item.y >> 16;
        for (int a = s1; a < e1; a++) {
            const int i1 = __builtin_amdgcn_readfirstlane((int)(unsigned)key1[a]);
            if (valid && !valid[(size_t)b1 * cap + i1]) continue;   // no (good) MapPoint: :193-199
            const uint32_t *qrow = reinterpret_cast<const uint32_t *>(d1 + (size_t)i1 * 32);
            uint32_t Q[8];
#pragma unroll
            for (int k = 0; k < 8; k++) Q[k] = qrow[k];
            BsBest B = {256, 0x7FFFFFFF, 256};
            for (int p = s2 + lane; p < e2; p += 64) {
                const int i2 = (int)(unsigned)key2[p];
                if ((claim[i2 >> 5] >> (i2 & 31)) & 1u) continue;   // already matched: :209-210
                if (th_mode && valid && !valid[(size_t)b * cap + i2]) continue;   // KF-KF variant: :572-578
                const uint4 r0 = reinterpret_cast<const uint4 *>(d2 + (size_t)i2 * 32)[0];
                const uint4 r1 = reinterpret_cast<const uint4 *>(d2 + (size_t)i2 * 32)[1];
                const int d = __popc(Q[0] ^ r0.x) + __popc(Q[1] ^ r0.y) + __popc(Q[2] ^ r0.z) + __popc(Q[3] ^ r0.w) +
                              __popc(Q[4] ^ r1.x) + __popc(Q[5] ^ r1.y) + __popc(Q[6] ^ r1.z) + __popc(Q[7] ^ r1.w);
                if (d < B.b1) {
                    B.b2 = B.b1;
                    B.b1 = d;
                    B.pos = p;
                } else if (d < B.b2) {
                    B.b2 = d;
                }
            }
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) {
                const int ob1 = __shfl_xor(B.b1, o), opos = __shfl_xor(B.pos, o), ob2 = __shfl_xor(B.b2, o);
                bs_merge(B, ob1, opos, ob2);
            }
            const bool pass = th_mode ? (B.b1 < th) : (B.b1 <= th);
            if (pass && (float)B.b1 < nnratio * (float)B.b2) {
                const int i2 = (int)(unsigned)key2[B.pos];
                if (lane == 0) {
                    m12[i1] = i2;
                    atomicOr(&claim[i2 >> 5], 1u << (i2 & 31));
                }
                // the claim must be visible to this wave's next side-1 feature
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            }
        }
    }
    __syncthreads();

    // ---- 4. rotation consistency ----
    const orbhip_keypoint *k1 = kps + (size_t)b1 * cap, *k2 = kps + (size_t)b * cap;
    int mybin0 = -1;   // this thread handles features tid, tid+256, ...: remember bins in registers
    if (check_ori) {
        for (int i1 = tid; i1 < n1; i1 += blockDim.x) {
            const int i2 = m12[i1];
            if (i2 >= 0) {
                float rot = k1[i1].angle - k2[i2].angle;
                if (rot < 0.0f) rot += 360.0f;
                int bin = (int)roundf(rot * (1.0f / BS_HISTO));
                if (bin == BS_HISTO) bin = 0;
                if (bin >= 0 && bin < BS_HISTO) atomicAdd(&s_hist[bin], 1);
            }
        }
        __syncthreads();
        if (tid == 0) {
            int max1 = 0, max2 = 0, max3 = 0, i1 = -1, i2 = -1, i3 = -1;
            for (int i = 0; i < BS_HISTO; i++) {
                const int s = s_hist[i];
                if (s > max1) {
                    max3 = max2; max2 = max1; max1 = s;
                    i3 = i2; i2 = i1; i1 = i;
                } else if (s > max2) {
                    max3 = max2; max2 = s;
                    i3 = i2; i2 = i;
                } else if (s > max3) {
                    max3 = s;
                    i3 = i;
                }
            }
            if ((float)max2 < 0.1f * (float)max1) {
                i2 = -1;
                i3 = -1;
            } else if ((float)max3 < 0.1f * (float)max1) {
                i3 = -1;
            }
            s_keep[0] = i1;
            s_keep[1] = i2;
            s_keep[2] = i3;
        }
        __syncthreads();
    }
    (void)mybin0;
    // ---- 5. outputs ----
    for (int i = tid; i < cap; i += blockDim.x) o21[i] = -1;
    __syncthreads();
    int local = 0;
    for (int i1 = tid; i1 < cap; i1 += blockDim.x) {
        int i2 = i1 < n1 ? m12[i1] : -1;
        if (i2 >= 0 && check_ori) {
            float rot = k1[i1].angle - k2[i2].angle;
            if (rot < 0.0f) rot += 360.0f;
            int bin = (int)roundf(rot * (1.0f / BS_HISTO));
            if (bin == BS_HISTO) bin = 0;
            if (bin >= 0 && bin < BS_HISTO && bin != s_keep[0] && bin != s_keep[1] && bin != s_keep[2]) i2 = -1;
        }
        o12[i1] = i2;
        if (i2 >= 0) {
            o21[i2] = i1;
            local++;
        }
    }
    atomicAdd(&s_nm, local);
    __syncthreads();
    if (tid == 0) nmatches[b] = s_nm;
}

void launch_bow_seq(hipStream_t s, const uint8_t *desc, const orbhip_keypoint *kps, const int32_t *counts,
                    const int32_t *node, const float *weight, const uint8_t *valid, int cap, int B, int lag, int th,
                    int th_mode, float nnratio, int check_ori, int32_t *match12, int32_t *match21,
                    int32_t *nmatches)
{
    if (B <= 0) return;
    int NP = 512;
    while (NP < cap) NP <<= 1;
    const size_t lds = (size_t)NP * (8 + 8 + 4 + 8) + (size_t)NP / 8 + 64;
    if (lds > 48 * 1024)
        (void)hipFuncSetAttribute((const void *)k_bow_seq, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(k_bow_seq, dim3(B, 1, 1), dim3(bs_threads(), 1, 1), lds, s, desc, kps, counts, node, weight, valid, cap, NP,
                       lag, th, th_mode, nnratio, check_ori, match12, match21, nmatches);
}
