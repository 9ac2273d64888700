// k_bowseq.hip -- batched ORBmatcher::SearchByBoW for a frame sequence that lives on the device
// (ref: src/ORBmatcher.cc:159-288 (KeyFrame, Frame) and :522-655 (KeyFrame, KeyFrame); called per
// frame by Tracking::TrackReferenceKeyFrame src/Tracking.cc:1881-1885).  One 1024-thread workgroup
// per frame pair (side 1 = frame b - lag acting as the key frame, side 2 = frame b), one workgroup per CU:
//   0. both descriptor sets (2 x 33 KB at 1000 features) and the validity bitmaps are staged in LDS: the
//      matching below is a chain of dependent reads, and an LDS round trip is ~20x shorter than one to L2;
//   1. both FeatureVectors are built in LDS: one bitonic sort pass over 64-bit keys (node id << 32 | feature
//      index) of both sides groups the features by vocabulary node with ascending feature index inside a
//      node -- the canonical FeatureVector order; stopped features (weight <= 0,
//      TemplatedVocabulary.h:1334) sort to the end and are ignored;
//   2. every node present on both sides is a work item (the reference's merge walk, :180-264);
//   3. items are taken in cost order, largest first: side-1 features are visited serially (the greedy claiming
//      of the reference is order dependent); the lanes of a wave -- or, for small nodes, of one of its four
//      16-lane DPP rows -- scan the node's unclaimed side-2 features and reduce (best, position, second) with
//      the lowest position winning ties, then the acceptance test best <= / < TH and best < ratio * second
//      (:228-230, :598-600);
//   4. rotation histogram of 30 bins, ComputeThreeMaxima (:1629-1670) and removal of the matches
//      outside the three dominant bins (:267-285), all in the workgroup.
// Integer/bitwise path (XOR + popcount); no MFMA.
#include "orbhip_internal.h"

#include <type_traits>

#include <cstdlib>

#define BS_HISTO 30

static int bs_threads()
{
    static const int n = ORB_TUNE("BOW_THREADS", 1024);
    return n;
}

// 16 bytes per lane from global memory straight into LDS at (ldsAddr + 16 * lane); M0 carries the LDS address and is restored
__device__ __forceinline__ void bs_glds16(const void *gsrc, uint32_t ldsAddr)
{
    uint32_t keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(gsrc), "s"(ldsAddr)
                 : "memory");
}

struct BsBest {
    int b1, pos, b2;
};

__device__ __forceinline__ int bs_wave_min(int v) { return orb_wave_min_i(v); }
// minimum over the 16 lanes of a DPP row, result in every lane of the row
__device__ __forceinline__ int bs_row_min(int v) { return orb_row_min_i(v); }

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

// bitonic sort of both key arrays at once.  A compare-exchange phase with distance j <= 32 stays inside aligned
// blocks of 64 keys, and the 32 pairs of such a block belong to 32 consecutive threads of ONE wave: those phases
// (45 of the 55 for 1024 keys) only need the wave to agree, not the workgroup.
__device__ void bs_sort2(unsigned long long *keysA, unsigned long long *keysB, int NP, int tid)
{
    auto exchange = [&](int t, int k, int j) {
        unsigned long long *keys = t < NP / 2 ? keysA : keysB;
        const int u = t < NP / 2 ? t : t - NP / 2;
        // u-th compare-exchange pair of this (k, j) phase; j is a power of two
        const int i = ((u & ~(j - 1)) << 1) | (u & (j - 1));
        const int p = i + j;
        const bool up = ((i & k) == 0);
        const unsigned long long a = keys[i], b = keys[p];
        if ((a > b) == up) {
            keys[i] = b;
            keys[p] = a;
        }
    };
    for (int k = 2; k <= NP; k <<= 1) {
        int j = k >> 1;
        for (; j >= 64; j >>= 1) {
            for (int t = tid; t < NP; t += blockDim.x) exchange(t, k, j);
            __syncthreads();
        }
        for (int t = tid; t < NP; t += blockDim.x)
            for (int jj = j; jj > 0; jj >>= 1) {
                exchange(t, k, jj);
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            }
        if (k >= 64) __syncthreads();   // the next phase (distance k >= 64) crosses waves
    }
    __syncthreads();
}

// LDSD: both descriptor sets are staged in LDS (fits for cap up to ~1500 keypoints per frame); otherwise they
// are read from global memory.
template <bool LDSD>
__global__ __launch_bounds__(1024) void k_bow_seq(const uint8_t *__restrict__ desc,
                                                 const orbhip_keypoint *__restrict__ kps,
                                                 const int32_t *__restrict__ counts,
                                                 const int32_t *__restrict__ node, const float *__restrict__ weight,
                                                 const uint8_t *__restrict__ valid, int cap, int NP, int lag, int th,
                                                 int th_mode, float nnratio, int check_ori,
                                                 int32_t *__restrict__ match12, int32_t *__restrict__ match21,
                                                 int32_t *__restrict__ nmatches ORB_ABL_PARAM)
{
    extern __shared__ __align__(16) uint8_t smem[];
    unsigned long long *key1 = reinterpret_cast<unsigned long long *>(smem);
    unsigned long long *key2 = key1 + NP;
    int *m12 = reinterpret_cast<int *>(key2 + NP);                  // [NP] side-1 feature -> side-2 feature or -1
    uint2 *items = reinterpret_cast<uint2 *>(m12 + NP);             // [NP] (s1 | e1 << 16, s2 | e2 << 16)
    uint2 *sorted = items + NP;                                     // [NP] the items ordered by cost
    unsigned *claim = reinterpret_cast<unsigned *>(sorted + NP);    // [NP / 32] side-2 feature claimed
    unsigned *vbit1 = claim + NP / 32, *vbit2 = vbit1 + NP / 32;    // [NP / 32] "has a good MapPoint" per side
    // both descriptor sets live in LDS: the greedy loop is a chain of dependent reads (key -> descriptor ->
    // distance -> claim) per side-1 feature, and an LDS round trip is ~20x shorter than one to L2 / HBM
    uint4 *ls1 = reinterpret_cast<uint4 *>(smem + (((size_t)NP * 36 + (size_t)NP / 32 * 12 + 15) & ~(size_t)15));   // [cap][2]
    uint4 *ls2 = ls1 + (size_t)cap * 2;
    __shared__ int s_n1v, s_n2v, s_nitems, s_next, s_nm;
    __shared__ int s_hist[BS_HISTO];
    __shared__ int s_keep[3];
    __shared__ int s_grpItem[64];

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
    for (int i = tid; i < NP / 32; i += blockDim.x) {
        claim[i] = 0;
        unsigned v1 = 0xFFFFFFFFu, v2 = 0xFFFFFFFFu;
        if (valid) {
            v1 = v2 = 0;
            for (int k = 0; k < 32; k++) {
                const int f = i * 32 + k;
                if (f < n1 && valid[(size_t)b1 * cap + f]) v1 |= 1u << k;
                if (f < n2 && valid[(size_t)b * cap + f]) v2 |= 1u << k;
            }
        }
        vbit1[i] = v1;
        vbit2[i] = v2;
    }
    if (tid < BS_HISTO) s_hist[tid] = 0;
    if (tid == 0) {
        s_n1v = 0;
        s_n2v = 0;
        s_nitems = 0;
        s_next = 0;
        s_nm = 0;
    }
    __syncthreads();
    ORB_ABL_STOP(phases < 1);
    if (LDSD) {
        // both descriptor sets travel to LDS by LDS-DMA (16 bytes per lane, 1 KB per wave transfer) WHILE the keys are sorted: the
        // sort and the item list only touch the key arrays, the descriptors are first read by the greedy phase (waited for before
        // its barrier).  Issued after the key-building loads above so that no wait of the compiler's covers them.
        const int wv = __builtin_amdgcn_readfirstlane(tid >> 6), nw = (int)(blockDim.x >> 6);
        for (int side = 0; side < 2; side++) {
            const uint8_t *src = side ? d2 : d1;
            const int nchunk = (side ? n2 : n1) * 2;
            const uint32_t ldsBase = (uint32_t)(uintptr_t)(side ? ls2 : ls1);
            for (int base = wv * 64; base < nchunk; base += nw * 64)
                if (base + lane < nchunk) bs_glds16(src + (size_t)(base + lane) * 16, ldsBase + (uint32_t)base * 16u);
        }
    }
    // only indices < max(n1, n2) hold keys: sort the smallest power of two that covers them
    int NS = 64;
    while (NS < max(n1, n2)) NS <<= 1;
    bs_sort2(key1, key2, NS, tid);
    ORB_ABL_STOP(phases < 2);
    // number of live entries per side
    for (int i = tid; i < NS; i += blockDim.x) {
        if (key1[i] != ~0ull && (i + 1 == NS || key1[i + 1] == ~0ull)) s_n1v = i + 1;
        if (key2[i] != ~0ull && (i + 1 == NS || key2[i + 1] == ~0ull)) s_n2v = i + 1;
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
    if (LDSD) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's descriptor transfers have landed
    __syncthreads();

    ORB_ABL_STOP(phases < 3);
    // ---- 3. greedy matching ----
    // The greedy claiming of the reference is order dependent, so the side-1 features of a node form a serial
    // chain (key -> descriptor -> distances -> minima -> claim); the kernel's time is the longest chain plus
    // how well the chains are packed.  Items are therefore ordered by cost, largest first (LPT); a wave works on
    // whole nodes (candidates one per lane) until it draws one with at most 16 side-2 features, and from then on
    // runs four nodes at a time, one per 16-lane DPP row (either form is correct for any node size).
    const int nitems = s_nitems;
    {
        // order by cost = n1 * ceil(n2 / 16), ties by slot: rank by counting (nitems is at most a few hundred)
        for (int i = tid; i < nitems; i += blockDim.x) {
            const uint2 me = items[i];
            const int c = (int)((me.x >> 16) - (me.x & 0xFFFF)) * (int)((((me.y >> 16) - (me.y & 0xFFFF)) + 15) >> 4);
            int rank = 0;
            for (int o = 0; o < nitems; o++) {
                const uint2 ot = items[o];
                const int oc = (int)((ot.x >> 16) - (ot.x & 0xFFFF)) * (int)((((ot.y >> 16) - (ot.y & 0xFFFF)) + 15) >> 4);
                rank += (oc > c) || (oc == c && o < i);
            }
            sorted[rank] = me;
        }
        __syncthreads();
        const int gl = lane & 15, grp = tid >> 4;      // lane inside the group, group inside the workgroup
        int a = 0, e1 = 0, s2 = 0, e2 = 0;             // state of the current item (wave- or group-uniform)
        bool have = false, done = false, groupMode = false;
        // -- wave mode: nodes with more than 16 side-2 features --
        while (!groupMode) {
            int it = 0;
            if (lane == 0) it = atomicAdd(&s_next, 1);
            it = __builtin_amdgcn_readfirstlane(it);
            if (it >= nitems) {
                done = true;
                break;
            }
            const uint2 item = sorted[it];
            a = item.x & 0xFFFF;
            e1 = item.x >> 16;
            s2 = item.y & 0xFFFF;
            e2 = item.y >> 16;
            if (e2 - s2 <= 16) {        // the large nodes are done (cost order): from here on four nodes per wave
                groupMode = true;
                have = lane < 16;
                break;
            }
            if (e2 - s2 <= 128) {
                auto chain = [&](auto ncTag) {
                constexpr int NC = decltype(ncTag)::value;   // candidates per lane: 1 for nodes up to 64 side-2 features
                // At most two candidates per lane: their indices, descriptors and "still free" flags stay in registers
                // for the whole node (a side-2 feature belongs to one node, so only this wave ever claims it), and the
                // next side-1 descriptor is fetched while the current one is reduced.  The serial step is then distances
                // + two wave minima with no memory round trip in it -- the largest node's chain is the critical path of
                // the workgroup.
                int pc[NC], i2c[NC];
                bool avail[NC];
                uint4 r0[NC], r1[NC];
#pragma unroll
                for (int c = 0; c < NC; c++) {
                    pc[c] = s2 + lane + 64 * c;
                    const bool hasC = pc[c] < e2;
                    i2c[c] = hasC ? (int)(unsigned)key2[pc[c]] : 0;
                    avail[c] = hasC && !((claim[i2c[c] >> 5] >> (i2c[c] & 31)) & 1u) &&
                               !(th_mode && !((vbit2[i2c[c] >> 5] >> (i2c[c] & 31)) & 1u));   // :209-210, :572-578
                    r0[c] = LDSD ? ls2[2 * i2c[c]] : reinterpret_cast<const uint4 *>(d2)[2 * i2c[c]];
                    r1[c] = LDSD ? ls2[2 * i2c[c] + 1] : reinterpret_cast<const uint4 *>(d2)[2 * i2c[c] + 1];
                }
                int ni1 = a < e1 ? __builtin_amdgcn_readfirstlane((int)(unsigned)key1[a]) : 0;
                uint4 nq0 = LDSD ? ls1[2 * ni1] : reinterpret_cast<const uint4 *>(d1)[2 * ni1];
                uint4 nq1 = LDSD ? ls1[2 * ni1 + 1] : reinterpret_cast<const uint4 *>(d1)[2 * ni1 + 1];
                bool nvalid = a < e1 && ((vbit1[ni1 >> 5] >> (ni1 & 31)) & 1u);
                for (; a < e1; a++) {
                    const int i1 = ni1;
                    const uint4 q0 = nq0, q1 = nq1;
                    const bool valid1 = nvalid;
                    if (a + 1 < e1) {
                        ni1 = __builtin_amdgcn_readfirstlane((int)(unsigned)key1[a + 1]);
                        nq0 = LDSD ? ls1[2 * ni1] : reinterpret_cast<const uint4 *>(d1)[2 * ni1];
                        nq1 = LDSD ? ls1[2 * ni1 + 1] : reinterpret_cast<const uint4 *>(d1)[2 * ni1 + 1];
                        nvalid = (vbit1[ni1 >> 5] >> (ni1 & 31)) & 1u;
                    }
                    if (!valid1) continue;   // no (good) MapPoint: :193-199
                    BsBest B = {256, 0x7FFFFFFF, 256};
#pragma unroll
                    for (int c = 0; c < NC; c++) {
                        const int d = __popc(q0.x ^ r0[c].x) + __popc(q0.y ^ r0[c].y) + __popc(q0.z ^ r0[c].z) + __popc(q0.w ^ r0[c].w) +
                                      __popc(q1.x ^ r1[c].x) + __popc(q1.y ^ r1[c].y) + __popc(q1.z ^ r1[c].z) + __popc(q1.w ^ r1[c].w);
                        if (avail[c]) {
                            if (d < B.b1) {
                                B.b2 = B.b1;
                                B.b1 = d;
                                B.pos = pc[c];
                            } else if (d < B.b2) {
                                B.b2 = d;
                            }
                        }
                    }
                    const int key = B.b1 < 256 ? ((B.b1 << 16) | B.pos) : 0x7FFFFFFF;
                    const int k1 = bs_wave_min(key);
                    const int k2 = bs_wave_min(key == k1 ? B.b2 : B.b1);
                    const int b1 = k1 == 0x7FFFFFFF ? 256 : (k1 >> 16), b2 = k2;
                    const bool pass = th_mode ? (b1 < th) : (b1 <= th);
                    if (pass && (float)b1 < nnratio * (float)b2) {
                        const int pw = k1 & 0xFFFF;                          // winning position
                        const int i2 = (int)(unsigned)key2[pw];
                        if (lane == 0) {
                            m12[i1] = i2;
                            atomicOr(&claim[i2 >> 5], 1u << (i2 & 31));
                        }
#pragma unroll
                        for (int c = 0; c < NC; c++)
                            if (pc[c] == pw) avail[c] = false;
                    }
                }
                };
                if (e2 - s2 <= 64)
                    chain(std::integral_constant<int, 1>{});
                else
                    chain(std::integral_constant<int, 2>{});
                continue;
            }
            for (; a < e1; a++) {
                const int i1 = __builtin_amdgcn_readfirstlane((int)(unsigned)key1[a]);
                if (!((vbit1[i1 >> 5] >> (i1 & 31)) & 1u)) continue;   // no (good) MapPoint: :193-199
                const uint4 q0 = LDSD ? ls1[2 * i1] : reinterpret_cast<const uint4 *>(d1)[2 * i1];
                const uint4 q1 = LDSD ? ls1[2 * i1 + 1] : reinterpret_cast<const uint4 *>(d1)[2 * i1 + 1];
                const uint32_t Q[8] = {q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, q1.z, q1.w};
                BsBest B = {256, 0x7FFFFFFF, 256};
                for (int p = s2 + lane; p < e2; p += 64) {
                    const int i2 = (int)(unsigned)key2[p];
                    if ((claim[i2 >> 5] >> (i2 & 31)) & 1u) continue;   // already matched: :209-210
                    if (th_mode && !((vbit2[i2 >> 5] >> (i2 & 31)) & 1u)) continue;   // KF-KF variant: :572-578
                    const uint4 r0 = LDSD ? ls2[2 * i2] : reinterpret_cast<const uint4 *>(d2)[2 * i2];
                    const uint4 r1 = LDSD ? ls2[2 * i2 + 1] : reinterpret_cast<const uint4 *>(d2)[2 * i2 + 1];
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
                // best = minimum of (distance, position) over the lanes; second = minimum of the best lane's own
                // second and the other lanes' bests
                const int key = B.b1 < 256 ? ((B.b1 << 16) | B.pos) : 0x7FFFFFFF;
                const int k1 = bs_wave_min(key);
                const int k2 = bs_wave_min(key == k1 ? B.b2 : B.b1);
                const int b1 = k1 == 0x7FFFFFFF ? 256 : (k1 >> 16), b2 = k2;
                const bool pass = th_mode ? (b1 < th) : (b1 <= th);
                if (pass && (float)b1 < nnratio * (float)b2) {
                    const int i2 = (int)(unsigned)key2[k1 & 0xFFFF];
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
        // -- group mode: four small nodes per wave at a time --
        while (groupMode) {
            if (!have && !done) {
                if (gl == 0) s_grpItem[grp] = atomicAdd(&s_next, 1);
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            if (!have && !done) {
                const int it = s_grpItem[grp];
                if (it < nitems) {
                    const uint2 item = sorted[it];
                    a = item.x & 0xFFFF;
                    e1 = item.x >> 16;
                    s2 = item.y & 0xFFFF;
                    e2 = item.y >> 16;
                    have = true;
                } else {
                    done = true;
                }
            }
            if (__all(done)) break;
            if (have) {
                const int i1 = (int)(unsigned)key1[a];
                if ((vbit1[i1 >> 5] >> (i1 & 31)) & 1u) {   // has a (good) MapPoint: :193-199
                    const uint4 q0 = LDSD ? ls1[2 * i1] : reinterpret_cast<const uint4 *>(d1)[2 * i1];
                    const uint4 q1 = LDSD ? ls1[2 * i1 + 1] : reinterpret_cast<const uint4 *>(d1)[2 * i1 + 1];
                    const uint32_t Q[8] = {q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, q1.z, q1.w};
                    BsBest B = {256, 0x7FFFFFFF, 256};
                    for (int p = s2 + gl; p < e2; p += 16) {   // usually one trip (small nodes come here)
                        const int i2 = (int)(unsigned)key2[p];
                        if ((claim[i2 >> 5] >> (i2 & 31)) & 1u) continue;   // already matched: :209-210
                        if (th_mode && !((vbit2[i2 >> 5] >> (i2 & 31)) & 1u)) continue;   // KF-KF variant: :572-578
                        const uint4 r0 = LDSD ? ls2[2 * i2] : reinterpret_cast<const uint4 *>(d2)[2 * i2];
                        const uint4 r1 = LDSD ? ls2[2 * i2 + 1] : reinterpret_cast<const uint4 *>(d2)[2 * i2 + 1];
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
                    const int key = B.b1 < 256 ? ((B.b1 << 16) | B.pos) : 0x7FFFFFFF;
                    const int k1 = bs_row_min(key);
                    const int k2 = bs_row_min(key == k1 ? B.b2 : B.b1);
                    const int b1 = k1 == 0x7FFFFFFF ? 256 : (k1 >> 16), b2 = k2;
                    const bool pass = th_mode ? (b1 < th) : (b1 <= th);
                    if (pass && (float)b1 < nnratio * (float)b2) {
                        const int i2 = (int)(unsigned)key2[k1 & 0xFFFF];
                        if (gl == 0) {
                            m12[i1] = i2;
                            atomicOr(&claim[i2 >> 5], 1u << (i2 & 31));
                        }
                    }
                }
                a++;
                if (a >= e1) have = false;
            }
            // the fence at the loop top makes the claims visible to the group's next side-1 feature
        }
    }
    __syncthreads();

    ORB_ABL_STOP(phases < 4);
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


// =====================================================================================================================
// k_bow_lane (r05): the same search with ONE LANE PER VOCABULARY NODE.
//
// What k_bow_seq above spends its time on is not distances: a frame pair holds ~10^4 (side-1, side-2) pairs inside shared nodes
// (1000 features over ~100 nodes), but every side-1 feature costs a wave two cross-lane minima, a claim and a hand-over to the
// next feature -- ~150 instructions for ~10 useful distances, 16 waves of one 144 KB workgroup per CU, 0.25 ms per 1024 frames
// (profiles/r04: greedy phase 0.142 of it).  The greedy claiming is serial only INSIDE a node (a side-2 feature belongs to one
// node: ref :180-264, the claims of two nodes never meet), so here
//   a. the distances of all pairs of a node are computed up front, densely (16-lane groups, side-2 descriptor in registers,
//      side-1 descriptors streamed), into a byte matrix in LDS -- rows = the node's side-1 features in FeatureVector order,
//      16-byte chunks of side-2 candidates, 255 = "no candidate here" (padding) or a distance >= 255;
//   b. a lane then walks ITS node alone: per side-1 feature the row's chunks (one 16-byte LDS read each), claimed candidates OR-ed
//      to 255, best / second as min / med3 over keys (distance << 8 | position: the lowest position wins a tie, as the reference's
//      strict '<' scan does), the acceptance test, the claim -- no cross-lane traffic at all; 64 nodes per wave side by side,
//      nodes dealt in cost order so that a wave's lanes finish together.
// Nodes with more than 64 side-2 features, and what does not fit the byte matrix, take the wave-cooperative path (descriptors
// from global memory) after it: degenerate vocabularies (levelsup above the tree's depth puts every feature under one node) and
// nothing else.  Exactness of the 255 clamp: a clamped or claimed candidate can only be the SECOND best, and the ratio test then
// reads b1 < nnratio * 255 instead of * 256; with b1 <= th both hold whenever th < nnratio * 255, which the launcher checks
// (th 50 / 100, nnratio 0.6 .. 0.9 in the reference) -- otherwise k_bow_seq runs.
// LDS per workgroup (512 threads): keys 2 x NS x 8 bytes (dead after the item list: the byte matrix takes their place) + ~20 KB
// of index arrays: three workgroups per CU at 1000 features.
// =====================================================================================================================
#define BL_THREADS 512
#define BL_TILES 640     // 16 x 16 tiles of the byte matrices per frame pair (~200 at 1000 features)
#define BL_MAXN2 128   // side-2 features of a node the matrix paths take (8 candidates per lane of a 16-lane group)

__device__ __forceinline__ int bl_dist(const uint4 &q0, const uint4 &q1, const uint4 &r0, const uint4 &r1)
{
    return __popc(q0.x ^ r0.x) + __popc(q0.y ^ r0.y) + __popc(q0.z ^ r0.z) + __popc(q0.w ^ r0.w) + __popc(q1.x ^ r1.x) +
           __popc(q1.y ^ r1.y) + __popc(q1.z ^ r1.z) + __popc(q1.w ^ r1.w);
}

__global__ __launch_bounds__(BL_THREADS, 3) void k_bow_lane(const uint8_t *__restrict__ desc, const orbhip_keypoint *__restrict__ kps,
                                                        const int32_t *__restrict__ counts, const int32_t *__restrict__ node,
                                                        const float *__restrict__ weight, const uint8_t *__restrict__ valid, int cap,
                                                        int NP, int lag, int th, int th_mode, float nnratio, int check_ori,
                                                        int32_t *__restrict__ match12, int32_t *__restrict__ match21,
                                                        int32_t *__restrict__ nmatches ORB_ABL_PARAM)
{
    extern __shared__ __align__(16) uint8_t smem[];
    // [keys | byte matrix] [idx1 idx2 m12 : u16 x capP] [items : uint2 x capP] [ioff : u16 x capP] [order : u16 x capP] [claim vbit1 vbit2]
    const int capP = (cap + 63) & ~63;
    unsigned long long *key1 = reinterpret_cast<unsigned long long *>(smem);
    unsigned long long *key2 = key1 + NP;
    uint8_t *dist = smem;                                             // takes the keys' place once the items exist
    const int dcap = NP * 16 - BL_TILES * 4;                          // bytes (offsets count 4-byte units); behind them the tile list:
    uint32_t *tiles = reinterpret_cast<uint32_t *>(smem + dcap);      // rank | tile row << 11 | tile column << 15 (written once the keys are dead)
    uint16_t *idx1 = reinterpret_cast<uint16_t *>(smem + (size_t)NP * 16);
    uint16_t *idx2 = idx1 + capP;
    uint16_t *m12 = idx2 + capP;                                      // side-1 feature -> side-2 feature, 0xFFFF = none
    uint2 *items = reinterpret_cast<uint2 *>(m12 + capP);             // (s1 | e1 << 16, s2 | e2 << 16)
    uint16_t *ioff = reinterpret_cast<uint16_t *>(items + capP);      // byte-matrix offset of the item / 16; 0xFFFF: cooperative path
    uint16_t *order = ioff + capP;                                    // item slots: lane items by cost, then the others
    unsigned *claim = reinterpret_cast<unsigned *>(order + capP);
    unsigned *vbit1 = claim + NP / 32, *vbit2 = vbit1 + NP / 32;
    __shared__ int s_n1v, s_n2v, s_nitems, s_next, s_nm, s_over, s_ntiles;
    __shared__ int s_cls[4], s_cur[4];
    __shared__ int s_hist[BS_HISTO];
    __shared__ int s_keep[3];

    const int b = blockIdx.x;
    const int tid = threadIdx.x, lane = tid & 63;
    const int n2 = min(counts[b], cap);
    int32_t *o12 = match12 + (size_t)b * cap;
    int32_t *o21 = match21 + (size_t)b * cap;
    if (b < lag) {
        for (int i = tid; i < cap; i += BL_THREADS) {
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
    int NS = 64;
    while (NS < max(n1, n2)) NS <<= 1;

    // ---- 1. keys (node << 32 | index); absent / stopped features sort last ----
    for (int i = tid; i < NS; i += BL_THREADS) {
        key1[i] = (i < n1 && w1[i] > 0.f) ? (((unsigned long long)(unsigned)nd1[i] << 32) | (unsigned)i) : ~0ull;
        key2[i] = (i < n2 && w2[i] > 0.f) ? (((unsigned long long)(unsigned)nd2[i] << 32) | (unsigned)i) : ~0ull;
    }
    for (int i = tid; i < capP; i += BL_THREADS) m12[i] = 0xFFFFu;
    for (int i = tid; i < NP / 32; i += BL_THREADS) {
        claim[i] = 0;
        unsigned v1 = 0xFFFFFFFFu, v2 = 0xFFFFFFFFu;
        if (valid) {
            v1 = v2 = 0;
            for (int k = 0; k < 32; k++) {
                const int f = i * 32 + k;
                if (f < n1 && valid[(size_t)b1 * cap + f]) v1 |= 1u << k;
                if (f < n2 && valid[(size_t)b * cap + f]) v2 |= 1u << k;
            }
        }
        vbit1[i] = v1;
        vbit2[i] = v2;
    }
    if (tid < BS_HISTO) s_hist[tid] = 0;
    if (tid == 0) {
        s_n1v = 0;
        s_n2v = 0;
        s_nitems = 0;
        s_cls[0] = s_cls[1] = s_cls[2] = s_cls[3] = 0;
        s_cur[0] = s_cur[1] = s_cur[2] = s_cur[3] = 0;
        s_next = 0;
        s_nm = 0;
        s_over = 0;
        s_ntiles = 0;
    }
    __syncthreads();
    ORB_ABL_STOP(phases < 1);
    bs_sort2(key1, key2, NS, tid);
    ORB_ABL_STOP(phases < 2);
    for (int i = tid; i < NS; i += BL_THREADS) {
        if (key1[i] != ~0ull && (i + 1 == NS || key1[i + 1] == ~0ull)) s_n1v = i + 1;
        if (key2[i] != ~0ull && (i + 1 == NS || key2[i + 1] == ~0ull)) s_n2v = i + 1;
    }
    __syncthreads();
    const int n1v = s_n1v, n2v = s_n2v;

    // ---- 2. the FeatureVectors as index lists; work items = nodes present on both sides ----
    for (int p = tid; p < n1v; p += BL_THREADS) {
        idx1[p] = (uint16_t)(unsigned)key1[p];
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
    for (int p = tid; p < n2v; p += BL_THREADS) idx2[p] = (uint16_t)(unsigned)key2[p];
    __syncthreads();
    ORB_ABL_STOP(phases < 3);
    const int nitems = s_nitems;

    // ---- 3. classes and order: [group items, 17 .. 128 side-2 features][lane items, <= 16][the rest]; inside a class the order is
    //         whatever the counters give (every group item gets a 16-lane row of its own, every lane item a lane: nothing to
    //         balance); offsets of the byte matrices (rows of ceil4(n2) bytes) in that order ----
    auto class_of = [&](const uint2 it) {
        const int c2 = (int)((it.y >> 16) - (it.y & 0xFFFF));
        return c2 > BL_MAXN2 ? 3 : c2 > 64 ? 0 : c2 > 16 ? 1 : 2;
    };
    for (int i = tid; i < nitems; i += BL_THREADS) atomicAdd(&s_cls[class_of(items[i])], 1);
    __syncthreads();
    // ranks [0, ngroup8): group items of 65 .. 128 side-2 features; [ngroup8, ngroup): of 17 .. 64; [ngroup, nmat): lane items; then the rest
    const int ngroup8 = s_cls[0], ngroup = ngroup8 + s_cls[1], nmat = ngroup + s_cls[2];
    for (int i = tid; i < nitems; i += BL_THREADS) {
        const int cls = class_of(items[i]);
        const int r = (cls == 0 ? 0 : cls == 1 ? ngroup8 : cls == 2 ? ngroup : nmat) + atomicAdd(&s_cur[cls], 1);
        order[r] = (uint16_t)i;
    }
    __syncthreads();
    if (tid < 64) {
        // offsets (4-byte units) in rank order by one wave; what no longer fits the matrix space goes to the cooperative path
        int base = 0, tbase = 0;
        for (int r0 = 0; r0 < nmat; r0 += 64) {
            const int r = r0 + lane;
            int units = 0;
            if (r < nmat) {
                const uint2 it = items[order[r]];
                units = (int)((it.x >> 16) - (it.x & 0xFFFF)) * (int)((((it.y >> 16) - (it.y & 0xFFFF)) + 3) >> 2);
            }
            int incl = units;
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) {
                const int o = __shfl_up(incl, d);
                if (lane >= d) incl += o;
            }
            const int start = base + incl - units;
            // the item's 16 x 16 tiles, appended to the tile list
            int ntl = 0, tR = 0, tC = 0;
            if (r < nmat) {
                const uint2 it = items[order[r]];
                tR = ((int)((it.x >> 16) - (it.x & 0xFFFF)) + 15) >> 4;
                tC = ((int)((it.y >> 16) - (it.y & 0xFFFF)) + 15) >> 4;
                ntl = tR * tC;
            }
            int tincl = ntl;
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) {
                const int o = __shfl_up(tincl, d);
                if (lane >= d) tincl += o;
            }
            const int tstart = tbase + tincl - ntl;
            const bool fits = (start + units) * 4 <= dcap && start + units < 0xFFFF && tR <= 16 && tstart + ntl <= BL_TILES;
            if (r < nmat) ioff[order[r]] = fits ? (uint16_t)start : (uint16_t)0xFFFFu;
            // (an item that does not fit leaves empty tiles in its part of the list: the prefix counted them before that was known)
            if (r < nmat)
                for (int k = 0; k < ntl && tstart + k < BL_TILES; k++)
                    tiles[tstart + k] = fits ? ((uint32_t)r | ((uint32_t)(k / tC) << 11) | ((uint32_t)(k % tC) << 15)) : 0xFFFFFFFFu;
            if (__ballot(r < nmat && !fits) && lane == 0) s_over = 1;
            base += __shfl(incl, 63);
            tbase += __shfl(tincl, 63);
        }
        if (lane == 0) s_ntiles = min(tbase, BL_TILES);
    }
    for (int r = nmat + tid; r < nitems; r += BL_THREADS) ioff[order[r]] = 0xFFFFu;
    __syncthreads();
    ORB_ABL_STOP(phases < 4);
    const int ntiles = s_ntiles;
    // (the keys are dead from here on: `dist` overwrites them)

    // ---- 4. byte matrices, by 16 x 16 tiles (rows x candidates) dealt over the 16-lane groups: lane gl fetches the descriptor of ITS row
    //         and of ITS candidate -- four independent loads, the next tile's in flight while this one is computed -- and the
    //         rows' descriptors then ROTATE through the group (DPP row_ror on the XOR's operand: no LDS, no exchange): in step u
    //         lane gl holds row (gl - u) mod 16 and scores it against its own candidate ----
    {
        const int gl = lane & 15, grp = tid >> 4, ngrp = BL_THREADS >> 4;
        const uint4 *D1 = reinterpret_cast<const uint4 *>(d1), *D2 = reinterpret_cast<const uint4 *>(d2);
        struct Tile {
            int off, stride, nrow, c, row0;
            bool hasC;
        };
        uint4 q0, q1, r0, r1, nq0, nq1, nr0, nr1;
        Tile T, NT;
        auto fetch = [&](int t, Tile &X, uint4 &a0, uint4 &a1, uint4 &b0, uint4 &b1) {
            const uint32_t e0 = tiles[t];
            const bool empty = e0 == 0xFFFFFFFFu;
            const uint32_t e = empty ? 0u : e0;
            const int slot = order[e & 0x7FF], tr = (e >> 11) & 15, tc = (e >> 15) & 15;
            const uint2 it = items[slot];
            const int s1 = it.x & 0xFFFF, n1i = (int)(it.x >> 16) - s1, s2 = it.y & 0xFFFF, n2i = (int)(it.y >> 16) - s2;
            X.off = ioff[slot];
            X.stride = (n2i + 3) & ~3;
            X.nrow = empty || X.off == 0xFFFF ? 0 : n1i;          // (no row: nothing is stored)
            X.row0 = tr * 16;
            X.c = tc * 16 + gl;
            X.hasC = X.c < n2i;
            const int i1 = idx1[s1 + min(tr * 16 + gl, n1i - 1)];
            const int i2 = idx2[s2 + min(X.c, n2i - 1)];
            a0 = D1[2 * i1];
            a1 = D1[2 * i1 + 1];
            b0 = D2[2 * i2];
            b1 = D2[2 * i2 + 1];
        };
        int t = grp;
        if (t < ntiles) fetch(t, NT, nq0, nq1, nr0, nr1);
        for (; t < ntiles; t += ngrp) {
            T = NT;
            q0 = nq0; q1 = nq1; r0 = nr0; r1 = nr1;
            if (t + ngrp < ntiles) fetch(t + ngrp, NT, nq0, nq1, nr0, nr1);
            uint8_t *dst = dist + (size_t)T.off * 4 + T.c;
            const bool inPad = T.c < T.stride;
            // One assembly block per step: a vector write of a register needs two wait states before a DPP read of it, and the
            // compiler neither sees the DPP reads inside inline assembly nor is kept from placing a register copy right in front of
            // one (the whole-batch check of bench.py caught exactly that: a copy of a descriptor dword ahead of its rotated read,
            // one frame pair in a thousand off by a match).  The s_nop at the top covers every operand; nothing can be scheduled
            // into the block.
#define BL_STEP(U)                                                                                                              \
    {                                                                                                                           \
        const int row = T.row0 + ((gl - (U)) & 15);                                                                            \
        uint32_t x0, x1, x2, x3, x4, x5, x6, x7;                                                                                \
        if ((U) == 0) {                                                                                                         \
            x0 = q0.x ^ r0.x; x1 = q0.y ^ r0.y; x2 = q0.z ^ r0.z; x3 = q0.w ^ r0.w;                                             \
            x4 = q1.x ^ r1.x; x5 = q1.y ^ r1.y; x6 = q1.z ^ r1.z; x7 = q1.w ^ r1.w;                                             \
        } else                                                                                                                  \
            asm volatile("s_nop 1\n\t"                                                                                          \
                         "v_xor_b32_dpp %0, %8, %16 row_ror:" #U " row_mask:0xf bank_mask:0xf\n\t"                              \
                         "v_xor_b32_dpp %1, %9, %17 row_ror:" #U " row_mask:0xf bank_mask:0xf\n\t"                              \
                         "v_xor_b32_dpp %2, %10, %18 row_ror:" #U " row_mask:0xf bank_mask:0xf\n\t"                             \
                         "v_xor_b32_dpp %3, %11, %19 row_ror:" #U " row_mask:0xf bank_mask:0xf\n\t"                             \
                         "v_xor_b32_dpp %4, %12, %20 row_ror:" #U " row_mask:0xf bank_mask:0xf\n\t"                             \
                         "v_xor_b32_dpp %5, %13, %21 row_ror:" #U " row_mask:0xf bank_mask:0xf\n\t"                             \
                         "v_xor_b32_dpp %6, %14, %22 row_ror:" #U " row_mask:0xf bank_mask:0xf\n\t"                             \
                         "v_xor_b32_dpp %7, %15, %23 row_ror:" #U " row_mask:0xf bank_mask:0xf"                                 \
                         : "=&v"(x0), "=&v"(x1), "=&v"(x2), "=&v"(x3), "=&v"(x4), "=&v"(x5), "=&v"(x6), "=&v"(x7)               \
                         : "v"(q0.x), "v"(q0.y), "v"(q0.z), "v"(q0.w), "v"(q1.x), "v"(q1.y), "v"(q1.z), "v"(q1.w), "v"(r0.x),   \
                           "v"(r0.y), "v"(r0.z), "v"(r0.w), "v"(r1.x), "v"(r1.y), "v"(r1.z), "v"(r1.w));                        \
        const int d = __popc(x0) + __popc(x1) + __popc(x2) + __popc(x3) + __popc(x4) + __popc(x5) + __popc(x6) + __popc(x7);    \
        if (inPad && row < T.nrow) dst[(size_t)row * T.stride] = (uint8_t)(T.hasC ? min(d, 255) : 255);                         \
    }
            BL_STEP(0) BL_STEP(1) BL_STEP(2) BL_STEP(3) BL_STEP(4) BL_STEP(5) BL_STEP(6) BL_STEP(7)
            BL_STEP(8) BL_STEP(9) BL_STEP(10) BL_STEP(11) BL_STEP(12) BL_STEP(13) BL_STEP(14) BL_STEP(15)
#undef BL_STEP
        }
    }
    __syncthreads();
    ORB_ABL_STOP(phases < 5);

    // ---- 5. greedy matching ----
    // 5a. group items: a 16-lane DPP row per node, four nodes per wave; lane gl owns the candidates gl, gl + 16, ... (J = 4 of them for
    //     nodes of up to 64 side-2 features, 8 up to 128), reads their bytes of the row, folds them into (best, second) keys, two row
    //     minima give the node's best / second
    {
        const int gl = lane & 15, grp = tid >> 4, ngrp = BL_THREADS >> 4;
        auto group_item = [&](auto jtag, int r) {
            constexpr int J = decltype(jtag)::value;
            const int slot = order[r];
            const int off = ioff[slot];
            if (off == 0xFFFF) return;
            const uint2 it = items[slot];
            const int s1 = it.x & 0xFFFF, nrow = (int)(it.x >> 16) - s1, s2 = it.y & 0xFFFF, n2i = (int)(it.y >> 16) - s2;
            const int stride = (n2i + 3) & ~3;
            uint32_t gm[J];                                  // 0xFF: my candidate gl + 16 j is claimed, invalid or absent
#pragma unroll
            for (int j = 0; j < J; j++) {
                const int c = gl + 16 * j;
                bool out = c >= n2i;
                if (!out && th_mode) {
                    const int i2 = idx2[s2 + c];
                    out = !((vbit2[i2 >> 5] >> (i2 & 31)) & 1u);   // :572-578
                }
                gm[j] = out ? 0xFFu : 0u;
            }
            // The next row's bytes and feature index are requested before this row is reduced (a row is then two DPP minima and a
            // chain of J med3 / min, no LDS round trip).  The index is fetched as the 32-bit word that holds it and taken apart
            // only where it is used, one row later: behind a 16-bit load the compiler zero-extends at once and puts its wait there
            // -- the round trip that was to be hidden.  (A first form issued the loads from inline assembly and waited at the loop
            // top; the compiler is free to copy such a register before the wait, and bench.py's whole-batch check found one frame
            // pair in a few thousand off by a match.)  A lane reads all J of its bytes whether the node has that many candidates or
            // not (what lies there is masked by `gm`, and the addresses stay inside the workgroup's LDS).
            const uint8_t *rowp = dist + (size_t)off * 4 + gl;
            const uint32_t *idx1w = reinterpret_cast<const uint32_t *>(idx1);
            const bool checkValid = valid != nullptr;
            uint32_t dn[J], iwn;
#pragma unroll
            for (int j = 0; j < J; j++) dn[j] = rowp[16 * j];
            iwn = idx1w[s1 >> 1];
            for (int a = 0; a < nrow; a++) {
                const int i1 = (int)((iwn >> (16 * ((s1 + a) & 1))) & 0xFFFFu);
                uint32_t dj[J];
#pragma unroll
                for (int j = 0; j < J; j++) dj[j] = dn[j] | gm[j];
                rowp += stride;
                if (a + 1 < nrow) {
#pragma unroll
                    for (int j = 0; j < J; j++) dn[j] = rowp[16 * j];
                    iwn = idx1w[(s1 + a + 1) >> 1];
                }
                if (checkValid && !((vbit1[i1 >> 5] >> (i1 & 31)) & 1u)) continue;   // (uniform over the row) no good MapPoint: :193-199
                unsigned k1 = 0xFFFFFFu, k2 = (256u << 8);
#pragma unroll
                for (int j = 0; j < J; j++) {
                    const unsigned key = (dj[j] << 8) | (unsigned)(gl + 16 * j);
                    unsigned nk2;
                    asm("v_med3_u32 %0, %1, %2, %3" : "=v"(nk2) : "v"(k1), "v"(k2), "v"(key));
                    k1 = min(k1, key);
                    k2 = nk2;
                }
                const int K1 = bs_row_min((int)k1);
                const int K2 = bs_row_min((int)(k1 == (unsigned)K1 ? k2 : k1));
                const int bd1 = K1 >> 8, bd2 = K2 >> 8, pos = K1 & 0xFF;
                const bool pass = th_mode ? (bd1 < th) : (bd1 <= th);
                if (pass && (float)bd1 < nnratio * (float)bd2) {        // ref: :228-230 / :598-600
                    if ((pos & 15) == gl) {
#pragma unroll
                        for (int j = 0; j < J; j++)
                            if ((pos >> 4) == j) gm[j] = 0xFFu;
                        m12[i1] = idx2[s2 + pos];
                    }
                }
            }
        };
        // (the two kinds never share a wave -- a wave would run them one after the other: the items of up to 64 candidates start at the
        // next multiple of four groups)
        const int g8pad = (ngroup8 + 3) & ~3, n4 = ngroup - ngroup8;
        for (int g = grp; g < g8pad + n4; g += ngrp) {
            if (g < ngroup8)
                group_item(std::integral_constant<int, 8>{}, g);
            else if (g >= g8pad)
                group_item(std::integral_constant<int, 4>{}, ngroup8 + (g - g8pad));
        }
    }
    // 5b. lane items: a lane walks its node alone (at most 16 candidates: up to four dwords per row); dealt from the last thread down,
    //     so that they start on the waves the group items left idle
    for (int r = ngroup + (BL_THREADS - 1 - tid); r < nmat; r += BL_THREADS) {
        const int slot = order[r];
        const int off = ioff[slot];
        if (off == 0xFFFF) continue;
        const uint2 it = items[slot];
        const int s1 = it.x & 0xFFFF, nrow = (int)(it.x >> 16) - s1, s2 = it.y & 0xFFFF, n2i = (int)(it.y >> 16) - s2;
        const int nw = (n2i + 3) >> 2;                      // dwords per row
        uint32_t cm[4] = {0u, 0u, 0u, 0u};                    // claimed / invalid candidates as bytes of 255
        if (th_mode) {
            for (int c = 0; c < n2i; c++) {
                const int i2 = idx2[s2 + c];
                if (!((vbit2[i2 >> 5] >> (i2 & 31)) & 1u)) {
                    const uint32_t m = 0xFFu << (8 * (c & 3));
#pragma unroll
                    for (int k = 0; k < 4; k++)
                        if ((c >> 2) == k) cm[k] |= m;
                }
            }
        }
        const uint32_t *row = reinterpret_cast<const uint32_t *>(dist + (size_t)off * 4);
        const uint32_t *idx1w = reinterpret_cast<const uint32_t *>(idx1);
        const bool checkValid = valid != nullptr;
        // (four dwords of every row are read whatever the node's width: the dwords beyond it belong to the next row and are masked;
        // the next row and its feature index -- as the 32-bit word that holds it, see 5a -- are requested before this row is reduced)
        const uint32_t padm[4] = {0u, nw > 1 ? 0u : 0xFFFFFFFFu, nw > 2 ? 0u : 0xFFFFFFFFu, nw > 3 ? 0u : 0xFFFFFFFFu};
        uint32_t wn[4], iwn;
#pragma unroll
        for (int k = 0; k < 4; k++) wn[k] = row[k];
        iwn = idx1w[s1 >> 1];
        for (int a = 0; a < nrow; a++) {
            const int i1 = (int)((iwn >> (16 * ((s1 + a) & 1))) & 0xFFFFu);
            uint32_t w[4];
#pragma unroll
            for (int k = 0; k < 4; k++) w[k] = wn[k] | cm[k] | padm[k];
            row += nw;
            if (a + 1 < nrow) {
#pragma unroll
                for (int k = 0; k < 4; k++) wn[k] = row[k];
                iwn = idx1w[(s1 + a + 1) >> 1];
            }
            if (checkValid && !((vbit1[i1 >> 5] >> (i1 & 31)) & 1u)) continue;   // no (good) MapPoint: :193-199
            unsigned k1 = 0xFFFFFFu, k2 = (256u << 8);          // best / second keys (distance << 8 | position)
#pragma unroll
            for (int k = 0; k < 4; k++) {
                if (k < nw) {
#pragma unroll
                    for (int q = 0; q < 4; q++) {
                        const unsigned key = (((w[k] >> (8 * q)) & 0xFFu) << 8) | (unsigned)(4 * k + q);
                        unsigned nk2;                            // second smallest of the three (k1 <= k2 after the first key)
                        asm("v_med3_u32 %0, %1, %2, %3" : "=v"(nk2) : "v"(k1), "v"(k2), "v"(key));
                        k1 = min(k1, key);
                        k2 = nk2;
                    }
                }
            }
            const int bd1 = (int)(k1 >> 8), bd2 = (int)(k2 >> 8), pos = (int)(k1 & 0xFFu);
            const bool pass = th_mode ? (bd1 < th) : (bd1 <= th);
            if (pass && (float)bd1 < nnratio * (float)bd2) {        // ref: :228-230 / :598-600
                m12[i1] = idx2[s2 + pos];
                const uint32_t m = 0xFFu << (8 * (pos & 3));
#pragma unroll
                for (int k = 0; k < 4; k++)
                    if ((pos >> 2) == k) cm[k] |= m;
            }
        }
    }
    __syncthreads();
    ORB_ABL_STOP(phases < 6);

    // ---- 6. what is left: nodes of more than 64 side-2 features or beyond the byte matrix -- a wave per node, lanes scan the node's
    //         unclaimed side-2 features, descriptors from global memory (k_bow_seq's general form) ----
    // (the ranks from nlane on; the lane items too only if one of them did not fit the byte matrix)
    const int firstCoop = s_over ? 0 : nmat;
    if (firstCoop >= nitems) goto rotation;
    if (tid == 0) s_next = firstCoop;
    __syncthreads();
    for (;;) {
        int rr = 0;
        if (lane == 0) rr = atomicAdd(&s_next, 1);
        rr = __builtin_amdgcn_readfirstlane(rr);
        if (rr >= nitems) break;
        const int slot = order[rr];
        if (ioff[slot] != 0xFFFF) continue;
        const uint2 it = items[slot];
        const int e1 = it.x >> 16, s2 = it.y & 0xFFFF, e2 = it.y >> 16;
        for (int a = it.x & 0xFFFF; a < e1; a++) {
            const int i1 = __builtin_amdgcn_readfirstlane((int)idx1[a]);
            if (!((vbit1[i1 >> 5] >> (i1 & 31)) & 1u)) continue;
            const uint4 q0 = reinterpret_cast<const uint4 *>(d1)[2 * i1], q1 = reinterpret_cast<const uint4 *>(d1)[2 * i1 + 1];
            BsBest B = {256, 0x7FFFFFFF, 256};
            for (int p = s2 + lane; p < e2; p += 64) {
                const int i2 = idx2[p];
                if ((claim[i2 >> 5] >> (i2 & 31)) & 1u) continue;
                if (th_mode && !((vbit2[i2 >> 5] >> (i2 & 31)) & 1u)) continue;
                const uint4 r0 = reinterpret_cast<const uint4 *>(d2)[2 * i2], r1 = reinterpret_cast<const uint4 *>(d2)[2 * i2 + 1];
                const int d = bl_dist(q0, q1, r0, r1);
                if (d < B.b1) {
                    B.b2 = B.b1;
                    B.b1 = d;
                    B.pos = p;
                } else if (d < B.b2) {
                    B.b2 = d;
                }
            }
            const int key = B.b1 < 256 ? ((B.b1 << 16) | B.pos) : 0x7FFFFFFF;
            const int k1 = bs_wave_min(key);
            const int k2 = bs_wave_min(key == k1 ? B.b2 : B.b1);
            const int bd1 = k1 == 0x7FFFFFFF ? 256 : (k1 >> 16), bd2 = k2;
            const bool pass = th_mode ? (bd1 < th) : (bd1 <= th);
            if (pass && (float)bd1 < nnratio * (float)bd2) {
                const int i2 = idx2[k1 & 0xFFFF];
                if (lane == 0) {
                    m12[i1] = (uint16_t)i2;
                    atomicOr(&claim[i2 >> 5], 1u << (i2 & 31));
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            }
        }
    }
    __syncthreads();
rotation:
    ORB_ABL_STOP(phases < 7);

    // ---- 7. rotation consistency (:267-285) and outputs ----
    const orbhip_keypoint *k1p = kps + (size_t)b1 * cap, *k2p = kps + (size_t)b * cap;
    auto bin_of = [&](int i1, int i2) {
        float rot = k1p[i1].angle - k2p[i2].angle;
        if (rot < 0.0f) rot += 360.0f;
        int bin = (int)roundf(rot * (1.0f / BS_HISTO));
        if (bin == BS_HISTO) bin = 0;
        return bin;
    };
    if (check_ori) {
        for (int i1 = tid; i1 < n1; i1 += BL_THREADS) {
            const int i2 = m12[i1];
            if (i2 != 0xFFFF) {
                const int bin = bin_of(i1, i2);
                if (bin >= 0 && bin < BS_HISTO) atomicAdd(&s_hist[bin], 1);
            }
        }
        __syncthreads();
        if (tid == 0) {
            int max1 = 0, max2 = 0, max3 = 0, i1 = -1, i2 = -1, i3 = -1;
            for (int i = 0; i < BS_HISTO; i++) {
                const int sz = s_hist[i];
                if (sz > max1) {
                    max3 = max2; max2 = max1; max1 = sz;
                    i3 = i2; i2 = i1; i1 = i;
                } else if (sz > max2) {
                    max3 = max2; max2 = sz;
                    i3 = i2; i2 = i;
                } else if (sz > max3) {
                    max3 = sz;
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
    for (int i = tid; i < cap; i += BL_THREADS) o21[i] = -1;
    __syncthreads();
    int local = 0;
    for (int i1 = tid; i1 < cap; i1 += BL_THREADS) {
        int i2 = i1 < n1 ? (int)m12[i1] : 0xFFFF;
        if (i2 == 0xFFFF) i2 = -1;
        if (i2 >= 0 && check_ori) {
            const int bin = bin_of(i1, i2);
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

// Returns the launch's error (hipSuccess when a kernel is in the stream).  A lane kernel that the runtime refuses (dynamic LDS
// attribute or launch) is not an error yet: the wave-per-node kernel takes the call.
hipError_t launch_bow_seq(hipStream_t s, const uint8_t *desc, const orbhip_keypoint *kps, const int32_t *counts,
                          const int32_t *node, const float *weight, const uint8_t *valid, int cap, int B, int lag, int th,
                          int th_mode, float nnratio, int check_ori, int32_t *match12, int32_t *match21,
                          int32_t *nmatches)
{
    if (B <= 0) return hipSuccess;
    int NP = 512;
    while (NP < cap) NP <<= 1;
    // the lane-per-node kernel (above) when its byte clamp is exact for these thresholds, the features fit its 16-bit indices and
    // its LDS fits; ORBHIP_BOW_LANE=0 (liborbhip_ablation.so): the wave-per-node kernel
    static const int laneEnv = ORB_TUNE("BOW_LANE", 1);
    {
        const int capP = (cap + 63) & ~63;
        const size_t ldsLane = (size_t)NP * 16 + (size_t)capP * (3 * 2 + 8 + 2 + 2) + (size_t)NP / 32 * 12 + 64;
        if (laneEnv && cap < 65535 && NP <= 4096 && ldsLane <= 150 * 1024 && (float)th < nnratio * 255.0f && th < 255) {
            static const int dbgL = ORB_TUNE("BOW_PHASES", 9);
            (void)dbgL;
            hipError_t e = hipSuccess;
            if (ldsLane > 48 * 1024) e = hipFuncSetAttribute((const void *)k_bow_lane, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsLane);
            if (e == hipSuccess) {
                (void)hipGetLastError();
                hipLaunchKernelGGL(k_bow_lane, dim3(B, 1, 1), dim3(BL_THREADS, 1, 1), ldsLane, s, desc, kps, counts, node, weight, valid, cap,
                                   NP, lag, th, th_mode, nnratio, check_ori, match12, match21, nmatches ORB_ABL_ARG(dbgL));
                e = hipGetLastError();
            }
            if (e == hipSuccess) {
                orb_path(ORB_PATH_BOW_LANE);
                return hipSuccess;
            }
            (void)hipGetLastError();   // refused: fall through to k_bow_seq
        }
    }
    const size_t base = (((size_t)NP * 36 + (size_t)NP / 32 * 12 + 15) & ~(size_t)15) + 64;
    const size_t full = base + (size_t)cap * 64;
    static const int forceGlobal = ORB_TUNE("BOW_GLOBAL_DESC", 0);
    const bool ldsd = full <= 150 * 1024 && !forceGlobal;
    static const int dbg = ORB_TUNE("BOW_PHASES", 9);   // timing ablation only (liborbhip_ablation.so): results are then invalid
    (void)dbg;
    const size_t lds = ldsd ? full : base;
    const void *fn = ldsd ? (const void *)k_bow_seq<true> : (const void *)k_bow_seq<false>;
    if (lds > 48 * 1024) {
        const hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
    }
    orb_path(ldsd ? ORB_PATH_BOW_SEQ_LDS : ORB_PATH_BOW_SEQ_GLOBAL);
    if (ldsd)
        hipLaunchKernelGGL(k_bow_seq<true>, dim3(B, 1, 1), dim3(bs_threads(), 1, 1), lds, s, desc, kps, counts, node, weight, valid,
                           cap, NP, lag, th, th_mode, nnratio, check_ori, match12, match21, nmatches ORB_ABL_ARG(dbg));
    else
        hipLaunchKernelGGL(k_bow_seq<false>, dim3(B, 1, 1), dim3(bs_threads(), 1, 1), lds, s, desc, kps, counts, node, weight, valid,
                           cap, NP, lag, th, th_mode, nnratio, check_ori, match12, match21, nmatches ORB_ABL_ARG(dbg));
    return hipGetLastError();
}
