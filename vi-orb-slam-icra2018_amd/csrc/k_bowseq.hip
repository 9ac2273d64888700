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

void launch_bow_seq(hipStream_t s, const uint8_t *desc, const orbhip_keypoint *kps, const int32_t *counts,
                    const int32_t *node, const float *weight, const uint8_t *valid, int cap, int B, int lag, int th,
                    int th_mode, float nnratio, int check_ori, int32_t *match12, int32_t *match21,
                    int32_t *nmatches)
{
    if (B <= 0) return;
    int NP = 512;
    while (NP < cap) NP <<= 1;
    const size_t base = (((size_t)NP * 36 + (size_t)NP / 32 * 12 + 15) & ~(size_t)15) + 64;
    const size_t full = base + (size_t)cap * 64;
    static const int forceGlobal = ORB_TUNE("BOW_GLOBAL_DESC", 0);
    const bool ldsd = full <= 150 * 1024 && !forceGlobal;
    static const int dbg = ORB_TUNE("BOW_PHASES", 9);   // timing ablation only (liborbhip_ablation.so): results are then invalid
    (void)dbg;
    const size_t lds = ldsd ? full : base;
    const void *fn = ldsd ? (const void *)k_bow_seq<true> : (const void *)k_bow_seq<false>;
    if (lds > 48 * 1024) (void)hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (ldsd)
        hipLaunchKernelGGL(k_bow_seq<true>, dim3(B, 1, 1), dim3(bs_threads(), 1, 1), lds, s, desc, kps, counts, node, weight, valid,
                           cap, NP, lag, th, th_mode, nnratio, check_ori, match12, match21, nmatches ORB_ABL_ARG(dbg));
    else
        hipLaunchKernelGGL(k_bow_seq<false>, dim3(B, 1, 1), dim3(bs_threads(), 1, 1), lds, s, desc, kps, counts, node, weight, valid,
                           cap, NP, lag, th, th_mode, nnratio, check_ori, match12, match21, nmatches ORB_ABL_ARG(dbg));
}
