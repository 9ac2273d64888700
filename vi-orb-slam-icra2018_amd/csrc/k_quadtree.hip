// k_quadtree.hip -- E4: DistributeOctTree on the device, one 256-thread workgroup per
// (frame, level) (ref: src/ORBextractor.cc:541-765).  The algorithm lives in quadtree_core.h
// (shared with the CPU test that checks it against the list-based oracle); this file supplies the
// workgroup execution model (LDS atomics, wave-shuffle scans) and the gather of the per-cell
// candidate slots written by k_fast into the compact, canonically ordered point array.
#include "orbhip_internal.h"
#include "quadtree_core.h"

#include <cstdlib>

#ifndef QT_BATCH_LAT
#define QT_BATCH_LAT 0   // 1: batches take the one-wave scan of the single-frame variant too (measured r06: exposed half 0.110 ms instead of 0.104)
#endif
#ifndef QT_GK
#define QT_GK 4    // batches, gather: (cell, lane) pairs per thread and trip
#endif
#ifndef QT_GL
#define QT_GL 16   // ... lanes per cell (measured: 8 / 16 / 32 / 64 lanes x 4 / 8 / 16 pairs -- 16 x 4; 16 pairs cost registers, 64 lanes idle ones)
#endif
#define QT_LDS_LIMIT ((size_t)156 * 1024)   // node tables beyond this go to global memory (k_quadtree<.., true>)

// inclusive prefix sum over the 64 lanes on the DPP network (row shifts, then the totals of the lower rows)
__device__ __forceinline__ int qt_wave_incl_scan(int v)
{
    v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xF, 0xF, true);   // row_shr:1
    v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xF, 0xF, true);   // row_shr:2
    v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xF, 0xF, true);   // row_shr:4
    v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xF, 0xF, true);   // row_shr:8
    const int t0 = __builtin_amdgcn_readlane(v, 15), t1 = __builtin_amdgcn_readlane(v, 31), t2 = __builtin_amdgcn_readlane(v, 47);
    const int row = (int)(threadIdx.x & 63) >> 4;
    return v + (row > 0 ? t0 : 0) + (row > 1 ? t1 : 0) + (row > 2 ? t2 : 0);
}

// LAT: the single-frame variant (a handful of workgroups on the whole chip: every step is latency); the batch variant runs
// thousands of workgroups beside the blur and wants the fewest instructions instead
template <bool LAT>
struct QtBlock {
    int *wtot;  // [waves + 2] LDS: wave totals, then two alternating result slots of the one-wave scan
    mutable int flip = 0;
    __device__ __forceinline__ int tid() const { return threadIdx.x; }
    __device__ __forceinline__ int nth() const { return blockDim.x; }
    __device__ __forceinline__ void sync() const { __syncthreads(); }
    __device__ __forceinline__ int atomic_add(int *p, int v) const { return atomicAdd(p, v); }
    __device__ __forceinline__ void atomic_min(int *p, int v) const { atomicMin(p, v); }
    __device__ __forceinline__ void atomic_max(unsigned *p, unsigned v) const { atomicMax(p, v); }
    // *p += sum of v over the workgroup; called by every thread (the wave sums go to the counter, one atomic per wave)
    __device__ __forceinline__ void reduce_add(int *p, int v) const
    {
        const int incl = qt_wave_incl_scan(v);
        if ((threadIdx.x & 63) == 63 && incl != 0) atomicAdd(p, incl);
    }
    // sum of v over each aligned group of 2^sl adjacent threads (sl <= 6: a group lies inside one wave); called by every thread
    __device__ __forceinline__ int group_sum(int v, int sl) const
    {
        for (int o = 1; o < (1 << sl); o <<= 1) v += __shfl_xor(v, o);
        return v;
    }
    // In-place exclusive scan of a[0..n) (LDS); returns the total.  Called by all threads.
    __device__ int scan_exclusive(int *a, int n) const
    {
        const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
        if (LAT && n <= 512) {
            // the node lists of a level (up to ~250 entries) and the cell counts: ONE wave scans K consecutive entries per
            // lane, the others only wait -- one barrier instead of two and no walk over the wave totals (a third of the
            // single-frame quadtree time went into the general path below: ~1.3 us per scan, a dozen scans per level)
            const int slot = 16 + (flip & 1);
            flip++;
            if (wave == 0) {
                const int K = (n + 63) >> 6;
                const int beg = min(lane * K, n), end = min(beg + K, n);
                int v[8], sum = 0;
#pragma unroll
                for (int k = 0; k < 8; k++) {
                    v[k] = beg + k < end ? a[beg + k] : 0;
                    sum += v[k];
                }
                const int incl = qt_wave_incl_scan(sum);
                int run = incl - sum;
#pragma unroll
                for (int k = 0; k < 8; k++) {
                    if (beg + k < end) a[beg + k] = run;
                    run += v[k];
                }
                if (lane == 63) wtot[slot] = incl;
            }
            __syncthreads();
            return wtot[slot];
        }
        const int nt = blockDim.x;
        const int K = (n + nt - 1) / nt;
        const int beg = min(t * K, n), end = min(beg + K, n);
        int sum = 0;
        for (int i = beg; i < end; i++) sum += a[i];
        const int incl = qt_wave_incl_scan(sum);
        if (lane == 63) wtot[wave] = incl;
        __syncthreads();
        int base = 0, total = 0;
        for (int w = 0; w < (int)(blockDim.x >> 6); w++) {
            const int v = wtot[w];
            if (w < wave) base += v;
            total += v;
        }
        int run = base + incl - sum;
        for (int i = beg; i < end; i++) {
            const int v = a[i];
            a[i] = run;
            run += v;
        }
        __syncthreads();
        return total;
    }
};

// LDSPTS: the variant for a frame or two (single-frame latency: the level-0 workgroup's chain of ~70 barrier-separated steps
// is the longest kernel of the call).  Its compact candidate array and the per-point node labels live in LDS when the level
// has at most `ldsPts` candidates, so that a step costs an LDS round trip instead of a trip to L2; larger levels use the
// global arrays as the batch variant always does (there the quadtree runs beside the blur and LDS is what it must not hog).
// GLOBALT: the node tables of a (frame, level) live in a global scratch block instead of LDS -- the slow path for feature
// quotas whose tables exceed the 160 KB of LDS (more than ~2000 features on ONE level, e.g. nfeatures 2500 with nlevels 1;
// the reference has no such limit).  Same algorithm, same results; the workgroup's barriers order the global accesses
// (one CU: its L1 is write-through, __syncthreads fences at workgroup scope).
template <bool LDSPTS, bool GLOBALT>
__global__ __launch_bounds__(LDSPTS ? 1024 : 256) void k_quadtree(const OrbLevels G, const uint32_t *__restrict__ cand,
                                                  const uint16_t *__restrict__ cellCnt,
                                                  uint32_t *__restrict__ pts, uint32_t *__restrict__ pnode,
                                                  int32_t *__restrict__ lvlCandCnt,
                                                  uint32_t *__restrict__ lvlKp,
                                                  int32_t *__restrict__ lvlKpCnt, int maxNodes, int qtBytes, int cellBytes,
                                                  int ldsPts, uint8_t *__restrict__ tableScratch ORB_ABL_PARAM)
{
    // timing ablation (liborbhip_ablation.so, ORBHIP_QT_PHASES; INVALID results): bits 0..7 = passes of the distribution (0: all),
    // bit 8 = stop behind the gather, bits 12..15 = only level (value - 1)
    ORB_ABL_STOP(((phases >> 12) & 15) != 0 && (int)blockIdx.y != ((phases >> 12) & 15) - 1);
    extern __shared__ __align__(16) uint8_t smem[];
    __shared__ int s_wtot[18];
    // level = blockIdx.y: workgroups are dispatched x-fastest, so every frame's level 0 (the longest chain) starts first
    // and the short upper levels fill the tail
    const int l = blockIdx.y, frame = blockIdx.x;
    const OrbLevel &L = G.lv[l];
    const int tid = threadIdx.x;
    QtBlock<LDSPTS || QT_BATCH_LAT> x;
    x.wtot = s_wtot;

    // ---- gather: per-cell slots -> compact array in canonical order ----
    int *cellOff = reinterpret_cast<int *>(smem + (GLOBALT ? 0 : qtBytes));
    const int ncells = L.nCols * L.nRows;
    const uint16_t *cc = cellCnt + (size_t)frame * G.totalCells + L.cellBase;
    const uint32_t *slots = cand + (size_t)frame * G.totalCands + L.candBase;
    // a frame or two: sixteen lanes per cell (a cell holds a dozen candidates on average, up to cellCap), and the first slot of
    // every (cell, lane) pair is requested together with the cell counts -- its latency passes during the scan instead of
    // after it (the gather was two dependent trips to memory)
    constexpr int QT_PRE = 6;
    uint32_t pre[QT_PRE];
    const int npairs = ncells * 16;
    if (LDSPTS) {
#pragma unroll
        for (int k = 0; k < QT_PRE; k++) {
            const int idx = min(tid + k * (int)blockDim.x, npairs - 1);
            pre[k] = slots[(size_t)(idx >> 4) * L.cellCap + min(idx & 15, L.cellCap - 1)];
        }
    }
    for (int c = tid; c < ncells; c += blockDim.x) cellOff[c] = cc[c];
    // batches, one root (nIni == 1): the gather below also labels the points for the first pass of the distribution and counts
    // them into the root's four children (qt_distribute's firstCounted) -- one trip over the points and one barrier fewer
    QtShared sh;
    qt_carve(sh, GLOBALT ? tableScratch + ((size_t)frame * gridDim.y + l) * (size_t)qtBytes : smem, maxNodes);
    const bool firstCounted = !LDSPTS && L.nIni == 1;
    if (firstCounted && tid < 4) sh.ccnt[tid] = 0;
    if (LDSPTS) {
        // (pins the prefetched values above the barrier: the compiler would otherwise sink each load to its use)
#pragma unroll
        for (int k = 0; k < QT_PRE; k++) asm volatile("" : "+v"(pre[k]));
    }
    __syncthreads();
    const int n = x.scan_exclusive(cellOff, ncells);
    // The rest of the workgroup's work as a function of where the candidates live.  It is instantiated twice in the
    // single-frame variant -- candidates in LDS / in memory -- instead of switching two pointers: a pointer that may be either
    // makes every access to the candidates a FLAT instruction (slower than both, and it waits on both counters).
    auto rest = [&](uint32_t *P, uint32_t *PN) {
        if (LDSPTS) {
#pragma unroll
            for (int k = 0; k < QT_PRE; k++) {
                const int idx = tid + k * (int)blockDim.x;
                if (idx < npairs) {
                    const int c = idx >> 4, j = idx & 15, o = cellOff[c];
                    const int kk = (c + 1 < ncells ? cellOff[c + 1] : n) - o;
                    if (j < kk) P[o + j] = pre[k];
                    const uint32_t *src = slots + (size_t)c * L.cellCap;
                    for (int jj = j + 16; jj < kk; jj += 16) P[o + jj] = src[jj];
                }
            }
            for (int idx = tid + QT_PRE * (int)blockDim.x; idx < npairs; idx += blockDim.x) {   // more cells than 64 x 6 per 1024 threads
                const int c = idx >> 4, o = cellOff[c];
                const int kk = (c + 1 < ncells ? cellOff[c + 1] : n) - o;
                const uint32_t *src = slots + (size_t)c * L.cellCap;
                for (int jj = idx & 15; jj < kk; jj += 16) P[o + jj] = src[jj];
            }
        } else {
            // batches: QT_GL lanes per cell (a cell of the bench's frames holds ~26 candidates; a thread per cell walked them one
            // dependent trip to memory after the other), QT_GK (cell, lane) pairs per thread and trip with the loads ahead of
            // the stores
            constexpr int GL = QT_GL, GK = QT_GK;
            const int npairsB = ncells * GL;
            const int midx = qt_ceil_half((int)(short)(int)(L.hX * 1.f)), midy = qt_ceil_half((int)(short)L.regh);   // the root: (0, 0) .. (hX, regh)
            const bool split = firstCounted && n > 1;
            int cq[4] = {0, 0, 0, 0};
            auto put = [&](int d, uint32_t v) {
                P[d] = v;
                if (firstCounted) {
                    const int q = split ? (QT_X(v) < midx ? 0 : 1) + (QT_Y(v) < midy ? 0 : 2) : 0;
                    PN[d] = (uint32_t)q << 30;
                    if (split) {
                        cq[0] += q == 0;
                        cq[1] += q == 1;
                        cq[2] += q == 2;
                        cq[3] += q == 3;
                    }
                }
            };
            for (int idx0 = tid; idx0 < npairsB; idx0 += GK * (int)blockDim.x) {
                uint32_t val[GK];
                int dst[GK], cnt[GK];
#pragma unroll
                for (int k = 0; k < GK; k++) {
                    const int idx = idx0 + k * (int)blockDim.x;
                    const bool live = idx < npairsB;
                    const int c = live ? idx / GL : 0, j = idx % GL;
                    const int o = cellOff[c];
                    cnt[k] = live ? (c + 1 < ncells ? cellOff[c + 1] : n) - o : 0;
                    dst[k] = o + j;
                    val[k] = j < cnt[k] ? slots[(size_t)c * L.cellCap + j] : 0u;
                }
#pragma unroll
                for (int k = 0; k < GK; k++) {
                    const int idx = idx0 + k * (int)blockDim.x;
                    const int c = idx < npairsB ? idx / GL : 0, j = idx % GL;
                    if (j < cnt[k]) put(dst[k], val[k]);
                    for (int jj = j + GL; jj < cnt[k]; jj += GL) put(dst[k] - j + jj, slots[(size_t)c * L.cellCap + jj]);   // (rare)
                }
            }
            if (split) {   // workgroup-uniform
#pragma unroll
                for (int q = 0; q < 4; q++) x.reduce_add(&sh.ccnt[q], cq[q]);
            }
        }
        if (tid == 0) lvlCandCnt[frame * ORBHIP_MAX_LEVELS + l] = n;
        // make the compact array visible to the whole workgroup (global memory, same CU)
        __threadfence_block();
        __syncthreads();
        ORB_ABL_STOP(phases & 256);

        QtParams Q;
        Q.N = L.N;
        Q.nIni = L.nIni;
        Q.hX = L.hX;
        Q.regw = L.regw;
        Q.regh = L.regh;
        Q.maxNodes = L.kpCap;
        Q.maxIter = 64;
#ifdef ORBHIP_ABLATION
        if ((phases & 255) != 0) Q.maxIter = (phases & 255) - 1;
#endif
        uint32_t *out = lvlKp + (size_t)frame * G.totalKps + L.kpBase;
        const int S = qt_distribute(x, Q, n, P, PN, sh, out, firstCounted);
        if (tid == 0) lvlKpCnt[frame * ORBHIP_MAX_LEVELS + l] = S;
    };
    if (LDSPTS && n <= ldsPts) {   // block-uniform
        uint32_t *P = reinterpret_cast<uint32_t *>(smem + (GLOBALT ? 0 : qtBytes) + cellBytes);
        rest(P, P + ldsPts);
    } else {
        rest(pts + (size_t)frame * G.totalPts + L.ptBase, pnode + (size_t)frame * G.totalPts + L.ptBase);
    }
}

size_t quadtree_lds_bytes(const OrbLevels &G)
{
    int maxNodes = 0, maxCells = 0;
    for (int l = 0; l < G.nlevels; l++) {
        maxNodes = std::max(maxNodes, G.lv[l].kpCap);
        maxCells = std::max(maxCells, G.lv[l].nCols * G.lv[l].nRows);
    }
    return ((qt_shared_bytes(maxNodes) + 15) & ~(size_t)15) + (size_t)maxCells * 4 + 16;
}

size_t quadtree_table_scratch_bytes(const OrbLevels &G, int B)
{
    if (quadtree_lds_bytes(G) <= QT_LDS_LIMIT) return 0;
    int maxNodes = 0;
    for (int l = 0; l < G.nlevels; l++) maxNodes = std::max(maxNodes, G.lv[l].kpCap);
    return (size_t)B * G.nlevels * ((qt_shared_bytes(maxNodes) + 15) & ~(size_t)15);
}

void launch_quadtree(hipStream_t s, const OrbLevels &G, const uint32_t *cand, const uint16_t *cellCnt,
                     uint32_t *pts, uint32_t *pnode, int32_t *lvlCandCnt, uint32_t *lvlKp,
                     int32_t *lvlKpCnt, int B, uint8_t *tableScratch)
{
    int maxNodes = 0;
    for (int l = 0; l < G.nlevels; l++) maxNodes = std::max(maxNodes, G.lv[l].kpCap);
    const int qtBytes = (int)((qt_shared_bytes(maxNodes) + 15) & ~(size_t)15);
    // more threads were measured not to shorten the level-0 workgroup (its passes are barrier / LDS-latency chains);
    // the switch accepts 64..256 (the kernel's launch bound)
    static const int forced = ORB_TUNE("QT_THREADS", 256);
    const int nthreads = forced >= 64 && forced <= 256 && forced % 64 == 0 ? forced : 256;
    dim3 grid(B, G.nlevels, 1), block(nthreads, 1, 1);
    static const int phases = ORB_TUNE("QT_PHASES", 0);
    (void)phases;
    const size_t base = quadtree_lds_bytes(G);
    const int cellBytes = (int)(base - (size_t)qtBytes);
    if (base > QT_LDS_LIMIT) {
        // tables in global memory (tableScratch sized by quadtree_table_scratch_bytes); LDS holds the cell offsets only
        orb_path(ORB_PATH_QT_GLOBAL);
        hipLaunchKernelGGL((k_quadtree<false, true>), grid, block, (size_t)cellBytes, s, G, cand, cellCnt, pts, pnode, lvlCandCnt, lvlKp,
                           lvlKpCnt, maxNodes, qtBytes, cellBytes, 0, tableScratch ORB_ABL_ARG(phases));
        return;
    }
    if (B < 8) {
        // a frame or two: candidates and labels in LDS (8 bytes per candidate) for levels of up to 6144 candidates
        static const int forcedPts = ORB_TUNE("QT_LDSPTS", 6144);
        int ldsPts = forcedPts;
        while (ldsPts > 0 && base + (size_t)ldsPts * 8 > 150 * 1024) ldsPts -= 256;
        if (ldsPts > 0) {
            // ... and 1024 threads: the steps are loops over a few thousand candidates between barriers
            static const int smallThreads = ORB_TUNE("QT_THREADS_SMALL", 1024);
            block = dim3(smallThreads >= 64 && smallThreads <= 1024 && smallThreads % 64 == 0 ? smallThreads : 1024, 1, 1);
            orb_path(ORB_PATH_QT_LDSPTS);
            hipLaunchKernelGGL((k_quadtree<true, false>), grid, block, base + (size_t)ldsPts * 8, s, G, cand, cellCnt, pts, pnode,
                               lvlCandCnt, lvlKp, lvlKpCnt, maxNodes, qtBytes, cellBytes, ldsPts, nullptr ORB_ABL_ARG(phases));
            return;
        }
    }
    orb_path(ORB_PATH_QT_LDS);
    hipLaunchKernelGGL((k_quadtree<false, false>), grid, block, base, s, G, cand, cellCnt, pts, pnode, lvlCandCnt, lvlKp, lvlKpCnt,
                       maxNodes, qtBytes, cellBytes, 0, nullptr ORB_ABL_ARG(phases));
}
