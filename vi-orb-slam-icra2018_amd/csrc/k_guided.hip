// k_guided.hip -- SURVEY.md section 8f row 3: the frame grid and the guided (projection) search
// (ref: src/Frame.cc:574-589 AssignFeaturesToGrid, :726-736 PosInGrid, :671-724 GetFeaturesInArea;
//  src/ORBmatcher.cc:45-129 and :1341-1498 SearchByProjection).
//   k_grid_build    per frame: CSR of the 64 x 48 grid (cell id = ix * 48 + iy, the visiting order of
//                   GetFeaturesInArea) with LDS counters; cells are then sorted ascending, which is the
//                   push_back order of the reference;
//   k_area_list     GetFeaturesInArea for a batch of windows (one thread per window);
//   k_proj_records  the frame's features in CSR order as 16-byte records {x, y, octave | index}, so that a window walk
//                   reads consecutive memory instead of chasing cell entry -> keypoint;
//   k_proj_cands    one thread per projected point: walks its window, keeps the features that pass the octave,
//                   distance and right-coordinate tests, in the reference's order, with their descriptor
//                   distance -- everything that does not depend on which features earlier points took;
//   k_proj_assign   one wave per frame: the points in index order; a point's candidates sit one per lane, the
//                   features taken so far are a bitmap in LDS, best / second come from two minima over
//                   (distance, list position).  Four points are evaluated at a time, one per 16-lane row, and
//                   accepted in order until a row considered a feature that an earlier row of the group has just
//                   closed (that row is redone); then the rotation histogram.  A point with more candidates than
//                   the list holds is rescanned exactly as the reference does it.
#include "orbhip_internal.h"

#define GCOLS ORBHIP_GRID_COLS
#define GROWS ORBHIP_GRID_ROWS
#define GCELLS ORBHIP_GRID_CELLS
#define PROJ_K 32   // candidate slots per point (one 64-point chunk of lists = 8 KB of LDS)

#define WAVE_LDS_SYNC()                                        \
    do {                                                       \
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); \
        __builtin_amdgcn_wave_barrier();                       \
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); \
    } while (0)

struct GridParams {
    float minX, minY, invW, invH;
};

__device__ __forceinline__ int grid_cell(const GridParams &gp, float x, float y)
{
    const int px = (int)roundf(__fmul_rn(__fsub_rn(x, gp.minX), gp.invW));   // :728-729
    const int py = (int)roundf(__fmul_rn(__fsub_rn(y, gp.minY), gp.invH));
    return (px < 0 || px >= GCOLS || py < 0 || py >= GROWS) ? -1 : px * GROWS + py;
}

// inclusive prefix sum over the 64 lanes on the DPP network
__device__ __forceinline__ int grid_wave_incl_scan(int v)
{
    v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xF, 0xF, true);   // row_shr:1
    v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xF, 0xF, true);
    v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xF, 0xF, true);
    v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xF, 0xF, true);
    const int t0 = __builtin_amdgcn_readlane(v, 15), t1 = __builtin_amdgcn_readlane(v, 31), t2 = __builtin_amdgcn_readlane(v, 47);
    const int row = (int)(threadIdx.x & 63) >> 4;
    return v + (row > 0 ? t0 : 0) + (row > 1 ? t1 : 0) + (row > 2 ? t2 : 0);
}

// LDSIDX: the cell entries are sorted in LDS (cap * 4 bytes of dynamic LDS) and written out once -- the insertion sort on the
// global array was a chain of dependent memory trips per cell (most of the 24 us a single frame's grid took); frames with
// more features than fit keep the in-place form.
// NT threads: 256 for batches (one workgroup per frame, many frames), 1024 for a frame or two -- the single workgroup of a frame is
// a chain of barrier-separated loops, and four times the threads make every loop a quarter as long (r04: 19.1 -> 16.0 us from keeping
// the cells in registers, -> 9.2 us with 1024 threads).
template <bool LDSIDX, int NT>
__global__ __launch_bounds__(NT) void k_grid_build(const orbhip_keypoint *__restrict__ kps,
                                                    const int32_t *__restrict__ cnt, int cap, const GridParams gp,
                                                    int32_t *__restrict__ cellOff, int32_t *__restrict__ cellIdx,
                                                    int32_t *__restrict__ cellOff2, int32_t *__restrict__ cellIdx2)
{
    // (cellOff2 / cellIdx2, frame 0 only: the page-locked twin of orbhip_frame_build's result block; LDSIDX form only)
    extern __shared__ int s_idx[];
    __shared__ int s_cnt[GCELLS];
    __shared__ int s_wtot[NT / 64];
    const int b = blockIdx.x, tid = threadIdx.x;
    const int n = min(cnt[b], cap);
    const orbhip_keypoint *K = kps + (size_t)b * cap;
    int32_t *O = cellOff + (size_t)b * (GCELLS + 1), *I = cellIdx + (size_t)b * cap;
    int *E = LDSIDX ? s_idx : I;
    // a thread's first GB_KEEP features: position read and cell computed ONCE, kept for the fill pass (the single workgroup of a
    // frame is a chain of latencies: the second read of the keypoints was one more global round trip of its ~19 us)
    constexpr int GB_KEEP = 2048 / NT;
    int myCell[GB_KEEP];
    {
        float kx[GB_KEEP], ky[GB_KEEP];
#pragma unroll
        for (int u = 0; u < GB_KEEP; u++) {
            const int i = min(tid + NT * u, max(n - 1, 0));
            kx[u] = K[i].x;
            ky[u] = K[i].y;
        }
        for (int c = tid; c < GCELLS; c += NT) s_cnt[c] = 0;
#pragma unroll
        for (int u = 0; u < GB_KEEP; u++) myCell[u] = tid + NT * u < n ? grid_cell(gp, kx[u], ky[u]) : -1;
    }
    __syncthreads();
#pragma unroll
    for (int u = 0; u < GB_KEEP; u++)
        if (myCell[u] >= 0) atomicAdd(&s_cnt[myCell[u]], 1);
    for (int i = tid + NT * GB_KEEP; i < n; i += NT) {
        const int c = grid_cell(gp, K[i].x, K[i].y);
        if (c >= 0) atomicAdd(&s_cnt[c], 1);
    }
    __syncthreads();
    constexpr int PER = GCELLS / NT;   // 12 (3) cells per thread
    int c[PER], sum = 0;
#pragma unroll
    for (int k = 0; k < PER; k++) {
        c[k] = s_cnt[tid * PER + k];
        sum += c[k];
    }
    const int incl = grid_wave_incl_scan(sum);
    if ((tid & 63) == 63) s_wtot[tid >> 6] = incl;
    __syncthreads();
    int run = incl - sum;
    for (int w = 0; w < (tid >> 6); w++) run += s_wtot[w];
    if (tid == NT - 1) {
        O[GCELLS] = run + sum;
        if (cellOff2) cellOff2[GCELLS] = run + sum;
    }
#pragma unroll
    for (int k = 0; k < PER; k++) {
        s_cnt[tid * PER + k] = run;   // fill cursor
        O[tid * PER + k] = run;
        if (cellOff2) cellOff2[tid * PER + k] = run;
        run += c[k];
    }
    __syncthreads();
#pragma unroll
    for (int u = 0; u < GB_KEEP; u++)
        if (myCell[u] >= 0) E[atomicAdd(&s_cnt[myCell[u]], 1)] = tid + NT * u;
    for (int i = tid + NT * GB_KEEP; i < n; i += NT) {
        const int cc = grid_cell(gp, K[i].x, K[i].y);
        if (cc >= 0) E[atomicAdd(&s_cnt[cc], 1)] = i;
    }
    if (!LDSIDX) __threadfence_block();
    __syncthreads();
    // ascending feature index inside each cell (cells hold a handful of features): the thread's PER cells are neighbours, their
    // ends read together; only a cell of two or more entries has anything to sort
    int ends[PER + 1];
    ends[0] = tid ? s_cnt[tid * PER - 1] : 0;
#pragma unroll
    for (int k = 0; k < PER; k++) ends[k + 1] = s_cnt[tid * PER + k];   // cursors now sit at the cell ends
#pragma unroll
    for (int k = 0; k < PER; k++) {
        const int s = ends[k], e = ends[k + 1];
        for (int a = s + 1; a < e; a++) {
            const int v = E[a];
            int p = a - 1;
            while (p >= s && E[p] > v) {
                E[p + 1] = E[p];
                p--;
            }
            E[p + 1] = v;
        }
    }
    if (LDSIDX) {
        __syncthreads();
        const int total = s_cnt[GCELLS - 1];
        for (int j = tid; j < total; j += NT) {
            I[j] = s_idx[j];
            if (cellIdx2) cellIdx2[j] = s_idx[j];
        }
    }
}

// The window of GetFeaturesInArea in cells; false = the early returns of :676-691.
__device__ __forceinline__ bool window_cells(const GridParams &gp, float x, float y, float r, int &x0, int &x1, int &y0,
                                             int &y1)
{
    x0 = max(0, (int)floorf(__fmul_rn(__fsub_rn(__fsub_rn(x, gp.minX), r), gp.invW)));
    if (x0 >= GCOLS) return false;
    x1 = min(GCOLS - 1, (int)ceilf(__fmul_rn(__fadd_rn(__fsub_rn(x, gp.minX), r), gp.invW)));
    if (x1 < 0) return false;
    y0 = max(0, (int)floorf(__fmul_rn(__fsub_rn(__fsub_rn(y, gp.minY), r), gp.invH)));
    if (y0 >= GROWS) return false;
    y1 = min(GROWS - 1, (int)ceilf(__fmul_rn(__fadd_rn(__fsub_rn(y, gp.minY), r), gp.invH)));
    if (y1 < 0) return false;
    return true;
}

// Calls f(feature index, octave) for the features of GetFeaturesInArea(q.u, q.v, q.radius, q.min_level,
// q.max_level) in the reference's order.  Cells (ix, y0..y1) are contiguous in the CSR.
template <typename F>
__device__ __forceinline__ void walk_window(const GridParams &gp, const orbhip_proj_query &q,
                                            const orbhip_keypoint *__restrict__ K, const int32_t *__restrict__ O,
                                            const int32_t *__restrict__ I, F f)
{
    int x0, x1, y0, y1;
    if (!window_cells(gp, q.u, q.v, q.radius, x0, x1, y0, y1)) return;
    for (int ix = x0; ix <= x1; ix++) {
        const int s = O[ix * GROWS + y0], e = O[ix * GROWS + y1 + 1];
        for (int j = s; j < e; j++) {
            const int idx = I[j];
            const float kx = K[idx].x, ky = K[idx].y;
            const int oct = K[idx].octave;
            if (oct < q.min_level) continue;                       // with min_level <= 0 never true, :693-703
            if (q.max_level >= 0 && oct > q.max_level) continue;
            if (fabsf(__fsub_rn(kx, q.u)) < q.radius && fabsf(__fsub_rn(ky, q.v)) < q.radius) f(idx, oct);
        }
    }
}

__global__ __launch_bounds__(256) void k_area_list(const orbhip_keypoint *__restrict__ K, const GridParams gp,
                                                   const int32_t *__restrict__ O, const int32_t *__restrict__ I,
                                                   const orbhip_proj_query *__restrict__ queries, int nq, int slots,
                                                   int32_t *__restrict__ outCnt, int32_t *__restrict__ outIdx)
{
    const int iq = blockIdx.x * 256 + threadIdx.x;
    if (iq >= nq) return;
    const orbhip_proj_query q = queries[iq];
    int n = 0;
    int32_t *out = outIdx + (size_t)iq * slots;
    walk_window(gp, q, K, O, I, [&](int idx, int) {
        if (n < slots) out[n] = idx;
        n++;
    });
    outCnt[iq] = n;
}

__device__ __forceinline__ int hamming256g(const uint4 a0, const uint4 a1, const uint4 r0, const uint4 r1)
{
    return __popc(a0.x ^ r0.x) + __popc(a0.y ^ r0.y) + __popc(a0.z ^ r0.z) + __popc(a0.w ^ r0.w) + __popc(a1.x ^ r1.x) +
           __popc(a1.y ^ r1.y) + __popc(a1.z ^ r1.z) + __popc(a1.w ^ r1.w);
}

// The features of a frame in CSR order as compact records {x, y, octave | index << 8}: the window walk of
// k_proj_cands then reads consecutive 16-byte records instead of following cell entry -> keypoint (two dependent
// memory round trips per examined feature).
__global__ __launch_bounds__(256) void k_proj_records(const orbhip_keypoint *__restrict__ kps, int cap,
                                                      const int32_t *__restrict__ cellOff,
                                                      const int32_t *__restrict__ cellIdx, float4 *__restrict__ rec)
{
    const int b = blockIdx.y, j = blockIdx.x * 256 + threadIdx.x;
    if (j >= cellOff[(size_t)b * (GCELLS + 1) + GCELLS]) return;
    const int idx = cellIdx[(size_t)b * cap + j];
    const orbhip_keypoint k = kps[(size_t)b * cap + idx];
    rec[(size_t)b * cap + j] = make_float4(k.x, k.y, __int_as_float((k.octave & 255) | (idx << 8)), 0.f);
}

template <typename F>
__device__ __forceinline__ void walk_window_rec(const GridParams &gp, const orbhip_proj_query &q,
                                                const float4 *__restrict__ R, const int32_t *__restrict__ O, F f)
{
    int x0, x1, y0, y1;
    if (!window_cells(gp, q.u, q.v, q.radius, x0, x1, y0, y1)) return;
    for (int ix = x0; ix <= x1; ix++) {
        const int s = O[ix * GROWS + y0], e = O[ix * GROWS + y1 + 1];
        for (int j = s; j < e; j++) {
            const float4 r = R[j];
            const int w = __float_as_int(r.z), oct = w & 255;
            if (oct < q.min_level) continue;
            if (q.max_level >= 0 && oct > q.max_level) continue;
            if (fabsf(__fsub_rn(r.x, q.u)) < q.radius && fabsf(__fsub_rn(r.y, q.v)) < q.radius) f(w >> 8, oct);
        }
    }
}

// candidate tuple: distance (9 bits) | octave << 9 (4 bits) | feature index << 13
__global__ __launch_bounds__(256) void k_proj_cands(const orbhip_keypoint *__restrict__ kps,
                                                    const uint8_t *__restrict__ desc, int cap,
                                                    const float *__restrict__ uRight, const GridParams gp,
                                                    const int32_t *__restrict__ cellOff,
                                                    const float4 *__restrict__ rec,
                                                    const orbhip_proj_query *__restrict__ queries,
                                                    const uint8_t *__restrict__ qdesc, const int32_t *__restrict__ nq,
                                                    int capQ, int capQpad, int keff, uint32_t *__restrict__ tuples,
                                                    int32_t *__restrict__ tcount)
{
    const int b = blockIdx.y, iq = blockIdx.x * 256 + threadIdx.x;
    if (iq >= capQpad) return;
    int count = 0;
    if (iq < min(nq[b], capQ)) {
        const orbhip_proj_query q = queries[(size_t)b * capQ + iq];
        if (q.flags & ORBHIP_Q_ACTIVE) {
            const uint4 *qd = reinterpret_cast<const uint4 *>(qdesc + ((size_t)b * capQ + iq) * 32);
            const uint4 a0 = qd[0], a1 = qd[1];
            const uint4 *D = reinterpret_cast<const uint4 *>(desc + (size_t)b * cap * 32);
            const float *UR = uRight ? uRight + (size_t)b * cap : nullptr;
            uint32_t *T = tuples + ((size_t)b * capQpad + iq) * PROJ_K;
            walk_window_rec(gp, q, rec + (size_t)b * cap, cellOff + (size_t)b * (GCELLS + 1), [&](int idx, int oct) {
                if (UR) {
                    const float ur = UR[idx];
                    if (ur > 0 && fabsf(__fsub_rn(q.proj_xr, ur)) > q.radius) return;   // :92-97, :1418-1424
                }
                if (count < keff) {
                    const int d = hamming256g(a0, a1, D[2 * idx], D[2 * idx + 1]);
                    T[count] = (uint32_t)d | (((uint32_t)oct & 15u) << 9) | ((uint32_t)idx << 13);
                }
                count++;
            });
        }
    }
    tcount[(size_t)b * capQpad + iq] = count;
}

// The same lists for ONE frame per call: a 16-lane row per point instead of a thread.  A thread walks its window record by
// record (20-40 dependent trips to memory: 85 us for the 1000 points of a frame, however few of them there are); a row reads
// the cell ranges of up to 16 window columns at once, flattens them (row prefix sum) and examines 16 records per trip; the
// records that pass keep their visiting order through a row ballot.  Same tuples, same counts.
__global__ __launch_bounds__(256) void k_proj_cands_row(const uint8_t *__restrict__ desc, int cap, const float *__restrict__ uRight,
                                                        const GridParams gp, const int32_t *__restrict__ cellOff,
                                                        const float4 *__restrict__ rec,
                                                        const orbhip_proj_query *__restrict__ queries,
                                                        const uint8_t *__restrict__ qdesc, const int32_t *__restrict__ nq,
                                                        int capQ, int capQpad, int keff, uint32_t *__restrict__ tuples,
                                                        int32_t *__restrict__ tcount)
{
    __shared__ int s_start[16][16], s_excl[16][17];
    const int b = blockIdx.y, tid = threadIdx.x, gl = tid & 15, row = tid >> 4;
    const int iq = blockIdx.x * 16 + row;
    if (iq >= capQpad) return;   // row-uniform
    int count = 0;
    if (iq < min(nq[b], capQ)) {
        const orbhip_proj_query q = queries[(size_t)b * capQ + iq];
        int x0, x1, y0, y1;
        if ((q.flags & ORBHIP_Q_ACTIVE) && window_cells(gp, q.u, q.v, q.radius, x0, x1, y0, y1)) {
            const uint4 *qd = reinterpret_cast<const uint4 *>(qdesc + ((size_t)b * capQ + iq) * 32);
            const uint4 a0 = qd[0], a1 = qd[1];
            const uint4 *D = reinterpret_cast<const uint4 *>(desc + (size_t)b * cap * 32);
            const float *UR = uRight ? uRight + (size_t)b * cap : nullptr;
            const float4 *R = rec + (size_t)b * cap;
            const int32_t *O = cellOff + (size_t)b * (GCELLS + 1);
            uint32_t *T = tuples + ((size_t)b * capQpad + iq) * PROJ_K;
            for (int cb = x0; cb <= x1; cb += 16) {
                const int ix = cb + gl;
                const int s = ix <= x1 ? O[ix * GROWS + y0] : 0, e = ix <= x1 ? O[ix * GROWS + y1 + 1] : 0;
                int incl = e - s;
                incl += __builtin_amdgcn_update_dpp(0, incl, 0x111, 0xF, 0xF, true);   // row_shr:1
                incl += __builtin_amdgcn_update_dpp(0, incl, 0x112, 0xF, 0xF, true);
                incl += __builtin_amdgcn_update_dpp(0, incl, 0x114, 0xF, 0xF, true);
                incl += __builtin_amdgcn_update_dpp(0, incl, 0x118, 0xF, 0xF, true);
                s_start[row][gl] = s;
                s_excl[row][gl + 1] = incl;
                if (gl == 0) s_excl[row][0] = 0;
                WAVE_LDS_SYNC();
                const int total = s_excl[row][16];
                for (int rb = 0; rb < total; rb += 16) {
                    const int r = rb + gl;
                    bool pass = false;
                    uint32_t tup = 0;
                    if (r < total) {
                        int c = 0;                               // the column whose range holds record r
#pragma unroll
                        for (int h = 8; h > 0; h >>= 1)
                            if (s_excl[row][c + h] <= r) c += h;
                        const float4 rr = R[s_start[row][c] + (r - s_excl[row][c])];
                        const int w = __float_as_int(rr.z), oct = w & 255, idx = w >> 8;
                        pass = !(oct < q.min_level) && !(q.max_level >= 0 && oct > q.max_level) &&
                               fabsf(__fsub_rn(rr.x, q.u)) < q.radius && fabsf(__fsub_rn(rr.y, q.v)) < q.radius;
                        if (pass && UR) {
                            const float ur = UR[idx];
                            if (ur > 0 && fabsf(__fsub_rn(q.proj_xr, ur)) > q.radius) pass = false;   // :92-97, :1418-1424
                        }
                        if (pass) {
                            const int d = hamming256g(a0, a1, D[2 * idx], D[2 * idx + 1]);
                            tup = (uint32_t)d | (((uint32_t)oct & 15u) << 9) | ((uint32_t)idx << 13);
                        }
                    }
                    const unsigned m = (unsigned)((__ballot(pass) >> (tid & 48)) & 0xFFFFull);
                    const int pos = count + __popc(m & ((1u << gl) - 1u));
                    if (pass && pos < keff) T[pos] = tup;
                    count += __popc(m);
                }
                WAVE_LDS_SYNC();
            }
        }
    }
    if (gl == 0) tcount[(size_t)b * capQpad + iq] = count;
}

// ---- per-query best feature of a KeyFrame window: the inner loop of ORBmatcher::Fuse (ref: src/ORBmatcher.cc:887-950 with
// the chi-square gate on the reprojection error, :1044-1075 without) and of SearchBySim3 (:1190-1224, :1270-1304).  The
// queries are independent (no feature is closed by an earlier point), so one thread walks one window; the first feature
// of smallest distance wins.  -1 / 256 when the query is inactive or nothing is closer than 256.
struct LevelGate {
    float invSigma2[16];
    int on;
};

__global__ __launch_bounds__(256) void k_window_best(const uint8_t *__restrict__ desc, int cap,
                                                     const float *__restrict__ uRight, const LevelGate gate,
                                                     const GridParams gp, const int32_t *__restrict__ cellOff,
                                                     const float4 *__restrict__ rec,
                                                     const orbhip_proj_query *__restrict__ queries,
                                                     const uint8_t *__restrict__ qdesc, const int32_t *__restrict__ nq,
                                                     int capQ, int32_t *__restrict__ bestIdx, int32_t *__restrict__ bestDist)
{
    const int b = blockIdx.y, iq = blockIdx.x * 256 + threadIdx.x;
    if (iq >= capQ) return;
    int bd = 256, bi = -1;
    if (iq < nq[b]) {
        const orbhip_proj_query q = queries[(size_t)b * capQ + iq];
        int x0, x1, y0, y1;
        if ((q.flags & ORBHIP_Q_ACTIVE) && window_cells(gp, q.u, q.v, q.radius, x0, x1, y0, y1)) {
            const uint4 *qd = reinterpret_cast<const uint4 *>(qdesc + ((size_t)b * capQ + iq) * 32);
            const uint4 a0 = qd[0], a1 = qd[1];
            const uint4 *D = reinterpret_cast<const uint4 *>(desc + (size_t)b * cap * 32);
            const float *UR = uRight ? uRight + (size_t)b * cap : nullptr;
            const float4 *R = rec + (size_t)b * cap;
            const int32_t *O = cellOff + (size_t)b * (GCELLS + 1);
            for (int ix = x0; ix <= x1; ix++) {
                const int s = O[ix * GROWS + y0], e = O[ix * GROWS + y1 + 1];
                for (int j = s; j < e; j++) {
                    const float4 r = R[j];
                    const int w = __float_as_int(r.z), oct = w & 255, idx = w >> 8;
                    if (!(fabsf(__fsub_rn(r.x, q.u)) < q.radius && fabsf(__fsub_rn(r.y, q.v)) < q.radius)) continue;
                    if (oct < q.min_level || oct > q.max_level) continue;
                    if (gate.on) {
                        const float ex = __fsub_rn(q.u, r.x), ey = __fsub_rn(q.v, r.y);
                        float e2 = __fadd_rn(__fmul_rn(ex, ex), __fmul_rn(ey, ey));
                        const float ur = UR ? UR[idx] : -1.0f;
                        double lim = 5.99;
                        if (ur >= 0) {
                            const float er = __fsub_rn(q.proj_xr, ur);
                            e2 = __fadd_rn(e2, __fmul_rn(er, er));
                            lim = 7.8;
                        }
                        if ((double)__fmul_rn(e2, gate.invSigma2[oct & 15]) > lim) continue;
                    }
                    const int d = hamming256g(a0, a1, D[2 * idx], D[2 * idx + 1]);
                    if (d < bd) {
                        bd = d;
                        bi = idx;
                    }
                }
            }
        }
    }
    bestIdx[(size_t)b * capQ + iq] = bi;
    bestDist[(size_t)b * capQ + iq] = bd;
}

// minimum over the 16 lanes of a DPP row, result in every lane of the row
__device__ __forceinline__ int row_min_i(int v)
{
    v = min(v, __builtin_amdgcn_update_dpp(v, v, 0xB1, 0xF, 0xF, false));
    v = min(v, __builtin_amdgcn_update_dpp(v, v, 0x4E, 0xF, 0xF, false));
    v = min(v, __builtin_amdgcn_update_dpp(v, v, 0x141, 0xF, 0xF, false));
    v = min(v, __builtin_amdgcn_update_dpp(v, v, 0x140, 0xF, 0xF, false));
    return v;
}

// k_window_best for ONE key frame per call: a 16-lane row per point (see k_proj_cands_row); the first feature of smallest
// distance in visiting order = the minimum of (distance, position).
__global__ __launch_bounds__(256) void k_window_best_row(const uint8_t *__restrict__ desc, int cap, const float *__restrict__ uRight,
                                                         const LevelGate gate, const GridParams gp,
                                                         const int32_t *__restrict__ cellOff, const float4 *__restrict__ rec,
                                                         const orbhip_proj_query *__restrict__ queries,
                                                         const uint8_t *__restrict__ qdesc, const int32_t *__restrict__ nq, int capQ,
                                                         int32_t *__restrict__ bestIdx, int32_t *__restrict__ bestDist)
{
    __shared__ int s_start[16][16], s_excl[16][17];
    const int b = blockIdx.y, tid = threadIdx.x, gl = tid & 15, row = tid >> 4;
    const int iq = blockIdx.x * 16 + row;
    if (iq >= capQ) return;   // row-uniform
    int key = 0x7FFFFFFF, myIdx = -1;   // (distance << 20 | position), feature of this lane's best
    if (iq < nq[b]) {
        const orbhip_proj_query q = queries[(size_t)b * capQ + iq];
        int x0, x1, y0, y1;
        if ((q.flags & ORBHIP_Q_ACTIVE) && window_cells(gp, q.u, q.v, q.radius, x0, x1, y0, y1)) {
            const uint4 *qd = reinterpret_cast<const uint4 *>(qdesc + ((size_t)b * capQ + iq) * 32);
            const uint4 a0 = qd[0], a1 = qd[1];
            const uint4 *D = reinterpret_cast<const uint4 *>(desc + (size_t)b * cap * 32);
            const float *UR = uRight ? uRight + (size_t)b * cap : nullptr;
            const float4 *R = rec + (size_t)b * cap;
            const int32_t *O = cellOff + (size_t)b * (GCELLS + 1);
            int seen = 0;
            for (int cb = x0; cb <= x1; cb += 16) {
                const int ix = cb + gl;
                const int s = ix <= x1 ? O[ix * GROWS + y0] : 0, e = ix <= x1 ? O[ix * GROWS + y1 + 1] : 0;
                int incl = e - s;
                incl += __builtin_amdgcn_update_dpp(0, incl, 0x111, 0xF, 0xF, true);
                incl += __builtin_amdgcn_update_dpp(0, incl, 0x112, 0xF, 0xF, true);
                incl += __builtin_amdgcn_update_dpp(0, incl, 0x114, 0xF, 0xF, true);
                incl += __builtin_amdgcn_update_dpp(0, incl, 0x118, 0xF, 0xF, true);
                s_start[row][gl] = s;
                s_excl[row][gl + 1] = incl;
                if (gl == 0) s_excl[row][0] = 0;
                WAVE_LDS_SYNC();
                const int total = s_excl[row][16];
                for (int r = gl; r < total; r += 16) {
                    int c = 0;
#pragma unroll
                    for (int h = 8; h > 0; h >>= 1)
                        if (s_excl[row][c + h] <= r) c += h;
                    const float4 rr = R[s_start[row][c] + (r - s_excl[row][c])];
                    const int w = __float_as_int(rr.z), oct = w & 255, idx = w >> 8;
                    if (!(fabsf(__fsub_rn(rr.x, q.u)) < q.radius && fabsf(__fsub_rn(rr.y, q.v)) < q.radius)) continue;
                    if (oct < q.min_level || oct > q.max_level) continue;
                    if (gate.on) {
                        const float ex = __fsub_rn(q.u, rr.x), ey = __fsub_rn(q.v, rr.y);
                        float e2 = __fadd_rn(__fmul_rn(ex, ex), __fmul_rn(ey, ey));
                        const float ur = UR ? UR[idx] : -1.0f;
                        double lim = 5.99;
                        if (ur >= 0) {
                            const float er = __fsub_rn(q.proj_xr, ur);
                            e2 = __fadd_rn(e2, __fmul_rn(er, er));
                            lim = 7.8;
                        }
                        if ((double)__fmul_rn(e2, gate.invSigma2[oct & 15]) > lim) continue;
                    }
                    const int d = hamming256g(a0, a1, D[2 * idx], D[2 * idx + 1]);
                    const int k = (d << 20) | (seen + r);
                    if (d < 256 && k < key) {
                        key = k;
                        myIdx = idx;
                    }
                }
                seen += total;
                WAVE_LDS_SYNC();
            }
        }
    }
    const int k1 = row_min_i(key);
    if (key == k1 && k1 != 0x7FFFFFFF) {   // one lane: positions are unique
        bestIdx[(size_t)b * capQ + iq] = myIdx;
        bestDist[(size_t)b * capQ + iq] = k1 >> 20;
    }
    if (k1 == 0x7FFFFFFF && gl == 0) {
        bestIdx[(size_t)b * capQ + iq] = -1;
        bestDist[(size_t)b * capQ + iq] = 256;
    }
}

__device__ __forceinline__ int wave_min_i(int v) { return orb_wave_min_i(v); }


__device__ __forceinline__ int wave_sum_g(int v)
{
    v += __builtin_amdgcn_update_dpp(0, v, 0xB1, 0xF, 0xF, true);
    v += __builtin_amdgcn_update_dpp(0, v, 0x4E, 0xF, 0xF, true);
    v += __builtin_amdgcn_update_dpp(0, v, 0x141, 0xF, 0xF, true);
    v += __builtin_amdgcn_update_dpp(0, v, 0x140, 0xF, 0xF, true);
    return __builtin_amdgcn_readlane(v, 0) + __builtin_amdgcn_readlane(v, 16) + __builtin_amdgcn_readlane(v, 32) +
           __builtin_amdgcn_readlane(v, 48);
}


__global__ __launch_bounds__(64) void k_proj_assign(const orbhip_keypoint *__restrict__ kps,
                                                    const uint8_t *__restrict__ desc,
                                                    const int32_t *__restrict__ cnt, int cap,
                                                    const float *__restrict__ uRight,
                                                    const uint8_t *__restrict__ occupied, const GridParams gp,
                                                    const int32_t *__restrict__ cellOff,
                                                    const int32_t *__restrict__ cellIdx,
                                                    const orbhip_proj_query *__restrict__ queries,
                                                    const uint8_t *__restrict__ qdesc, const int32_t *__restrict__ nq,
                                                    int capQ, int capQpad, int keff, const uint32_t *__restrict__ tuples,
                                                    const int32_t *__restrict__ tcount, int32_t *__restrict__ qfeat,
                                                    int use_ratio, float nnratio, int check_ori, int th_high,
                                                    int32_t *__restrict__ match, int32_t *__restrict__ nmatches,
                                                    const int32_t *__restrict__ fallback)
{
    extern __shared__ uint32_t s_dyn[];
    __shared__ uint32_t s_tup[64 * PROJ_K];
    if (fallback && fallback[blockIdx.x] == 0) return;   // k_proj_assign_par has done this frame
    __shared__ int s_hist[30];
    __shared__ int s_keep[3];
    const int b = blockIdx.x, lane = threadIdx.x;
    const int n = min(cnt[b], cap), NQ = min(nq[b], capQ);
    int32_t *s_match = reinterpret_cast<int32_t *>(s_dyn);      // [cap]
    uint32_t *s_occ = s_dyn + cap;                              // [(cap + 31) / 32]
    const orbhip_keypoint *K = kps + (size_t)b * cap;
    const uint4 *D = reinterpret_cast<const uint4 *>(desc + (size_t)b * cap * 32);
    const float *UR = uRight ? uRight + (size_t)b * cap : nullptr;
    const int32_t *O = cellOff + (size_t)b * (GCELLS + 1), *I = cellIdx + (size_t)b * cap;
    const orbhip_proj_query *Q = queries + (size_t)b * capQ;
    for (int i = lane; i < cap; i += 64) s_match[i] = -1;
    for (int w = lane; w < (cap + 31) / 32; w += 64) {
        uint32_t bits = 0;
        if (occupied)
            for (int k = 0; k < 32; k++) {
                const int i = w * 32 + k;
                if (i < n && occupied[(size_t)b * cap + i]) bits |= 1u << k;
            }
        s_occ[w] = bits;
    }
    if (lane < 30) s_hist[lane] = 0;
    WAVE_LDS_SYNC();
    int nm = 0;
    const uint4 *Tg = reinterpret_cast<const uint4 *>(tuples + (size_t)b * capQpad * PROJ_K);
    for (int base = 0; base < NQ; base += 64) {
        const int myq = base + lane;
        const int myc = myq < NQ ? tcount[(size_t)b * capQpad + myq] : 0;
        const int myflags = myq < NQ ? Q[myq].flags : 0;
        int myfeat = -1;
        unsigned long long todo = __ballot(myc > 0);
        if (todo) {
            // the chunk's candidate lists -> LDS (64 lists x 32 slots, contiguous in memory)
            uint4 *s4 = reinterpret_cast<uint4 *>(s_tup);
            const uint4 *src = Tg + (size_t)base * (PROJ_K / 4);
#pragma unroll
            for (int k = 0; k < PROJ_K / 4; k++) s4[k * 64 + lane] = src[k * 64 + lane];
            WAVE_LDS_SYNC();
        }
        while (todo) {
            // ---- four points at a time, one per 16-lane row, when the next points have at most 16 candidates ----
            // Each row finds its best / second among the features that are free NOW; the rows are then accepted
            // in point order as long as no earlier row of the group has just closed a feature that a later row
            // also considered -- that row and the ones after it are simply taken up again in the next round.
            {
                const int gl = lane & 15, g = lane >> 4;
                int jq[4] = {0, 0, 0, 0}, cq[4] = {0, 0, 0, 0}, fq[4] = {0, 0, 0, 0}, nb = 0;
                unsigned long long m = todo;
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    if (m != 0 && nb == r) {
                        const int j = __builtin_ctzll(m);
                        const int c = __builtin_amdgcn_readlane(myc, j);
                        if (c <= 16 && c <= keff) {
                            jq[r] = j;
                            cq[r] = c;
                            fq[r] = __builtin_amdgcn_readlane(myflags, j);
                            nb = r + 1;
                            m &= m - 1;
                        }
                    }
                }
                if (nb >= 2) {
                    const int myj = g == 0 ? jq[0] : (g == 1 ? jq[1] : (g == 2 ? jq[2] : jq[3]));
                    const int mycq = g == 0 ? cq[0] : (g == 1 ? cq[1] : (g == 2 ? cq[2] : cq[3]));
                    const bool rowLive = g < nb;
                    const uint32_t t = (rowLive && gl < mycq) ? s_tup[myj * PROJ_K + gl] : 0u;
                    const int idx = (int)(t >> 13);
                    const bool ok = rowLive && gl < mycq && !((s_occ[idx >> 5] >> (idx & 31)) & 1u);
                    const int key = ok ? (int)(((t & 511u) << 6) | (uint32_t)gl) : 0x7FFFFFFF;
                    const int k1 = row_min_i(key);
                    const int l1 = k1 & 15;
                    const int k2 = row_min_i((gl == l1 || k1 == 0x7FFFFFFF) ? 0x7FFFFFFF : key);
                    const uint32_t t1 = (uint32_t)__shfl((int)t, (g << 4) + l1);
                    const uint32_t t2 = (uint32_t)__shfl((int)t, (g << 4) + (k2 & 15));
                    const int bDist = k1 == 0x7FFFFFFF ? 256 : (k1 >> 6), bLevel = (int)((t1 >> 9) & 15u), bIdx = (int)(t1 >> 13);
                    const int bDist2 = k2 == 0x7FFFFFFF ? 256 : (k2 >> 6), bLevel2 = k2 == 0x7FFFFFFF ? -1 : (int)((t2 >> 9) & 15u);
                    const bool match = rowLive && k1 != 0x7FFFFFFF && bDist <= th_high &&
                                       !(use_ratio && bLevel == bLevel2 && (float)bDist > __fmul_rn(nnratio, (float)bDist2));
                    // scalars of every row (row-uniform values read from the row's first lane)
                    int rMatch[4], rIdx[4];
#pragma unroll
                    for (int r = 0; r < 4; r++) {
                        rMatch[r] = __builtin_amdgcn_readlane((int)match, r * 16);
                        rIdx[r] = __builtin_amdgcn_readlane(bIdx, r * 16);
                    }
                    // first row whose candidates contain a feature closed by an earlier row of this group
                    int firstBad = nb;
#pragma unroll
                    for (int e = 0; e < 3; e++) {
                        const bool closes = rMatch[e] && (fq[e] & ORBHIP_Q_OBSERVED) && e < nb;
                        const unsigned long long hit = __ballot(closes && ok && g > e && idx == rIdx[e]);
                        if (hit) firstBad = min(firstBad, (int)(__builtin_ctzll(hit) >> 4));
                    }
                    // commit rows [0, firstBad)
#pragma unroll
                    for (int r = 0; r < 4; r++) {
                        if (r < firstBad) {
                            if (rMatch[r]) {
                                if (lane == r * 16) {
                                    s_match[rIdx[r]] = base + jq[r];
                                    if (fq[r] & ORBHIP_Q_OBSERVED) s_occ[rIdx[r] >> 5] |= 1u << (rIdx[r] & 31);
                                }
                                if (lane == jq[r]) myfeat = rIdx[r];
                                nm++;
                            }
                            todo &= ~(1ull << jq[r]);
                        }
                    }
                    WAVE_LDS_SYNC();
                    continue;
                }
            }
            // ---- one point with the whole wave (more than 16 candidates, or the only one left) ----
            const int j = __builtin_ctzll(todo);
            todo &= todo - 1;
            const int c = __builtin_amdgcn_readlane(myc, j);
            const int flags = __builtin_amdgcn_readlane(myflags, j);
            int bestDist = 256, bestLevel = -1, bestDist2 = 256, bestLevel2 = -1, bestIdx = -1;
            if (c <= keff) {
                const uint32_t t = lane < c ? s_tup[j * PROJ_K + lane] : 0u;
                const int idx = (int)(t >> 13);
                const bool ok = lane < c && !((s_occ[idx >> 5] >> (idx & 31)) & 1u);
                const int key = ok ? (int)(((t & 511u) << 6) | (uint32_t)lane) : 0x7FFFFFFF;
                const int k1 = wave_min_i(key);
                if (k1 != 0x7FFFFFFF) {
                    const int l1 = k1 & 63;
                    const int k2 = wave_min_i(lane == l1 ? 0x7FFFFFFF : key);
                    const uint32_t t1 = (uint32_t)__builtin_amdgcn_readlane((int)t, l1);
                    bestDist = k1 >> 6;
                    bestLevel = (int)((t1 >> 9) & 15u);
                    bestIdx = (int)(t1 >> 13);
                    if (k2 != 0x7FFFFFFF) {
                        const uint32_t t2 = (uint32_t)__builtin_amdgcn_readlane((int)t, k2 & 63);
                        bestDist2 = k2 >> 6;
                        bestLevel2 = (int)((t2 >> 9) & 15u);
                    }
                }
            } else {
                // more candidates than the list holds: the reference's scan, identically in every lane
                const orbhip_proj_query q = Q[base + j];
                const uint4 *qd = reinterpret_cast<const uint4 *>(qdesc + ((size_t)b * capQ + base + j) * 32);
                const uint4 a0 = qd[0], a1 = qd[1];
                walk_window(gp, q, K, O, I, [&](int idx, int oct) {
                    if ((s_occ[idx >> 5] >> (idx & 31)) & 1u) return;
                    if (UR) {
                        const float ur = UR[idx];
                        if (ur > 0 && fabsf(__fsub_rn(q.proj_xr, ur)) > q.radius) return;
                    }
                    const int d = hamming256g(a0, a1, D[2 * idx], D[2 * idx + 1]);
                    if (d < bestDist) {
                        bestDist2 = bestDist;
                        bestDist = d;
                        bestLevel2 = bestLevel;
                        bestLevel = oct;
                        bestIdx = idx;
                    } else if (d < bestDist2) {
                        bestLevel2 = oct;
                        bestDist2 = d;
                    }
                });
            }
            if (bestIdx >= 0 && bestDist <= th_high) {
                if (use_ratio && bestLevel == bestLevel2 && (float)bestDist > __fmul_rn(nnratio, (float)bestDist2)) continue;
                if (lane == 0) {
                    s_match[bestIdx] = base + j;
                    if (flags & ORBHIP_Q_OBSERVED) s_occ[bestIdx >> 5] |= 1u << (bestIdx & 31);
                }
                if (lane == j) myfeat = bestIdx;
                nm++;
                WAVE_LDS_SYNC();
            }
        }
        if (myq < NQ) qfeat[(size_t)b * capQpad + myq] = myfeat;
    }
    // rotation consistency (:1467-1494): bins of the accepted matches, the three maxima, removal
    if (!use_ratio && check_ori) {
        __threadfence_block();
        WAVE_LDS_SYNC();
        for (int iq = lane; iq < NQ; iq += 64) {
            const int f = qfeat[(size_t)b * capQpad + iq];
            if (f < 0) continue;
            float rot = __fsub_rn(Q[iq].angle, K[f].angle);
            if (rot < 0.0f) rot = __fadd_rn(rot, 360.0f);
            int bin = (int)roundf(__fmul_rn(rot, 1.0f / 30));
            if (bin == 30) bin = 0;
            if (bin >= 0 && bin < 30) atomicAdd(&s_hist[bin], 1);
        }
        WAVE_LDS_SYNC();
        if (lane == 0) {
            int max1 = 0, max2 = 0, max3 = 0, i1 = -1, i2 = -1, i3 = -1;
            for (int i = 0; i < 30; i++) {
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
        WAVE_LDS_SYNC();
        int removed = 0;
        for (int iq = lane; iq < NQ; iq += 64) {
            const int f = qfeat[(size_t)b * capQpad + iq];
            if (f < 0) continue;
            float rot = __fsub_rn(Q[iq].angle, K[f].angle);
            if (rot < 0.0f) rot = __fadd_rn(rot, 360.0f);
            int bin = (int)roundf(__fmul_rn(rot, 1.0f / 30));
            if (bin == 30) bin = 0;
            if (bin >= 0 && bin < 30 && bin != s_keep[0] && bin != s_keep[1] && bin != s_keep[2]) {
                s_match[f] = -2;   // assigned, then removed: the reference stores NULL here (:1489)
                removed++;
            }
        }
        nm -= wave_sum_g(removed);
        WAVE_LDS_SYNC();
    }
    for (int i = lane; i < cap; i += 64) match[(size_t)b * cap + i] = s_match[i];
    if (lane == 0) nmatches[b] = nm;
}

// ---- the same assignment for ONE frame per call (a frame or two per launch): parallel fixed point ----------------------
// k_proj_assign walks the points of a frame in order in one wave: ~0.28 us per point, 281 us for the 1000 points of a
// SearchByProjection(CurrentFrame, LastFrame) -- fine beside 511 other frames, slower than a host core when it is the only
// frame.  The order only matters through the features that EARLIER points closed (a point whose MapPoint has observations
// keeps its feature, :1411-1413).  So every point first picks its best / second as if nothing were closed, in parallel (one
// point per 16-lane row, 64 points per trip of a 1024-thread workgroup); then, round after round, close[f] = the first point
// that holds feature f (observed points only) and every point picks again among the features with close[f] >= its own index.
// When a round changes nothing the state is the sequential result: the first point whose pick could be wrong has only
// correct points before it, and then its pick is the sequential one by construction -- so each round fixes at least the first
// wrong point, and a fixed point has none.  Conflicts are rare (a few per cent of the points), chains short: 2-4 rounds.
// Frames with a point of more than 32 candidates, or without a fixed point after PAR_ROUNDS rounds, are left to
// k_proj_assign (fallback[b] = 1); the tail (last writer per feature, rotation histogram) is the same.
#define PAR_ROUNDS 48
__global__ __launch_bounds__(1024) void k_proj_assign_par(const orbhip_keypoint *__restrict__ kps, const int32_t *__restrict__ cnt,
                                                          int cap, const uint8_t *__restrict__ occupied,
                                                          const orbhip_proj_query *__restrict__ queries,
                                                          const int32_t *__restrict__ nq, int capQ, int capQpad, int keff,
                                                          const uint32_t *__restrict__ tuples, const int32_t *__restrict__ tcount,
                                                          int32_t *__restrict__ qfeat, int use_ratio, float nnratio, int check_ori,
                                                          int th_high, int32_t *__restrict__ match, int32_t *__restrict__ nmatches,
                                                          int32_t *__restrict__ fallback, int maxRounds)
{
    extern __shared__ uint32_t s_dyn[];
    __shared__ int s_hist[30];
    __shared__ int s_keep[3];
    __shared__ int s_flag[3];   // changed | leave the frame to k_proj_assign | accepted matches
    const int b = blockIdx.x, tid = threadIdx.x, gl = tid & 15, row = tid >> 4, nth = blockDim.x;
    const int n = min(cnt[b], cap), NQ = min(nq[b], capQ);
    int32_t *s_match = reinterpret_cast<int32_t *>(s_dyn);      // [cap]   last point that took the feature
    int32_t *s_close = s_match + cap;                           // [cap]   first observed point holding the feature
    int32_t *s_feat = s_close + cap;                            // [capQpad] feature picked by the point, -1 = none
    uint32_t *s_occ = reinterpret_cast<uint32_t *>(s_feat + capQpad);   // [(cap + 31) / 32] features closed on entry
    const orbhip_keypoint *K = kps + (size_t)b * cap;
    const orbhip_proj_query *Q = queries + (size_t)b * capQ;
    const uint32_t *T = tuples + (size_t)b * capQpad * PROJ_K;
    const int32_t *TC = tcount + (size_t)b * capQpad;
    for (int i = tid; i < cap; i += nth) {
        s_match[i] = -1;
        s_close[i] = 0x7FFFFFFF;
    }
    for (int i = tid; i < capQpad; i += nth) s_feat[i] = -1;
    for (int w = tid; w < (cap + 31) / 32; w += nth) {
        uint32_t bits = 0;
        if (occupied)
            for (int k = 0; k < 32; k++) {
                const int i = w * 32 + k;
                if (i < n && occupied[(size_t)b * cap + i]) bits |= 1u << k;
            }
        s_occ[w] = bits;
    }
    if (tid < 30) s_hist[tid] = 0;
    if (tid < 3) s_flag[tid] = 0;
    __syncthreads();
    for (int q = tid; q < NQ; q += nth)
        if (TC[q] > min(keff, 32)) s_flag[1] = 1;
    __syncthreads();
    bool done = false;
    for (int round = 0; round < maxRounds && !s_flag[1]; round++) {
        bool changed = false;
        for (int q0 = 0; q0 < NQ; q0 += (nth >> 4)) {
            const int q = q0 + row;
            const bool live = q < NQ;
            const int c = live ? TC[q] : 0;
            const uint32_t t0 = gl < c ? T[(size_t)q * PROJ_K + gl] : 0u, t1 = gl + 16 < c ? T[(size_t)q * PROJ_K + gl + 16] : 0u;
            const int i0 = (int)(t0 >> 13), i1 = (int)(t1 >> 13);
            const bool ok0 = gl < c && !((s_occ[i0 >> 5] >> (i0 & 31)) & 1u) && s_close[i0] >= q;
            const bool ok1 = gl + 16 < c && !((s_occ[i1 >> 5] >> (i1 & 31)) & 1u) && s_close[i1] >= q;
            // (distance, position in the reference's visiting order): the first feature of smallest distance wins
            const int key0 = ok0 ? (int)(((t0 & 511u) << 6) | (uint32_t)gl) : 0x7FFFFFFF;
            const int key1 = ok1 ? (int)(((t1 & 511u) << 6) | (uint32_t)(gl + 16)) : 0x7FFFFFFF;
            const int mine = min(key0, key1), other = max(key0, key1);
            const int k1 = row_min_i(mine);
            const int k2 = row_min_i(mine == k1 ? other : mine);       // keys are unique (position bits) unless both are "none"
            const int p1 = k1 & 31, p2 = k2 & 31;
            const uint32_t w1 = (uint32_t)__shfl((int)(p1 >= 16 ? t1 : t0), (tid & 48) + (p1 & 15));
            const uint32_t w2 = (uint32_t)__shfl((int)(p2 >= 16 ? t1 : t0), (tid & 48) + (p2 & 15));
            const int bDist = k1 == 0x7FFFFFFF ? 256 : (k1 >> 6), bLevel = (int)((w1 >> 9) & 15u), bIdx = (int)(w1 >> 13);
            const int bDist2 = k2 == 0x7FFFFFFF ? 256 : (k2 >> 6), bLevel2 = k2 == 0x7FFFFFFF ? -1 : (int)((w2 >> 9) & 15u);
            const bool acc = live && k1 != 0x7FFFFFFF && bDist <= th_high &&
                             !(use_ratio && bLevel == bLevel2 && (float)bDist > __fmul_rn(nnratio, (float)bDist2));
            const int f = acc ? bIdx : -1;
            if (live && gl == 0 && s_feat[q] != f) {
                s_feat[q] = f;
                changed = true;
            }
        }
        if (changed) s_flag[0] = 1;
        __syncthreads();
        const bool any = s_flag[0] != 0;
        __syncthreads();
        if (!any) {
            done = true;
            break;
        }
        if (tid == 0) s_flag[0] = 0;
        for (int i = tid; i < cap; i += nth) s_close[i] = 0x7FFFFFFF;
        __syncthreads();
        for (int q = tid; q < NQ; q += nth) {
            const int f = s_feat[q];
            if (f >= 0 && (Q[q].flags & ORBHIP_Q_OBSERVED)) atomicMin(&s_close[f], q);
        }
        __syncthreads();
    }
    if (!done) {   // a point with more candidates than a row holds, or no fixed point yet: the sequential kernel does this frame
        if (tid == 0) fallback[b] = 1;
        return;
    }
    if (tid == 0) fallback[b] = 0;
    // the last point that took a feature is the one the reference's vector ends up with; every accepted pick counts (:1461-1463)
    int mine = 0;
    for (int q = tid; q < capQpad; q += nth) {
        const int f = q < NQ ? s_feat[q] : -1;
        if (f >= 0) {
            atomicMax(&s_match[f], q);
            mine++;
        }
        qfeat[(size_t)b * capQpad + q] = f;
    }
    if (mine) atomicAdd(&s_flag[2], mine);
    __syncthreads();
    // rotation consistency (:1467-1494): bins of the accepted matches, the three maxima, removal
    if (!use_ratio && check_ori) {
        for (int iq = tid; iq < NQ; iq += nth) {
            const int f = s_feat[iq];
            if (f < 0) continue;
            float rot = __fsub_rn(Q[iq].angle, K[f].angle);
            if (rot < 0.0f) rot = __fadd_rn(rot, 360.0f);
            int bin = (int)roundf(__fmul_rn(rot, 1.0f / 30));
            if (bin == 30) bin = 0;
            if (bin >= 0 && bin < 30) atomicAdd(&s_hist[bin], 1);
        }
        __syncthreads();
        if (tid == 0) {
            int max1 = 0, max2 = 0, max3 = 0, i1 = -1, i2 = -1, i3 = -1;
            for (int i = 0; i < 30; i++) {
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
        int removed = 0;
        for (int iq = tid; iq < NQ; iq += nth) {
            const int f = s_feat[iq];
            if (f < 0) continue;
            float rot = __fsub_rn(Q[iq].angle, K[f].angle);
            if (rot < 0.0f) rot = __fadd_rn(rot, 360.0f);
            int bin = (int)roundf(__fmul_rn(rot, 1.0f / 30));
            if (bin == 30) bin = 0;
            if (bin >= 0 && bin < 30 && bin != s_keep[0] && bin != s_keep[1] && bin != s_keep[2]) {
                s_match[f] = -2;   // assigned, then removed: the reference stores NULL here (:1489)
                removed++;
            }
        }
        if (removed) atomicSub(&s_flag[2], removed);
        __syncthreads();
    }
    for (int i = tid; i < cap; i += nth) match[(size_t)b * cap + i] = s_match[i];
    if (tid == 0) nmatches[b] = s_flag[2];
}

// ---- ORBmatcher::SearchForInitialization (ref: src/ORBmatcher.cc:405-520) ----------------------------------
//   k_init_cands   one wave per level-0 feature of frame 1: the window of GetFeaturesInArea(prev_matched[i1], windowSize,
//                  0, 0) over frame 2's records, 64 at a time; the features that pass keep the reference's order through
//                  a ballot prefix count; tuple = distance << 23 | position << 16 | index; lists of up to 64 leave sorted (r05).
//   k_init_assign  one wave per frame pair, features of frame 1 in index order: a candidate is skipped when frame 2's
//                  feature is already matched at a distance <= this one (vMatchedDistance, LDS), best / second are two
//                  minima over (distance, list position), an accepted match displaces the earlier owner.  Then the rotation
//                  histogram over every accepted i1 (displaced ones included, as in the reference) and the removal.
#define INIT_K 128   // candidate slots per feature (64 lists = 32 KB of LDS)

__global__ __launch_bounds__(256) void k_init_cands(const orbhip_keypoint *__restrict__ kps1,
                                                    const uint8_t *__restrict__ desc1, const int32_t *__restrict__ cnt1,
                                                    int cap1, const float2 *__restrict__ prev, float radius,
                                                    const GridParams gp, const int32_t *__restrict__ cellOff2,
                                                    const float4 *__restrict__ rec2, const uint8_t *__restrict__ desc2,
                                                    int cap2, int cap1pad, int keff, uint32_t *__restrict__ tuples,
                                                    int32_t *__restrict__ tcount)
{
    __shared__ int s_pre[4][65];
    __shared__ int s_beg[4][64];
    __shared__ uint32_t s_lst[4][INIT_K];   // the feature's tuples in visiting order, before they are (sorted and) stored
    const int b = blockIdx.y, i1 = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (i1 >= cap1pad) return;
    int count = 0;
    const int n1 = min(cnt1[b], cap1);
    if (i1 < n1) {
        const int level1 = kps1[(size_t)b * cap1 + i1].octave;
        const float2 pm = prev[(size_t)b * cap1 + i1];
        int x0, x1, y0, y1;
        if (level1 <= 0 && window_cells(gp, pm.x, pm.y, radius, x0, x1, y0, y1)) {   // :420-421
            const uint4 *qd = reinterpret_cast<const uint4 *>(desc1 + ((size_t)b * cap1 + i1) * 32);
            const uint4 a0 = qd[0], a1 = qd[1];
            const uint4 *D = reinterpret_cast<const uint4 *>(desc2 + (size_t)b * cap2 * 32);
            const float4 *R = rec2 + (size_t)b * cap2;
            const int32_t *O = cellOff2 + (size_t)b * (GCELLS + 1);
            // a round's 64 lists (features i1 & ~63 ...) are stored TRANSPOSED: entry p of list l at [p * 64 + l] -- in k_init_assign a
            // lane walks its own list, and the round's rows [0, longest list) are one contiguous piece
            uint32_t *T = tuples + ((size_t)b * cap1pad + (i1 & ~63)) * INIT_K + (i1 & 63);
            // The window's cell columns are contiguous record ranges; the wave walks their concatenation 64 records at
            // a time (lane c holds column c's range, a prefix sum over the lengths maps a flat position back to its
            // column), which is the visiting order of GetFeaturesInArea.
            const int wv = threadIdx.x >> 6, ncols = x1 - x0 + 1;
            int cs = 0, ce = 0;
            if (lane < ncols) {
                cs = O[(x0 + lane) * GROWS + y0];
                ce = O[(x0 + lane) * GROWS + y1 + 1];
            }
            int incl = ce - cs;
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) {
                const int v = __shfl_up(incl, d);
                if (lane >= d) incl += v;
            }
            s_pre[wv][lane + 1] = incl;
            if (lane == 0) s_pre[wv][0] = 0;
            s_beg[wv][lane] = cs;
            const int total = __builtin_amdgcn_readlane(incl, 63);
            WAVE_LDS_SYNC();
            for (int t0 = 0; t0 < total; t0 += 64) {
                const int t = min(t0 + lane, total - 1);
                int c = 0;
#pragma unroll
                for (int step = 32; step >= 1; step >>= 1)
                    if (c + step < ncols && s_pre[wv][c + step] <= t) c += step;
                const float4 r = R[s_beg[wv][c] + (t - s_pre[wv][c])];
                const int w = __float_as_int(r.z), oct = w & 255;
                const bool pass = t0 + lane < total && !(oct < level1) && !(level1 >= 0 && oct > level1) &&
                                  fabsf(__fsub_rn(r.x, pm.x)) < radius && fabsf(__fsub_rn(r.y, pm.y)) < radius;
                const unsigned long long m = __ballot(pass);
                const int pos = count + __popcll(m & ((1ull << lane) - 1ull));
                if (pass && pos < keff) {
                    const int idx = w >> 8;
                    // tuple = distance << 23 | position in visiting order << 16 | feature of frame 2 (< 2^16): as an integer it
                    // orders by distance, then by position -- the reference's strict '<' scan takes the first of equal distances
                    s_lst[wv][pos] = ((uint32_t)hamming256g(a0, a1, D[2 * idx], D[2 * idx + 1]) << 23) | ((uint32_t)pos << 16) | (uint32_t)idx;
                }
                count += __popcll(m);
            }
            WAVE_LDS_SYNC();
            const int stored = min(count, keff);
            if (count <= 64 && count <= keff) {
                // r05: a list of up to 64 candidates leaves SORTED (bitonic network over the wave): k_init_assign then finds best and
                // second as the first two entries that the state of vMatchedDistance does not skip -- one ballot, no reduction
                uint32_t v = lane < count ? s_lst[wv][lane] : 0xFFFFFFFFu;
#pragma unroll
                for (int k = 2; k <= 64; k <<= 1)
#pragma unroll
                    for (int j = k >> 1; j > 0; j >>= 1) {
                        const uint32_t o = (uint32_t)__shfl_xor((int)v, j);
                        const bool up = (lane & k) == 0, lower = (lane & j) == 0;
                        v = (lower == up) ? min(v, o) : max(v, o);
                    }
                if (lane < count) T[lane * 64] = v;
            } else {
                for (int p = lane; p < stored; p += 64) T[p * 64] = s_lst[wv][p];
            }
        }
    }
    if (lane == 0) tcount[(size_t)b * cap1pad + i1] = count;
}

#define INIT_NT 256   // k_init_assign: wave 0 walks the features, all four waves stage, prefetch the lists, check the rotation, write
#define INIT_ROUND (64 * INIT_K)   // words of one round's 64 lists

__global__ __launch_bounds__(INIT_NT) void k_init_assign(const orbhip_keypoint *__restrict__ kps1,
                                                         const uint8_t *__restrict__ desc1, const int32_t *__restrict__ cnt1,
                                                         int cap1, const orbhip_keypoint *__restrict__ kps2,
                                                         const uint8_t *__restrict__ desc2, const int32_t *__restrict__ cnt2,
                                                         int cap2, float2 *__restrict__ prev, float radius, const GridParams gp,
                                                         const int32_t *__restrict__ cellOff2,
                                                         const int32_t *__restrict__ cellIdx2, int cap1pad, int keff,
                                                         const uint32_t *__restrict__ tuples, const int32_t *__restrict__ tcount,
                                                         float nnratio, int check_ori, int th_low,
                                                         int32_t *__restrict__ matches12, int32_t *__restrict__ nmatches, int mode)
{
    extern __shared__ uint32_t s_dyn[];
    __shared__ int s_hist[30];
    __shared__ int s_keep[3];
    __shared__ int s_rm;
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const bool staged = mode & 1;   // the tables below fit the LDS beside the match tables (frames of up to ~3000 features)
    const bool dbl = mode & 2;      // ... and so does a second buffer for the lists: the next round's arrive while this one is walked
    int n1 = min(cnt1[b], cap1);   // (features of frame 2 beyond cnt2 are not in its grid)
    uint32_t *s_tup = s_dyn;                                       // the lists of a round, one or two buffers
    int *s_md = reinterpret_cast<int *>(s_dyn + (dbl ? 2 : 1) * INIT_ROUND);   // vMatchedDistance [cap2]
    int *s_m21 = s_md + cap2;                     // vnMatches21 [cap2]
    int *s_m12 = s_m21 + cap2;                    // vnMatches12 [cap1]
    int *s_acc = s_m12 + cap1;                    // feature of frame 2 at the time i1 was accepted, or -1 [cap1]
    int *s_stamp = s_acc + cap1;                  // who accepted a feature of frame 2 in the current trip of the walk [cap2]
    // What the walk would otherwise fetch one dependent round trip at a time (r04 trace: 106 us per call, most of it ~60 such
    // trips of ~1.5 us -- the list lengths per round of 64 features, the two angles per accepted feature in both passes of the
    // rotation check, the matched keypoint's position in the last loop) is read once, coalesced, with the loads in flight together:
    int *s_tc = s_stamp + cap2;                                       // list lengths [cap1pad]
    float *s_a1 = reinterpret_cast<float *>(s_tc + cap1pad);          // angles of frame 1 [cap1]
    float *s_k2 = s_a1 + cap1;                                        // x, y, angle of frame 2 [3 * cap2]
    const orbhip_keypoint *K1 = kps1 + (size_t)b * cap1, *K2 = kps2 + (size_t)b * cap2;
    const uint4 *D2 = reinterpret_cast<const uint4 *>(desc2 + (size_t)b * cap2 * 32);
    const int32_t *O = cellOff2 + (size_t)b * (GCELLS + 1), *I = cellIdx2 + (size_t)b * cap2;
    float2 *PM = prev + (size_t)b * cap1;
    const int32_t *TC = tcount + (size_t)b * cap1pad;
    const uint4 *Tg = reinterpret_cast<const uint4 *>(tuples + (size_t)b * cap1pad * INIT_K);
    // rows of a round that hold entries: the longest stored list of its 64 features (every wave computes the same number)
    auto round_rows = [&](int base) -> int {
        const int q = base + lane;
        const int c = q < n1 ? (staged ? s_tc[q] : TC[q]) : 0;
        return -wave_min_i(-min(c, keff));
    };
    // a round's 64 lists, transposed (entry p of list l at [p * 64 + l]): rows [0, rows) are contiguous (a window of 100 pixels
    // holds ~30 level-0 features: ~10 of the 32 KB).  Thread t of nt; eight loads in flight and no branch around any of them
    // (a guarded load is compiled as load, wait, next load): what lies beyond the last piece reads and writes the last piece again
    auto load_round = [&](int base, uint32_t *buf, int rows, int t, int nt) {
        uint4 *s4 = reinterpret_cast<uint4 *>(buf);
        const uint4 *src = Tg + (size_t)base * (INIT_K / 4);
        const int nq = rows * 16;
        for (int k0 = 0; k0 < nq; k0 += nt * 8) {
            uint4 v[8];
#pragma unroll
            for (int u = 0; u < 8; u++) v[u] = src[min(k0 + nt * u + t, nq - 1)];
#pragma unroll
            for (int u = 0; u < 8; u++) s4[min(k0 + nt * u + t, nq - 1)] = v[u];
        }
    };
    if (staged) {
        // one trip for frames of up to 1024 features: the five loads of a thread's four positions are issued before the first store
        const int n2 = min(cnt2[b], cap2), n2c = max(n2 - 1, 0);
        constexpr int U = 4;
        for (int i0 = 0; i0 < max(cap1pad, n2); i0 += INIT_NT * U) {
            int v[U];
            float w[U], x[U], y[U], a[U];
#pragma unroll
            for (int u = 0; u < U; u++) {
                const int i = i0 + INIT_NT * u + tid, i2 = min(i, n2c);
                v[u] = i < n1 ? TC[i] : 0;
                w[u] = i < n1 ? K1[i].angle : 0.f;
                x[u] = K2[i2].x;
                y[u] = K2[i2].y;
                a[u] = K2[i2].angle;
            }
#pragma unroll
            for (int u = 0; u < U; u++) {
                const int i = i0 + INIT_NT * u + tid;
                if (i < cap1pad) s_tc[i] = v[u];
                if (i < n1) s_a1[i] = w[u];
                if (i < n2) {
                    s_k2[3 * i] = x[u];
                    s_k2[3 * i + 1] = y[u];
                    s_k2[3 * i + 2] = a[u];
                }
            }
        }
    }
    for (int i = tid; i < cap2; i += INIT_NT) {
        s_md[i] = 0x7FFFFFFF;
        s_m21[i] = -1;
        s_stamp[i] = 0;
    }
    for (int i = tid; i < cap1; i += INIT_NT) {
        s_m12[i] = -1;
        s_acc[i] = -1;
    }
    if (tid < 30) s_hist[tid] = 0;
    if (tid == 0) s_rm = 0;
    __syncthreads();
    const int rows0 = n1 > 0 ? round_rows(0) : 0;
    load_round(0, s_tup, rows0, tid, INIT_NT);
    __syncthreads();
    const int stop = mode >> 4;            (void)stop;   // timing ablation only (liborbhip_ablation.so, ORBHIP_INIT_STOP); 0 in the shipped library
    ORB_ABL_IF(stop == 1) return;           // tables staged
    ORB_ABL_IF(stop == 3) n1 = min(n1, 64); // one round of the feature loop
    int nm = 0, trip = 0;
    const int limS = min(64, keff);   // lists of up to limS candidates arrive sorted by (distance, position)
    // the last feature with a candidate (every wave computes it): only level-0 features have any, and the extractor puts them
    // first -- three quarters of the rounds would be a barrier and nothing else
    int lastq = -1;
    for (int i = lane; i < n1; i += 64) lastq = (staged ? s_tc[i] : TC[i]) > 0 ? i : lastq;
    lastq = -wave_min_i(-lastq);
    const int nWalk = min(n1, lastq + 1);
    int rows = rows0;
    for (int base = 0, r = 0; base < nWalk; base += 64, r++) {
        const uint32_t *tupb = s_tup + (dbl ? (r & 1) * INIT_ROUND : 0);
        const int rowsNext = base + 64 < nWalk ? round_rows(base + 64) : 0;
        if (wv != 0) {
            if (dbl) load_round(base + 64, s_tup + ((r + 1) & 1) * INIT_ROUND, rowsNext, tid - 64, INIT_NT - 64);
        } else if (rows > 0) {
            const int myq = base + lane;
            const int myc = myq < n1 ? (staged ? s_tc[myq] : TC[myq]) : 0;
            const unsigned long long todo = __ballot(myc > 0);
            auto accept = [&](int i1, int bestIdx, int bestDist, int bestDist2) -> bool {
                if (!(bestIdx >= 0 && bestDist <= th_low && (float)bestDist < __fmul_rn((float)bestDist2, nnratio))) return false;   // :458-460
                const int old = s_m21[bestIdx];
                if (old >= 0) nm--;
                if (lane == 0) {
                    if (old >= 0) s_m12[old] = -1;
                    s_m12[i1] = bestIdx;
                    s_m21[bestIdx] = i1;
                    s_md[bestIdx] = bestDist;
                    s_acc[i1] = bestIdx;
                }
                nm++;
                WAVE_LDS_SYNC();
                return true;
            };
            // The reference walks the features of frame 1 in index order, and vMatchedDistance makes a feature depend on every earlier
            // one that was accepted with one of its candidates.  Through r05's first half one wave walked them one at a time: ~100
            // instructions and three dependent LDS round trips per feature (0.4 us; written without a single branch it took the same
            // time).  Now a LANE walks its own feature and the wave keeps the order: in one trip every undecided feature of the round
            // takes best / second as the first two entries of its sorted list that the state BEFORE the trip does not skip (held from
            // trip to trip, below), and a
            // feature that would be accepted stamps its best candidate (trip, lowest such lane: one atomicMax).  vMatchedDistance only
            // ever decreases, so a feature's decision stands unless an earlier feature of the same trip stamped one of the entries
            // it looked at: the features before the first such one (the first undecided one never is) are committed together -- their
            // best candidates are distinct, their displaced owners were accepted before the trip -- and the rest decide again in the
            // next trip.  Features with a longer list (unsorted, or longer than the table: the rescan) take a trip of their own over the
            // whole wave, as before.
            const bool mine = myc > 0 && myc <= limS;
            // what a lane knows of its list: the first two entries not skipped so far (h1, h2; held = how many), and where its scan
            // stands (next).  An entry once skipped stays skipped -- vMatchedDistance only decreases -- so a trip only looks at
            // h1 and h2 again and scans on from `next` when one of them has gone: every entry is read once per round, not once per trip
            uint32_t h1 = 0, h2 = 0;
            int held = 0, next = 0;
            unsigned long long rem = todo;
            while (rem) {
                const int j = (int)__builtin_ctzll(rem);
                const int c = __builtin_amdgcn_readlane(myc, j);
                if (c > limS) {
                    const int i1 = base + j;
                    int bestDist = 0x7FFFFFFF, bestDist2 = 0x7FFFFFFF, bestIdx = -1;
                    if (c <= keff) {
                        int m1 = 0x7FFFFFFF, m2 = 0x7FFFFFFF;
                        for (int p = lane; p < c; p += 64) {
                            const uint32_t t = tupb[p * 64 + j];         // (in visiting order, position = slot)
                            const int d = (int)(t >> 23), idx = (int)(t & 0xFFFFu);
                            const int key = s_md[idx] <= d ? 0x7FFFFFFF : (int)(t >> 16);   // :443-444; distance << 7 | position
                            if (key < m1) {
                                m2 = m1;
                                m1 = key;
                            } else if (key < m2) {
                                m2 = key;
                            }
                        }
                        const int k1 = wave_min_i(m1);
                        if (k1 != 0x7FFFFFFF) {
                            const int k2 = wave_min_i(m1 == k1 ? m2 : m1);   // list positions are unique: one lane holds k1
                            bestDist = k1 >> 7;
                            bestIdx = (int)(tupb[(k1 & 127) * 64 + j] & 0xFFFFu);
                            if (k2 != 0x7FFFFFFF) bestDist2 = k2 >> 7;
                        }
                    } else {
                        // more candidates than the list holds: the reference's scan, identically in every lane
                        orbhip_proj_query q;
                        const float2 pm = PM[i1];
                        q.u = pm.x;
                        q.v = pm.y;
                        q.radius = radius;
                        q.min_level = q.max_level = K1[i1].octave;
                        const uint4 *qd = reinterpret_cast<const uint4 *>(desc1 + ((size_t)b * cap1 + i1) * 32);
                        const uint4 a0 = qd[0], a1 = qd[1];
                        walk_window(gp, q, K2, O, I, [&](int idx, int) {
                            const int d = hamming256g(a0, a1, D2[2 * idx], D2[2 * idx + 1]);
                            if (s_md[idx] <= d) return;
                            if (d < bestDist) {
                                bestDist2 = bestDist;
                                bestDist = d;
                                bestIdx = idx;
                            } else if (d < bestDist2) {
                                bestDist2 = d;
                            }
                        });
                    }
                    (void)accept(i1, bestIdx, bestDist, bestDist2);
                    rem &= rem - 1ull;
                    continue;
                }
                trip++;
                const bool undecided = (rem >> lane) & 1ull;
                const bool act = mine && undecided;
                {
                    // (no branch around the reads: a lane without an entry reads feature 0 and does not use it)
                    const int m1 = s_md[h1 & 0xFFFFu], m2 = s_md[h2 & 0xFFFFu];
                    const bool k1 = held >= 1 && !(m1 <= (int)(h1 >> 23)), k2 = held >= 2 && !(m2 <= (int)(h2 >> 23));   // :443-444
                    h1 = k1 ? h1 : h2;
                    held = (k1 ? 1 : 0) + (k2 ? 1 : 0);
                    // scan on, four entries at a time (their reads in flight together); a lane stops behind its second entry
                    while (__ballot(act && held < 2 && next < myc)) {
                        uint32_t t4[4];
                        int md[4];
    #pragma unroll
                        for (int k = 0; k < 4; k++) t4[k] = next + k < myc ? tupb[(next + k) * 64 + lane] : 0u;
    #pragma unroll
                        for (int k = 0; k < 4; k++) md[k] = s_md[t4[k] & 0xFFFFu];
                        const int from = next;
    #pragma unroll
                        for (int k = 0; k < 4; k++) {
                            const bool there = act && held < 2 && from + k < myc;
                            const bool take = there && !(md[k] <= (int)(t4[k] >> 23));
                            next = there ? from + k + 1 : next;
                            h2 = take && held == 1 ? t4[k] : h2;
                            h1 = take && held == 0 ? t4[k] : h1;
                            held += take ? 1 : 0;
                        }
                    }
                }
                const int bD = held >= 1 ? (int)(h1 >> 23) : 0x7FFFFFFF, bI = (int)(h1 & 0xFFFFu);
                const int bD2 = held >= 2 ? (int)(h2 >> 23) : 0x7FFFFFFF;
                const bool ok = act && held >= 1 && bD <= th_low && (float)bD < __fmul_rn((float)bD2, nnratio);   // :458-460
                if (ok) atomicMax(&s_stamp[bI], trip * 64 + 63 - lane);
                WAVE_LDS_SYNC();
                // the decision rests on h1 and h2 alone: what lies before them is skipped for good, what lies behind them only
                // counts once one of them goes
                const int st1 = s_stamp[h1 & 0xFFFFu], st2 = s_stamp[h2 & 0xFFFFu];
                const int old = s_m21[bI];
                bool dirty = undecided && !mine;   // a long list: decided in a trip of its own, when it is the first undecided one
                dirty |= act & (held >= 1) & ((st1 >> 6) == trip) & (63 - (st1 & 63) < lane);
                dirty |= act & (held >= 2) & ((st2 >> 6) == trip) & (63 - (st2 & 63) < lane);
                const unsigned long long dm = __ballot(dirty);
                const unsigned long long commit = dm ? rem & ((1ull << __builtin_ctzll(dm)) - 1ull) : rem;
                const bool w = ok && ((commit >> lane) & 1ull);
                if (w) {
                    if (old >= 0) s_m12[old] = -1;   // the feature's previous owner loses it (:462-466)
                    s_m12[myq] = bI;
                    s_m21[bI] = myq;
                    s_md[bI] = bD;
                    s_acc[myq] = bI;
                }
                nm += __popcll(__ballot(w)) - __popcll(__ballot(w && old >= 0));
                rem &= ~commit;
                WAVE_LDS_SYNC();
            }
        }
        __syncthreads();
        if (!dbl) {
            load_round(base + 64, s_tup, rowsNext, tid, INIT_NT);
            __syncthreads();
        }
        rows = rowsNext;
    }
    ORB_ABL_IF(stop == 2) return;
    if (check_ori) {
        for (int i1 = tid; i1 < n1; i1 += INIT_NT) {
            const int f = s_acc[i1];
            if (f < 0) continue;
            float rot = staged ? __fsub_rn(s_a1[i1], s_k2[3 * f + 2]) : __fsub_rn(K1[i1].angle, K2[f].angle);
            if (rot < 0.0f) rot = __fadd_rn(rot, 360.0f);
            int bin = (int)roundf(__fmul_rn(rot, 1.0f / 30));
            if (bin == 30) bin = 0;
            if (bin >= 0 && bin < 30) atomicAdd(&s_hist[bin], 1);
        }
        __syncthreads();
        if (tid == 0) {
            int max1 = 0, max2 = 0, max3 = 0, i1 = -1, i2 = -1, i3 = -1;
            for (int i = 0; i < 30; i++) {
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
        int removed = 0;
        for (int i1 = tid; i1 < n1; i1 += INIT_NT) {
            const int f = s_acc[i1];
            if (f < 0) continue;
            float rot = staged ? __fsub_rn(s_a1[i1], s_k2[3 * f + 2]) : __fsub_rn(K1[i1].angle, K2[f].angle);
            if (rot < 0.0f) rot = __fadd_rn(rot, 360.0f);
            int bin = (int)roundf(__fmul_rn(rot, 1.0f / 30));
            if (bin == 30) bin = 0;
            if (bin >= 0 && bin < 30 && bin != s_keep[0] && bin != s_keep[1] && bin != s_keep[2] && s_m12[i1] >= 0) {
                s_m12[i1] = -1;
                removed++;
            }
        }
        if (removed) atomicAdd(&s_rm, removed);
        __syncthreads();
    }
    for (int i1 = tid; i1 < cap1; i1 += INIT_NT) {
        const int m = i1 < n1 ? s_m12[i1] : -1;
        matches12[(size_t)b * cap1 + i1] = m;
        if (m >= 0) PM[i1] = staged ? make_float2(s_k2[3 * m], s_k2[3 * m + 1]) : make_float2(K2[m].x, K2[m].y);   // :512-515
    }
    ORB_ABL_IF(stop == 4) nm = trip + s_rm;   // (ablation only: the walk's trips instead of the match count)
    if (tid == 0) nmatches[b] = nm - s_rm;
}

static int init_keff()
{
    static int v = -1;
    if (v < 0) {
        v = ORB_TUNE("INIT_K", INIT_K);   // tests force the rescan path with a small value
        if (v < 1) v = 1;
        if (v > INIT_K) v = INIT_K;
    }
    return v;
}

size_t init_scratch_bytes(int B, int cap1, int cap2)
{
    const size_t cap1pad = ((size_t)cap1 + 63) / 64 * 64;
    return (size_t)B * cap2 * 16 + (size_t)B * cap1pad * (INIT_K * 4 + 4) + 256;
}

size_t init_assign_lds(int cap1, int cap2) { return ((size_t)cap1 * 2 + (size_t)cap2 * 3) * 4; }   // the match tables (required)
// ... plus list lengths [cap1pad], angles of frame 1 [cap1] and x, y, angle of frame 2 [3 cap2], staged when they fit as well
static size_t init_assign_lds_staged(int cap1, int cap2)
{
    const size_t cap1pad = ((size_t)cap1 + 63) / 64 * 64;
    return init_assign_lds(cap1, cap2) + (cap1pad + (size_t)cap1 + 3 * (size_t)cap2) * 4;
}

int launch_search_for_initialization(hipStream_t s, const orbhip_keypoint *kps1, const uint8_t *desc1, const int32_t *cnt1,
                                     int cap1, const orbhip_keypoint *kps2, const uint8_t *desc2, const int32_t *cnt2, int cap2,
                                     int B, float minX, float minY, float invW, float invH, const int32_t *cellOff2,
                                     const int32_t *cellIdx2, float *prev, int windowSize, float nnratio, int check_ori,
                                     int th_low, int32_t *matches12, int32_t *nmatches, void *scratch)
{
    const GridParams gp = {minX, minY, invW, invH};
    const int cap1pad = (cap1 + 63) / 64 * 64;
    float4 *rec = (float4 *)scratch;
    uint32_t *tuples = (uint32_t *)(rec + (size_t)B * cap2);
    int32_t *tcount = (int32_t *)(tuples + (size_t)B * cap1pad * INIT_K);
    const int keff = init_keff();
    const float radius = (float)windowSize;
    hipLaunchKernelGGL(k_proj_records, dim3((cap2 + 255) / 256, B, 1), dim3(256, 1, 1), 0, s, kps2, cap2, cellOff2, cellIdx2, rec);
    hipLaunchKernelGGL(k_init_cands, dim3(cap1pad / 4, B, 1), dim3(256, 1, 1), 0, s, kps1, desc1, cnt1, cap1, (const float2 *)prev,
                       radius, gp, cellOff2, rec, desc2, cap2, cap1pad, keff, tuples, tcount);
    // LDS: one buffer for a round's lists (32 KB) + the match tables are required (the API checks); the staged tables and a second
    // list buffer when they fit as well
    const size_t kLds = 144 * 1024, round = (size_t)INIT_ROUND * 4;
    int mode = round + init_assign_lds_staged(cap1, cap2) <= kLds ? 1 : 0;
    const size_t tables = mode ? init_assign_lds_staged(cap1, cap2) : init_assign_lds(cap1, cap2);
    if (2 * round + tables <= kLds) mode |= 2;
    mode |= ORB_TUNE("INIT_STOP", 0) << 4;
    const size_t lds = ((mode & 2) ? 2 : 1) * round + tables;
    if (lds > 16 * 1024) (void)hipFuncSetAttribute((const void *)k_init_assign, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(k_init_assign, dim3(B, 1, 1), dim3(INIT_NT, 1, 1), lds, s, kps1, desc1, cnt1, cap1, kps2,
                       desc2, cnt2, cap2, (float2 *)prev, radius, gp, cellOff2, cellIdx2, cap1pad, keff, tuples, tcount, nnratio,
                       check_ori, th_low, matches12, nmatches, mode);
    return ORBHIP_OK;
}

static int proj_keff()
{
    static int v = -1;
    if (v < 0) {
        v = ORB_TUNE("PROJ_K", PROJ_K);   // tests force the rescan path with a small value
        if (v < 1) v = 1;
        if (v > PROJ_K) v = PROJ_K;
    }
    return v;
}

int launch_grid_build(hipStream_t s, const orbhip_keypoint *kps, const int32_t *cnt, int cap, int B, float minX,
                      float minY, float invW, float invH, int32_t *cellOff, int32_t *cellIdx, int32_t *cellOff2, int32_t *cellIdx2)
{
    const GridParams gp = {minX, minY, invW, invH};
    if ((size_t)cap * 4 <= 48 * 1024)
        if (B < 8)
            hipLaunchKernelGGL((k_grid_build<true, 1024>), dim3(B, 1, 1), dim3(1024, 1, 1), (size_t)cap * 4, s, kps, cnt, cap, gp, cellOff,
                               cellIdx, cellOff2, cellIdx2);
        else
            hipLaunchKernelGGL((k_grid_build<true, 256>), dim3(B, 1, 1), dim3(256, 1, 1), (size_t)cap * 4, s, kps, cnt, cap, gp, cellOff,
                               cellIdx, cellOff2, cellIdx2);
    else {
        if (cellOff2 || cellIdx2) return ORBHIP_E_SIZE;   // (the twin is written by the LDS form)
        hipLaunchKernelGGL((k_grid_build<false, 256>), dim3(B, 1, 1), dim3(256, 1, 1), 0, s, kps, cnt, cap, gp, cellOff, cellIdx,
                           (int32_t *)nullptr, (int32_t *)nullptr);
    }
    return ORBHIP_OK;
}

int launch_area_list(hipStream_t s, const orbhip_keypoint *kps, float minX, float minY, float invW, float invH,
                     const int32_t *cellOff, const int32_t *cellIdx, const orbhip_proj_query *queries, int nq, int slots,
                     int32_t *outCnt, int32_t *outIdx)
{
    const GridParams gp = {minX, minY, invW, invH};
    hipLaunchKernelGGL(k_area_list, dim3((nq + 255) / 256, 1, 1), dim3(256, 1, 1), 0, s, kps, gp, cellOff, cellIdx, queries, nq,
                       slots, outCnt, outIdx);
    return ORBHIP_OK;
}

size_t window_best_scratch_bytes(int B, int cap) { return (size_t)B * cap * 16 + 256; }

int launch_window_best(hipStream_t s, const orbhip_keypoint *kps, const uint8_t *desc, int cap, int B, const float *uRight,
                       const float *invLevelSigma2, int nlevels, float minX, float minY, float invW, float invH,
                       const int32_t *cellOff, const int32_t *cellIdx, const orbhip_proj_query *queries,
                       const uint8_t *qdesc, const int32_t *nq, int capQ, int32_t *bestIdx, int32_t *bestDist, void *scratch)
{
    const GridParams gp = {minX, minY, invW, invH};
    LevelGate gate;
    gate.on = invLevelSigma2 != nullptr;
    for (int i = 0; i < 16; i++) gate.invSigma2[i] = (gate.on && i < nlevels) ? invLevelSigma2[i] : 0.f;
    float4 *rec = (float4 *)scratch;
    hipLaunchKernelGGL(k_proj_records, dim3((cap + 255) / 256, B, 1), dim3(256, 1, 1), 0, s, kps, cap, cellOff, cellIdx, rec);
    static const bool seqOnly = ORB_TUNE("PROJ_SEQ", 0) != 0;
    if (!seqOnly && cap < (1 << 20))   // a 16-lane row per point (written for one key frame per call; batches gain 3 % too)
        hipLaunchKernelGGL(k_window_best_row, dim3((capQ + 15) / 16, B, 1), dim3(256, 1, 1), 0, s, desc, cap, uRight, gate, gp,
                           cellOff, rec, queries, qdesc, nq, capQ, bestIdx, bestDist);
    else
        hipLaunchKernelGGL(k_window_best, dim3((capQ + 255) / 256, B, 1), dim3(256, 1, 1), 0, s, desc, cap, uRight, gate, gp,
                           cellOff, rec, queries, qdesc, nq, capQ, bestIdx, bestDist);
    return ORBHIP_OK;
}

size_t proj_scratch_bytes(int B, int capQ, int cap)
{
    const size_t capQpad = ((size_t)capQ + 63) / 64 * 64;
    return (size_t)B * capQpad * (PROJ_K * 4 + 4 + 4) + (size_t)B * cap * 16 + (size_t)B * 4 + 512;
}

size_t proj_assign_par_lds(int cap, int capQpad) { return (size_t)cap * 8 + (size_t)capQpad * 4 + (size_t)((cap + 31) / 32) * 4; }

size_t proj_assign_lds(int cap) { return (size_t)cap * 4 + (size_t)((cap + 31) / 32) * 4; }

int launch_search_by_projection(hipStream_t s, const orbhip_keypoint *kps, const uint8_t *desc, const int32_t *cnt, int cap,
                                int B, const float *uRight, const uint8_t *occupied, float minX, float minY, float invW,
                                float invH, const int32_t *cellOff, const int32_t *cellIdx, const orbhip_proj_query *queries,
                                const uint8_t *qdesc, const int32_t *nq, int capQ, int use_ratio, float nnratio,
                                int check_ori, int th_high, int32_t *match, int32_t *nmatches, void *scratch)
{
    const GridParams gp = {minX, minY, invW, invH};
    const int capQpad = (capQ + 63) / 64 * 64;
    float4 *rec = (float4 *)scratch;                                   // scratch base is 256-byte aligned
    uint32_t *tuples = (uint32_t *)(rec + (size_t)B * cap);
    int32_t *tcount = (int32_t *)(tuples + (size_t)B * capQpad * PROJ_K);
    int32_t *qfeat = tcount + (size_t)B * capQpad;
    const int keff = proj_keff();
    hipLaunchKernelGGL(k_proj_records, dim3((cap + 255) / 256, B, 1), dim3(256, 1, 1), 0, s, kps, cap, cellOff, cellIdx, rec);
    static const bool seqOnly = ORB_TUNE("PROJ_SEQ", 0) != 0;
    if (!seqOnly)   // a 16-lane row per point (written for the single-frame call; 512-frame batches gain 3-4 % from it too)
        hipLaunchKernelGGL(k_proj_cands_row, dim3((capQpad + 15) / 16, B, 1), dim3(256, 1, 1), 0, s, desc, cap, uRight, gp, cellOff, rec,
                           queries, qdesc, nq, capQ, capQpad, keff, tuples, tcount);
    else
        hipLaunchKernelGGL(k_proj_cands, dim3((capQpad + 255) / 256, B, 1), dim3(256, 1, 1), 0, s, kps, desc, cap, uRight, gp,
                           cellOff, rec, queries, qdesc, nq, capQ, capQpad, keff, tuples, tcount);
    // the parallel fixed-point kernel first; frames it cannot do (a point with more than 32
    // candidates, no fixed point yet) are left to the sequential one through fallback[] (ORBHIP_PROJ_SEQ=1: sequential only)
    static const int maxRounds = ORB_TUNE("PROJ_ROUNDS", PAR_ROUNDS);   // tests force the hand-over with 1
    int32_t *fallback = nullptr;
    const size_t parLds = proj_assign_par_lds(cap, capQpad);
    if (!seqOnly && parLds <= 150 * 1024) {   // (batches: +2 % on the tracking front-end row of configs.md)
        fallback = (int32_t *)(((uintptr_t)(qfeat + (size_t)B * capQpad) + 255) & ~(uintptr_t)255);
        if (parLds > 32 * 1024)
            (void)hipFuncSetAttribute((const void *)k_proj_assign_par, hipFuncAttributeMaxDynamicSharedMemorySize, (int)parLds);
        hipLaunchKernelGGL(k_proj_assign_par, dim3(B, 1, 1), dim3(1024, 1, 1), parLds, s, kps, cnt, cap, occupied, queries, nq, capQ,
                           capQpad, keff, tuples, tcount, qfeat, use_ratio, nnratio, check_ori, th_high, match, nmatches, fallback, maxRounds);
    }
    if (proj_assign_lds(cap) > 32 * 1024)
        (void)hipFuncSetAttribute((const void *)k_proj_assign, hipFuncAttributeMaxDynamicSharedMemorySize,
                                  (int)proj_assign_lds(cap));
    hipLaunchKernelGGL(k_proj_assign, dim3(B, 1, 1), dim3(64, 1, 1), proj_assign_lds(cap), s, kps, desc, cnt, cap, uRight,
                       occupied, gp, cellOff, cellIdx, queries, qdesc, nq, capQ, capQpad, keff, tuples, tcount, qfeat,
                       use_ratio, nnratio, check_ori, th_high, match, nmatches, fallback);
    return ORBHIP_OK;
}
