// k_hamming.hip -- M0/M1/M2/M3: 256-bit Hamming matching (ref: src/ORBmatcher.cc:1675-1691
// DescriptorDistance; :205-226 and the other search routines' best / second-best bookkeeping;
// :159-288 and :522-655 SearchByBoW).  Bitwise path (XOR + v_bcnt_u32_b32) everywhere except the brute force over many queries,
// which runs on the matrix pipe.
//
//  k_knn2_mfma   brute force, many queries: |a & b| as an FP4 matrix product of the bits written out as numbers (see the kernel);
//                k_knn2_seq_mfma the same for the frame sequence form.  ORBHIP_KNN2_MFMA=0 selects the scalar kernels below.
//  k_knn2        brute force, query tile x database split; a thread owns one query (8 VGPRs),
//                the database row is wave-uniform and arrives through scalar loads; partial
//                (best, index, second) per split, merged in split order by k_knn2_merge so that
//                the lowest index wins ties exactly like the reference's strict '<' loop.
//                Roofline: 16 integer ops per 32-byte pair => VALU-bound for many queries,
//                HBM-bound (database streamed once) for few.
//  k_knn2_lists  the same bookkeeping over explicit candidate lists (guided search).
//  k_bow_match   SearchByBoW: one wave per shared vocabulary node; side-1 features are visited
//                serially (the reference's greedy claiming is order dependent), the 64 lanes scan
//                the node's side-2 features in parallel and reduce with a tie-aware merge.
#include "orbhip_internal.h"

#include <algorithm>
#include <cstdlib>

struct Best {
    int b1, idx, b2;
};

__device__ __forceinline__ int hamming256(const uint32_t q[8], const uint32_t r[8])
{
    int d = 0;
#pragma unroll
    for (int k = 0; k < 8; k++) d += __popc(q[k] ^ r[k]);
    return d;
}

// sequential update of ref: :216-226
__device__ __forceinline__ void best_update(Best &B, int d, int j)
{
    // branch-free form of: if (d < b1) { b2 = b1; b1 = d; idx = j; } else if (d < b2) b2 = d;
    const bool lt = d < B.b1;
    B.b2 = lt ? B.b1 : min(B.b2, d);
    B.idx = lt ? j : B.idx;
    B.b1 = min(B.b1, d);
}

// A holds candidates that come BEFORE all of B's in the reference's visiting order.
__device__ __forceinline__ Best best_merge_ordered(const Best &A, const Best &B)
{
    Best R;
    if (B.b1 < A.b1) {
        R.b1 = B.b1;
        R.idx = B.idx;
        R.b2 = min(A.b1, B.b2);
    } else {
        R.b1 = A.b1;
        R.idx = A.idx;
        R.b2 = min(A.b2, B.b1);
    }
    return R;
}

// Visits database rows j0..j1-1 in order, four rows per trip so that four wave-uniform (scalar)
// row loads are in flight while the previous rows are being compared.
__device__ __forceinline__ void scan_rows(Best &B, const uint32_t Q[8], const uint8_t *__restrict__ db, int j0, int j1)
{
    int j = j0;
    if (j + 4 <= j1) {
        // software pipeline: the next four rows are requested before the current four are compared
        uint32_t R[32];
        {
            const uint32_t *row = reinterpret_cast<const uint32_t *>(db + (size_t)j * 32);
#pragma unroll
            for (int k = 0; k < 32; k++) R[k] = row[k];
        }
        for (; j + 8 <= j1; j += 4) {
            const uint32_t *nrow = reinterpret_cast<const uint32_t *>(db + (size_t)(j + 4) * 32);
            uint32_t N[32];
#pragma unroll
            for (int k = 0; k < 32; k++) N[k] = nrow[k];
#pragma unroll
            for (int u = 0; u < 4; u++) best_update(B, hamming256(Q, R + 8 * u), j + u);
#pragma unroll
            for (int k = 0; k < 32; k++) R[k] = N[k];
        }
#pragma unroll
        for (int u = 0; u < 4; u++) best_update(B, hamming256(Q, R + 8 * u), j + u);
        j += 4;
    }
    for (; j < j1; j++) {
        const uint32_t *row = reinterpret_cast<const uint32_t *>(db + (size_t)j * 32);
        uint32_t R[8];
#pragma unroll
        for (int k = 0; k < 8; k++) R[k] = row[k];
        best_update(B, hamming256(Q, R), j);
    }
}

__global__ __launch_bounds__(256) void k_knn2(const uint8_t *__restrict__ q, int nq,
                                              const uint8_t *__restrict__ db, int ndb, int rowsPerSplit,
                                              int4 *__restrict__ partial)
{
    const int qi = blockIdx.x * 256 + threadIdx.x;
    const int split = blockIdx.y;
    uint32_t Q[8];
    if (qi < nq) {
        const uint4 a = reinterpret_cast<const uint4 *>(q + (size_t)qi * 32)[0];
        const uint4 b = reinterpret_cast<const uint4 *>(q + (size_t)qi * 32)[1];
        Q[0] = a.x; Q[1] = a.y; Q[2] = a.z; Q[3] = a.w;
        Q[4] = b.x; Q[5] = b.y; Q[6] = b.z; Q[7] = b.w;
    } else {
#pragma unroll
        for (int k = 0; k < 8; k++) Q[k] = 0;
    }
    Best B = {256, -1, 256};
    const int j0 = split * rowsPerSplit;
    const int j1 = min(ndb, j0 + rowsPerSplit);
    scan_rows(B, Q, db, j0, j1);   // wave-uniform addresses: the compiler emits scalar loads
    if (qi < nq) partial[(size_t)split * nq + qi] = make_int4(B.b1, B.idx, B.b2, 0);
}

// 16 queries x 16 runs of consecutive splits per workgroup (thread per query alone: 16 workgroups walking 256 partial results
// each took 73 us behind a 0.71 ms k_knn2_mfma); the runs are merged in split order, then the 16 run results in run order.
#define KMERGE_Q 16
#define KMERGE_P 16
__global__ __launch_bounds__(256) void k_knn2_merge(const int4 *__restrict__ partial, int nq, int nsplit,
                                                    int32_t *__restrict__ best_idx,
                                                    int32_t *__restrict__ best_d,
                                                    int32_t *__restrict__ second_d)
{
    __shared__ int4 s_run[KMERGE_P][KMERGE_Q];
    const int ql = threadIdx.x & (KMERGE_Q - 1), run = threadIdx.x / KMERGE_Q;
    const int qi = blockIdx.x * KMERGE_Q + ql;
    const int per = (nsplit + KMERGE_P - 1) / KMERGE_P;
    const int s0 = run * per, s1 = min(nsplit, s0 + per);
    Best A = {256, -1, 256};
    if (qi < nq)
        for (int s = s0; s < s1; s++) {
            const int4 p = partial[(size_t)s * nq + qi];
            Best Bp = {p.x, p.y, p.z};
            A = best_merge_ordered(A, Bp);
        }
    s_run[run][ql] = make_int4(A.b1, A.idx, A.b2, 0);
    __syncthreads();
    if (run != 0 || qi >= nq) return;
    for (int r = 1; r < KMERGE_P; r++) {
        const int4 p = s_run[r][ql];
        Best Bp = {p.x, p.y, p.z};
        A = best_merge_ordered(A, Bp);
    }
    best_idx[qi] = A.idx;
    best_d[qi] = A.b1;
    second_d[qi] = A.b2;
}

#ifndef KM_WANT
#define KM_WANT 1024      // workgroups a query is cut into (query tiles x database splits): one round of four per CU -- a workgroup's
                          // set-up and its rescan (below) are ~15 us in which its waves issue no matrix instruction, and the
                          // workgroups of a round reach them together (4096 workgroups of 4096-row key ranges: 0.51 ms for 4000 x 1 M; 1024 of 16384: 0.47)
#endif
static void knn2_shape(int nq, int ndb, int *qTiles, int *nsplit, int *rows)
{
    *qTiles = (nq + 255) / 256;
    int want = KM_WANT / (*qTiles > 0 ? *qTiles : 1);
    if (want < 1) want = 1;
    int maxSplit = (ndb + 63) / 64;
    if (maxSplit < 1) maxSplit = 1;
    int ns = want < maxSplit ? want : maxSplit;
    if (ns > 65535) ns = 65535;
    *rows = (ndb + ns - 1) / ns;
    if (*rows < 1) *rows = 1;
    *nsplit = ndb > 0 ? (ndb + *rows - 1) / *rows : 1;
}

// ---- few queries against a large database: the HBM-bound regime ----
// Lanes own database rows (one coalesced 32-byte row per lane and trip), the <= FQ queries of a pass
// sit in LDS and are broadcast; every lane keeps (best, index, second) per query and the results are
// merged across lanes, waves and workgroups with the lowest index winning ties.
#define FQ 8
#define FQ_BLOCKS 2048

__device__ __forceinline__ void best_merge_unordered(Best &A, int ob1, int oidx, int ob2)
{
    const bool mine = (A.b1 < ob1) || (A.b1 == ob1 && (unsigned)A.idx < (unsigned)oidx);
    const int nb2 = mine ? min(A.b2, ob1) : min(ob2, A.b1);
    A.b1 = mine ? A.b1 : ob1;
    A.idx = mine ? A.idx : oidx;
    A.b2 = nb2;
}

__global__ __launch_bounds__(256) void k_knn2_fewq(const uint8_t *__restrict__ q, int nq, int q0,
                                                   const uint8_t *__restrict__ db, int ndb,
                                                   int4 *__restrict__ partial)
{
    __shared__ uint32_t s_q[FQ][8];
    __shared__ int4 s_red[4][FQ];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int nloc = min(FQ, nq - q0);
    if (tid < FQ * 8) {
        const int k = tid >> 3, w = tid & 7;
        s_q[k][w] = k < nloc ? reinterpret_cast<const uint32_t *>(q + (size_t)(q0 + k) * 32)[w] : 0u;
    }
    __syncthreads();
    Best B[FQ];
#pragma unroll
    for (int k = 0; k < FQ; k++) B[k] = {256, -1, 256};
    const int stride = gridDim.x * 256;
    for (int j = blockIdx.x * 256 + tid; j < ndb; j += stride) {
        const uint4 r0 = reinterpret_cast<const uint4 *>(db + (size_t)j * 32)[0];
        const uint4 r1 = reinterpret_cast<const uint4 *>(db + (size_t)j * 32)[1];
        const uint32_t R[8] = {r0.x, r0.y, r0.z, r0.w, r1.x, r1.y, r1.z, r1.w};
#pragma unroll
        for (int k = 0; k < FQ; k++) {
            uint32_t Q[8];
#pragma unroll
            for (int w = 0; w < 8; w++) Q[w] = s_q[k][w];
            best_update(B[k], hamming256(Q, R), j);
        }
    }
    // lanes -> wave -> workgroup
#pragma unroll
    for (int k = 0; k < FQ; k++) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const int ob1 = __shfl_xor(B[k].b1, o), oidx = __shfl_xor(B[k].idx, o), ob2 = __shfl_xor(B[k].b2, o);
            best_merge_unordered(B[k], ob1, oidx, ob2);
        }
        if (lane == 0) s_red[wave][k] = make_int4(B[k].b1, B[k].idx, B[k].b2, 0);
    }
    __syncthreads();
    if (tid < FQ) {
        Best A = {s_red[0][tid].x, s_red[0][tid].y, s_red[0][tid].z};
        for (int w = 1; w < 4; w++) best_merge_unordered(A, s_red[w][tid].x, s_red[w][tid].y, s_red[w][tid].z);
        partial[(size_t)blockIdx.x * FQ + tid] = make_int4(A.b1, A.idx, A.b2, 0);
    }
}

__global__ __launch_bounds__(64) void k_knn2_fewq_merge(const int4 *__restrict__ partial, int nblocks, int nq, int q0,
                                                        int32_t *__restrict__ best_idx, int32_t *__restrict__ best_d,
                                                        int32_t *__restrict__ second_d)
{
    const int k = blockIdx.x, lane = threadIdx.x;
    if (q0 + k >= nq) return;
    Best A = {256, -1, 256};
    for (int b = lane; b < nblocks; b += 64) {
        const int4 p = partial[(size_t)b * FQ + k];
        best_merge_unordered(A, p.x, p.y, p.z);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const int ob1 = __shfl_xor(A.b1, o), oidx = __shfl_xor(A.idx, o), ob2 = __shfl_xor(A.b2, o);
        best_merge_unordered(A, ob1, oidx, ob2);
    }
    if (lane == 0) {
        best_idx[q0 + k] = A.idx;
        best_d[q0 + k] = A.b1;
        second_d[q0 + k] = A.b2;
    }
}

static bool knn2_few(int nq, int ndb) { return nq <= 4 * FQ && ndb >= 16384; }

size_t knn2_scratch_bytes(int nq, int ndb)
{
    if (knn2_few(nq, ndb)) return (size_t)FQ_BLOCKS * FQ * sizeof(int4);
    int qt, ns, rows;
    knn2_shape(nq, ndb, &qt, &ns, &rows);
    return (size_t)ns * (size_t)(nq > 0 ? nq : 1) * sizeof(int4);
}

#ifndef KM_SUB
#define KM_SUB 2          // 32-row tiles of the matrix instruction per staged tile (one barrier per 64 rows)
#endif
#ifndef KM_QT
#define KM_QT 2           // 32-query tiles per wave (each A fragment read from LDS feeds KM_QT matrix instructions)
#endif
#define KM_QPB (4 * 32 * KM_QT)   // queries per workgroup of four waves
__global__ void k_knn2_mfma(const uint8_t *__restrict__ q, int nq, const uint8_t *__restrict__ db, int ndb, int rowsPerSplit,
                            int4 *__restrict__ partial);   // (below)

void launch_knn2(hipStream_t s, const uint8_t *q, int nq, const uint8_t *db, int ndb, int32_t *best_idx,
                 int32_t *best_d, int32_t *second_d, void *scratch, size_t scratch_bytes)
{
    if (nq <= 0) return;
    (void)scratch_bytes;
    int4 *partial = reinterpret_cast<int4 *>(scratch);
    if (knn2_few(nq, ndb)) {
        const int nblocks = std::min(FQ_BLOCKS, (ndb + 255) / 256);
        for (int q0 = 0; q0 < nq; q0 += FQ) {
            hipLaunchKernelGGL(k_knn2_fewq, dim3(nblocks, 1, 1), dim3(256, 1, 1), 0, s, q, nq, q0, db, ndb, partial);
            hipLaunchKernelGGL(k_knn2_fewq_merge, dim3(FQ, 1, 1), dim3(64, 1, 1), 0, s, partial, nblocks, nq, q0,
                               best_idx, best_d, second_d);
        }
        return;
    }
    int qt, ns, rows;
    knn2_shape(nq, ndb, &qt, &ns, &rows);
    static const int mfmaEnv = ORB_SWITCH("KNN2_MFMA", 1);
    // (the matrix kernel reads queries as 16-byte words and database rows as dwords: a caller's device pointer that is not
    // aligned like that takes the scalar kernel, whose accesses are dword / byte-safe scalar loads)
    const bool aligned = (reinterpret_cast<uintptr_t>(q) & 15u) == 0 && (reinterpret_cast<uintptr_t>(db) & 3u) == 0;
    if (mfmaEnv && aligned)
        hipLaunchKernelGGL(k_knn2_mfma, dim3((nq + KM_QPB - 1) / KM_QPB, ns, 1), dim3(256, 1, 1), 0, s, q, nq, db, ndb, rows, partial);
    else
        hipLaunchKernelGGL(k_knn2, dim3(qt, ns, 1), dim3(256, 1, 1), 0, s, q, nq, db, ndb, rows, partial);
    hipLaunchKernelGGL(k_knn2_merge, dim3((nq + KMERGE_Q - 1) / KMERGE_Q, 1, 1), dim3(256, 1, 1), 0, s, partial, nq, ns, best_idx,
                       best_d, second_d);
}

// ---- brute force on the matrix pipe (many queries) ----
// d(a, b) = |a| + |b| - 2 |a & b|, and |a & b| is a dot product of the bits written out as numbers: one
// v_mfma_scale_f32_32x32x64_f8f6f4 with FP4 (e2m1) operands takes 64 bits of 32 database rows (A operand) against 64 bits of 32
// queries (B operand), four of them a tile of 1024 distances -- 4 matrix instructions instead of 1024 x (8 v_xor + 8 v_bcnt + 7
// adds).  (r03: v_mfma_i32_32x32x32_i8 on the bits written out as bytes, 8 instructions per tile at the same 32 cycles each; the
// FP4 form has twice the k per instruction, half the LDS image and half the operand registers.)
//  * Exactness.  A row bit is the FP4 number 0.5 (0b0001) with block scale 2^9, a query bit +1 (clear, 0b0010) or -1 (set, 0b1010)
//    with block scale 2^6: a set row bit contributes +-2^14, every partial sum is an integer below 2^24, the f32 accumulation is exact
//    (tools/microbench/fp4_dot.hip checks the instruction against the integer formula; the parity tests check the kernel).
//  * The instruction also builds the comparison key.  |b| - 2 |a & b| = sum over the set bits of b of (1 - 2 a_k), and the
//    accumulator starts at the row's index inside a chunk of 16384 database rows plus a bias, so the result is
//    ((|b| - 2 |a & b|) << 14) + row + bias, a POSITIVE float: ordered by distance, then by index, and its bit pattern orders like
//    its value, so the epilogue is integer min / med3 on the raw registers (|a| is the same for all keys of a query and added at the end).
//    What must come out is the two smallest keys -- (best, lowest index) and the second smallest distance of the multiset, exactly
//    what the strict '<' loop of the reference leaves (:205-226).  A lane takes the minimum of its 16 keys of a 32-row tile (8
//    v_min3_i32) and updates the running pair with that one key: m1 is exact, m2 is the second smallest GROUP minimum, and the true
//    second key is min(m2, second key of the winner's group) -- 15 rows per query and 16384-row chunk, recomputed by popcount after
//    the chunk (the "rescan").  (r03 kept the pair per value: v_min + v_med3 per key, 136 of 238 vector instructions per 64 rows.)
//  * Which bit goes to which k-slot is free as long as rows and queries agree: slot j of dword d of a word's 16-byte fragment holds
//    bit 4 j + d, i.e. dword d = (word >> d) & 0x11111111 -- 7 vector instructions per 32 bits (the byte form took 24).
//  * Result layout: column = lane & 31 = query, the 16 registers of a lane = 16 of the tile's rows (8 g + 4 h + e for register 4 g + e,
//    h = lane >> 5): a lane folds its own values, no cross-lane work; the two lanes of a query merge once at the end.
//  * A workgroup = 4 waves x KM_QT tiles of 32 queries x the rows of one split; the FP4 form of a 64-row database tile (8 KB) and
//    the rows' start values are built once per workgroup in LDS, thread = one word of a row, double-buffered behind one barrier per
//    tile.  Output = the split's partial result in the format of k_knn2, merged in split order by k_knn2_merge.
typedef int v4i_h __attribute__((ext_vector_type(4)));
typedef int v8i_h __attribute__((ext_vector_type(8)));
typedef float v16f_h __attribute__((ext_vector_type(16)));
#ifndef KM_SHIFT
#define KM_SHIFT 14          // index bits of a key = log2 of the database rows per key range (14 is the most an exact f32 key holds)
#endif
#define KM_BIAS ((1 << (KM_SHIFT + 8)) + (1 << (KM_SHIFT + 2)))   // > 256 << KM_SHIFT, a multiple of the chunk: every key is positive
#define KM_NONE 0x4C000000   // the f32 2^25 as a bit pattern: above every key
#define KM_SCALE_A (127 + KM_SHIFT - 5)   // E8M0: a row's set bit, 0.5, counts 2^(KM_SHIFT - 6) (2^7 x 0.5 = 64 for 12 index bits)
#define KM_SCALE_B 133       // E8M0 2^6: a query's bit, +-1, counts +-64
#define KM_CHUNK (1 << KM_SHIFT)   // database rows per key range
#ifndef KM_WAVES
#define KM_WAVES 4           // waves per SIMD the register budget is set for
#endif

// 32 bits of a descriptor word -> 32 four-bit slots holding 0 or 1
__device__ __forceinline__ v4i_h bits32_to_slots(uint32_t x)
{
    v4i_h r;
    r.x = (int)(x & 0x11111111u);
    r.y = (int)((x >> 1) & 0x11111111u);
    r.z = (int)((x >> 2) & 0x11111111u);
    r.w = (int)((x >> 3) & 0x11111111u);
    return r;
}

__device__ __forceinline__ int min3_after(int first, int b, int c)
{
    int r;
    asm("v_min3_i32 %0, %1, %2, %3" : "=v"(r) : "v"(first), "v"(b), "v"(c));
    return r;
}

__device__ __forceinline__ void top2_update(int &m1, int &m2, int key)
{
    int med;   // (key is the result of a vector instruction here, never a matrix result: nothing to wait for)
    asm("v_med3_i32 %0, %1, %2, %3" : "=v"(med) : "v"(m1), "v"(m2), "v"(key));   // second smallest of (m1 <= m2, key)
    m1 = min(m1, key);
    m2 = med;
}

// the key of (distance term s = |b| - 2 |a & b|, row r of the chunk) as the matrix instruction leaves it, and back
static_assert(KM_SHIFT >= 6 && 2 * (256 << KM_SHIFT) + (1 << (KM_SHIFT + 2)) + KM_CHUNK < (1 << 24), "keys must be integers an f32 holds exactly");
__device__ __forceinline__ int km_key(int s, int r) { return __float_as_int((float)(s * KM_CHUNK + r + KM_BIAS)); }
__device__ __forceinline__ int km_value(int key) { return (int)__int_as_float(key) - KM_BIAS; }

// KM_QPB queries q[qbase ..] against database rows j0 .. j1 - 1: query qbase + 32 KM_QT wave + 32 t + (lane & 31) -> out[t] (valid in lanes < 32)
__device__ __forceinline__ void knn2_mfma_core(const uint8_t *__restrict__ q, int nq, int qbase, const uint8_t *__restrict__ db,
                                               int j0, int j1, v4i_h (*s_A)[KM_SUB][8][32], int (*s_T)[KM_SUB * 32], Best out[KM_QT])
{
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, c = lane & 31, h = lane >> 5;
    int pa[KM_QT];
    v4i_h Bq[KM_QT][4];
#pragma unroll
    for (int t = 0; t < KM_QT; t++) {
        const int qi = qbase + 32 * KM_QT * wave + 32 * t + c;
        uint32_t Q[8];
        if (qi < nq) {
            const uint4 a = reinterpret_cast<const uint4 *>(q + (size_t)qi * 32)[0];
            const uint4 b = reinterpret_cast<const uint4 *>(q + (size_t)qi * 32)[1];
            Q[0] = a.x; Q[1] = a.y; Q[2] = a.z; Q[3] = a.w;
            Q[4] = b.x; Q[5] = b.y; Q[6] = b.z; Q[7] = b.w;
        } else {
#pragma unroll
            for (int k = 0; k < 8; k++) Q[k] = 0;
        }
        pa[t] = 0;
#pragma unroll
        for (int k = 0; k < 8; k++) pa[t] += __popc(Q[k]);
#pragma unroll
        for (int ks = 0; ks < 4; ks++)   // k-step ks of this lane: word 2 ks + h; -1 (0b1010) per set bit, +1 (0b0010) per clear one
            Bq[t][ks] = (bits32_to_slots(h ? Q[2 * ks + 1] : Q[2 * ks]) << 3) | 0x22222222;
        out[t].b1 = 256;
        out[t].idx = -1;
        out[t].b2 = 256;
    }
    // staging role of this thread: word sw of the rows srow, srow + 32, ... of the staged tile (the 32 lanes of a half-wave write 32
    // consecutive 16-byte slots)
    const int srow = tid & 31, sw = tid >> 5;
    constexpr int TR = 32 * KM_SUB;
    for (int c0 = j0; c0 < j1; c0 += KM_CHUNK) {   // (one trip unless the database has more rows than a key can index)
        const int c1 = min(j1, c0 + KM_CHUNK);
        auto fetch = [&](int jt, uint32_t word[KM_SUB]) {
#pragma unroll
            for (int u = 0; u < KM_SUB; u++) {
                const int j = jt + 32 * u + srow;
                word[u] = j < c1 ? reinterpret_cast<const uint32_t *>(db + (size_t)j * 32)[sw] : 0u;
            }
        };
        auto stage = [&](int buf, int jt, const uint32_t word[KM_SUB]) {
#pragma unroll
            for (int u = 0; u < KM_SUB; u++) {
                s_A[buf][u][sw][srow] = bits32_to_slots(word[u]);
                const int j = jt + 32 * u + srow;
                if (sw == 0) s_T[buf][32 * u + srow] = j < c1 ? km_key(0, j - c0) : KM_NONE;
            }
        };
        int m1[KM_QT], m2[KM_QT];
#pragma unroll
        for (int t = 0; t < KM_QT; t++) m1[t] = m2[t] = KM_NONE;
        const int ntiles = (c1 - c0 + TR - 1) / TR;
        uint32_t word[KM_SUB];
        __syncthreads();   // (a previous chunk's last tile may still be read)
        fetch(c0, word);
        stage(0, c0, word);
        for (int n = 0; n < ntiles; n++) {
            const int buf = n & 1;
            __syncthreads();   // tile n is staged; every wave is done with tile n - 1 (whose buffer the staging below overwrites)
            if (n + 1 < ntiles) fetch(c0 + TR * (n + 1), word);
            // the accumulators start at the rows' keys: register 4 g + e of this lane = row 8 g + 4 h + e of the 32-row tile.  The
            // start values and the first A fragment of a 32-row tile are requested BEFORE the keys of the tile in front of it
            // are folded (into registers of their own): the LDS round trip runs beside the fold instead of behind it.
            struct Start {
                v4i_h T[4], a0;
            };
            auto request = [&](int u) {
                Start S;
#pragma unroll
                for (int g = 0; g < 4; g++) S.T[g] = reinterpret_cast<const v4i_h *>(s_T[buf] + 32 * u)[2 * g + h];
                S.a0 = s_A[buf][u][h][c];
                return S;
            };
            auto issue = [&](int u, const Start &S, v16f_h (&acc)[KM_QT]) {
#pragma unroll
                for (int g = 0; g < 4; g++)
#pragma unroll
                    for (int e = 0; e < 4; e++)
#pragma unroll
                        for (int t = 0; t < KM_QT; t++) acc[t][4 * g + e] = __int_as_float(S.T[g][e]);
                v4i_h ring[2];
                ring[0] = S.a0;
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    if (k + 1 < 4) ring[(k + 1) & 1] = s_A[buf][u][2 * (k + 1) + h][c];
                    const v4i_h a = ring[k & 1];
                    const v8i_h A8 = {a.x, a.y, a.z, a.w, 0, 0, 0, 0};
#pragma unroll
                    for (int t = 0; t < KM_QT; t++) {
                        const v8i_h B8 = {Bq[t][k].x, Bq[t][k].y, Bq[t][k].z, Bq[t][k].w, 0, 0, 0, 0};
                        acc[t] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(A8, B8, acc[t], 4, 4, 0, KM_SCALE_A, 0, KM_SCALE_B);
                    }
                }
            };
            // the smallest of the lane's 16 keys, and only IT enters the running pair (see above).  Each chain starts with a
            // v_min the compiler sees -- it knows how long a matrix result must not be read after its instruction issued and
            // waits in front of it; the inline v_min3 that follow depend on its result.
            auto fold = [&](const v16f_h (&acc)[KM_QT]) {
#pragma unroll
                for (int t = 0; t < KM_QT; t++) {
                    // a tree, not a chain: five independent minima of three, two of those, one of two (depth 3 + the update)
                    int x[16];
#pragma unroll
                    for (int r = 0; r < 16; r++) x[r] = __float_as_int(acc[t][r]);
                    const int a0 = min(min(x[0], x[1]), x[2]), a1 = min(min(x[3], x[4]), x[5]), a2 = min(min(x[6], x[7]), x[8]);
                    const int a3 = min(min(x[9], x[10]), x[11]), a4 = min(min(x[12], x[13]), x[14]);
                    const int b0 = min(min(a0, a1), a2), b1 = min(min(a3, a4), x[15]);
                    top2_update(m1[t], m2[t], min(b0, b1));
                }
            };
            Start S = request(0);
#pragma unroll
            for (int u = 0; u < KM_SUB; u++) {
                v16f_h acc[KM_QT];
                issue(u, S, acc);
                if (u + 1 < KM_SUB) S = request(u + 1);
                fold(acc);
            }
            if (n + 1 < ntiles) stage(buf ^ 1, c0 + TR * (n + 1), word);
        }
        // rescan: the other 15 rows of the group the smallest key came from, by popcount
#pragma unroll
        for (int t = 0; t < KM_QT; t++) {
            const int qi = qbase + 32 * KM_QT * wave + 32 * t + c;
            if (m1[t] < KM_NONE && qi < nq) {
                const uint4 qa = reinterpret_cast<const uint4 *>(q + (size_t)qi * 32)[0];
                const uint4 qb = reinterpret_cast<const uint4 *>(q + (size_t)qi * 32)[1];
                const int win = km_value(m1[t]) & (KM_CHUNK - 1), base = (win & ~31) + 4 * h;
                const uint8_t *rows = db + (size_t)(c0 + base) * 32;
                int m2v = m2[t];
#pragma unroll 4
                for (int ge = 0; ge < 16; ge++) {
                    const int ro = 8 * (ge >> 2) + (ge & 3), rr = base + ro;
                    if (rr != win && c0 + rr < c1) {
                        const uint4 ra = reinterpret_cast<const uint4 *>(rows + ro * 32)[0];
                        const uint4 rb = reinterpret_cast<const uint4 *>(rows + ro * 32)[1];
                        const int nb = __popc(ra.x) + __popc(ra.y) + __popc(ra.z) + __popc(ra.w) + __popc(rb.x) + __popc(rb.y) +
                                       __popc(rb.z) + __popc(rb.w);
                        const int nab = __popc(ra.x & qa.x) + __popc(ra.y & qa.y) + __popc(ra.z & qa.z) + __popc(ra.w & qa.w) +
                                        __popc(rb.x & qb.x) + __popc(rb.y & qb.y) + __popc(rb.z & qb.z) + __popc(rb.w & qb.w);
                        m2v = min(m2v, km_key(nb - 2 * nab, rr));
                    }
                }
                m2[t] = m2v;
            }
        }
#pragma unroll
        for (int t = 0; t < KM_QT; t++) {
            // the other half of this query's rows, then this chunk behind the earlier ones
            const int o1 = __shfl_xor(m1[t], 32), o2 = __shfl_xor(m2[t], 32);
            const int k1 = min(m1[t], o1), k2 = min(max(m1[t], o1), min(m2[t], o2));
            const int add = pa[t] << KM_SHIFT;
            Best C;
            C.b1 = k1 >= KM_NONE ? 256 : (km_value(k1) + add) >> KM_SHIFT;
            C.idx = k1 >= KM_NONE ? -1 : c0 + (km_value(k1) & (KM_CHUNK - 1));
            C.b2 = k2 >= KM_NONE ? 256 : (km_value(k2) + add) >> KM_SHIFT;
            out[t] = best_merge_ordered(out[t], C);
        }
    }
}

__global__ __launch_bounds__(256, KM_WAVES) void k_knn2_mfma(const uint8_t *__restrict__ q, int nq, const uint8_t *__restrict__ db, int ndb,
                                                          int rowsPerSplit, int4 *__restrict__ partial)
{
    __shared__ v4i_h s_A[2][KM_SUB][8][32];
    __shared__ __align__(16) int s_T[2][KM_SUB * 32];
    const int split = blockIdx.y;
    const int j0 = split * rowsPerSplit, j1 = min(ndb, j0 + rowsPerSplit);
    Best B[KM_QT];
    knn2_mfma_core(q, nq, blockIdx.x * KM_QPB, db, j0, j1, s_A, s_T, B);
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int t = 0; t < KM_QT; t++) {
        const int qi = blockIdx.x * KM_QPB + 32 * KM_QT * (threadIdx.x >> 6) + 32 * t + (lane & 31);
        if (lane < 32 && qi < nq) partial[(size_t)split * nq + qi] = make_int4(B[t].b1, B[t].idx, B[t].b2, 0);
    }
}

__global__ __launch_bounds__(256, KM_WAVES) void k_knn2_seq_mfma(const uint8_t *__restrict__ desc, const int32_t *__restrict__ counts, int cap,
                                                              int lag, int32_t *__restrict__ best_idx, int32_t *__restrict__ best_d,
                                                              int32_t *__restrict__ second_d)
{
    __shared__ v4i_h s_A[2][KM_SUB][8][32];
    __shared__ __align__(16) int s_T[2][KM_SUB * 32];
    const int b = blockIdx.y;
    const int nq = min(counts[b], cap);
    if (blockIdx.x * KM_QPB >= nq) return;   // whole block idle (uniform)
    const int ndb = b >= lag ? min(counts[b - lag], cap) : 0;
    Best B[KM_QT];
    knn2_mfma_core(desc + (size_t)b * cap * 32, nq, blockIdx.x * KM_QPB, desc + (size_t)(b >= lag ? b - lag : 0) * cap * 32, 0, ndb,
                   s_A, s_T, B);
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int t = 0; t < KM_QT; t++) {
        const int qi = blockIdx.x * KM_QPB + 32 * KM_QT * (threadIdx.x >> 6) + 32 * t + (lane & 31);
        if (lane < 32 && qi < nq) {
            const size_t o = (size_t)b * cap + qi;
            best_idx[o] = B[t].idx;
            best_d[o] = B[t].b1;
            second_d[o] = B[t].b2;
        }
    }
}

// B independent (query set, database set) problems of a frame sequence in one launch:
// queries = descriptors of frame b, database = descriptors of frame b - lag.
__global__ __launch_bounds__(256) void k_knn2_seq(const uint8_t *__restrict__ desc,
                                                  const int32_t *__restrict__ counts, int cap, int lag,
                                                  int32_t *__restrict__ best_idx, int32_t *__restrict__ best_d,
                                                  int32_t *__restrict__ second_d)
{
    const int b = blockIdx.y;
    const int qi = blockIdx.x * 256 + threadIdx.x;
    const int nq = min(counts[b], cap);
    if (blockIdx.x * 256 >= nq) return;  // whole block idle (uniform)
    Best B = {256, -1, 256};
    if (b >= lag) {
        const int ndb = min(counts[b - lag], cap);
        const uint8_t *q = desc + ((size_t)b * cap + (qi < nq ? qi : 0)) * 32;
        const uint4 a0 = reinterpret_cast<const uint4 *>(q)[0], a1 = reinterpret_cast<const uint4 *>(q)[1];
        const uint32_t Q[8] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w};
        const uint8_t *db = desc + (size_t)(b - lag) * cap * 32;
        scan_rows(B, Q, db, 0, ndb);
    }
    if (qi < nq) {
        const size_t o = (size_t)b * cap + qi;
        best_idx[o] = B.idx;
        best_d[o] = B.b1;
        second_d[o] = B.b2;
    }
}

void launch_knn2_seq(hipStream_t s, const uint8_t *desc, const int32_t *counts, int cap, int B, int lag,
                     int32_t *best_idx, int32_t *best_d, int32_t *second_d)
{
    if (B <= 0) return;
    static const int mfmaEnv = ORB_SWITCH("KNN2_MFMA", 1);
    if (mfmaEnv && (reinterpret_cast<uintptr_t>(desc) & 15u) == 0)
        hipLaunchKernelGGL(k_knn2_seq_mfma, dim3((cap + KM_QPB - 1) / KM_QPB, B, 1), dim3(256, 1, 1), 0, s, desc, counts, cap, lag, best_idx,
                           best_d, second_d);
    else
        hipLaunchKernelGGL(k_knn2_seq, dim3((cap + 255) / 256, B, 1), dim3(256, 1, 1), 0, s, desc, counts, cap, lag,
                           best_idx, best_d, second_d);
}

// Min-merge of per-shard brute-force results (database rows sharded over ranks): parts[s] =
// best_idx[nq] | best_d[nq] | second_d[nq] | shard_offset, indices local to the shard.  Shards are visited in order of
// increasing rank = increasing database rows, so strict '<' keeps the lowest global index on ties, as one pass over the
// whole database would (src/ORBmatcher.cc:205-226 bookkeeping).
__global__ __launch_bounds__(256) void k_knn2_merge(const int32_t *__restrict__ parts, int nshards, int nq,
                                                    int32_t *__restrict__ best_idx, int32_t *__restrict__ best_d,
                                                    int32_t *__restrict__ second_d)
{
    const int qi = blockIdx.x * 256 + threadIdx.x;
    if (qi >= nq) return;
    const size_t part = (size_t)3 * nq + 1;
    int bi = -1, bd = 256, sd = 256;
    for (int s = 0; s < nshards; s++) {
        const int32_t *P = parts + (size_t)s * part;
        const int li = P[qi], ld = P[nq + qi], ls = P[2 * (size_t)nq + qi], off = P[3 * (size_t)nq];
        if (ld < bd) {
            sd = min(bd, ls);
            bd = ld;
            bi = li >= 0 ? li + off : -1;
        } else {
            sd = min(sd, ld);
        }
    }
    best_idx[qi] = bi;
    best_d[qi] = bd;
    second_d[qi] = sd;
}

void launch_knn2_merge(hipStream_t s, const int32_t *parts, int nshards, int nq, int32_t *best_idx, int32_t *best_d,
                       int32_t *second_d)
{
    hipLaunchKernelGGL(k_knn2_merge, dim3((nq + 255) / 256, 1, 1), dim3(256, 1, 1), 0, s, parts, nshards, nq, best_idx, best_d,
                       second_d);
}

__global__ void k_fill_i32(int32_t *p, int32_t v, int n)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) p[i] = v;
}

void launch_fill_i32(hipStream_t s, int32_t *p, int32_t v, int n)
{
    hipLaunchKernelGGL(k_fill_i32, dim3((n + 255) / 256, 1, 1), dim3(256, 1, 1), 0, s, p, v, n);
}

__global__ __launch_bounds__(256) void k_knn2_lists(const uint8_t *__restrict__ q, int nq,
                                                    const uint8_t *__restrict__ db,
                                                    const int32_t *__restrict__ off,
                                                    const int32_t *__restrict__ cand,
                                                    int32_t *__restrict__ best_idx,
                                                    int32_t *__restrict__ best_d,
                                                    int32_t *__restrict__ second_d)
{
    const int qi = blockIdx.x * 256 + threadIdx.x;
    if (qi >= nq) return;
    uint32_t Q[8];
    const uint4 a = reinterpret_cast<const uint4 *>(q + (size_t)qi * 32)[0];
    const uint4 b = reinterpret_cast<const uint4 *>(q + (size_t)qi * 32)[1];
    Q[0] = a.x; Q[1] = a.y; Q[2] = a.z; Q[3] = a.w;
    Q[4] = b.x; Q[5] = b.y; Q[6] = b.z; Q[7] = b.w;
    Best B = {256, -1, 256};
    for (int t = off[qi]; t < off[qi + 1]; t++) {
        const int j = cand[t];
        const uint4 r0 = reinterpret_cast<const uint4 *>(db + (size_t)j * 32)[0];
        const uint4 r1 = reinterpret_cast<const uint4 *>(db + (size_t)j * 32)[1];
        const uint32_t R[8] = {r0.x, r0.y, r0.z, r0.w, r1.x, r1.y, r1.z, r1.w};
        best_update(B, hamming256(Q, R), j);
    }
    best_idx[qi] = B.idx;
    best_d[qi] = B.b1;
    second_d[qi] = B.b2;
}

void launch_knn2_lists(hipStream_t s, const uint8_t *q, int nq, const uint8_t *db, const int32_t *off,
                       const int32_t *cand, int32_t *best_idx, int32_t *best_d, int32_t *second_d)
{
    if (nq <= 0) return;
    hipLaunchKernelGGL(k_knn2_lists, dim3((nq + 255) / 256, 1, 1), dim3(256, 1, 1), 0, s, q, nq, db, off, cand,
                       best_idx, best_d, second_d);
}

// ---- SearchByBoW ----
#define BOW_CLAIM_BITS 4096  // claimed-flags kept in LDS per wave (side-2 features of one node)

// (b1, pos, b2) merge for two disjoint candidate sets in arbitrary order; ties -> lower position.
__device__ __forceinline__ void bow_merge(int &b1, int &pos, int &b2, int ob1, int opos, int ob2)
{
    const bool mine = (b1 < ob1) || (b1 == ob1 && pos < opos);
    const int nb1 = mine ? b1 : ob1;
    const int npos = mine ? pos : opos;
    const int nb2 = mine ? min(b2, ob1) : min(ob2, b1);
    b1 = nb1;
    pos = npos;
    b2 = nb2;
}

__global__ __launch_bounds__(256) void k_bow_match(const uint8_t *__restrict__ desc1,
                                                   const uint8_t *__restrict__ valid1,
                                                   const int32_t *__restrict__ off1,
                                                   const int32_t *__restrict__ idx1,
                                                   const uint8_t *__restrict__ desc2,
                                                   const uint8_t *__restrict__ valid2,
                                                   const int32_t *__restrict__ off2,
                                                   const int32_t *__restrict__ idx2,
                                                   const int2 *__restrict__ pairs, int npairs, int th,
                                                   int th_mode, float nnratio, int32_t *__restrict__ match12,
                                                   int32_t *__restrict__ match21)
{
    __shared__ uint32_t s_claim[4][BOW_CLAIM_BITS / 32];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int pi = blockIdx.x * 4 + wave;
    if (pi >= npairs) return;
    const int2 pr = pairs[pi];
    const int a0 = off1[pr.x], a1 = off1[pr.x + 1];
    const int b0 = off2[pr.y], b1e = off2[pr.y + 1];
    const int n2 = b1e - b0;
    if (n2 <= 0 || a1 <= a0) return;
    if (n2 <= 128) {
        // The usual node (a dozen features a side): the side-2 descriptors and the node's side-1 indices are loaded ONCE into
        // registers (lane p holds candidates p and p + 64), the claimed flags are lane-local, and the descriptor of the next
        // side-1 feature is requested while the current one is reduced -- a side-1 feature then costs one 6-step wave
        // reduction instead of two dependent trips to memory (a single SearchByBoW(KF, F) call: 97 -> ~25 us).
        uint32_t R[2][8];
        int I2[2];
        bool live[2];
#pragma unroll
        for (int c = 0; c < 2; c++) {
            const int p = c * 64 + lane;
            live[c] = p < n2;
            I2[c] = idx2[b0 + min(p, n2 - 1)];
            if (valid2 && !valid2[I2[c]]) live[c] = false;
            const uint4 r0 = reinterpret_cast<const uint4 *>(desc2 + (size_t)I2[c] * 32)[0];
            const uint4 r1 = reinterpret_cast<const uint4 *>(desc2 + (size_t)I2[c] * 32)[1];
            R[c][0] = r0.x; R[c][1] = r0.y; R[c][2] = r0.z; R[c][3] = r0.w;
            R[c][4] = r1.x; R[c][5] = r1.y; R[c][6] = r1.z; R[c][7] = r1.w;
        }
        for (int abase = a0; abase < a1; abase += 64) {
            const int n1c = min(64, a1 - abase);
            const int myI1 = idx1[abase + min(lane, n1c - 1)];
            const int myV1 = valid1[myI1];
            // descriptor of the chunk's first feature, then always one ahead
            int i1n = __builtin_amdgcn_readlane(myI1, 0);
            uint4 q0n = reinterpret_cast<const uint4 *>(desc1 + (size_t)i1n * 32)[0];
            uint4 q1n = reinterpret_cast<const uint4 *>(desc1 + (size_t)i1n * 32)[1];
            for (int k = 0; k < n1c; k++) {
                const int i1 = i1n;
                const uint32_t Q[8] = {q0n.x, q0n.y, q0n.z, q0n.w, q1n.x, q1n.y, q1n.z, q1n.w};
                const int v1 = __builtin_amdgcn_readlane(myV1, k);   // (k is wave-uniform: v_readlane, not a trip through ds_bpermute)
                if (k + 1 < n1c) {
                    i1n = __builtin_amdgcn_readlane(myI1, k + 1);
                    q0n = reinterpret_cast<const uint4 *>(desc1 + (size_t)i1n * 32)[0];
                    q1n = reinterpret_cast<const uint4 *>(desc1 + (size_t)i1n * 32)[1];
                }
                if (!v1) continue;   // wave-uniform
                int bd1 = 256, bpos = 0x7FFFFFFF, bd2 = 256;
#pragma unroll
                for (int c = 0; c < 2; c++) {
                    if (live[c]) {
                        const int d = hamming256(Q, R[c]);
                        if (d < bd1) {
                            bd2 = bd1;
                            bd1 = d;
                            bpos = c * 64 + lane;
                        } else if (d < bd2) {
                            bd2 = d;
                        }
                    }
                }
                {
                    // (best distance, lowest position) and the second smallest distance of the whole node: two wave minima --
                    // the winner by (distance, position), then every lane's best, the winner lane's second (r04: the butterfly
                    // of three shuffled values and six tie-aware merges was two thirds of a side-1 feature's instructions, and this
                    // loop is the serial chain the call waits for)
                    const int key = bpos != 0x7FFFFFFF ? ((bd1 << 8) | bpos) : 0x7FFFFFFF;
                    const int k1 = orb_wave_min_i(key);
                    const int k2 = orb_wave_min_i(key == k1 ? bd2 : bd1);
                    bd1 = k1 != 0x7FFFFFFF ? k1 >> 8 : 256;
                    bpos = k1 != 0x7FFFFFFF ? (k1 & 255) : 0x7FFFFFFF;
                    bd2 = k2;
                }
                const bool pass = th_mode ? (bd1 < th) : (bd1 <= th);
                if (pass && (float)bd1 < nnratio * (float)bd2) {  // ref: :228-230 / :598-600
#pragma unroll
                    for (int c = 0; c < 2; c++)
                        if (bpos == c * 64 + lane) {
                            live[c] = false;            // claimed (:233 / :603)
                            match12[i1] = I2[c];
                            match21[I2[c]] = i1;
                        }
                }
            }
        }
        return;
    }
    uint32_t *claim = s_claim[wave];
    const bool useLds = n2 <= BOW_CLAIM_BITS;
    if (useLds)
        for (int k = lane; k < (n2 + 31) / 32; k += 64) claim[k] = 0;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");

    for (int a = a0; a < a1; a++) {
        const int i1 = idx1[a];
        if (!valid1[i1]) continue;  // wave-uniform
        uint32_t Q[8];
        {
            const uint32_t *row = reinterpret_cast<const uint32_t *>(desc1 + (size_t)i1 * 32);
#pragma unroll
            for (int k = 0; k < 8; k++) Q[k] = row[k];
        }
        int bd1 = 256, bpos = 0x7FFFFFFF, bd2 = 256;
        for (int p = lane; p < n2; p += 64) {
            const int i2 = idx2[b0 + p];
            bool skip;
            if (useLds)
                skip = (claim[p >> 5] >> (p & 31)) & 1u;
            else
                skip = __atomic_load_n(&match21[i2], __ATOMIC_RELAXED) >= 0;
            if (valid2 && !valid2[i2]) skip = true;
            if (skip) continue;
            const uint4 r0 = reinterpret_cast<const uint4 *>(desc2 + (size_t)i2 * 32)[0];
            const uint4 r1 = reinterpret_cast<const uint4 *>(desc2 + (size_t)i2 * 32)[1];
            const uint32_t R[8] = {r0.x, r0.y, r0.z, r0.w, r1.x, r1.y, r1.z, r1.w};
            const int d = hamming256(Q, R);
            if (d < bd1) {
                bd2 = bd1;
                bd1 = d;
                bpos = p;
            } else if (d < bd2) {
                bd2 = d;
            }
        }
        {
            const int key = bpos != 0x7FFFFFFF ? ((bd1 << 20) | bpos) : 0x7FFFFFFF;   // (a FeatureVector of < 2^20 entries: orbhip_search_by_bow / _sets check)
            const int k1 = orb_wave_min_i(key);
            const int k2 = orb_wave_min_i(key == k1 ? bd2 : bd1);
            bd1 = k1 != 0x7FFFFFFF ? k1 >> 20 : 256;
            bpos = k1 != 0x7FFFFFFF ? (k1 & 0xFFFFF) : 0x7FFFFFFF;
            bd2 = k2;
        }
        const bool pass = th_mode ? (bd1 < th) : (bd1 <= th);
        if (pass && (float)bd1 < nnratio * (float)bd2) {  // ref: :228-230 / :598-600
            const int i2 = idx2[b0 + bpos];
            if (lane == 0) {
                match12[i1] = i2;
                if (useLds)
                    claim[bpos >> 5] |= 1u << (bpos & 31);
                __atomic_store_n(&match21[i2], i1, __ATOMIC_RELAXED);
            }
            if (!useLds) __threadfence();
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        }
    }
}

void launch_bow_match(hipStream_t s, const uint8_t *desc1, const uint8_t *valid1, const int32_t *off1,
                      const int32_t *idx1, const uint8_t *desc2, const uint8_t *valid2,
                      const int32_t *off2, const int32_t *idx2, const int32_t *pairs, int npairs,
                      int th, int th_mode, float nnratio, int32_t *match12, int32_t *match21)
{
    if (npairs <= 0) return;
    hipLaunchKernelGGL(k_bow_match, dim3((npairs + 3) / 4, 1, 1), dim3(256, 1, 1), 0, s, desc1, valid1, off1,
                       idx1, desc2, valid2, off2, idx2, reinterpret_cast<const int2 *>(pairs), npairs, th,
                       th_mode, nnratio, match12, match21);
}

// ---- ORBmatcher::SearchForTriangulation (ref: src/ORBmatcher.cc:657-827, CheckDistEpipolarLine :140-157) ----
// One wave per shared vocabulary node; the node's side-1 features one after the other, the side-2 features one per
// lane.  A candidate counts when its distance is <= TH_LOW, it is not too close to the epipole (mono pairs) and it lies
// within 3.84 sigma2 of the epipolar line; among those the reference keeps the smallest distance and, at equal
// distance, the LAST one in the node's list (":719 dist>bestDist continue").  This fork never marks a side-2 feature as
// taken, so the side-1 features are independent of each other.  Float expressions in the reference's order, no
// contraction; the final comparison is in double as in the reference (3.84 is a double literal).
struct TriParams {
    float F[9];
    float ex, ey;
    int only_stereo, th_low;
};

__global__ __launch_bounds__(256) void k_tri_match(const orbhip_keypoint *__restrict__ kps1,
                                                   const uint8_t *__restrict__ desc1, const uint8_t *__restrict__ skip1,
                                                   const float *__restrict__ ur1, const int32_t *__restrict__ off1,
                                                   const int32_t *__restrict__ idx1,
                                                   const orbhip_keypoint *__restrict__ kps2,
                                                   const uint8_t *__restrict__ desc2, const uint8_t *__restrict__ skip2,
                                                   const float *__restrict__ ur2, const int32_t *__restrict__ off2,
                                                   const int32_t *__restrict__ idx2, const int2 *__restrict__ pairs,
                                                   int npairs, const TriParams P, const float *__restrict__ scale2,
                                                   const float *__restrict__ sigma2, int32_t *__restrict__ match12)
{
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int pi = blockIdx.x * 4 + wave;
    if (pi >= npairs) return;
    const int2 pr = pairs[pi];
    const int a0 = off1[pr.x], a1 = off1[pr.x + 1];
    const int b0 = off2[pr.y], n2 = off2[pr.y + 1] - b0;
    if (n2 <= 0 || a1 <= a0) return;
    if (n2 <= 128) {
        // The usual node: the side-2 features (descriptor, position, octave thresholds) are loaded ONCE into registers, lane p
        // holding candidates p and p + 64; the side-1 features of the node are loaded 64 at a time, one per lane, and
        // broadcast; the next side-1 descriptor is requested while the current one is reduced.  (A call for one key-frame
        // pair was a chain of dependent trips to memory per side-1 feature: ~90 us for ~100 nodes of ten features.)
        uint32_t R[2][8];
        float x2[2], y2[2], thr2[2];
        double lim2[2];
        bool live[2], st2[2];
        int I2[2];
#pragma unroll
        for (int c = 0; c < 2; c++) {
            const int p = c * 64 + lane;
            live[c] = p < n2;
            I2[c] = idx2[b0 + min(p, n2 - 1)];
            if (skip2[I2[c]]) live[c] = false;
            st2[c] = ur2 && ur2[I2[c]] >= 0.f;
            if (P.only_stereo && !st2[c]) live[c] = false;
            const uint4 r0 = reinterpret_cast<const uint4 *>(desc2 + (size_t)I2[c] * 32)[0];
            const uint4 r1 = reinterpret_cast<const uint4 *>(desc2 + (size_t)I2[c] * 32)[1];
            R[c][0] = r0.x; R[c][1] = r0.y; R[c][2] = r0.z; R[c][3] = r0.w;
            R[c][4] = r1.x; R[c][5] = r1.y; R[c][6] = r1.z; R[c][7] = r1.w;
            const orbhip_keypoint k2 = kps2[I2[c]];
            x2[c] = k2.x;
            y2[c] = k2.y;
            thr2[c] = __fmul_rn(100.f, scale2[k2.octave]);
            lim2[c] = __dmul_rn(3.84, (double)sigma2[k2.octave]);
        }
        for (int abase = a0; abase < a1; abase += 64) {
            const int n1c = min(64, a1 - abase);
            const int myI1 = idx1[abase + min(lane, n1c - 1)];
            const int mySkip = skip1[myI1];
            const int mySt1 = (ur1 && ur1[myI1] >= 0.f) ? 1 : 0;
            const float myX1 = kps1[myI1].x, myY1 = kps1[myI1].y;
            int i1n = __builtin_amdgcn_readlane(myI1, 0);
            uint4 q0n = reinterpret_cast<const uint4 *>(desc1 + (size_t)i1n * 32)[0];
            uint4 q1n = reinterpret_cast<const uint4 *>(desc1 + (size_t)i1n * 32)[1];
            for (int k = 0; k < n1c; k++) {
                const int i1 = i1n;
                const uint32_t Q[8] = {q0n.x, q0n.y, q0n.z, q0n.w, q1n.x, q1n.y, q1n.z, q1n.w};
                const int skip = __builtin_amdgcn_readlane(mySkip, k);   // (k is wave-uniform: v_readlane, not ds_bpermute)
                const bool stereo1 = __builtin_amdgcn_readlane(mySt1, k) != 0;
                const float x1 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(myX1), k));
                const float y1 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(myY1), k));
                if (k + 1 < n1c) {
                    i1n = __builtin_amdgcn_readlane(myI1, k + 1);
                    q0n = reinterpret_cast<const uint4 *>(desc1 + (size_t)i1n * 32)[0];
                    q1n = reinterpret_cast<const uint4 *>(desc1 + (size_t)i1n * 32)[1];
                }
                if (skip) continue;                              // wave-uniform
                if (P.only_stereo && !stereo1) continue;
                const float la = __fadd_rn(__fadd_rn(__fmul_rn(x1, P.F[0]), __fmul_rn(y1, P.F[3])), P.F[6]);
                const float lb = __fadd_rn(__fadd_rn(__fmul_rn(x1, P.F[1]), __fmul_rn(y1, P.F[4])), P.F[7]);
                const float lc = __fadd_rn(__fadd_rn(__fmul_rn(x1, P.F[2]), __fmul_rn(y1, P.F[5])), P.F[8]);
                const float den = __fadd_rn(__fmul_rn(la, la), __fmul_rn(lb, lb));
                int key = 0x7FFFFFFF;
#pragma unroll
                for (int c = 0; c < 2; c++) {
                    if (!live[c]) continue;
                    const int d = hamming256(Q, R[c]);
                    if (d > P.th_low) continue;
                    if (!stereo1 && !st2[c]) {
                        const float dx = __fsub_rn(P.ex, x2[c]), dy = __fsub_rn(P.ey, y2[c]);
                        if (__fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy)) < thr2[c]) continue;
                    }
                    const float num = __fadd_rn(__fadd_rn(__fmul_rn(la, x2[c]), __fmul_rn(lb, y2[c])), lc);
                    if (den == 0.f) continue;
                    const float dsqr = __fdiv_rn(__fmul_rn(num, num), den);
                    if (!((double)dsqr < lim2[c])) continue;
                    key = min(key, (d << 16) | (0xFFFF - (c * 64 + lane)));
                }
                key = orb_wave_min_i(key);
                if (key != 0x7FFFFFFF) {
                    const int pos = 0xFFFF - (key & 0xFFFF);
#pragma unroll
                    for (int c = 0; c < 2; c++)
                        if (pos == c * 64 + lane) match12[i1] = I2[c];
                }
            }
        }
        return;
    }
    for (int a = a0; a < a1; a++) {
        const int i1 = idx1[a];
        if (skip1[i1]) continue;   // wave-uniform
        const bool stereo1 = ur1 && ur1[i1] >= 0.f;
        if (P.only_stereo && !stereo1) continue;
        const float x1 = kps1[i1].x, y1 = kps1[i1].y;
        // epipolar line in the second image: l = x1' F12 = [la lb lc]
        const float la = __fadd_rn(__fadd_rn(__fmul_rn(x1, P.F[0]), __fmul_rn(y1, P.F[3])), P.F[6]);
        const float lb = __fadd_rn(__fadd_rn(__fmul_rn(x1, P.F[1]), __fmul_rn(y1, P.F[4])), P.F[7]);
        const float lc = __fadd_rn(__fadd_rn(__fmul_rn(x1, P.F[2]), __fmul_rn(y1, P.F[5])), P.F[8]);
        const float den = __fadd_rn(__fmul_rn(la, la), __fmul_rn(lb, lb));
        uint32_t Q[8];
        {
            const uint32_t *row = reinterpret_cast<const uint32_t *>(desc1 + (size_t)i1 * 32);
#pragma unroll
            for (int k = 0; k < 8; k++) Q[k] = row[k];
        }
        int key = 0x7FFFFFFF;   // distance << 16 | (0xFFFF - list position): smallest distance, then the last position
        for (int p = lane; p < n2; p += 64) {
            const int i2 = idx2[b0 + p];
            if (skip2[i2]) continue;
            const bool stereo2 = ur2 && ur2[i2] >= 0.f;
            if (P.only_stereo && !stereo2) continue;
            const uint4 r0 = reinterpret_cast<const uint4 *>(desc2 + (size_t)i2 * 32)[0];
            const uint4 r1 = reinterpret_cast<const uint4 *>(desc2 + (size_t)i2 * 32)[1];
            const uint32_t R[8] = {r0.x, r0.y, r0.z, r0.w, r1.x, r1.y, r1.z, r1.w};
            const int d = hamming256(Q, R);
            if (d > P.th_low) continue;
            const orbhip_keypoint k2 = kps2[i2];
            if (!stereo1 && !stereo2) {
                const float dx = __fsub_rn(P.ex, k2.x), dy = __fsub_rn(P.ey, k2.y);
                if (__fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy)) < __fmul_rn(100.f, scale2[k2.octave])) continue;
            }
            const float num = __fadd_rn(__fadd_rn(__fmul_rn(la, k2.x), __fmul_rn(lb, k2.y)), lc);
            if (den == 0.f) continue;
            const float dsqr = __fdiv_rn(__fmul_rn(num, num), den);
            if (!((double)dsqr < __dmul_rn(3.84, (double)sigma2[k2.octave]))) continue;
            key = min(key, (d << 16) | (0xFFFF - min(p, 0xFFFF)));
        }
        key = orb_wave_min_i(key);
        if (lane == 0 && key != 0x7FFFFFFF) match12[i1] = idx2[b0 + (0xFFFF - (key & 0xFFFF))];
    }
}

void launch_tri_match(hipStream_t s, const orbhip_keypoint *kps1, const uint8_t *desc1, const uint8_t *skip1, const float *ur1,
                      const int32_t *off1, const int32_t *idx1, const orbhip_keypoint *kps2, const uint8_t *desc2,
                      const uint8_t *skip2, const float *ur2, const int32_t *off2, const int32_t *idx2, const int32_t *pairs,
                      int npairs, const float F12[9], float ex, float ey, int only_stereo, int th_low, const float *scale2,
                      const float *sigma2, int32_t *match12)
{
    if (npairs <= 0) return;
    TriParams P;
    for (int i = 0; i < 9; i++) P.F[i] = F12[i];
    P.ex = ex;
    P.ey = ey;
    P.only_stereo = only_stereo;
    P.th_low = th_low;
    hipLaunchKernelGGL(k_tri_match, dim3((npairs + 3) / 4, 1, 1), dim3(256, 1, 1), 0, s, kps1, desc1, skip1, ur1, off1, idx1, kps2,
                       desc2, skip2, ur2, off2, idx2, reinterpret_cast<const int2 *>(pairs), npairs, P, scale2, sigma2, match12);
}

// ---- MapPoint::ComputeDistinctiveDescriptors (ref: src/MapPoint.cc:283-349) for many points at once ----------------
// Sixteen lanes per point (most points have 2..16 observers, so four points share a wave): lane i owns row i of the N x N
// distance matrix (rows beyond 16 in further rounds).  The median the
// reference reads, sorted_row[(size_t)(0.5 * (N - 1))], is the k-th smallest of the row; distances are integers in
// 0..256, so it is found by bisection on the value (9 counts of "row entries <= v", the row's distances recomputed from
// the descriptors each time: they are broadcast reads of a few hundred bytes, cheaper than parking rows in LDS and
// without a bound on N).  The first row of least median wins: min over median << 20 | row within the 16 lanes.
__global__ __launch_bounds__(256) void k_distinctive(const uint8_t *__restrict__ desc, const int32_t *__restrict__ off, int P,
                                                     int32_t *__restrict__ best, int32_t *__restrict__ bestMedian)
{
    const int p = blockIdx.x * 16 + (threadIdx.x >> 4), lane = threadIdx.x & 15;
    if (p >= P) return;
    const int s = off[p], N = off[p + 1] - s;
    if (N <= 0) {
        if (lane == 0) {
            best[p] = -1;
            bestMedian[p] = 0x7fffffff;
        }
        return;
    }
    const uint4 *D = reinterpret_cast<const uint4 *>(desc) + (size_t)s * 2;
    const int k = (int)(0.5 * (double)(N - 1));
    unsigned key = 0xffffffffu;
    for (int i0 = 0; i0 < N; i0 += 16) {
        const int i = i0 + lane;
        if (i < N) {
            const uint4 a0 = D[2 * i], a1 = D[2 * i + 1];
            int lo = 0, hi = 256;
            while (lo < hi) {
                const int mid = (lo + hi) >> 1;
                int cnt = 0;
                for (int j = 0; j < N; j++) {
                    const uint4 b0 = D[2 * j], b1 = D[2 * j + 1];
                    const int d = __popc(a0.x ^ b0.x) + __popc(a0.y ^ b0.y) + __popc(a0.z ^ b0.z) + __popc(a0.w ^ b0.w) +
                                  __popc(a1.x ^ b1.x) + __popc(a1.y ^ b1.y) + __popc(a1.z ^ b1.z) + __popc(a1.w ^ b1.w);
                    cnt += d <= mid;
                }
                if (cnt > k) hi = mid;
                else lo = mid + 1;
            }
            key = min(key, ((unsigned)lo << 20) | (unsigned)i);
        }
    }
#pragma unroll
    for (int d = 8; d >= 1; d >>= 1) key = min(key, (unsigned)__shfl_xor((int)key, d));
    if (lane == 0) {
        best[p] = (int)(key & 0xfffffu);
        bestMedian[p] = (int)(key >> 20);
    }
}

void launch_distinctive(hipStream_t s, const uint8_t *desc, const int32_t *off, int P, int32_t *best, int32_t *bestMedian)
{
    if (P <= 0) return;
    hipLaunchKernelGGL(k_distinctive, dim3((P + 15) / 16, 1, 1), dim3(256, 1, 1), 0, s, desc, off, P, best, bestMedian);
}
