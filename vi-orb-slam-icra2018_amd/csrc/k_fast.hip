// k_fast.hip -- E3 + E3f: per-cell FAST-9/16 with two thresholds, for all levels and all frames
// of a batch in ONE launch (ref: src/ORBextractor.cc:767-831; cv::FAST TYPE_9_16 with
// non-max suppression, OpenCV 2.4 fast.cpp / fast_score.cpp).
//
// Whole-level formulation: the detection domains of the reference's
// overlapping 36x36 sub-images tile the level without overlap, the FAST score does not depend on
// the detection threshold and "corner at t" <=> "score >= t".  So per pixel we compute the score
// once (0 when < minThFAST), suppress non-maxima among the 8 neighbours THAT BELONG TO THE SAME
// CELL (other cells' pixels count as 0, exactly like the zeroed score rows/columns outside a
// sub-image), and per cell keep the survivors >= iniThFAST if there is any, else all survivors.
//
// One 256-thread workgroup per (frame, run of <= FAST_TILE_CELLS cells of one cell-row):
//   1. stage the run's pixels (+3 px halo) into LDS with 16-byte row-coalesced loads;
//   2. compass pre-test on EVERY domain pixel, 4 pixels per lane from aligned LDS dwords, in packed
//      16-bit arithmetic (v_pk_max/min/sub_u16, 2 pixels per instruction): a 9-arc always contains
//      ring pixel 0 or 8 and ring pixel 4 or 12, so a corner needs
//      min(max(q0,q8), max(q4,q12)) > v + t   or   max(min(q0,q8), min(q4,q12)) < v - t.
//      (the two conditions are sign bits of wrapped 16-bit differences).  Survivors (18 % of the pixels at
//      level 0, 60 % at level 7) are appended to an LDS work list;
//   3. full score on the work list, dense lanes, on packed halves: the ring as 8 registers (q_k, q_k+8) and the
//      gfx950 three-input packed minimum / maximum (fast_score_pol below), one polarity unless both are
//      possible; corners (score >= minThFAST) are compacted into a corner list;
//   4. non-max suppression over the corner list only; survivors get a sortable key
//      (cell, row, column, score) in a survivor list, plus a per-cell ">= iniThFAST" flag;
//   5. survivors that pass their cell's threshold set a bit in a per-(cell, row) bitmap; a survivor's rank
//      is the number of bits before it (row prefix + popcount), and it is written to that slot of the
//      cell's fixed range: cells row-major, raster inside a cell = the reference's candidate order.  No
//      global atomics, deterministic.
// HBM traffic: each level pixel inside [16, w-16) x [16, h-16) is read once per tile that needs it;
// the 6-row vertical halo (hCell ~ 30) is re-read by the tile below.  Roofline: nominally HBM read
// (algorithmic bytes = sum_l (w_l-32)(h_l-32) per frame); measured bound is integer VALU + LDS
// (DESIGN.md section 4).
#include "orbhip_internal.h"

#include <cstdlib>

typedef unsigned short us2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ const uint8_t *level_ptr(const OrbLevels &G, int l, int frame,
                                                    const uint8_t *lvl0, int stride0,
                                                    unsigned long long frame0, const uint8_t *pyr,
                                                    unsigned long long pyrFrame, int &stride)
{
    if (l == 0) {
        stride = stride0;
        return lvl0 + (size_t)frame * frame0;
    }
    stride = G.lv[l].stride;
    return pyr + (size_t)frame * pyrFrame + G.lv[l].imgOff;
}

// bytes (0,1) / (2,3) of w zero-extended into the two 16-bit halves
__device__ __forceinline__ us2 lo2(uint32_t w)
{
    return __builtin_bit_cast(us2, __builtin_amdgcn_perm(0u, w, 0x0c010c00u));
}
__device__ __forceinline__ us2 hi2(uint32_t w)
{
    return __builtin_bit_cast(us2, __builtin_amdgcn_perm(0u, w, 0x0c030c02u));
}

// Compass test of two pixels: bit 15 of a half is set <=> that pixel can be a FAST-9 corner at threshold t (the
// differences are below 2^15 in magnitude, so the wrapped 16-bit difference carries the sign).
__device__ __forceinline__ uint32_t compass2(us2 v, us2 qt, us2 qb, us2 ql, us2 qr, us2 tt)
{
    const us2 mb = __builtin_elementwise_min(__builtin_elementwise_max(qt, qb), __builtin_elementwise_max(ql, qr));
    const us2 md = __builtin_elementwise_max(__builtin_elementwise_min(qt, qb), __builtin_elementwise_min(ql, qr));
    const us2 hi = v + tt;
    const us2 lo = __builtin_elementwise_sub_sat(v, tt);
    const us2 f = (us2)(hi - mb) | (us2)(md - lo);   // mb > v + t  or  md < v - t  (lo = max(v - t, 0); md < 0 never holds)
    return __builtin_bit_cast(uint32_t, f);
}

// ---- score of one polarity on packed halves (gfx950: v_pk_minimum3_f16 / v_pk_maximum3_f16) ----
// The 16 ring pixels sit as 8 registers P[k] = (q_k, q_{k+8}) whose halves are the pixel values with bit 14 set:
// as binary16 bit patterns these are the normal numbers 2.0 .. 2.498, ordered like the integers, so the packed
// three-input float minimum / maximum order them exactly.  op_sel feeds a register with its halves exchanged,
// which is how "index + 8" is read.
#define PKOP3(name, op, sel)                                                                            \
    __device__ __forceinline__ uint32_t name(uint32_t a, uint32_t b, uint32_t c)                      \
    {                                                                                                   \
        uint32_t r;                                                                                     \
        asm("v_pk_" op "_f16 %0, %1, %2, %3" sel : "=v"(r) : "v"(a), "v"(b), "v"(c));                   \
        return r;                                                                                       \
    }
PKOP3(pkmin3, "minimum3", "")
PKOP3(pkmin3_x3, "minimum3", " op_sel:[0,0,1] op_sel_hi:[1,1,0]")      // third operand with exchanged halves
PKOP3(pkmin3_x23, "minimum3", " op_sel:[0,1,1] op_sel_hi:[1,0,0]")    // second and third
PKOP3(pkmax3, "maximum3", "")
PKOP3(pkmax3_x2, "maximum3", " op_sel:[0,1,0] op_sel_hi:[1,0,1]")

// max over the 16 cyclic 9-arcs of the arc minimum of the ring P (pairs (k, k + 8)) after P ^= C, minus v, minus 1
__device__ __forceinline__ int arcs_score(const uint32_t P[8], uint32_t C, int v)
{
    uint32_t Q[8];
#pragma unroll
    for (int k = 0; k < 8; k++) Q[k] = P[k] ^ C;
    // minima of 3 consecutive ring pixels: M[k] = (m3[k], m3[k + 8])
    uint32_t M[8];
#pragma unroll
    for (int k = 0; k < 6; k++) M[k] = pkmin3(Q[k], Q[k + 1], Q[k + 2]);
    M[6] = pkmin3_x3(Q[6], Q[7], Q[0]);
    M[7] = pkmin3_x23(Q[7], Q[0], Q[1]);
    // minima of the 9-arcs: X[k] = (arc starting at k, arc starting at k + 8)
    uint32_t X[8];
    X[0] = pkmin3(M[0], M[3], M[6]);
    X[1] = pkmin3(M[1], M[4], M[7]);
    X[2] = pkmin3_x3(M[2], M[5], M[0]);
    X[3] = pkmin3_x3(M[3], M[6], M[1]);
    X[4] = pkmin3_x3(M[4], M[7], M[2]);
    X[5] = pkmin3_x23(M[5], M[0], M[3]);
    X[6] = pkmin3_x23(M[6], M[1], M[4]);
    X[7] = pkmin3_x23(M[7], M[2], M[5]);
    const uint32_t R0 = pkmax3(X[0], X[1], X[2]), R1 = pkmax3(X[3], X[4], X[5]), R2 = pkmax3(X[6], X[7], X[7]);
    const uint32_t R = pkmax3(R0, R1, R2);
    const uint32_t Rm = pkmax3_x2(R, R, R);                      // both halves = max(low, high)
    return (int)(Rm & 0xFFu) - v - 1;
}

// FAST score of the pixel at p, 0 if it is not a corner at threshold t (t >= 1).  bright (q - v > t): max_arcs min_arc (q - v) - 1 = (max_arcs min_arc q) - v - 1; dark: the same
// on the complemented bytes, v - q = (255 - q) - (255 - v).
__device__ __forceinline__ int fast_score_pol(const uint8_t *p, int pitch, int t)
{
    const uint8_t *rm3 = p - 3 * pitch - 3, *rm2 = p - 2 * pitch - 3, *rm1 = p - pitch - 3, *r0 = p - 3;
    const uint8_t *rp1 = p + pitch - 3, *rp2 = p + 2 * pitch - 3, *rp3 = p + 3 * pitch - 3;
    // ring pixel k (OpenCV's order: k = 0 at (0, +3), then clockwise in image coordinates) paired with pixel k + 8
    const int v0 = p[0];
    const int q0 = rp3[3], q8 = rm3[3], q4 = r0[6], q12 = r0[0];
    // polarity that can hold a 9-arc (it contains ring pixel 0 or 8 and ring pixel 4 or 12)
    const bool pb = min(max(q0, q8), max(q4, q12)) > v0 + t;    // bright: q - v > t
    const bool pd = max(min(q0, q8), min(q4, q12)) < v0 - t;    // dark:   v - q > t
    if (!pb && !pd) return 0;
    const bool dark = !pb;
    const uint32_t C = dark ? 0x40FF40FFu : 0x40004000u;
    uint32_t P[8];
    P[0] = ((uint32_t)q0 | ((uint32_t)q8 << 16));               // (0, +3)  | (0, -3)
    P[1] = ((uint32_t)rp3[4] | ((uint32_t)rm3[2] << 16));       // (+1, +3) | (-1, -3)
    P[2] = ((uint32_t)rp2[5] | ((uint32_t)rm2[1] << 16));       // (+2, +2) | (-2, -2)
    P[3] = ((uint32_t)rp1[6] | ((uint32_t)rm1[0] << 16));       // (+3, +1) | (-3, -1)
    P[4] = ((uint32_t)q4 | ((uint32_t)q12 << 16));              // (+3, 0)  | (-3, 0)
    P[5] = ((uint32_t)rm1[6] | ((uint32_t)rp1[0] << 16));       // (+3, -1) | (-3, +1)
    P[6] = ((uint32_t)rm2[5] | ((uint32_t)rp2[1] << 16));       // (+2, -2) | (-2, +2)
    P[7] = ((uint32_t)rm3[4] | ((uint32_t)rp3[2] << 16));       // (+1, -3) | (-1, +3)
    int sc = arcs_score(P, C, v0 ^ (dark ? 0xFF : 0));
    if (pb && pd) sc = max(sc, arcs_score(P, 0x40FF40FFu, v0 ^ 0xFF));   // both possible (rare): the dark one as well
    return sc >= t ? sc : 0;
}

// inclusive wave prefix sum
__device__ __forceinline__ int wave_incl_scan(int v, int lane)
{
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int u = __shfl_up(v, o);
        if (lane >= o) v += u;
    }
    return v;
}

// inclusive wave prefix sum on the DPP network (row shifts, then the row totals of the lower rows)
__device__ __forceinline__ int wave_incl_scan_dpp(int v)
{
    v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xF, 0xF, true);   // row_shr:1
    v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xF, 0xF, true);   // row_shr:2
    v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xF, 0xF, true);   // row_shr:4
    v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xF, 0xF, true);   // row_shr:8
    const int t0 = __builtin_amdgcn_readlane(v, 15), t1 = __builtin_amdgcn_readlane(v, 31), t2 = __builtin_amdgcn_readlane(v, 47);
    const int row = (int)(threadIdx.x & 63) >> 4;
    return v + (row > 0 ? t0 : 0) + (row > 1 ? t1 : 0) + (row > 2 ? t2 : 0);
}

// Append the lanes with `flag` set to an LDS list (order irrelevant); returns the slot or -1.
__device__ __forceinline__ int wave_append(bool flag, int *counter, int lane)
{
    const unsigned long long m = __ballot(flag);
    if (m == 0) return -1;
    const int n = __popcll(m);
    const int leader = __ffsll((long long)m) - 1;
    int base = 0;
    if (lane == leader) base = atomicAdd(counter, n);
    base = __shfl(base, leader);
    return flag ? base + __popcll(m & ((1ull << lane) - 1ull)) : -1;
}

// Non-max suppression of the corner at (r, c) of the tile's score map against its 8 neighbours that
// lie in the same cell; returns the survivor's sortable key (cell | row | column | score).
__device__ __forceinline__ uint32_t nms_key(const uint8_t *s_score, int SP, int r, int c, int DH, int TW, int wCell,
                                            unsigned cellMagic, int iniTh, int *s_cellAny, bool &surv)
{
    const uint8_t *sp = s_score + r * SP + c;
    const int s = sp[0];
    const int cj = (int)(((unsigned)c * cellMagic) >> 16);
    const int cx0 = cj * wCell;
    int cx1 = cx0 + wCell;
    if (cx1 > TW) cx1 = TW;
    const bool up = r > 0, dn = r < DH - 1, lf = c > cx0, rt = c < cx1 - 1;
    int m = 0;
    if (lf) m = max(m, (int)sp[-1]);
    if (rt) m = max(m, (int)sp[1]);
    if (up) {
        m = max(m, (int)sp[-SP]);
        if (lf) m = max(m, (int)sp[-SP - 1]);
        if (rt) m = max(m, (int)sp[-SP + 1]);
    }
    if (dn) {
        m = max(m, (int)sp[SP]);
        if (lf) m = max(m, (int)sp[SP - 1]);
        if (rt) m = max(m, (int)sp[SP + 1]);
    }
    surv = s > m;
    if (surv && s >= iniTh) s_cellAny[cj] = 1;   // benign race: every writer stores 1
    return ((uint32_t)cj << 28) | ((uint32_t)r << 21) | ((uint32_t)c << 8) | (uint32_t)s;
}

__global__ __launch_bounds__(256) void k_fast(const OrbLevels G, const uint8_t *__restrict__ lvl0,
                                              int stride0, unsigned long long frame0,
                                              const uint8_t *__restrict__ pyr,
                                              unsigned long long pyrFrame,
                                              const FastTile *__restrict__ tiles,
                                              uint32_t *__restrict__ cand,
                                              uint16_t *__restrict__ cellCnt, int pixBytes, int scoreBytes,
                                              int listBytes, int listCap, int cornerCap, int phases, int xcdMap, int ntiles)
{
    extern __shared__ __align__(16) uint8_t smem[];
    __shared__ int s_cellAny[FAST_TILE_CELLS];
    __shared__ int s_cellCnt[FAST_TILE_CELLS];
    __shared__ int s_listCount, s_cornerCount, s_survCount;

    const int tileId = xcd_tile(xcdMap), frame = blockIdx.y;
    if (tileId >= ntiles) return;   // grid padded to a multiple of 8 (orbhip_internal.h, xcd_tile)
    const FastTile T = tiles[tileId];
    const OrbLevel &L = G.lv[T.level];
    const int tid = threadIdx.x, lane = tid & 63;

    const int maxBX = L.w - ORB_MIN_BORDER, maxBY = L.h - ORB_MIN_BORDER;
    const int iniY = ORB_MIN_BORDER + T.row * L.hCell;
    const int X0 = ORB_MIN_BORDER + T.c0 * L.wCell;
    int maxY = iniY + L.hCell + 6;
    if (maxY > maxBY) maxY = maxBY;
    int X1 = ORB_MIN_BORDER + (T.c0 + T.ncells) * L.wCell + 6;
    if (X1 > maxBX) X1 = maxBX;
    // :797-798 / :805-806 -- rows and columns the reference skips produce nothing
    const bool rowLive = iniY < maxBY - 3;
    const int DH = rowLive ? maxY - iniY - 6 : 0;       // domain rows
    const int TW = X1 - X0 - 6;                          // domain columns of the whole run
    uint16_t *cnt = cellCnt + (size_t)frame * G.totalCells + L.cellBase + T.row * L.nCols + T.c0;
    if (DH <= 0 || TW <= 0) {
        if (tid < T.ncells) cnt[tid] = 0;
        return;
    }
    int stride;
    const uint8_t *img = level_ptr(G, T.level, frame, lvl0, stride0, frame0, pyr, pyrFrame, stride);

    // ---- 1. stage pixels [iniY, maxY) x [XA, X1) into LDS, 16 bytes per lane per load ----
    const int XA = X0 & ~15;
    const int nchunk = (X1 - XA + 15) >> 4;
    const int pitch = nchunk << 4;
    const int RH = maxY - iniY;
    uint8_t *s_pix = smem;
    uint8_t *s_score = smem + pixBytes;
    uint16_t *s_list = reinterpret_cast<uint16_t *>(smem + pixBytes + scoreBytes);
    uint16_t *s_corner = reinterpret_cast<uint16_t *>(smem + pixBytes + scoreBytes + listBytes);
    // the survivor list reuses the work list's storage (the work list is dead after phase 3)
    uint32_t *s_surv = reinterpret_cast<uint32_t *>(smem + pixBytes + scoreBytes);
    const int SP = (TW + 3) & ~3;
    uint16_t *s_ent = reinterpret_cast<uint16_t *>(s_score);   // phase 2 only: u16 per item (2 * items <= DH * SP)
    const float invNchunk = 1.0f / (float)nchunk;   // i / nchunk = floor((i + 0.5) * invNchunk), exact for i < 2^16
    for (int i = tid; i < RH * nchunk; i += 256) {
        const int r = (int)(((float)i + 0.5f) * invNchunk), c = i - r * nchunk;
        const uint4 v = *reinterpret_cast<const uint4 *>(img + (size_t)(iniY + r) * stride + XA + (c << 4));
        *reinterpret_cast<uint4 *>(s_pix + r * pitch + (c << 4)) = v;
    }
    if (tid < FAST_TILE_CELLS) {
        s_cellAny[tid] = 0;
        s_cellCnt[tid] = 0;
    }
    if (tid == 0) {
        s_listCount = 0;
        s_cornerCount = 0;
        s_survCount = 0;
    }
    __syncthreads();
    if (phases < 2) return;   // timing ablation only (ORBHIP_FAST_PHASES), results are then invalid

    // ---- 2. compass pre-test, 4 pixels per item; survivors -> work list ----
    // items = (domain row, aligned dword column) pairs, dealt round-robin to the 256 threads so that
    // consecutive lanes read consecutive LDS dwords of one row (conflict-free) whatever the tile width
    const int t = G.minTh;
    const us2 tt = {(unsigned short)t, (unsigned short)t};
    const int j0 = X0 + 3 - XA;            // LDS column of domain column 0
    const int jd0 = j0 & ~3;               // first aligned dword column touching the domain
    const int GPR = ((j0 + TW + 3) >> 2) - (j0 >> 2);   // dword groups per row
    const int nitems = DH * GPR;
    const float invGPR = 1.0f / (float)GPR;   // item / GPR = floor((item + 0.5) * invGPR), exact for item < 2^16
    // domain masks of a row's first and last dword group (pixel k at bit 8k + 7)
    uint32_t domFirst = 0, domLast = 0;
    {
        const int cLast = jd0 + ((GPR - 1) << 2) - j0;
        for (int k = 0; k < 4; k++) {
            if (jd0 - j0 + k >= 0 && jd0 - j0 + k < TW) domFirst |= 0x80u << (8 * k);
            if (cLast + k >= 0 && cLast + k < TW) domLast |= 0x80u << (8 * k);
        }
    }
    for (int ibase = 0; ibase < nitems; ibase += 8 * 256) {
        // acc: bit (8 * k + 7 - i) = pixel k of this thread's i-th item of the chunk
        uint32_t acc = 0;
#pragma unroll
        for (int i = 0; i < 8; i++) {
            const int item = ibase + i * 256 + tid;
            if (item < nitems) {
                const int r = (int)(((float)item + 0.5f) * invGPR);
                const int g = item - __mul24(r, GPR);
                const int jd = jd0 + (g << 2);
                s_ent[item] = (uint16_t)((r << 9) | jd);   // list entry of the item's pixel 0
                const uint8_t *row = s_pix + __mul24(r + 3, pitch) + jd;
                const uint32_t Cw = *reinterpret_cast<const uint32_t *>(row);
                const uint32_t Lw = *reinterpret_cast<const uint32_t *>(row - 4);
                const uint32_t Rw = *reinterpret_cast<const uint32_t *>(row + 4);
                const uint32_t Tw = *reinterpret_cast<const uint32_t *>(row - 3 * pitch);
                const uint32_t Bw = *reinterpret_cast<const uint32_t *>(row + 3 * pitch);
                const uint32_t lft = __builtin_amdgcn_alignbyte(Cw, Lw, 1);   // bytes L1 L2 L3 C0 (column - 3)
                const uint32_t rgt = __builtin_amdgcn_alignbyte(Rw, Cw, 3);   // bytes C3 R0 R1 R2 (column + 3)
                const uint32_t fA = compass2(lo2(Cw), lo2(Tw), lo2(Bw), lo2(lft), lo2(rgt), tt);   // px 0,1
                const uint32_t fB = compass2(hi2(Cw), hi2(Tw), hi2(Bw), hi2(lft), hi2(rgt), tt);   // px 2,3
                // the sign bytes of px 0..3 into bytes 0..3; pixels outside the domain (first / last group of a row) are
                // masked; item slot i keeps bit 7 - i of each byte
                const uint32_t z = __builtin_amdgcn_perm(fB, fA, 0x07050301u);
                const uint32_t dom = (g == 0 ? domFirst : 0x80808080u) & (g == GPR - 1 ? domLast : 0x80808080u);
                acc |= (z & dom) >> i;
            }
        }
        // append this thread's survivors to the work list (order is irrelevant)
        const int n = __popc(acc);
        const int incl = wave_incl_scan_dpp(n);
        const int total = __builtin_amdgcn_readlane(incl, 63);
        int base = 0;
        if (lane == 63 && total > 0) base = atomicAdd(&s_listCount, total);
        base = __builtin_amdgcn_readlane(base, 63);
        int pos = base + incl - n;
        while (acc) {
            const int b = __ffs(acc) - 1;
            acc &= acc - 1;
            const int ent = s_ent[ibase + (7 - (b & 7)) * 256 + tid] + (b >> 3);   // written by this thread above
            if (pos < listCap) s_list[pos] = (uint16_t)ent;
            pos++;
        }
    }
    __syncthreads();
    // the score tile (its storage held the items' entries until here) starts at zero
    for (int i = tid; i < (DH * SP) >> 2; i += 256) reinterpret_cast<uint32_t *>(s_score)[i] = 0;
    __syncthreads();
    if (phases < 3) return;

    // ---- 3. full score on the work list; corners -> corner list ----
    // The lists have a fixed LDS budget.  If a tile has more compass survivors than the work list
    // holds (noise-like images), every domain pixel is scored instead (the compass test is the
    // early-out of fast_score_pol); if it has more corners than the corner list holds, phase 4 scans
    // the score tile.  Both fallbacks produce the same result as the list paths.
    const int nlist = s_listCount;
    if (nlist <= listCap) {
        for (int e = tid; e < nlist; e += 256) {
            const int ent = s_list[e];
            const int r = ent >> 9, j = ent & 511;
            const int s = fast_score_pol(s_pix + __mul24(r + 3, pitch) + j, pitch, t);
            if (s > 0) {
                s_score[__mul24(r, SP) + (j - j0)] = (uint8_t)s;
                const int slot = atomicAdd(&s_cornerCount, 1);   // hipcc aggregates this per wave
                if (slot < cornerCap) s_corner[slot] = (uint16_t)ent;
            }
        }
    } else {
        const float invTW = 1.0f / (float)TW;   // px / TW = floor((px + 0.5) * invTW): exact for every px < DH * TW (a 20-bit
                                                 // integer reciprocal is NOT: it fails from px ~ 2^20 / TW on, e.g. TW 155, DH 45)
        for (int p0 = 0; p0 < DH * TW; p0 += 256) {
            const int px = p0 + tid;
            int ent = 0, s = 0;
            if (px < DH * TW) {
                const int r = (int)(((float)px + 0.5f) * invTW);
                const int c = px - r * TW;
                s = fast_score_pol(s_pix + (r + 3) * pitch + j0 + c, pitch, t);
                if (s > 0) s_score[r * SP + c] = (uint8_t)s;
                ent = (r << 9) | (j0 + c);
            }
            const int slot = wave_append(s > 0, &s_cornerCount, lane);
            if (slot >= 0 && slot < cornerCap) s_corner[slot] = (uint16_t)ent;
        }
    }
    __syncthreads();
    if (phases < 4) return;

    // ---- 4. NMS over the corners (cell-local neighbourhood); survivors -> keyed list ----
    const unsigned cellMagic = 65536u / (unsigned)L.wCell + 1u;   // c / wCell for c < 65536 / wCell
    const int ncorner = s_cornerCount;
    if (ncorner <= cornerCap) {
        for (int e = tid; e < ncorner; e += 256) {
            const int ent = s_corner[e];
            const int r = ent >> 9, c = (ent & 511) - j0;
            bool surv = false;
            const uint32_t key = nms_key(s_score, SP, r, c, DH, TW, L.wCell, cellMagic, G.iniTh, s_cellAny, surv);
            if (surv) s_surv[atomicAdd(&s_survCount, 1)] = key;
        }
    } else {
        // fallback: scan the score tile, 4 pixels per dword
        const int SPW = SP >> 2;                                       // score dwords per row
        const unsigned spwMagic = (1u << 20) / (unsigned)SPW + 1u;
        const int nwords = DH * SPW;
        for (int i0 = 0; i0 < nwords; i0 += 256) {
            const int i = i0 + tid;
            uint32_t w = i < nwords ? reinterpret_cast<const uint32_t *>(s_score)[i] : 0u;
            const int r = (int)(((unsigned)(i < nwords ? i : 0) * spwMagic) >> 20);
            const int cb = ((i < nwords ? i : 0) - r * SPW) << 2;
            // every lane runs the loop body the same number of times (wave-wide append inside)
            for (int k = 0; k < 4; k++) {
                const int s = (w >> (8 * k)) & 0xFF;
                bool surv = false;
                uint32_t key = 0;
                if (__ballot(s > 0) == 0) continue;   // wave-uniform
                if (s > 0) key = nms_key(s_score, SP, r, cb + k, DH, TW, L.wCell, cellMagic, G.iniTh, s_cellAny, surv);
                const int slot = wave_append(surv, &s_survCount, lane);
                if (slot >= 0) s_surv[slot] = key;
            }
        }
    }
    __syncthreads();
    if (phases < 5) return;

    // ---- 5. per-cell threshold, rank inside the cell (= raster order), write the slots ----
    // The kept survivors set one bit per (cell, row, column-in-cell) in a bitmap that reuses the score tile (dead
    // after the NMS); the rank of a survivor is then the number of bits before it: a prefix over the rows of its
    // cell plus a popcount inside its row -- no survivor is ever compared with another one.
    const int nsurv = s_survCount;
    unsigned long long *s_bits = reinterpret_cast<unsigned long long *>(s_score);   // [ncells][DH]; wCell < 64
    int *s_pre = reinterpret_cast<int *>(s_bits + T.ncells * DH);                   // [ncells][DH]
    const int nrowsAll = T.ncells * DH;
    for (int i = tid; i < nrowsAll; i += 256) s_bits[i] = 0ull;
    __syncthreads();
    for (int e = tid; e < nsurv; e += 256) {
        const uint32_t key = s_surv[e];
        const int cj = key >> 28;
        const int thr = s_cellAny[cj] ? G.iniTh : G.minTh;
        if ((int)(key & 0xFF) >= thr) {
            const int r = (key >> 21) & 127, c = (key >> 8) & 0x1FFF;
            atomicOr(&s_bits[cj * DH + r], 1ull << (c - cj * L.wCell));
        }
    }
    __syncthreads();
    if (DH <= 64) {
        // one wave per cell, one lane per row: the row prefix is a wave scan of the row popcounts
        for (int cj = tid >> 6; cj < T.ncells; cj += 4) {
            const int n = lane < DH ? __popcll(s_bits[cj * DH + lane]) : 0;
            const int incl = wave_incl_scan_dpp(n);
            if (lane < DH) s_pre[cj * DH + lane] = incl - n;
            if (lane == 63) s_cellCnt[cj] = incl;
        }
    } else {
        const unsigned dhMagic = 65536u / (unsigned)DH + 1u;   // i / DH for i < 65536 / DH
        for (int i = tid; i < nrowsAll; i += 256) {
            const int cj = (int)(((unsigned)i * dhMagic) >> 16), r = i - cj * DH;
            int pre = 0;
            for (int rr = 0; rr < r; rr++) pre += __popcll(s_bits[cj * DH + rr]);
            s_pre[i] = pre;
            if (r == DH - 1) s_cellCnt[cj] = pre + __popcll(s_bits[i]);
        }
    }
    __syncthreads();
    const size_t candFrame = (size_t)frame * G.totalCands + L.candBase;
    for (int e = tid; e < nsurv; e += 256) {
        const uint32_t key = s_surv[e];
        const int cj = key >> 28;
        const int thr = s_cellAny[cj] ? G.iniTh : G.minTh;
        const int sc = key & 0xFF;
        if (sc >= thr) {
            const int r = (key >> 21) & 127, c = (key >> 8) & 0x1FFF;
            const int cl = c - cj * L.wCell;
            const int rank = s_pre[cj * DH + r] + __popcll(s_bits[cj * DH + r] & ((1ull << cl) - 1ull));
            const int px = X0 + 3 + c - ORB_MIN_BORDER;   // relative to (16,16), :824-825
            const int py = iniY + 3 + r - ORB_MIN_BORDER;
            uint32_t *slot = cand + candFrame + (size_t)(T.row * L.nCols + T.c0 + cj) * L.cellCap;
            slot[rank] = (uint32_t)px | ((uint32_t)py << 12) | ((uint32_t)sc << 24);
        }
    }
    __syncthreads();
    // cells whose iniX >= maxBorderX-6 are skipped by the reference (:805): their domain is empty -> 0
    if (tid < T.ncells) cnt[tid] = (uint16_t)s_cellCnt[tid];
}

void launch_fast(hipStream_t s, const OrbLevels &G, const uint8_t *lvl0, int stride0, size_t frame0,
                 const uint8_t *pyr, size_t pyrFrame, const FastTile *tiles, int ntiles,
                 uint32_t *cand, uint16_t *cellCnt, int B)
{
    // LDS: pixel tile + score tile + work list of the largest run over all levels
    int pixBytes = 0, scoreBytes = 0, listBytes = 0, survBytes = 0;
    for (int l = 0; l < G.nlevels; l++) {
        const OrbLevel &L = G.lv[l];
        int tileCells = fast_tile_cells();
        while (tileCells > 1 && tileCells * L.wCell + 6 + 16 > FAST_MAX_TILE_W) tileCells--;
        const int regw = tileCells * L.wCell + 6;
        const int pitch = ((regw + 15 + 15) >> 4) << 4;
        const int rh = L.hCell + 6;
        const int sp = (tileCells * L.wCell + 3) & ~3;
        pixBytes = std::max(pixBytes, pitch * rh);
        scoreBytes = std::max(scoreBytes, sp * L.hCell);
        // work list: u16 per domain pixel; the survivor list (u32 per strict local maximum, at most
        // a quarter of the pixels plus cell seams) reuses the same storage
        listBytes = std::max(listBytes, sp * L.hCell * 2);
        survBytes = std::max(survBytes, tileCells * L.cellCap * 4);
    }
    pixBytes = (pixBytes + 15) & ~15;
    scoreBytes = (scoreBytes + 15) & ~15;
    // fixed list budgets (entries): work list = half of the tile's pixels, corner list = an eighth;
    // tiles that exceed them take the exact fallback paths.  ORBHIP_FAST_LISTCAP forces tiny lists
    // (tests exercise the fallbacks with it).
    static const int forced = getenv("ORBHIP_FAST_LISTCAP") ? atoi(getenv("ORBHIP_FAST_LISTCAP")) : 0;
    int listCap = listBytes / 4, cornerCap = listBytes / 16;
    if (forced > 0) listCap = cornerCap = forced;
    // the survivor list (u32 per strict local maximum, at most survBytes/4 of them) shares the work list
    listBytes = std::max(listCap * 2, survBytes);
    listBytes = (listBytes + 15) & ~15;
    const int cornerBytes = (cornerCap * 2 + 15) & ~15;
    static const int phases = getenv("ORBHIP_FAST_PHASES") ? atoi(getenv("ORBHIP_FAST_PHASES")) : 5;
    dim3 grid(orb_xcd_grid(ntiles), B, 1), block(256, 1, 1);
    hipLaunchKernelGGL(k_fast, grid, block, (size_t)(pixBytes + scoreBytes + listBytes + cornerBytes), s, G, lvl0, stride0,
                       (unsigned long long)frame0, pyr, (unsigned long long)pyrFrame, tiles, cand, cellCnt,
                       pixBytes, scoreBytes, listBytes, listCap, cornerCap, phases, orb_xcd_arg(), ntiles);
}
