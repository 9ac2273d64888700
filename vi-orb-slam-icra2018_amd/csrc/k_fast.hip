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
// One 256-thread workgroup per (frame, run of <= 4 cells of one cell-row): stage the run's pixels (+3 px halo) into LDS, then
// the passes and phases described above the kernels below.  No global atomics, deterministic.  Two kernels, same results:
//   k_fast_fix<PITCH>  the usual cell grid (rows of <= 208 staged bytes, cells of <= 34 rows): fixed LDS layout, geometry from the
//                      run's descriptor, LDS-DMA staging, second pass and candidate order by one wave per cell (round 3);
//   k_fast<PITCH>      any grid (round 2); also ORBHIP_FAST_FIX=0.
// HBM traffic: each level pixel inside [16, w-16) x [16, h-16) is read once per tile that needs it;
// the 6-row vertical halo (hCell ~ 30) is re-read by the tile below.  Roofline: nominally HBM read
// (algorithmic bytes = sum_l (w_l-32)(h_l-32) per frame); measured bound is vector-instruction issue
// (DESIGN.md section 4).
#include "orbhip_internal.h"

#include <cstdlib>
#include <type_traits>

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
template <bool DEFER>
__device__ __forceinline__ int fast_score_win(const uint8_t *pix, int off, int pitch, int t, bool *other)
{
    // pix + off = top-left corner of the pixel's 7x7 window.  Every ring pixel as a non-negative offset from the top-left corner of the 7x7 window: with a compile-time pitch the 17
    // loads share one address register (ds_read_u8 offset:imm), with a run-time pitch they need one add per window row.
    // (the empty asm keeps the compiler from re-deriving the addresses from p, which costs an add per negative offset)
    // (on the offset, not the pointer: an asm on the pointer would lose its LDS address space and the loads would be flat loads)
    asm volatile("" : "+v"(off));
    const uint8_t *w = pix + off;
    const int P1 = pitch, P2 = 2 * pitch, P3 = 3 * pitch, P4 = 4 * pitch, P5 = 5 * pitch, P6 = 6 * pitch;
    // ring pixel k (OpenCV's order: k = 0 at (0, +3), then clockwise in image coordinates) paired with pixel k + 8
    const int v0 = w[P3 + 3];
    const int q0 = w[P6 + 3], q8 = w[3], q4 = w[P3 + 6], q12 = w[P3];
    // polarity that can hold a 9-arc (it contains ring pixel 0 or 8 and ring pixel 4 or 12)
    const bool pb = min(max(q0, q8), max(q4, q12)) > v0 + t;    // bright: q - v > t
    const bool pd = max(min(q0, q8), min(q4, q12)) < v0 - t;    // dark:   v - q > t
    if (!pb && !pd) return 0;
    const bool dark = !pb;
    const uint32_t C = dark ? 0x40FF40FFu : 0x40004000u;
    uint32_t P[8];
    P[0] = ((uint32_t)q0 | ((uint32_t)q8 << 16));                       // (0, +3)  | (0, -3)
    P[1] = ((uint32_t)w[P6 + 4] | ((uint32_t)w[2] << 16));              // (+1, +3) | (-1, -3)
    P[2] = ((uint32_t)w[P5 + 5] | ((uint32_t)w[P1 + 1] << 16));         // (+2, +2) | (-2, -2)
    P[3] = ((uint32_t)w[P4 + 6] | ((uint32_t)w[P2] << 16));             // (+3, +1) | (-3, -1)
    P[4] = ((uint32_t)q4 | ((uint32_t)q12 << 16));                      // (+3, 0)  | (-3, 0)
    P[5] = ((uint32_t)w[P2 + 6] | ((uint32_t)w[P4] << 16));             // (+3, -1) | (-3, +1)
    P[6] = ((uint32_t)w[P1 + 5] | ((uint32_t)w[P5 + 1] << 16));         // (+2, -2) | (-2, +2)
    P[7] = ((uint32_t)w[4] | ((uint32_t)w[P6 + 2] << 16));              // (+1, -3) | (-1, +3)
    int sc = arcs_score(P, C, v0 ^ (dark ? 0xFF : 0));
    if (DEFER) {
        // The dark arcs of a pixel whose compass points admit both polarities are left to the caller (a 9-arc of one
        // polarity excludes one of the other, so a bright corner is final): 7-12 % of the work-list entries are such
        // pixels, i.e. practically every wave holds one, and evaluating them in place makes every wave pay both polarities.
        *other = pb && pd && sc < t;
    } else if (pb && pd)
        sc = max(sc, arcs_score(P, 0x40FF40FFu, v0 ^ 0xFF));   // both possible: the dark one as well
    return sc >= t ? sc : 0;
}

// the dark-polarity score alone (second half of a deferred entry)
__device__ __forceinline__ int fast_score_dark(const uint8_t *pix, int off, int pitch, int t)
{
    asm volatile("" : "+v"(off));
    const uint8_t *w = pix + off;
    const int P1 = pitch, P2 = 2 * pitch, P3 = 3 * pitch, P4 = 4 * pitch, P5 = 5 * pitch, P6 = 6 * pitch;
    const int v0 = w[P3 + 3];
    uint32_t P[8];
    P[0] = ((uint32_t)w[P6 + 3] | ((uint32_t)w[3] << 16));
    P[1] = ((uint32_t)w[P6 + 4] | ((uint32_t)w[2] << 16));
    P[2] = ((uint32_t)w[P5 + 5] | ((uint32_t)w[P1 + 1] << 16));
    P[3] = ((uint32_t)w[P4 + 6] | ((uint32_t)w[P2] << 16));
    P[4] = ((uint32_t)w[P3 + 6] | ((uint32_t)w[P3] << 16));
    P[5] = ((uint32_t)w[P2 + 6] | ((uint32_t)w[P4] << 16));
    P[6] = ((uint32_t)w[P1 + 5] | ((uint32_t)w[P5 + 1] << 16));
    P[7] = ((uint32_t)w[4] | ((uint32_t)w[P6 + 2] << 16));
    const int sc = arcs_score(P, 0x40FF40FFu, v0 ^ 0xFF);
    return sc >= t ? sc : 0;
}

__device__ __forceinline__ int fast_score_pol(const uint8_t *p, int pitch, int t)
{
    return fast_score_win<false>(p, -3 * pitch - 3, pitch, t, nullptr);
}

// 16 bytes per lane from global memory straight into LDS at (ldsAddr + 16 * lane): global_load_lds_dwordx4 in assembly (the
// builtin makes hipcc wait vmcnt(0) at every LDS access that might alias); M0 carries the LDS address and is restored.
__device__ __forceinline__ void glds16(const void *gsrc, uint32_t ldsAddr)
{
    uint32_t keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(gsrc), "s"(ldsAddr)
                 : "memory");
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


// =====================================================================================================================
// k_fast (r02): the reference's own control flow -- FAST at iniThFAST first, FAST at minThFAST only for the cells the first
// run left empty (src/ORBextractor.cc:811-818) -- on the whole-level formulation.
//
// Exactness: the score of a pixel does not depend on the detection threshold and "corner at t" <=>
// "score >= t", so cv::FAST(cell, t, nonmax) = { p : s(p) >= t and s(p) > s(q) for the 8 neighbours q of the same cell with
// s(q) >= t }, and a neighbour below t is smaller than s(p) anyway: the suppression at threshold t only needs the scores
// >= t.  Pass 0 therefore computes only the scores >= iniThFAST (compass test, work list, arc score: all at iniThFAST),
// suppresses among them, and the survivors are final: their cell is not empty.  Pass 1 repeats the three steps at
// minThFAST for the pixels of the cells without a survivor (NOT "without a corner": two equal neighbouring maxima suppress
// each other, and the reference then reruns the cell) and every survivor of it is final as well.  On textured frames pass 0
// scores 14 % of the pixels instead of 24-28 %, and pass 1 touches a fifth of the cells (low contrast: 6 % of their pixels
// pass the compass test).
//
// Phases of a pass (one 256-thread workgroup per run of <= 5 cells, as before):
//   2. compass test, 4 pixels per item.  A thread keeps ONE dword column of the tile and walks down the rows in steps of
//      RS = 256 / (active dword columns): no division per item, the domain / active-cell mask is a per-thread constant, the
//      list entry of a survivor is a multiply-add of its bit position (r01 parked an entry per item in LDS and read it back
//      per survivor: a dependent LDS round trip in a divergent loop).  Bytes are split into even / odd halves with one
//      v_and and one v_perm per dword instead of two v_perm.
//   3. arc score on the work list (fast_score_pol: packed halves, v_pk_minimum3/maximum3_f16), corners -> score tile +
//      corner list.
//   4. suppression over the corner list; a survivor sets its bit in the per-(cell, row) bitmap directly -- every survivor
//      of either pass is kept, so the keyed survivor list and the per-cell threshold of r01 are gone.
//   5. (after both passes) row prefix of the bitmap, survivors written to their cell's slots in raster order.
// Bounded lists with exact fallbacks as before (dense scoring of the active cells; scan of the score tile).
// =====================================================================================================================

// ---- compass test of the 4 pixels of an aligned dword in byte arithmetic (v_lerp_u8) ----
// v_lerp_u8 d, a, b, c computes per byte (a + b + (c & 1)) >> 1 in 9 bits: with b = ~v it is floor((q - v + 255 + c) / 2), a
// monotone map of the difference d = q - v into a byte, and with b = ~K, c = 1 it is (m - K + 256) >> 1 whose bit 7 says
// m >= K -- a bytewise unsigned compare in one instruction.  Halving loses the parity of d, so the rounding bit is chosen
// per threshold such that the cut falls between two values of the halved quantity:
//   bright  q - v > t   <=>  d + 255 + c  >= t + 256 + c (even for c = t & 1)        <=>  floor((d + 255 + c) / 2)  >= KB
//   dark    q - v < -t  <=>  d + 255 + c' <= 254 + c' - t (odd for c' = (t + 1) & 1)  <=>  floor((d + 255 + c') / 2) <= KD
// (exhaustive check over all q, v, t: tests/test_host_logic.py).  Four instructions per neighbour dword instead of the
// unpack / packed-16-bit sequence of r01 (16 v_lerp + 5 bit operations per 4 pixels instead of 22 v_pk + 10 unpack).
struct CompassK {
    uint32_t cb, cd;     // rounding bits (0x01010101 or 0) of the bright / dark map
    uint32_t nkb, nkd;   // ~KB, ~(KD + 1) in every byte
};
__device__ __forceinline__ CompassK compass_consts(int t)
{
    // 1 <= t <= 254 (t = 255 admits no corner; the caller skips the pass)
    const uint32_t ONES = 0x01010101u;
    const int c = t & 1, c2 = c ^ 1;
    const int KB = ((t + c) >> 1) + 128;          // <= 255
    const int KD = (253 + c2 - t) >> 1;           // >= 0
    CompassK K;
    K.cb = c ? ONES : 0u;
    K.cd = c2 ? ONES : 0u;
    K.nkb = ~((uint32_t)KB * ONES);
    K.nkd = ~((uint32_t)(KD + 1) * ONES);
    return K;
}
// bit 7 of byte k of the result <=> pixel k of Cw passes the compass test (before the domain mask)
__device__ __forceinline__ uint32_t compass4(uint32_t Cw, uint32_t Tw, uint32_t Bw, uint32_t lft, uint32_t rgt, const CompassK &K)
{
    const uint32_t ONES = 0x01010101u;
    const uint32_t nv = ~Cw;
#define ORB_BRIGHT(q) __builtin_amdgcn_lerp(__builtin_amdgcn_lerp((q), nv, K.cb), K.nkb, ONES)   /* bit 7: q - v > t      */
#define ORB_NOTDARK(q) __builtin_amdgcn_lerp(__builtin_amdgcn_lerp((q), nv, K.cd), K.nkd, ONES)  /* bit 7: !(q - v < -t) */
    const uint32_t bT = ORB_BRIGHT(Tw), bB = ORB_BRIGHT(Bw), bL = ORB_BRIGHT(lft), bR = ORB_BRIGHT(rgt);
    const uint32_t dT = ORB_NOTDARK(Tw), dB = ORB_NOTDARK(Bw), dL = ORB_NOTDARK(lft), dR = ORB_NOTDARK(rgt);
#undef ORB_BRIGHT
#undef ORB_NOTDARK
    const uint32_t bright = (bT | bB) & (bL | bR);
    const uint32_t notdark = (dT & dB) | (dL & dR);
    return bright | ~notdark;
}

__device__ __forceinline__ bool nms_survives(const uint8_t *s_score, int SP, int r, int c, int DH, int TW, int wCell, int cj)
{
    const uint8_t *sp = s_score + r * SP + c;
    const int s = sp[0];
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
    return s > m;
}

// nms_survives for the fixed-layout kernel: the eight neighbours are read unconditionally (one wait instead of one per
// neighbour; a row above / below the tile lies inside the workgroup's LDS -- the score tile is padded by a row -- and reads
// whatever is there) and the ones outside the cell or the tile are masked to 0.
__device__ __forceinline__ bool nms_survives_fix(const uint8_t *s_score, int SP, int r, int c, int DH, int TW, int wCell, int cj)
{
    const uint8_t *sp = s_score + r * SP + c;
    const int s = sp[0];
    const int cx0 = cj * wCell, cx1 = min(cx0 + wCell, TW);
    const int a = sp[-1], b = sp[1], u0 = sp[-SP - 1], u1 = sp[-SP], u2 = sp[-SP + 1], d0 = sp[SP - 1], d1 = sp[SP], d2 = sp[SP + 1];
    const bool up = r > 0, dn = r < DH - 1, lf = c > cx0, rt = c < cx1 - 1;
    int m = max(lf ? a : 0, rt ? b : 0);
    m = max(m, up ? max(u1, max(lf ? u0 : 0, rt ? u2 : 0)) : 0);
    m = max(m, dn ? max(d1, max(lf ? d0 : 0, rt ? d2 : 0)) : 0);
    return s > m;
}

// PITCH: the LDS row pitch of the pixel tile as a compile-time constant (0 = from the tile's width at run time).  With it the
// five dwords of a compass item and the 17 bytes of a score window are loads at immediate offsets from one address register.
template <int PITCH>
__global__ __launch_bounds__(256, 8) void k_fast(const OrbLevels G, const uint8_t *__restrict__ lvl0, int stride0,
                                              unsigned long long frame0, const uint8_t *__restrict__ pyr,
                                              unsigned long long pyrFrame, const FastTile *__restrict__ tiles,
                                              uint32_t *__restrict__ cand, uint16_t *__restrict__ cellCnt, int pixBytes,
                                              int scoreBytes, int listBytes, int cornerBytes, int bitsBytes, int listCap,
                                              int cornerCap, int xcdMap, int ntiles ORB_ABL_PARAM)
{
    extern __shared__ __align__(16) uint8_t smem[];
    __shared__ int s_cellAny[FAST_TILE_CELLS];
    __shared__ int s_cellCnt[FAST_TILE_CELLS];
    __shared__ int s_listCount, s_cornerCount, s_nAct;
    __shared__ uint8_t s_grp[FAST_MAX_TILE_W / 4 + 8];
    __shared__ uint32_t s_dom[FAST_MAX_TILE_W / 4 + 8];   // per dword group of the pass: which of its 4 pixels are tested (bit 8k + 7)

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

    // ---- 1. stage pixels [iniY, maxY) x [XA, X1) into LDS, 16 bytes per lane per load; clear the score tile and the bitmap ----
    const int XA = X0 & ~15;
    const int nchunk = (X1 - XA + 15) >> 4;
    const int pitch = PITCH ? PITCH : nchunk << 4;
    const int RH = maxY - iniY;
    uint8_t *s_pix = smem;
    uint8_t *s_score = smem + pixBytes;
    uint16_t *s_list = reinterpret_cast<uint16_t *>(smem + pixBytes + scoreBytes);
    uint16_t *s_corner = reinterpret_cast<uint16_t *>(smem + pixBytes + scoreBytes + listBytes);
    unsigned long long *s_bits = reinterpret_cast<unsigned long long *>(smem + pixBytes + scoreBytes + listBytes + cornerBytes);   // [ncells][DH]; wCell < 64
    int *s_pre = reinterpret_cast<int *>(smem + pixBytes + scoreBytes + listBytes + cornerBytes + bitsBytes);                       // [ncells][DH]
    const int SP = (TW + 3) & ~3;
    // i / nchunk = floor((i + 0.5) * invNchunk): v_rcp_f32 is within 1 ulp, the product is off by < 2^-9 for i < 2^13, the
    // quotient is >= 1 / (2 nchunk) >= 2^-6 away from an integer
    const float invNchunk = __builtin_amdgcn_rcpf((float)nchunk);
    const uint8_t *img0 = img + (size_t)iniY * stride + XA;   // workgroup-uniform base, 32-bit offsets from it
    const int nstage = RH * nchunk;
    for (int i = tid; i < nstage; i += 512) {
        // two loads in flight per thread (a tile of the usual grid is 1.8 x 256 chunks)
        const int ib = min(i + 256, nstage - 1);
        const int ra = (int)(((float)i + 0.5f) * invNchunk), ca = i - ra * nchunk;
        const int rb = (int)(((float)ib + 0.5f) * invNchunk), cb = ib - rb * nchunk;
        const uint4 va = *reinterpret_cast<const uint4 *>(img0 + (unsigned)(__mul24(ra, stride) + (ca << 4)));
        const uint4 vb = *reinterpret_cast<const uint4 *>(img0 + (unsigned)(__mul24(rb, stride) + (cb << 4)));
        *reinterpret_cast<uint4 *>(s_pix + __mul24(ra, pitch) + (ca << 4)) = va;
        if (i + 256 < nstage) *reinterpret_cast<uint4 *>(s_pix + __mul24(rb, pitch) + (cb << 4)) = vb;
    }
    const int nrowsAll = T.ncells * DH;
    for (int i = tid; i < (DH * SP + 15) >> 4; i += 256) reinterpret_cast<uint4 *>(s_score)[i] = make_uint4(0u, 0u, 0u, 0u);   // scoreBytes is a multiple of 16
    for (int i = tid; i < nrowsAll; i += 256) s_bits[i] = 0ull;
    const int j0 = X0 + 3 - XA;            // LDS column of domain column 0
    const int jd0 = j0 & ~3;               // first aligned dword column touching the domain
    const int GPR = ((j0 + TW + 3) >> 2) - (j0 >> 2);   // dword groups per row
    // pass 0 tests every domain pixel: the mask of a group is its overlap with [0, TW)
    for (int g = 255 - tid; g < GPR; g += 256) {   // (the last wave: it has the fewest staging loads)
        uint32_t m = 0;
#pragma unroll
        for (int k = 0; k < 4; k++)
            if ((unsigned)(jd0 + (g << 2) + k - j0) < (unsigned)TW) m |= 0x80u << (8 * k);
        s_dom[g] = m;
    }
    if (tid < FAST_TILE_CELLS) {
        s_cellAny[tid] = 0;
        s_cellCnt[tid] = 0;
    }
    if (tid == 0) {
        s_listCount = 0;
        s_cornerCount = 0;
        s_nAct = 0;
    }
    __syncthreads();
    ORB_ABL_STOP(phases < 2);   // timing ablation only (liborbhip_ablation.so, ORBHIP_FAST_PHASES): results are then invalid

    const int wCell = L.wCell;
    const unsigned cellMagic = 65536u / (unsigned)wCell + 1u;   // c / wCell for c < 65536 / wCell
    const unsigned allCells = (1u << T.ncells) - 1u;

    for (int pass = 0; pass < 2; pass++) {
        // pass 0: every cell at iniThFAST; pass 1: the cells without a survivor at minThFAST (block-uniform decisions)
        const int t = pass == 0 ? G.iniTh : G.minTh;
        unsigned cellMask = allCells;
        if (pass == 1) {
            if (G.minTh >= G.iniTh) break;   // FAST(ini) empty => FAST(min >= ini) empty
            cellMask = 0;
            for (int cj = 0; cj < T.ncells; cj++)
                if (!s_cellAny[cj]) cellMask |= 1u << cj;
            if (cellMask == 0) break;
        }
        if (t >= 255) continue;   // no pixel differs from its centre by more than 255: no corner at all (block-uniform)
        const CompassK CK = compass_consts(t);

        // ---- 2. compass pre-test, 4 pixels per item; survivors -> work list ----
        int nAct = GPR;
        if (pass == 1) {
            // dword groups that hold a domain pixel of an active cell (any order)
            for (int g = tid; g < GPR; g += 256) {
                uint32_t m = 0;
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    const int c = jd0 + (g << 2) + k - j0;
                    if ((unsigned)c < (unsigned)TW && ((cellMask >> ((unsigned)c * cellMagic >> 16)) & 1u)) m |= 0x80u << (8 * k);
                }
                if (m) {
                    const int a = atomicAdd(&s_nAct, 1);
                    s_grp[a] = (uint8_t)g;
                    s_dom[a] = m;
                }
            }
            if (tid == 0) {
                s_listCount = 0;
                s_cornerCount = 0;
            }
            __syncthreads();
            nAct = s_nAct;
        }
        // threads taking part: all of them in pass 0; in pass 1 about one per five items (whole waves), so that a tile with one
        // empty cell does not pay four waves' worth of fixed cost (scan, masks) for 300 items
        int nthr = 256;
        if (pass == 1) {
            const int want = (__mul24(nAct, DH) + 319) / 320;            // waves at ~5 items per thread
            const int need = (nAct + 63) >> 6;                           // at least one thread per column
            nthr = min(4, max(want, need)) << 6;
        }
        if ((tid & ~63) < nthr) {
            // thread -> (dword column, first row): tid = r0 * nAct + slot; rows r0, r0 + RS, ... (RS = nthr / nAct >= 1)
            // (v_rcp_f32 is within 1 ulp: the products below are off by < 2^-13, the quotients are >= 1 / (2 nAct) away from an integer)
            const float invAct = __builtin_amdgcn_rcpf((float)nAct);
            const int r0 = (int)(((float)tid + 0.5f) * invAct);
            const int slot = tid - r0 * nAct;
            const int RS = (int)(((float)nthr + 0.5f) * invAct);
            const bool mine = r0 < RS;
            const int g = pass == 0 ? slot : (int)s_grp[slot];
            const int jd = jd0 + (g << 2);
            const uint32_t dom = s_dom[slot];   // domain / active-cell mask of my four pixels (pixel k at bit 8k + 7)
            const int rowStep = __mul24(RS, pitch);
            const int RSsh = RS << 9;
            const uint32_t listCountAddr = (uint32_t)(uintptr_t)&s_listCount;   // LDS byte address (low half of the flat address)
            for (int rbase = 0; rbase < DH; rbase += 8 * RS) {
                // acc: bit (8 * k + i) = pixel k of this thread's i-th row of the chunk
                uint32_t acc = 0;
                // the item's five dwords at non-negative offsets from (row - 3, column - 4): top | left, centre, right | bottom
                const uint8_t *win = s_pix + __mul24(rbase + r0, pitch) + (jd - 4);
                const int rlim = mine ? DH - rbase - r0 : 0;   // item i exists <=> i * RS < rlim
#pragma unroll
                for (int i = 0; i < 8; i++) {
                    if (__mul24(i, RS) < rlim) {
                        const uint32_t Tw = *reinterpret_cast<const uint32_t *>(win + 4);
                        const uint32_t Lw = *reinterpret_cast<const uint32_t *>(win + 3 * pitch);
                        const uint32_t Cw = *reinterpret_cast<const uint32_t *>(win + 3 * pitch + 4);
                        const uint32_t Rw = *reinterpret_cast<const uint32_t *>(win + 3 * pitch + 8);
                        const uint32_t Bw = *reinterpret_cast<const uint32_t *>(win + 6 * pitch + 4);
                        const uint32_t lft = __builtin_amdgcn_alignbyte(Cw, Lw, 1);   // bytes L1 L2 L3 C0 (column - 3)
                        const uint32_t rgt = __builtin_amdgcn_alignbyte(Rw, Cw, 3);   // bytes C3 R0 R1 R2 (column + 3)
                        const uint32_t z = compass4(Cw, Tw, Bw, lft, rgt, CK);
                        acc |= (z & dom) >> (7 - i);
                    }
                    win += rowStep;
                }
                // append this thread's survivors to the work list (order is irrelevant): one returning LDS add per thread
                // claims its range (r01 ran a DPP wave scan + one add per wave: ~45 vector instructions per chunk; the LDS
                // unit serialises the 64 adds instead, and it has the cycles to spare).  A thread whose entries do not all
                // fit writes none: s_listCount then exceeds listCap and phase 3 takes the fallback.
                const int n = __popc(acc);
                ORB_ABL_IF(phases == 12) {   // ablation only: the compass items without the list
                    if (n == 77) s_list[0] = (uint16_t)n;
                    continue;
                }
                if (n > 0) {
                    // (inline asm: hipcc's atomic optimiser would turn a plain atomicAdd of a per-lane value back into a DPP scan)
                    int base;
                    asm volatile("ds_add_rtn_u32 %0, %1, %2\n\ts_waitcnt lgkmcnt(0)" : "=v"(base) : "v"(listCountAddr), "v"(n) : "memory");
                    if (base + n <= listCap) {
                        uint16_t *dst = s_list + base;
                        const int entBase = ((rbase + r0) << 9) | jd;
                        while (acc) {
                            const int b = __ffs(acc) - 1;
                            acc &= acc - 1;
                            // (row << 9) | LDS column = entBase + (b & 7) * (RS << 9) + (b >> 3); one v_mad_u32_u24 (the compiler
                            // picks the 64-bit multiply-add for the plain expression)
                            int ent;
                            asm("v_mad_u32_u24 %0, %1, %2, %3" : "=v"(ent) : "v"(b & 7), "v"(RSsh), "v"(entBase));
                            *dst++ = (uint16_t)(ent | (b >> 3));
                        }
                    }
                }
            }
        }
        __syncthreads();
        ORB_ABL_STOP(phases < 3 + 3 * pass || phases == 12);   // ablation stops: 2-4 = phases of pass 0, 5-7 = of pass 1

        // ---- 3. full score on the work list; corners (score >= t) -> score tile + corner list ----
        // If a tile has more compass survivors than the work list holds (noise-like images), every domain pixel of the
        // active cells is scored instead (the compass test is the early-out of fast_score_pol); if it has more corners
        // than the corner list holds, phase 4 scans the score tile.  Both fallbacks produce the same result.
        const int nlist = s_listCount;
        const uint32_t cornerCountAddr = (uint32_t)(uintptr_t)&s_cornerCount;
        if (nlist <= listCap) {
            for (int e = tid; e < nlist; e += 256) {
                const int ent = s_list[e];
                const int r = ent >> 9, j = ent & 511;
                const int s = fast_score_win<false>(s_pix, __mul24(r, pitch) + (j - 3), pitch, t, nullptr);
                if (s > 0) {
                    s_score[__mul24(r, SP) + (j - j0)] = (uint8_t)s;
                    // one returning LDS add per corner (a fifth of the lanes; hipcc's wave aggregation of atomicAdd(p, 1) costs
                    // a dozen vector instructions per iteration for every lane)
                    int slot;
                    asm volatile("ds_add_rtn_u32 %0, %1, %2\n\ts_waitcnt lgkmcnt(0)" : "=v"(slot) : "v"(cornerCountAddr), "v"(1) : "memory");
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
                    if ((cellMask >> ((unsigned)c * cellMagic >> 16)) & 1u) {
                        s = fast_score_pol(s_pix + (r + 3) * pitch + j0 + c, pitch, t);
                        if (s > 0) s_score[r * SP + c] = (uint8_t)s;
                    }
                    ent = (r << 9) | (j0 + c);
                }
                const int slot = wave_append(s > 0, &s_cornerCount, lane);
                if (slot >= 0 && slot < cornerCap) s_corner[slot] = (uint16_t)ent;
            }
        }
        __syncthreads();
        ORB_ABL_STOP(phases < 4 + 3 * pass);

        // ---- 4. NMS over the corners (cell-local neighbourhood); every survivor is final: set its bit ----
        const int ncorner = s_cornerCount;
        if (ncorner <= cornerCap) {
            for (int e = tid; e < ncorner; e += 256) {
                const int ent = s_corner[e];
                const int r = ent >> 9, c = (ent & 511) - j0;
                const int cj = (int)(((unsigned)c * cellMagic) >> 16);
                if (nms_survives(s_score, SP, r, c, DH, TW, wCell, cj)) {
                    atomicOr(&s_bits[cj * DH + r], 1ull << (c - cj * wCell));
                    s_cellAny[cj] = 1;   // benign race: every writer stores 1
                }
            }
        } else {
            // fallback: scan the score tile, 4 pixels per dword; only the active cells' corners of this pass (score >= t)
            const int SPW = SP >> 2;                                       // score dwords per row
            const unsigned spwMagic = (1u << 20) / (unsigned)SPW + 1u;
            const int nwords = DH * SPW;
            for (int i = tid; i < nwords; i += 256) {
                const uint32_t w = reinterpret_cast<const uint32_t *>(s_score)[i];
                if (w == 0) continue;
                const int r = (int)(((unsigned)i * spwMagic) >> 20);
                const int cb = (i - r * SPW) << 2;
                for (int k = 0; k < 4; k++) {
                    const int s = (w >> (8 * k)) & 0xFF;
                    const int c = cb + k;
                    if (s < t || c >= TW) continue;
                    const int cj = (int)(((unsigned)c * cellMagic) >> 16);
                    if (!((cellMask >> cj) & 1u)) continue;
                    if (nms_survives(s_score, SP, r, c, DH, TW, wCell, cj)) {
                        atomicOr(&s_bits[cj * DH + r], 1ull << (c - cj * wCell));
                        s_cellAny[cj] = 1;
                    }
                }
            }
        }
        __syncthreads();
        ORB_ABL_STOP(phases < 5 + 3 * pass);
    }

    // ---- 5. rank inside the cell (= raster order) from the bitmap, write the slots ----
    // The rank of a survivor is the number of bits before it: a prefix over the rows of its cell plus a popcount inside its
    // row -- no survivor is ever compared with another one.
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
    {
        const unsigned dhMagic = 65536u / (unsigned)DH + 1u;
        for (int i = tid; i < nrowsAll; i += 256) {
            unsigned long long bits = s_bits[i];
            if (bits == 0ull) continue;
            const int cj = (int)(((unsigned)i * dhMagic) >> 16), r = i - cj * DH;
            int rank = s_pre[i];
            uint32_t *slot = cand + candFrame + (size_t)(T.row * L.nCols + T.c0 + cj) * L.cellCap;
            const int py = iniY + 3 + r - ORB_MIN_BORDER;                     // relative to (16,16), :824-825
            const int pxBase = X0 + 3 + cj * wCell - ORB_MIN_BORDER;
            const uint8_t *sc = s_score + r * SP + cj * wCell;
            while (bits) {
                const int cl = __ffsll((long long)bits) - 1;
                bits &= bits - 1ull;
                slot[rank++] = (uint32_t)(pxBase + cl) | ((uint32_t)py << 12) | ((uint32_t)sc[cl] << 24);
            }
        }
    }
    // cells whose iniX >= maxBorderX-6 are skipped by the reference (:805): their domain is empty -> 0
    if (tid < T.ncells) cnt[tid] = (uint16_t)s_cellCnt[tid];
}


// =====================================================================================================================
// k_fast_fix (r03): the same algorithm as k_fast above for the usual grid (30-pixel cells: runs of <= 6 cells, staged rows of
// <= 208 bytes, cells <= 34 rows), written against what the r02 kernel's counters showed in round 3 -- a sixth of its vector
// instructions were not FAST at all but the prologue: the run's geometry derived in every workgroup, everything invariant
// hoisted out of the two-pass loop (including the divisions and per-thread tables of paths that hardly ever run) and,
// because that keeps ~130 uniform values alive, 54 of them parked in lanes of a vector register and fetched back one
// v_readlane at a time.  Here
//   * the geometry of the run is read from its FastTile (orb_build_geometry computes it once per image size);
//   * every LDS array has a compile-time address (only the score tile's extent varies): no base registers, immediate offsets;
//   * the two passes are two instantiations of one body, and what only pass 1 or a fallback path needs is computed there
//     (an empty asm on its inputs keeps the compiler from hoisting it into the common path);
//   * the pixel tile arrives by LDS-DMA (global_load_lds_dwordx4: no registers, no vector work per chunk);
//   * compass pass: a thread owns one dword column and a SEGMENT of consecutive rows, so every load of every item is at an
//     immediate offset from one register and "does item i exist" is a workgroup-uniform question; the survivor bits of an
//     item are gathered into a nibble by v_dot4_u32_u8, which makes a list entry (group << 9 | row << 2 | pixel) the
//     entry of the thread's first pixel plus the bit position: one add;
//   * score pass: of a pixel whose compass points admit both polarities only the bright arcs are evaluated in place; if
//     they make no corner the entry is parked and the dark arcs of all parked entries are evaluated by dense lanes afterwards
//     (7-12 % of the entries are such pixels, so practically every wave paid for both polarities of all its 64 entries).
// Tiles outside these bounds (other cell sizes, ORBHIP_FAST_PITCH=0) run k_fast<0>.
// =====================================================================================================================
// workgroup -> (run, frame) of the fixed-layout kernel: 4 = a whole frame's runs on one XCD (grid (8, runs, B / 8), see the kernel).
// Measured (r04, gpurun_out/r04_fast2 -> profiles/r04): fabric reads 1.91 -> 0.92 GB per 1024 frames (1.13 x the algorithmic bytes)
// at the same 1.02 ms -- the kernel is bound by vector and LDS issue, not by its loads; the r01 mappings paid an integer division
// per thread for the same traffic and were slower.
#define FAST_DEFAULT_XCD 4
#ifndef ORB_FAST_TILE_BY_VALUE
#define ORB_FAST_TILE_BY_VALUE 1
#endif
#define FF_RHM FAST_FIX_ROWS            // staged rows (hCell + 6) the fixed layout holds: cells of up to 34 rows, every level of 640 x 480 / 752 x 480 ...
#define FF_RHM_TALL 48                 // ... and the instance for taller cells (a level of two or three cell rows rounds its cell height up:
                                       // 1241 x 376 has cells of 40 rows at its smallest level); 1.7 KB more LDS: seven workgroups per CU
#define FF_NCM 5                       // cells per run
#define FF_LISTCAP 2176                // work list entries: what the eighth workgroup per CU leaves at the 176-byte pitch (8 x 20 KB of LDS; ff_max_lds below): 44 % of the largest
                                       // tile's pixels.  (r04: a third, 1664 -- on the photographs 15 % of the runs exceeded it and took the
                                       // every-pixel path, 9 % exceed 2176: k_fast 1.55 -> 1.46 ms per 1024 frames there, the textured class
                                       // 1.25 -> 1.22; taking the space from the corner list instead (2496 / 320) costs more in suppression
                                       // scans than it saves: 1.60 / 1.26)
#define FF_CORNERCAP 640               // corner list entries (an eighth)
#define FF_GRPM 64                     // dword groups per row

// ---- the compass test of k_fast_fix: one halving per neighbour for both polarities ----
// compass4 (above) halves q - v twice, with the rounding bit that makes each polarity's cut exact.  The compass test only
// has to be NECESSARY (every entry of the work list is scored exactly, and fast_score_win starts with the exact compass
// conditions), so here both polarities share the halving that is exact for "bright" (c = t & 1): the dark cut then admits
// q - v = -t as well as q - v < -t -- one value more -- and an item costs 12 v_lerp_u8 instead of 16.
// (tests/test_host_logic.py: bright exact, dark a superset by exactly that value, for every q, v, t.)
struct CompassL {
    uint32_t c;          // rounding bits (0x01010101 or 0)
    uint32_t nkb, nkd;   // ~KB, ~(KD + 1) in every byte
};
__device__ __forceinline__ CompassL compass_loose_consts(int t)
{
    // 1 <= t <= 254
    const uint32_t ONES = 0x01010101u;
    const int c = t & 1;
    const int KB = ((t + c) >> 1) + 128;          // bright <=> floor((d + 255 + c) / 2) >= KB                 (exact)
    const int KD = (254 + c - t) >> 1;            // dark   =>  floor((d + 255 + c) / 2) <= KD                 (d = -t passes too)
    CompassL K;
    K.c = c ? ONES : 0u;
    K.nkb = ~((uint32_t)KB * ONES);
    K.nkd = ~((uint32_t)(KD + 1) * ONES);
    return K;
}
__device__ __forceinline__ uint32_t compass4_loose(uint32_t Cw, uint32_t Tw, uint32_t Bw, uint32_t lft, uint32_t rgt, const CompassL &K)
{
    const uint32_t ONES = 0x01010101u;
    const uint32_t nv = ~Cw;
    const uint32_t aT = __builtin_amdgcn_lerp(Tw, nv, K.c), aB = __builtin_amdgcn_lerp(Bw, nv, K.c);
    const uint32_t aL = __builtin_amdgcn_lerp(lft, nv, K.c), aR = __builtin_amdgcn_lerp(rgt, nv, K.c);
    const uint32_t bT = __builtin_amdgcn_lerp(aT, K.nkb, ONES), bB = __builtin_amdgcn_lerp(aB, K.nkb, ONES);
    const uint32_t bL = __builtin_amdgcn_lerp(aL, K.nkb, ONES), bR = __builtin_amdgcn_lerp(aR, K.nkb, ONES);
    const uint32_t dT = __builtin_amdgcn_lerp(aT, K.nkd, ONES), dB = __builtin_amdgcn_lerp(aB, K.nkd, ONES);
    const uint32_t dL = __builtin_amdgcn_lerp(aL, K.nkd, ONES), dR = __builtin_amdgcn_lerp(aR, K.nkd, ONES);
    const uint32_t bright = (bT | bB) & (bL | bR);
    const uint32_t notdark = (dT & dB) | (dL & dR);
    return bright | ~notdark;
}

// The survivors of a thread (bit b of acc set <=> list entry entBase + b) as consecutive 16-bit entries from the LDS byte
// address dst on.  The loop runs as long as ANY lane of the wave has bits left, so its body is what counts: find-first-bit,
// clear-lowest (2), entry, store -- two entries per trip so that the pointer moves once for both (the compiler's loop has a
// pointer increment and a register copy per entry: 7 vector instructions instead of 5.5).  exec is narrowed inside and restored.
__device__ __forceinline__ void list_write_entries(uint32_t acc, uint32_t dst, uint32_t entBase)
{
    uint32_t b, tmp;
    unsigned long long sx;
    asm volatile("s_mov_b64 %[sx], exec\n\t"
                 "v_cmp_ne_u32_e32 vcc, 0, %[acc]\n\t"
                 "s_and_b64 exec, exec, vcc\n\t"
                 "s_cbranch_execz 2f\n"
                 "1:\n\t"
                 "v_ffbl_b32_e32 %[b], %[acc]\n\t"
                 "v_add_u32_e32 %[tmp], -1, %[acc]\n\t"
                 "v_add_u32_e32 %[b], %[b], %[eb]\n\t"
                 "v_and_b32_e32 %[acc], %[tmp], %[acc]\n\t"
                 "ds_write_b16 %[dst], %[b]\n\t"
                 "v_cmp_ne_u32_e32 vcc, 0, %[acc]\n\t"
                 "s_and_b64 exec, exec, vcc\n\t"
                 "s_cbranch_execz 2f\n\t"
                 "v_ffbl_b32_e32 %[b], %[acc]\n\t"
                 "v_add_u32_e32 %[tmp], -1, %[acc]\n\t"
                 "v_add_u32_e32 %[b], %[b], %[eb]\n\t"
                 "v_and_b32_e32 %[acc], %[tmp], %[acc]\n\t"
                 "ds_write_b16 %[dst], %[b] offset:2\n\t"
                 "v_add_u32_e32 %[dst], 4, %[dst]\n\t"
                 "v_cmp_ne_u32_e32 vcc, 0, %[acc]\n\t"
                 "s_and_b64 exec, exec, vcc\n\t"
                 "s_cbranch_execnz 1b\n"
                 "2:\n\t"
                 "s_mov_b64 exec, %[sx]"
                 : [acc] "+v"(acc), [dst] "+v"(dst), [b] "=&v"(b), [tmp] "=&v"(tmp), [sx] "=&s"(sx)
                 : [eb] "v"(entBase)
                 : "vcc", "memory");
}

// Compass items of one thread: rows [row0 + cb, row0 + cb + min(8, seg - cb)) of the dword column at `win` (= the address of
// (row0 + cb - 3, column - 4)), PITCH a compile-time constant.  Returns the survivors of item i in nibble i (pixel k at bit
// 4 i + k), gathered from the bits 7 of the four bytes by v_dot4_u32_u8 with the weights 1 2 4 8 (even items) / 16 32 64 128
// (odd items): the sum lands 7 bits up.
template <int PITCH>
__device__ __forceinline__ uint32_t compass_items(const uint8_t *win, int nitems /* uniform */, uint32_t dom, const CompassL &CK)
{
    uint32_t a01 = 0, a23 = 0, a45 = 0, a67 = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        if (i < nitems) {   // uniform
            const uint8_t *wi = win + i * PITCH;
            // the item's five dwords at non-negative offsets from (row - 3, column - 4): top | left, centre, right | bottom
            const uint32_t Tw = *reinterpret_cast<const uint32_t *>(wi + 4);
            const uint32_t Lw = *reinterpret_cast<const uint32_t *>(wi + 3 * PITCH);
            const uint32_t Cw = *reinterpret_cast<const uint32_t *>(wi + 3 * PITCH + 4);
            const uint32_t Rw = *reinterpret_cast<const uint32_t *>(wi + 3 * PITCH + 8);
            const uint32_t Bw = *reinterpret_cast<const uint32_t *>(wi + 6 * PITCH + 4);
            const uint32_t lft = __builtin_amdgcn_alignbyte(Cw, Lw, 1);   // bytes L1 L2 L3 C0 (column - 3)
            const uint32_t rgt = __builtin_amdgcn_alignbyte(Rw, Cw, 3);   // bytes C3 R0 R1 R2 (column + 3)
            const uint32_t z = compass4_loose(Cw, Tw, Bw, lft, rgt, CK) & dom;   // bits 7 of the bytes only
            const uint32_t W = (i & 1) ? 0x80402010u : 0x08040201u;
            uint32_t &a = i < 2 ? a01 : i < 4 ? a23 : i < 6 ? a45 : a67;
            a = __builtin_amdgcn_udot4(z, W, a, false);
        }
    }
    return (a01 >> 7) | (a23 << 1) | (a45 << 9) | (a67 << 17);
}


// ---- the ring gather of k_fast_fix: 17 byte loads at immediate offsets from one address, one wait ----
// Ring pixel k comes by ds_read_u8, its opposite k + 8 by ds_read_u8_d16_hi: the byte lands in bits 16..23 and, on this target
// (SRAM-ECC on: a d16 load rewrites the whole register -- the library is built for gfx950:sramecc+, csrc/Makefile, so a device
// that preserved the other half would refuse the code object instead of scoring garbage), the other half is cleared -- so P[k] = lo | hi needs no
// shift (v_lshlrev_b32 is a four-cycle instruction, profiles/r03/valu_rates.txt: eight of them per work-list entry), and the
// OR folds into the v_bitop3 that applies the polarity constant.  Issued from one asm block because the compiler cannot be
// told about d16 loads of separate registers; the offsets are spelled per pitch (an asm operand list holds 30 entries).
#define ORB_RING_ASM(PS)                                                                                                          \
    asm volatile("ds_read_u8 %0, %17 offset:6*" PS "+3\n\t"                                                                       \
                 "ds_read_u8 %1, %17 offset:6*" PS "+4\n\t"                                                                       \
                 "ds_read_u8 %2, %17 offset:5*" PS "+5\n\t"                                                                       \
                 "ds_read_u8 %3, %17 offset:4*" PS "+6\n\t"                                                                       \
                 "ds_read_u8 %4, %17 offset:3*" PS "+6\n\t"                                                                       \
                 "ds_read_u8 %5, %17 offset:2*" PS "+6\n\t"                                                                       \
                 "ds_read_u8 %6, %17 offset:" PS "+5\n\t"                                                                         \
                 "ds_read_u8 %7, %17 offset:4\n\t"                                                                                \
                 "ds_read_u8_d16_hi %8, %17 offset:3\n\t"                                                                         \
                 "ds_read_u8_d16_hi %9, %17 offset:2\n\t"                                                                         \
                 "ds_read_u8_d16_hi %10, %17 offset:" PS "+1\n\t"                                                                 \
                 "ds_read_u8_d16_hi %11, %17 offset:2*" PS "\n\t"                                                                 \
                 "ds_read_u8_d16_hi %12, %17 offset:3*" PS "\n\t"                                                                 \
                 "ds_read_u8_d16_hi %13, %17 offset:4*" PS "\n\t"                                                                 \
                 "ds_read_u8_d16_hi %14, %17 offset:5*" PS "+1\n\t"                                                               \
                 "ds_read_u8_d16_hi %15, %17 offset:6*" PS "+2\n\t"                                                               \
                 "ds_read_u8 %16, %17 offset:3*" PS "+3\n\t"                                                                      \
                 "s_waitcnt lgkmcnt(0)"                                                                                           \
                 : "=&v"(lo[0]), "=&v"(lo[1]), "=&v"(lo[2]), "=&v"(lo[3]), "=&v"(lo[4]), "=&v"(lo[5]), "=&v"(lo[6]), "=&v"(lo[7]), \
                   "=&v"(hi[0]), "=&v"(hi[1]), "=&v"(hi[2]), "=&v"(hi[3]), "=&v"(hi[4]), "=&v"(hi[5]), "=&v"(hi[6]), "=&v"(hi[7]), \
                   "=&v"(v0)                                                                                                      \
                 : "v"(a)                                                                                                         \
                 : "memory")
template <int PITCH>
__device__ __forceinline__ void ring_gather(uint32_t a, uint32_t (&lo)[8], uint32_t (&hi)[8], uint32_t &v0)
{
    static_assert(PITCH == 160 || PITCH == 176 || PITCH == 192 || PITCH == 208, "ring_gather: pitch without an instance");
    if constexpr (PITCH == 160) ORB_RING_ASM("160");
    else if constexpr (PITCH == 176) ORB_RING_ASM("176");
    else if constexpr (PITCH == 192) ORB_RING_ASM("192");
    else ORB_RING_ASM("208");
}
#undef ORB_RING_ASM

// fast_score_win / fast_score_dark on the gathered ring: a = LDS byte address of the top-left corner of the pixel's 7x7 window
template <int PITCH, bool DEFER>
__device__ __forceinline__ int fast_score_ring(uint32_t a, int t, bool *other)
{
    uint32_t lo[8], hi[8], v0u;
    ring_gather<PITCH>(a, lo, hi, v0u);
    const int v0 = (int)v0u;
    const int q0 = (int)lo[0], q4 = (int)lo[4], q8 = (int)(hi[0] >> 16), q12 = (int)(hi[4] >> 16);
    const bool pb = min(max(q0, q8), max(q4, q12)) > v0 + t;    // bright: q - v > t
    const bool pd = max(min(q0, q8), min(q4, q12)) < v0 - t;    // dark:   v - q > t
    if (!pb && !pd) return 0;
    const bool dark = !pb;
    const uint32_t C = dark ? 0x40FF40FFu : 0x40004000u;
    uint32_t P[8];
#pragma unroll
    for (int k = 0; k < 8; k++) P[k] = lo[k] | hi[k];
    int sc = arcs_score(P, C, v0 ^ (dark ? 0xFF : 0));
    if (DEFER) {
        *other = pb && pd && sc < t;
    } else if (pb && pd)
        sc = max(sc, arcs_score(P, 0x40FF40FFu, v0 ^ 0xFF));
    return sc >= t ? sc : 0;
}
template <int PITCH>
__device__ __forceinline__ int fast_score_ring_dark(uint32_t a, int t)
{
    uint32_t lo[8], hi[8], v0u;
    ring_gather<PITCH>(a, lo, hi, v0u);
    uint32_t P[8];
#pragma unroll
    for (int k = 0; k < 8; k++) P[k] = lo[k] | hi[k];
    const int sc = arcs_score(P, 0x40FF40FFu, (int)v0u ^ 0xFF);
    return sc >= t ? sc : 0;
}

// LDS of a workgroup of the fixed-layout kernel: the static arrays below + the score tile (largest: SP x DH with SP <= PITCH - 20,
// DH <= RHM - 6, + a padding row and the launcher's 16 + 256 bytes).  Eight workgroups per CU (8 x 20 KB of the 160 KB) hold
// for the 160- and 176-byte pitches at 40 staged rows -- every level of 640 x 480 and 752 x 480; the wider pitches and the tall
// instance run seven.
constexpr int ff_static_lds(int pitch, int rhm) { return pitch * rhm + FF_NCM * (rhm - 6) * 8 + FF_LISTCAP * 2 + FF_CORNERCAP * 2 + FF_GRPM * 4 + 8 * 4 + 3 * 4 + 4 * 2 * 4; }
constexpr int ff_max_lds(int pitch, int rhm) { return ff_static_lds(pitch, rhm) + (pitch - 20) * (rhm - 6) + 16 + 256; }
static_assert(ff_max_lds(176, FAST_FIX_ROWS) <= 160 * 1024 / 8, "k_fast_fix<176>: the eighth workgroup per CU no longer fits (FF_LISTCAP / FF_CORNERCAP)");
static_assert(ff_max_lds(160, FAST_FIX_ROWS) <= 160 * 1024 / 8, "k_fast_fix<160>: the eighth workgroup per CU no longer fits");
static_assert(ff_max_lds(208, FF_RHM_TALL) <= 160 * 1024 / 6, "k_fast_fix<208, tall>: fewer than six workgroups per CU");

template <int PITCH, bool DEFER, int RHM>
__global__ __launch_bounds__(256, 8) void k_fast_fix(const uint8_t *__restrict__ lvl0, int stride0, unsigned long long frame0,
                                                     const uint8_t *__restrict__ pyr, unsigned long long pyrFrame,
                                                     const FastTile *__restrict__ tiles, uint32_t *__restrict__ cand,
                                                     uint16_t *__restrict__ cellCnt, int totalCells, int totalCands, int iniTh,
                                                     int minTh, int listCap, int cornerCap, int xcdMap, int ntiles ORB_ABL_PARAM)
{
    __shared__ __align__(16) uint8_t s_pix[PITCH * RHM];
    __shared__ unsigned long long s_bits[FF_NCM * (RHM - 6)];   // [cell][row]: survivors of the row (wCell < 64)
    __shared__ uint16_t s_list[FF_LISTCAP];
    __shared__ uint16_t s_corner[FF_CORNERCAP];
    __shared__ uint32_t s_dom[FF_GRPM];   // per dword group: which of its 4 pixels lie in the domain (bit 8k + 7)
    __shared__ int s_cellAny[8];
    __shared__ int s_listCount, s_cornerCount, s_deferCount;
    __shared__ int s_wcnt[4][2];          // pass 1, per wave: work-list and corner-list counts
    extern __shared__ __align__(16) uint8_t s_score[];       // [DH][SP]

#if ORB_FAST_TILE_BY_VALUE
    // (every kernel argument fetched together, in front of the descriptor: the compiler otherwise fetches each where it is first used)
    asm volatile("" ::"s"(stride0), "s"(frame0), "s"(pyrFrame), "s"(totalCells), "s"(totalCands), "s"(iniTh), "s"(minTh), "s"(listCap),
                 "s"(cornerCap), "s"(xcdMap), "s"(ntiles), "s"((unsigned long long)(uintptr_t)lvl0), "s"((unsigned long long)(uintptr_t)pyr),
                 "s"((unsigned long long)(uintptr_t)cand), "s"((unsigned long long)(uintptr_t)cellCnt));
#endif
    int tileId, frame;
    if ((xcdMap & 255) == 4) {
        // grid (8, runs, ceil(B / 8)): workgroups are dealt to the XCDs round-robin by linear id = (z * runs + y) * 8 + x, so XCD x
        // receives run y of frame x + 8 z, y = 0, 1, ... -- a whole frame's runs on one XCD, whose L2 then reads the frame's
        // levels (1 MB of its 4 MB) once; the eight XCDs work on eight frames.  No division: the ids are the grid's own.
        tileId = blockIdx.y;
        frame = blockIdx.x + 8 * blockIdx.z;
        if (frame >= (xcdMap >> 8)) return;
    } else {
        tileId = xcd_tile(xcdMap);
        frame = blockIdx.y;
    }
    if (tileId >= ntiles) return;   // grid padded to a multiple of 8 (orbhip_internal.h, xcd_tile)
#if ORB_FAST_TILE_BY_VALUE
    // The run's descriptor in ONE scalar round trip: read field by field where it is first used (the reference below) the
    // prologue is a chain of seven dependent scalar loads in front of the first LDS-DMA -- ~1 us of a workgroup's ~8 us.
    FastTile T;
    {
        const uint4 *tp = reinterpret_cast<const uint4 *>(tiles + tileId);
        uint4 q0 = tp[0], q1 = tp[1], q2 = tp[2], q3 = tp[3], q4 = tp[4], q5 = tp[5];
        asm volatile("" : "+s"(q0.x), "+s"(q0.y), "+s"(q0.z), "+s"(q0.w), "+s"(q1.x), "+s"(q1.y), "+s"(q1.z), "+s"(q1.w), "+s"(q2.x), "+s"(q2.y),
                          "+s"(q2.z), "+s"(q2.w), "+s"(q3.x), "+s"(q3.y), "+s"(q3.z), "+s"(q3.w), "+s"(q4.x), "+s"(q4.y), "+s"(q4.z), "+s"(q4.w),
                          "+s"(q5.x), "+s"(q5.y), "+s"(q5.z), "+s"(q5.w));
        const uint4 q[6] = {q0, q1, q2, q3, q4, q5};
        __builtin_memcpy(&T, q, sizeof(T));
    }
#else
    const FastTile &T = tiles[tileId];
#endif
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int DH = T.DH, TW = T.TW, ncells = T.nc;
    if (DH == 0) {
        if (tid < ncells) cellCnt[(size_t)frame * totalCells + T.cntOff + tid] = 0;
        return;
    }
    constexpr int pitch = PITCH;
    const uint32_t pixAddr = (uint32_t)(uintptr_t)s_pix;   // LDS byte address of the pixel tile
    const int RH = T.RH, nchunk = T.nchunk, j0 = T.j0, jd0 = j0 & ~3, GPR = T.GPR;
    const int SP = (TW + 3) & ~3;
    const int nrowsAll = ncells * DH;

    // ---- 1. stage pixels [iniY, iniY + RH) x [xa, xa + 16 nchunk) by LDS-DMA; clear the score tile and the bitmap ----
    {
        // a wave transfer writes 64 x 16 bytes to consecutive LDS addresses, so with the row pitch a multiple of 16
        // lane = row * (PITCH / 16) + chunk covers 64 / (PITCH / 16) whole rows; the address per lane is built once, a round
        // adds a multiple of the stride.  Waited for (vmcnt) before the barrier below: the compiler does not know of them.
        const bool isL0 = T.lvlOff == 0xFFFFFFFFu;
        const int stride = isL0 ? stride0 : T.stride;
        const uint8_t *img = isL0 ? lvl0 + (size_t)frame * frame0 : pyr + (size_t)frame * pyrFrame + T.lvlOff;
        const uint8_t *img0 = img + (unsigned)(__mul24(T.iniY, stride) + T.xa);
        constexpr int CPW = PITCH / 16, RPW = 64 / CPW;
        const int rl = lane / CPW, ch = lane - rl * CPW;
        const bool laneOk = rl < RPW && ch < nchunk;
        const uint8_t *src = img0 + (unsigned)(__mul24(rl, stride) + (ch << 4));
        const uint32_t ldsBase = (uint32_t)(uintptr_t)s_pix;
        for (int rb = wv * RPW; rb < RH; rb += 4 * RPW)
            if (laneOk && rb + rl < RH) glds16(src + (unsigned)__mul24(rb, stride), ldsBase + (uint32_t)(rb * PITCH));
    }
    for (int i = tid; i < (DH * SP + 15) >> 4; i += 256) reinterpret_cast<uint4 *>(s_score)[i] = make_uint4(0u, 0u, 0u, 0u);
    for (int i = tid; i < nrowsAll; i += 256) s_bits[i] = 0ull;
    // the mask of a group is its overlap with the domain columns [0, TW)
    if (tid < GPR) {
        uint32_t m = 0;
#pragma unroll
        for (int k = 0; k < 4; k++)
            if ((unsigned)(jd0 + (tid << 2) + k - j0) < (unsigned)TW) m |= 0x80u << (8 * k);
        s_dom[tid] = m;
    }
    if (tid < 8) s_cellAny[tid] = 0;
    if (tid == 0) {
        s_listCount = 0;
        s_cornerCount = 0;
        s_deferCount = 0;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    ORB_ABL_STOP(phases < 2);   // timing ablation only (liborbhip_ablation.so, ORBHIP_FAST_PHASES): results are then invalid

    const int wCell = T.wCell;
    const unsigned cellMagic = (unsigned)T.cellMagic;
    // list entries (16 bits): (dword group of the row << 9) | (row << 2) | pixel within the group; LDS column = jd0 + 4 group + pixel
#define ENT_ROW(e) (((e) >> 2) & 127)
#define ENT_COL(e) (jd0 + (((e) >> 9) << 2) + ((e) & 3))
#define ENT_MAKE(r, j) (((((j) - jd0) >> 2) << 9) | ((r) << 2) | (((j) - jd0) & 3))

    // =================================== pass 0: every cell at iniThFAST, the whole workgroup ===================================
    // (a threshold of 255 admits no corner: no pixel differs from its centre by more)
    if (iniTh < 255) {
        const int t = iniTh;
        // ---- 2. compass pre-test, 4 pixels per item; survivors -> work list ----
        {
            const CompassL CK = compass_loose_consts(t);
            // thread -> (dword column, segment of consecutive rows): tid = sidx * GPR + slot
            const int sidx = (int)(((unsigned)tid * (unsigned)T.grpMagic) >> 16);
            const int seg = T.seg;
            const int slot = tid - __mul24(sidx, GPR);
            const int rs = __mul24(sidx, seg);               // first row of my segment
            const bool mine = rs < DH;
            const int jd = jd0 + (slot << 2);
            const uint32_t dom = mine ? s_dom[slot] : 0u;   // domain mask of my four pixels (pixel k at bit 8k + 7)
            const uint32_t listCountAddr = (uint32_t)(uintptr_t)&s_listCount;   // LDS byte address (low half of the flat address)
            const uint8_t *win = s_pix + __mul24(mine ? rs : 0, pitch) + (jd - 4);
            ORB_ABL_STOP(phases == 11);   // ablation only: the set-up of the compass phase
            for (int cb = 0; cb < seg; cb += 8) {
                uint32_t acc = compass_items<PITCH>(win, seg - cb, dom, CK);
                // rows of the last segment below the domain were computed on whatever lies there: drop them
                const int live = DH - rs - cb;   // (<= 0 only for threads that are not `mine`)
                if (live < 8) acc &= (1u << (4 * max(live, 0))) - 1u;
                win += 8 * pitch;
                // append this thread's survivors to the work list (order is irrelevant): one returning LDS add per thread
                // claims its range.  A thread whose entries do not all fit writes none: s_listCount then exceeds listCap
                // and phase 3 takes the fallback.
                const int n = __popc(acc);
                ORB_ABL_IF(phases == 12) {   // ablation only: the compass items without the list
                    if (n == 77) s_list[0] = (uint16_t)n;
                    continue;
                }
                if (n > 0) {
                    // (inline asm: hipcc's atomic optimiser would turn a plain atomicAdd of a per-lane value back into a DPP scan)
                    int base;
                    asm volatile("ds_add_rtn_u32 %0, %1, %2\n\ts_waitcnt lgkmcnt(0)" : "=v"(base) : "v"(listCountAddr), "v"(n) : "memory");
                    if (base + n <= listCap)
                        list_write_entries(acc, (uint32_t)(uintptr_t)s_list + 2u * (uint32_t)base, (uint32_t)((slot << 9) | ((rs + cb) << 2)));
                }
            }
        }
        __syncthreads();
        ORB_ABL_STOP(phases < 3 || phases == 12);   // ablation stops: 2-4 = phases of pass 0
        // Ablation only (liborbhip_ablation.so, ORBHIP_FAST_PHASES=13 / 14; r05 review item 1a): the work list regrouped BY ROW in front
        // of the score phase -- a counting sort over the entries' rows through the (still empty) score tile -- so that a wave's 64
        // entries come from two or three rows.  Stop 13 ends after the sort, stop 14 after the score phase on the sorted list: the
        // difference of their LDS counters is the ring gather's cost with a row-grouped list, to set beside (stop 3 - stop 2).
        ORB_ABL_IF(phases == 13 || phases == 14) {
            const int nl = s_listCount;
            if (nl <= listCap && nl * 2 <= DH * SP) {
                uint32_t *hist = reinterpret_cast<uint32_t *>(s_corner);   // [64] rows' counts, [64] rows' cursors
                uint16_t *tmp = reinterpret_cast<uint16_t *>(s_score);
                if (tid < 128) hist[tid] = 0;
                __syncthreads();
                for (int e = tid; e < nl; e += 256) atomicAdd(&hist[ENT_ROW(s_list[e])], 1u);
                __syncthreads();
                if (tid == 0) {
                    uint32_t a = 0;
                    for (int r = 0; r < 64; r++) {
                        hist[64 + r] = a;
                        a += hist[r];
                    }
                }
                __syncthreads();
                for (int e = tid; e < nl; e += 256) {
                    const int ent = s_list[e];
                    tmp[atomicAdd(&hist[64 + ENT_ROW(ent)], 1u)] = (uint16_t)ent;
                }
                __syncthreads();
                for (int e = tid; e < nl; e += 256) s_list[e] = tmp[e];
                __syncthreads();
                for (int i = tid; i < (DH * SP + 15) >> 4; i += 256) reinterpret_cast<uint4 *>(s_score)[i] = make_uint4(0u, 0u, 0u, 0u);
                __syncthreads();
            }
        }
        ORB_ABL_STOP(phases == 13);

        // ---- 3. full score on the work list; corners (score >= t) -> score tile + corner list ----
        // If a tile has more compass survivors than the work list holds (noise-like images), every domain pixel is scored
        // instead (the compass test is the early-out of fast_score_pol); if it has more corners than the corner list holds,
        // phase 4 scans the score tile.  Both fallbacks produce the same result.
        const int nlist = s_listCount;
        const uint32_t cornerCountAddr = (uint32_t)(uintptr_t)&s_cornerCount;
        auto put_corner = [&](int ent, int r, int j, int s) {
            s_score[__mul24(r, SP) + (j - j0)] = (uint8_t)s;
            // one returning LDS add per corner (a fifth of the lanes; hipcc's wave aggregation of atomicAdd(p, 1) costs
            // a dozen vector instructions per iteration for every lane)
            int slot;
            asm volatile("ds_add_rtn_u32 %0, %1, %2\n\ts_waitcnt lgkmcnt(0)" : "=v"(slot) : "v"(cornerCountAddr), "v"(1) : "memory");
            if (slot < cornerCap) s_corner[slot] = (uint16_t)ent;
        };
        if (nlist <= listCap) {
            if (DEFER) {
                const uint32_t deferCountAddr = (uint32_t)(uintptr_t)&s_deferCount;
                for (int e = tid; e < nlist; e += 256) {
                    const int ent = s_list[e];
                    const int r = ENT_ROW(ent), j = ENT_COL(ent);
                    bool other;
                    int s = fast_score_ring<PITCH, true>(pixAddr + (uint32_t)(__mul24(r, pitch) + (j - 3)), t, &other);
                    if (other) {
                        // park the entry at the unused end of the work list (slots >= nlist are never read by this loop);
                        // if the list is full to that point, finish the pixel here
                        int d;
                        asm volatile("ds_add_rtn_u32 %0, %1, %2\n\ts_waitcnt lgkmcnt(0)" : "=v"(d) : "v"(deferCountAddr), "v"(1) : "memory");
                        const int idx = listCap - 1 - d;
                        if (idx >= nlist) s_list[idx] = (uint16_t)ent;
                        else s = fast_score_ring_dark<PITCH>(pixAddr + (uint32_t)(__mul24(r, pitch) + (j - 3)), t);
                    }
                    if (s > 0) put_corner(ent, r, j, s);
                }
                __syncthreads();
                // the parked entries, dark polarity only (dense lanes again)
                const int ndefer = min(s_deferCount, listCap - nlist);
                for (int e = tid; e < ndefer; e += 256) {
                    const int ent = s_list[listCap - 1 - e];
                    const int r = ENT_ROW(ent), j = ENT_COL(ent);
                    const int s = fast_score_ring_dark<PITCH>(pixAddr + (uint32_t)(__mul24(r, pitch) + (j - 3)), t);
                    if (s > 0) put_corner(ent, r, j, s);
                }
            } else {
                for (int e = tid; e < nlist; e += 256) {
                    const int ent = s_list[e];
                    const int r = ENT_ROW(ent), j = ENT_COL(ent);
                    const int s = fast_score_ring<PITCH, false>(pixAddr + (uint32_t)(__mul24(r, pitch) + (j - 3)), t, nullptr);
                    if (s > 0) put_corner(ent, r, j, s);
                }
            }
        } else {
            int TWx = TW;
            asm volatile("" : "+s"(TWx));            // (keeps the division below inside this branch)
            const float invTW = 1.0f / (float)TWx;   // px / TW = floor((px + 0.5) * invTW): exact for every px < DH * TW (a 20-bit
                                                      // integer reciprocal is NOT: it fails from px ~ 2^20 / TW on, e.g. TW 155, DH 45)
            for (int p0 = 0; p0 < DH * TWx; p0 += 256) {
                const int px = p0 + tid;
                int ent = 0, s = 0;
                if (px < DH * TWx) {
                    const int r = (int)(((float)px + 0.5f) * invTW);
                    const int c = px - r * TWx;
                    s = fast_score_pol(s_pix + (r + 3) * pitch + j0 + c, pitch, t);
                    if (s > 0) s_score[r * SP + c] = (uint8_t)s;
                    ent = ENT_MAKE(r, j0 + c);
                }
                const int slot = wave_append(s > 0, &s_cornerCount, lane);
                if (slot >= 0 && slot < cornerCap) s_corner[slot] = (uint16_t)ent;
            }
        }
        __syncthreads();
        ORB_ABL_STOP(phases < 4 || phases == 14);

        // ---- 4. NMS over the corners (cell-local neighbourhood); every survivor is final: set its bit ----
        const int ncorner = s_cornerCount;
        if (ncorner <= cornerCap) {
            for (int e = tid; e < ncorner; e += 256) {
                const int ent = s_corner[e];
                const int r = ENT_ROW(ent), c = ENT_COL(ent) - j0;
                const int cj = (int)(((unsigned)c * cellMagic) >> 16);
                if (nms_survives_fix(s_score, SP, r, c, DH, TW, wCell, cj)) {
                    atomicOr(&s_bits[cj * DH + r], 1ull << (c - cj * wCell));
                    s_cellAny[cj] = 1;   // benign race: every writer stores 1
                }
            }
        } else {
            // fallback: scan the score tile, 4 pixels per dword (every non-zero score is a corner of this pass)
            int SPx = SP;
            asm volatile("" : "+s"(SPx));
            const int SPW = SPx >> 2;                                       // score dwords per row
            const unsigned spwMagic = (1u << 20) / (unsigned)SPW + 1u;
            const int nwords = DH * SPW;
            for (int i = tid; i < nwords; i += 256) {
                const uint32_t w = reinterpret_cast<const uint32_t *>(s_score)[i];
                if (w == 0) continue;
                const int r = (int)(((unsigned)i * spwMagic) >> 20);
                const int cb = (i - r * SPW) << 2;
                for (int k = 0; k < 4; k++) {
                    const int s = (w >> (8 * k)) & 0xFF;
                    const int c = cb + k;
                    if (s < t || c >= TW) continue;
                    const int cj = (int)(((unsigned)c * cellMagic) >> 16);
                    if (nms_survives_fix(s_score, SP, r, c, DH, TW, wCell, cj)) {
                        atomicOr(&s_bits[cj * DH + r], 1ull << (c - cj * wCell));
                        s_cellAny[cj] = 1;
                    }
                }
            }
        }
    }
    __syncthreads();
    ORB_ABL_STOP(phases < 5);

    // ============ pass 1 and the output: ONE WAVE PER CELL, no workgroup barrier from here on ============
    // The cells without a survivor are searched again at minThFAST (:814-818).  A cell's second pass touches nothing outside the
    // cell -- its pixels' scores, the suppression among them, its rows of the bitmap -- so a wave does all of it alone: compass
    // test over the cell's dword groups (lane = group x row segment), its own quarter of the work list and of the corner list,
    // score, suppression; then, for every cell (searched again or not), the ranks and the candidate slots.  (The r02 kernel ran
    // pass 1 as three more workgroup phases with four barriers: half of the runs have an empty cell, and a run's second pass
    // is ~300 items -- one wave's worth -- so three waves waited at every one of them.)
    const bool pass1 = minTh < iniTh && minTh < 255;   // FAST(ini) empty => FAST(min >= ini) empty
    constexpr int WLCAP = FF_LISTCAP / 4, WCCAP = FF_CORNERCAP / 4;
    uint16_t *const wlist = s_list + wv * WLCAP;
    uint16_t *const wcorner = s_corner + wv * WCCAP;
    const int wlcap = min(listCap, WLCAP), wccap = min(cornerCap, WCCAP);   // (forced small in the tests)
    uint32_t *candRun = cand + (size_t)frame * totalCands + T.candOff;
    uint16_t *cntRun = cellCnt + (size_t)frame * totalCells + T.cntOff;
    for (int cj = wv; cj < ncells; cj += 4) {
        const int cx0 = __mul24(cj, wCell), cx1 = min(cx0 + wCell, TW);   // the cell's domain columns
        if (pass1 && !__builtin_amdgcn_readfirstlane(s_cellAny[cj])) {
            // (empty asm: what follows belongs to this branch -- the compiler otherwise computes the lanes' set-up for every
            // cell, searched again or not: 40 vector instructions per cell, a twelfth of the kernel)
            int t = minTh, lanep = lane, cxa = cx0;
            asm volatile("" : "+s"(t), "+v"(lanep), "+s"(cxa));
            const CompassL CK = compass_loose_consts(t);
            // the cell's dword groups (absolute index in the staged row) and the lanes' (group, row segment)
            const int gA = (j0 + cxa) >> 2, ng = ((j0 + cx1 - 1) >> 2) - gA + 1;
            const float invNg = __builtin_amdgcn_rcpf((float)ng);
            // (v_rcp_f32 is within 1 ulp: the products below are off by < 2^-13, the quotients are >= 1 / (2 n) away from an integer)
            const int S = __builtin_amdgcn_readfirstlane((int)(64.5f * invNg));                                           // >= 1 (ng <= 64 / 4 + 2)
            const int seg = __builtin_amdgcn_readfirstlane((int)(((float)(DH + S - 1) + 0.5f) * __builtin_amdgcn_rcpf((float)S)));
            const int sidx = (int)(((float)lanep + 0.5f) * invNg);
            const int slot = lanep - __mul24(sidx, ng);
            const int rs = __mul24(sidx, seg);
            const bool mine = rs < DH;
            const int jg = (gA + slot) << 2;                  // staged column of my group's pixel 0
            uint32_t dom = 0;
#pragma unroll
            for (int k = 0; k < 4; k++)
                if (mine && jg + k >= j0 + cxa && jg + k < j0 + cx1) dom |= 0x80u << (8 * k);
            if (lanep == 0) {
                s_wcnt[wv][0] = 0;
                s_wcnt[wv][1] = 0;
            }
            const uint32_t wlAddr = (uint32_t)(uintptr_t)&s_wcnt[wv][0], wcAddr = (uint32_t)(uintptr_t)&s_wcnt[wv][1];
            const uint8_t *win = s_pix + __mul24(mine ? rs : 0, pitch) + (jg - 4);
            for (int cb = 0; cb < seg; cb += 8) {
                uint32_t acc = compass_items<PITCH>(win, seg - cb, dom, CK);
                const int live = DH - rs - cb;
                if (live < 8) acc &= (1u << (4 * max(live, 0))) - 1u;
                win += 8 * pitch;
                const int n = __popc(acc);
                if (n > 0) {
                    int base;
                    asm volatile("ds_add_rtn_u32 %0, %1, %2\n\ts_waitcnt lgkmcnt(0)" : "=v"(base) : "v"(wlAddr), "v"(n) : "memory");
                    if (base + n <= wlcap)
                        list_write_entries(acc, (uint32_t)(uintptr_t)wlist + 2u * (uint32_t)base,
                                           (uint32_t)(((gA + slot - (jd0 >> 2)) << 9) | ((rs + cb) << 2)));
                }
            }
            // (LDS operations of one wave complete in order: the counts read below include every lane's add above)
            const int nl = __builtin_amdgcn_readfirstlane(s_wcnt[wv][0]);
            auto put_corner1 = [&](int ent, int r, int j, int s) {
                s_score[__mul24(r, SP) + (j - j0)] = (uint8_t)s;
                int slot1;
                asm volatile("ds_add_rtn_u32 %0, %1, %2\n\ts_waitcnt lgkmcnt(0)" : "=v"(slot1) : "v"(wcAddr), "v"(1) : "memory");
                if (slot1 < wccap) wcorner[slot1] = (uint16_t)ent;
            };
            const int cw = cx1 - cx0;
            if (nl <= wlcap) {
                for (int e = lane; e < nl; e += 64) {
                    const int ent = wlist[e];
                    const int r = ENT_ROW(ent), j = ENT_COL(ent);
                    const int s = fast_score_ring<PITCH, false>(pixAddr + (uint32_t)(__mul24(r, pitch) + (j - 3)), t, nullptr);
                    if (s > 0) put_corner1(ent, r, j, s);
                }
            } else {
                // more survivors than this wave's part of the list holds: every pixel of the cell
                const float invCw = __builtin_amdgcn_rcpf((float)cw);
                for (int p0 = 0; p0 < DH * cw; p0 += 64) {
                    const int px = p0 + lane;
                    if (px < DH * cw) {
                        const int r = (int)(((float)px + 0.5f) * invCw);   // (exact: px < 64 * 64, see above)
                        const int c = cx0 + px - r * cw;
                        const int s = fast_score_pol(s_pix + (r + 3) * pitch + j0 + c, pitch, t);
                        if (s > 0) put_corner1(ENT_MAKE(r, j0 + c), r, j0 + c, s);
                    }
                }
            }
            const int ncor = __builtin_amdgcn_readfirstlane(s_wcnt[wv][1]);
            if (ncor <= wccap) {
                for (int e = lane; e < ncor; e += 64) {
                    const int ent = wcorner[e];
                    const int r = ENT_ROW(ent), c = ENT_COL(ent) - j0;
                    if (nms_survives_fix(s_score, SP, r, c, DH, TW, wCell, cj)) atomicOr(&s_bits[cj * DH + r], 1ull << (c - cx0));
                }
            } else {
                // more corners than this wave's part of the corner list holds: scan the cell's scores
                const float invCw = __builtin_amdgcn_rcpf((float)cw);
                for (int p0 = 0; p0 < DH * cw; p0 += 64) {
                    const int px = p0 + lane;
                    if (px < DH * cw) {
                        const int r = (int)(((float)px + 0.5f) * invCw);
                        const int c = cx0 + px - r * cw;
                        if (s_score[r * SP + c] >= t && nms_survives_fix(s_score, SP, r, c, DH, TW, wCell, cj))
                            atomicOr(&s_bits[cj * DH + r], 1ull << (c - cx0));
                    }
                }
            }
        }
        ORB_ABL_IF(phases < 8) continue;
        // ---- 5. rank inside the cell (= raster order) from the bitmap, write the slots: lane = row (DH <= 34) ----
        // The rank of a survivor is the number of bits before it: a prefix over the rows of its cell plus a popcount inside its
        // row -- no survivor is ever compared with another one.
        unsigned long long bits = lane < DH ? s_bits[cj * DH + lane] : 0ull;
        const int n = __popcll(bits);
        const int incl = wave_incl_scan_dpp(n);
        int rank = incl - n;
        if (lane == 63) cntRun[cj] = (uint16_t)incl;
        uint32_t *slotp = candRun + __mul24(cj, T.cellCap);
        const int py = T.py0 + lane;                                      // relative to (16,16), :824-825
        const int pxBase = T.px0 + cx0;
        const uint8_t *sc = s_score + lane * SP + cx0;
        // (the row's bits as two 32-bit halves: cells of the usual grid are 31 or 32 pixels wide, the 64-bit walk costs twice)
        const uint32_t posBase = (uint32_t)pxBase | ((uint32_t)py << 12);
        uint32_t half = (uint32_t)bits;
        while (half) {
            const int cl = __ffs(half) - 1;
            half &= half - 1u;
            slotp[rank++] = (posBase + (uint32_t)cl) | ((uint32_t)sc[cl] << 24);
        }
        half = (uint32_t)(bits >> 32);
        while (half) {
            const int cl = __ffs(half) + 31;
            half &= half - 1u;
            slotp[rank++] = (posBase + (uint32_t)cl) | ((uint32_t)sc[cl] << 24);
        }
    }
    ORB_ABL_IF(phases < 8 && tid < ncells) cntRun[tid] = 0;
#undef ENT_ROW
#undef ENT_COL
#undef ENT_MAKE
}

// the runs tiles[0 .. ntiles) of the levels in levelMask, one launch
static void launch_fast_levels(hipStream_t s, const OrbLevels &G, const uint8_t *lvl0, int stride0, size_t frame0,
                               const uint8_t *pyr, size_t pyrFrame, const FastTile *tiles, int ntiles,
                               uint32_t *cand, uint16_t *cellCnt, int B, unsigned levelMask)
{
    // LDS: pixel tile + score tile + work list of the largest run over the levels of this launch
    int pixBytes = 0, scoreBytes = 0, listBytes = 0, survBytes = 0, bitsRows = 0, maxPitch = 0, maxRh = 0;
    for (int l = 0; l < G.nlevels; l++) {
        if (!((levelMask >> l) & 1u)) continue;
        const OrbLevel &L = G.lv[l];
        int tileCells = fast_tile_cells();
        while (tileCells > 1 && tileCells * L.wCell + 6 + 16 > FAST_MAX_TILE_W) tileCells--;
        const int regw = tileCells * L.wCell + 6;
        const int pitch = ((regw + 15 + 15) >> 4) << 4;
        const int rh = L.hCell + 6;
        const int sp = (tileCells * L.wCell + 3) & ~3;
        pixBytes = std::max(pixBytes, pitch * rh);
        maxPitch = std::max(maxPitch, pitch);
        maxRh = std::max(maxRh, rh);
        scoreBytes = std::max(scoreBytes, sp * L.hCell);
        // work list: u16 per domain pixel; the survivor list (u32 per strict local maximum, at most
        // a quarter of the pixels plus cell seams) reuses the same storage
        listBytes = std::max(listBytes, sp * L.hCell * 2);
        survBytes = std::max(survBytes, tileCells * L.cellCap * 4);
        bitsRows = std::max(bitsRows, tileCells * L.hCell);
    }
    pixBytes = (pixBytes + 15) & ~15;
    scoreBytes = (scoreBytes + 15) & ~15;
    static const int forced = ORB_TUNE("FAST_LISTCAP", 0);   // tests force every list to overflow
    static const int phases = ORB_TUNE("FAST_PHASES", 99);
    (void)phases;
    static const int fastXcd = ORB_TUNE("FAST_XCD", FAST_DEFAULT_XCD);   // workgroup -> (run, frame) of the fixed-layout kernel
    dim3 grid(orb_xcd_grid(ntiles), B, 1), block(256, 1, 1);
    // r02 kernel: work list = a quarter of the tile's pixels (the first pass runs at iniThFAST), corner list = a sixteenth;
    // bitmap and row prefix per (cell, row).  Tiles that exceed the lists take the exact fallback paths.
    const int px = listBytes / 2;                      // sp * hCell of the largest tile
    int listCap = px / 4, cornerCap = px / 16;
    // (the fixed-layout kernel: up to a half (FF_LISTCAP) and an eighth -- on densely textured frames 5 % of the runs exceeded a
    // quarter / a sixteenth, and their fallbacks (every pixel scored; score tile scanned) were a third of the kernel's score work)
    const int listCapFix = std::min(px / 2, FF_LISTCAP), cornerCapFix = std::min(px / 8, FF_CORNERCAP);
    if (forced > 0) listCap = cornerCap = forced;
    const int lBytes = (listCap * 2 + 15) & ~15, cBytes = (cornerCap * 2 + 15) & ~15;
    const int bitsBytes = (bitsRows * 8 + 15) & ~15, preBytes = (bitsRows * 4 + 15) & ~15;
    // the kernel with a compile-time pitch (176 / 192 / 208: five cells of 31..36 pixels + halo + alignment, i.e. every level of
    // the usual 30-pixel cell grid) when no level needs more, the run-time pitch otherwise (ORBHIP_FAST_PITCH=0 forces the latter)
    static const int pitchEnv = ORB_TUNE("FAST_PITCH", 1);
    const int fixed = pitchEnv == 0 ? 0 : maxPitch <= 176 ? 176 : maxPitch <= 192 ? 192 : maxPitch <= 208 ? 208 : 0;
    const int fixedFix = fixed && maxPitch <= 160 ? 160 : fixed;   // the fixed-layout kernel also has a 160-byte instance
    if (fixed) pixBytes = (fixed * maxRh + 15) & ~15;
    const size_t lds = (size_t)(pixBytes + scoreBytes + lBytes + cBytes + bitsBytes + preBytes);
    // the fixed-layout kernel when every level fits its bounds (ORBHIP_FAST_FIX=0: the generic kernel)
    static const int fixEnv = ORB_SWITCH("FAST_FIX", 1);
    static const int deferEnv = ORB_TUNE("FAST_DEFER", 1);
    const int maxCells = fast_tile_cells();   // the configured run length bounds every tile's
    if (fixEnv && fixed && maxRh <= FF_RHM_TALL && maxCells <= FF_NCM) {
        const bool tall = maxRh > FF_RHM;
        const int lc = forced > 0 ? std::min(forced, FF_LISTCAP) : listCapFix, cc = forced > 0 ? std::min(forced, FF_CORNERCAP) : cornerCapFix;
        static const int ldsPad = ORB_TUNE("FAST_LDS_PAD", 0);   // occupancy experiments (liborbhip_ablation.so)
        const size_t ldsScore = (size_t)scoreBytes + 16 + 256 + (size_t)ldsPad;   // + a row: nms_survives_fix reads one below the tile
        const bool perXcd = fastXcd == 4 && B >= 8;   // (a frame or two: the runs over all XCDs)
        if (perXcd) grid = dim3(8, ntiles, (B + 7) / 8);
        orb_path(ORB_PATH_FAST_FIX | (tall ? ORB_PATH_FAST_TALL : 0u));
#define ORB_LAUNCH_FIX(P, D, R)                                                                                              \
    hipLaunchKernelGGL((k_fast_fix<P, D, R>), grid, block, ldsScore, s, lvl0, stride0, (unsigned long long)frame0, pyr,      \
                       (unsigned long long)pyrFrame, tiles, cand, cellCnt, G.totalCells, G.totalCands, G.iniTh, G.minTh, lc, \
                       cc, perXcd ? (4 | (B << 8)) : orb_xcd_arg(), ntiles ORB_ABL_ARG(phases))
#define ORB_LAUNCH_FIX_P(D, R)                             \
    switch (fixedFix) {                                    \
    case 160: ORB_LAUNCH_FIX(160, D, R); break;            \
    case 176: ORB_LAUNCH_FIX(176, D, R); break;            \
    case 192: ORB_LAUNCH_FIX(192, D, R); break;            \
    default: ORB_LAUNCH_FIX(208, D, R); break;             \
    }
        if (deferEnv) {
            if (tall) {
                ORB_LAUNCH_FIX_P(true, FF_RHM_TALL)
            } else {
                ORB_LAUNCH_FIX_P(true, FF_RHM)
            }
        } else {
            if (tall) {
                ORB_LAUNCH_FIX_P(false, FF_RHM_TALL)
            } else {
                ORB_LAUNCH_FIX_P(false, FF_RHM)
            }
        }
#undef ORB_LAUNCH_FIX_P
#undef ORB_LAUNCH_FIX
        return;
    }
#define ORB_LAUNCH_FAST(P)                                                                                                   \
    hipLaunchKernelGGL(k_fast<P>, grid, block, lds, s, G, lvl0, stride0, (unsigned long long)frame0, pyr,                    \
                       (unsigned long long)pyrFrame, tiles, cand, cellCnt, pixBytes, scoreBytes, lBytes, cBytes, bitsBytes,  \
                       listCap, cornerCap, orb_xcd_arg(), ntiles ORB_ABL_ARG(phases))
    orb_path(ORB_PATH_FAST_GENERIC);
    switch (fixed) {
    case 176: ORB_LAUNCH_FAST(176); break;
    case 192: ORB_LAUNCH_FAST(192); break;
    case 208: ORB_LAUNCH_FAST(208); break;
    default: ORB_LAUNCH_FAST(0); break;
    }
#undef ORB_LAUNCH_FAST
}

// Batches: the run list is ordered [levels whose cells fit the 34-row instance | levels with taller cells] (orb_geometry.hip), ntall =
// length of the second part.  When both parts exist they are two launches, so that the taller cells of one or two small levels
// (1241 x 376, 1280 x 720, 960 x 540 ... end in them) do not put every level on the instance with seven workgroups per CU.
void launch_fast(hipStream_t s, const OrbLevels &G, const uint8_t *lvl0, int stride0, size_t frame0,
                 const uint8_t *pyr, size_t pyrFrame, const FastTile *tiles, int ntiles,
                 uint32_t *cand, uint16_t *cellCnt, int B, int ntall)
{
    unsigned all = 0, tallMask = 0;
    for (int l = 0; l < G.nlevels; l++) {
        all |= 1u << l;
        if (G.lv[l].hCell + 6 > FAST_FIX_ROWS) tallMask |= 1u << l;
    }
    if (ntall > 0 && ntall < ntiles && B >= 8) {
        launch_fast_levels(s, G, lvl0, stride0, frame0, pyr, pyrFrame, tiles, ntiles - ntall, cand, cellCnt, B, all & ~tallMask);
        launch_fast_levels(s, G, lvl0, stride0, frame0, pyr, pyrFrame, tiles + (ntiles - ntall), ntall, cand, cellCnt, B, tallMask);
    } else
        launch_fast_levels(s, G, lvl0, stride0, frame0, pyr, pyrFrame, tiles, ntiles, cand, cellCnt, B, all);
}
