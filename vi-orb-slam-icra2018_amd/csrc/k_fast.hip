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
// One 256-thread workgroup per (frame, run of <= 8 cells of one cell-row); five phases:
//   1. stage the run's pixels (+3 px halo) into LDS with 16-byte row-coalesced loads;
//   2. compass pre-test on EVERY domain pixel, 4 pixels per lane from aligned LDS dwords: a 9-arc
//      always contains ring pixel 0 or 8 and ring pixel 4 or 12, so a corner needs
//      (q0|q8) & (q4|q12) beyond the threshold with one polarity.  ~10 VALU ops per pixel (SDWA
//      byte compares, lane masks combined on the scalar unit).  Survivors (typically 10-20 %) are
//      appended to an LDS work list;
//   3. full score on the work list, dense lanes: arc minima with min3 trees, one polarity unless
//      both are possible;
//   4. non-max suppression over the scored corners only (sparse), survivors marked in an LDS bitmap,
//      per-cell "has a corner >= iniThFAST" flag;
//   5. per cell, one wave walks the bitmap rows in raster order and writes the kept candidates into
//      the cell's fixed slot range: cells row-major, raster inside a cell = the reference's
//      candidate order.  No global atomics, no sorting, deterministic.
// HBM traffic: each level pixel inside [16, w-16) x [16, h-16) is read once per tile that needs it;
// the 6-row vertical halo (hCell ~ 30) is re-read by the tile below.  Roofline: HBM read,
// algorithmic bytes = sum_l (w_l-32)(h_l-32) per frame.
#include "orbhip_internal.h"

#include <cstdlib>

#define FAST_BM_WORDS 12   // bitmap words per domain row (domain width <= 384)
#define FAST_MAX_DH 66     // domain rows per tile (hCell <= 66)

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

__device__ __forceinline__ int min3i(int a, int b, int c) { return min(min(a, b), c); }
__device__ __forceinline__ int max3i(int a, int b, int c) { return max(max(a, b), c); }

// max over the 16 cyclic 9-arcs of the arc minimum of e[]
__device__ __forceinline__ int max_arc_min(const int e[16])
{
    int m3[16];
#pragma unroll
    for (int k = 0; k < 16; k++) m3[k] = min3i(e[k], e[(k + 1) & 15], e[(k + 2) & 15]);
    int a = -256;
#pragma unroll
    for (int k = 0; k < 16; k += 2) {
        const int x = min3i(m3[k], m3[(k + 3) & 15], m3[(k + 6) & 15]);
        const int y = min3i(m3[k + 1], m3[(k + 4) & 15], m3[(k + 7) & 15]);
        a = max3i(a, x, y);
    }
    return a;
}

// FAST score of the pixel at p (LDS), 0 if it is not a corner at threshold t (t >= 1).
// score = max over the 16 cyclic 9-arcs of min_{q in arc} (v - q), same for (q - v), minus 1.
__device__ __forceinline__ int fast_score_lds(const uint8_t *p, int pitch, int t)
{
    const int v = p[0];
    const int q0 = p[3 * pitch], q8 = p[-3 * pitch], q4 = p[3], q12 = p[-3];
    // polarity that can hold a 9-arc: dark (v - q > t) / bright (q - v > t)
    const bool pd = ((v - q0 > t) || (v - q8 > t)) && ((v - q4 > t) || (v - q12 > t));
    const bool pb = ((q0 - v > t) || (q8 - v > t)) && ((q4 - v > t) || (q12 - v > t));
    if (!pd && !pb) return 0;
    const int sgn = (pb && !pd) ? -1 : 1;   // evaluate e = sgn * (v - q)
    const int sv = sgn * v;
    int e[16];
    e[0] = sv - sgn * q0;
    e[1] = sv - sgn * (int)p[3 * pitch + 1];
    e[2] = sv - sgn * (int)p[2 * pitch + 2];
    e[3] = sv - sgn * (int)p[pitch + 3];
    e[4] = sv - sgn * q4;
    e[5] = sv - sgn * (int)p[-pitch + 3];
    e[6] = sv - sgn * (int)p[-2 * pitch + 2];
    e[7] = sv - sgn * (int)p[-3 * pitch + 1];
    e[8] = sv - sgn * q8;
    e[9] = sv - sgn * (int)p[-3 * pitch - 1];
    e[10] = sv - sgn * (int)p[-2 * pitch - 2];
    e[11] = sv - sgn * (int)p[-pitch - 3];
    e[12] = sv - sgn * q12;
    e[13] = sv - sgn * (int)p[pitch - 3];
    e[14] = sv - sgn * (int)p[2 * pitch - 2];
    e[15] = sv - sgn * (int)p[3 * pitch - 1];
    int a = max_arc_min(e);
    if (pd && pb) {   // both polarities possible (rare): evaluate the other one as well
        int f[16];
#pragma unroll
        for (int k = 0; k < 16; k++) f[k] = -e[k];
        a = max(a, max_arc_min(f));
    }
    const int s = a - 1;
    return s >= t ? s : 0;
}

__global__ __launch_bounds__(256) void k_fast(const OrbLevels G, const uint8_t *__restrict__ lvl0,
                                              int stride0, unsigned long long frame0,
                                              const uint8_t *__restrict__ pyr,
                                              unsigned long long pyrFrame,
                                              const FastTile *__restrict__ tiles,
                                              uint32_t *__restrict__ cand,
                                              uint16_t *__restrict__ cellCnt, int pixBytes, int scoreBytes,
                                              int phases)
{
    extern __shared__ __align__(16) uint8_t smem[];
    __shared__ uint32_t s_bitmap[FAST_MAX_DH][FAST_BM_WORDS];
    __shared__ int s_cellAny[FAST_TILE_CELLS];
    __shared__ int s_listCount;

    const FastTile T = tiles[blockIdx.x];
    const int frame = blockIdx.y;
    const OrbLevel &L = G.lv[T.level];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;

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
    const int SP = (TW + 3) & ~3;
    for (int i = tid; i < RH * nchunk; i += 256) {
        const int r = i / nchunk, c = i - r * nchunk;
        const uint4 v = *reinterpret_cast<const uint4 *>(img + (size_t)(iniY + r) * stride + XA + (c << 4));
        *reinterpret_cast<uint4 *>(s_pix + r * pitch + (c << 4)) = v;
    }
    for (int i = tid; i < (DH * SP) >> 2; i += 256) reinterpret_cast<uint32_t *>(s_score)[i] = 0;
    for (int i = tid; i < FAST_MAX_DH * FAST_BM_WORDS; i += 256) (&s_bitmap[0][0])[i] = 0;
    if (tid < FAST_TILE_CELLS) s_cellAny[tid] = 0;
    if (tid == 0) s_listCount = 0;
    __syncthreads();
    if (phases < 2) return;   // timing ablation only (ORBHIP_FAST_PHASES), results are then invalid

    // ---- 2. compass pre-test, 4 pixels per lane; survivors -> work list ----
    const int t = G.minTh;
    const int j0 = X0 + 3 - XA;            // LDS column of domain column 0
    const int jd0 = j0 & ~3;               // first aligned dword column touching the domain
    const int GPR = ((j0 + TW + 3) >> 2) - (j0 >> 2);   // dword groups per row
    for (int rbase = 0; rbase < DH; rbase += 32) {
        for (int gi0 = 0; gi0 < GPR; gi0 += 64) {      // normally a single trip (GPR <= 64)
            // every lane runs the body (wave-wide shuffles below); idle lanes contribute nothing
            const int gi = gi0 + lane;
            const bool live = gi < GPR;
            const int jd = jd0 + ((live ? gi : 0) << 2);
            unsigned mask = 0;
#pragma unroll
            for (int i = 0; i < 8; i++) {
                const int r = rbase + wave + 4 * i;
                if (live && r < DH) {
                    const uint8_t *row = s_pix + (r + 3) * pitch + jd;
                    const uint32_t Cw = *reinterpret_cast<const uint32_t *>(row);
                    const uint32_t Lw = *reinterpret_cast<const uint32_t *>(row - 4);
                    const uint32_t Rw = *reinterpret_cast<const uint32_t *>(row + 4);
                    const uint32_t Tw = *reinterpret_cast<const uint32_t *>(row - 3 * pitch);
                    const uint32_t Bw = *reinterpret_cast<const uint32_t *>(row + 3 * pitch);
                    // pixel k of the group: left = column-3, right = column+3
                    const uint32_t lft = (Lw >> 8) | (Cw << 24);      // bytes: L1 L2 L3 C0
                    const uint32_t rgt = (Cw >> 24) | (Rw << 8);      // bytes: C3 R0 R1 R2
#pragma unroll
                    for (int k = 0; k < 4; k++) {
                        const int v = (Cw >> (8 * k)) & 0xFF;
                        const int hi = v + t, lo = v - t;
                        const int qt = (Tw >> (8 * k)) & 0xFF, qb = (Bw >> (8 * k)) & 0xFF;
                        const int ql = (lft >> (8 * k)) & 0xFF, qr = (rgt >> (8 * k)) & 0xFF;
                        const bool br = ((qt > hi) | (qb > hi)) & ((ql > hi) | (qr > hi));
                        const bool dk = ((qt < lo) | (qb < lo)) & ((ql < lo) | (qr < lo));
                        const int c = jd + k - j0;
                        const bool in = (c >= 0) & (c < TW);
                        if ((br | dk) & in) mask |= 1u << (4 * i + k);
                    }
                }
            }
            // append this lane's survivors to the work list (order is irrelevant)
            const int n = __popc(mask);
            int incl = n;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) {
                const int u = __shfl_up(incl, o);
                if (lane >= o) incl += u;
            }
            const int total = __shfl(incl, 63);
            int base = 0;
            if (lane == 63 && total > 0) base = atomicAdd(&s_listCount, total);
            base = __shfl(base, 63);
            int pos = base + incl - n;
            while (mask) {
                const int b = __ffs(mask) - 1;
                mask &= mask - 1;
                const int r = rbase + wave + 4 * (b >> 2);
                const int j = jd + (b & 3);
                s_list[pos++] = (uint16_t)((r << 9) | j);
            }
        }
    }
    __syncthreads();
    if (phases < 3) return;

    // ---- 3. full score on the work list ----
    const int nlist = s_listCount;
    for (int e = tid; e < nlist; e += 256) {
        const int ent = s_list[e];
        const int r = ent >> 9, j = ent & 511;
        const int s = fast_score_lds(s_pix + (r + 3) * pitch + j, pitch, t);
        if (s > 0) s_score[r * SP + (j - j0)] = (uint8_t)s;
    }
    __syncthreads();
    if (phases < 4) return;

    // ---- 4. NMS over scored corners (cell-local neighbourhood), survivors -> bitmap ----
    const unsigned cellMagic = 65536u / (unsigned)L.wCell + 1u;   // c / wCell for c < 65536 / wCell
    for (int e = tid; e < nlist; e += 256) {
        const int ent = s_list[e];
        const int r = ent >> 9, c = (ent & 511) - j0;
        const uint8_t *sp = s_score + r * SP + c;
        const int s = sp[0];
        if (s == 0) continue;
        const int cj = (int)(((unsigned)c * cellMagic) >> 16);
        const int cx0 = cj * L.wCell;
        int cx1 = cx0 + L.wCell;
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
        if (s > m) {
            atomicOr(&s_bitmap[r][c >> 5], 1u << (c & 31));
            if (s >= G.iniTh) s_cellAny[cj] = 1;   // benign race: every writer stores 1
        }
    }
    __syncthreads();
    if (phases < 5) return;

    // ---- 5. per cell: threshold choice, raster-ordered extraction from the bitmap ----
    const size_t candFrame = (size_t)frame * G.totalCands + L.candBase;
    for (int cj = wave; cj < T.ncells; cj += 4) {
        const int cx0 = cj * L.wCell;
        int cx1 = cx0 + L.wCell;
        if (cx1 > TW) cx1 = TW;
        // a cell whose iniX >= maxBorderX-6 is skipped by the reference (:805); its domain is empty
        if (cx1 <= cx0) {
            if (lane == 0) cnt[cj] = 0;
            continue;
        }
        const int thr = s_cellAny[cj] ? G.iniTh : G.minTh;
        uint32_t *slot = cand + candFrame + (size_t)(T.row * L.nCols + T.c0 + cj) * L.cellCap;
        int count = 0;
        for (int rb = 0; rb < DH; rb += 64) {
            const int r = rb + lane;
            // this lane's row: kept columns as a 64-bit mask relative to cx0 (cell width <= 64)
            unsigned long long keep = 0;
            if (r < DH) {
                const int w0 = cx0 >> 5;
                unsigned long long bits = (unsigned long long)s_bitmap[r][w0];
                bits |= (unsigned long long)s_bitmap[r][w0 + 1] << 32;
                bits >>= (cx0 & 31);
                if ((cx0 & 31) && w0 + 2 < FAST_BM_WORDS)
                    bits |= (unsigned long long)s_bitmap[r][w0 + 2] << (64 - (cx0 & 31));
                const int cw = cx1 - cx0;
                if (cw < 64) bits &= (1ull << cw) - 1ull;
                unsigned long long it = bits;
                while (it) {
                    const int b = __ffsll((long long)it) - 1;
                    it &= it - 1;
                    if (s_score[r * SP + cx0 + b] >= thr) keep |= 1ull << b;
                }
            }
            const int n = __popcll(keep);
            int incl = n;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) {
                const int u = __shfl_up(incl, o);
                if (lane >= o) incl += u;
            }
            int pos = count + incl - n;
            while (keep) {
                const int b = __ffsll((long long)keep) - 1;
                keep &= keep - 1;
                const int s = s_score[r * SP + cx0 + b];
                const int px = X0 + 3 + cx0 + b - ORB_MIN_BORDER;   // relative to (16,16), :824-825
                const int py = iniY + 3 + r - ORB_MIN_BORDER;
                slot[pos++] = (uint32_t)px | ((uint32_t)py << 12) | ((uint32_t)s << 24);
            }
            count += __shfl(incl, 63);
        }
        if (lane == 0) cnt[cj] = (uint16_t)count;
    }
}

void launch_fast(hipStream_t s, const OrbLevels &G, const uint8_t *lvl0, int stride0, size_t frame0,
                 const uint8_t *pyr, size_t pyrFrame, const FastTile *tiles, int ntiles,
                 uint32_t *cand, uint16_t *cellCnt, int B)
{
    // LDS: pixel tile + score tile + work list of the largest run over all levels
    int pixBytes = 0, scoreBytes = 0, listBytes = 0;
    for (int l = 0; l < G.nlevels; l++) {
        const OrbLevel &L = G.lv[l];
        int tileCells = FAST_TILE_CELLS;
        while (tileCells > 1 && tileCells * L.wCell + 6 + 16 > FAST_MAX_TILE_W) tileCells--;
        const int regw = tileCells * L.wCell + 6;
        const int pitch = ((regw + 15 + 15) >> 4) << 4;
        const int rh = L.hCell + 6;
        const int sp = (tileCells * L.wCell + 3) & ~3;
        pixBytes = std::max(pixBytes, pitch * rh);
        scoreBytes = std::max(scoreBytes, sp * L.hCell);
        listBytes = std::max(listBytes, sp * L.hCell * 2);
    }
    pixBytes = (pixBytes + 15) & ~15;
    scoreBytes = (scoreBytes + 15) & ~15;
    static const int phases = getenv("ORBHIP_FAST_PHASES") ? atoi(getenv("ORBHIP_FAST_PHASES")) : 5;
    dim3 grid(ntiles, B, 1), block(256, 1, 1);
    hipLaunchKernelGGL(k_fast, grid, block, (size_t)(pixBytes + scoreBytes + listBytes), s, G, lvl0, stride0,
                       (unsigned long long)frame0, pyr, (unsigned long long)pyrFrame, tiles, cand, cellCnt,
                       pixBytes, scoreBytes, phases);
}
