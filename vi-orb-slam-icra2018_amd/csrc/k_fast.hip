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
// Work decomposition: one 256-thread workgroup per (frame, run of <= 8 cells of one cell-row).
//   1. the pixel region of the run (+3 px halo) is staged into LDS with 16-byte row-coalesced
//      loads (rows of a level are 64-byte aligned);
//   2. every thread scores pixels of the run's domain from LDS into an LDS score tile;
//   3. each wave takes whole cells: NMS + ballot compaction in raster order, which IS the
//      reference's candidate order (cells row-major, raster inside a cell), into the cell's fixed
//      slot range of the candidate array.  No atomics, no sorting, deterministic.
// HBM traffic: each level pixel inside [16, w-16) x [16, h-16) is read once per tile that needs
// it: the 6-row vertical halo (hCell ~ 30) is re-read by the tile below (L2 hit when co-resident).
// Bound: HBM read (algorithmic bytes = sum_l (w_l-32)(h_l-32) per frame) -- see DESIGN.md.
#include "orbhip_internal.h"

#define FAST_MASK_SLOTS 64  // 64-pixel chunks per cell kept per wave (cells up to 4096 px)

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

// FAST score of the pixel at p (LDS), 0 if it is not a corner at threshold t (t >= 1).
// score = max over the 16 cyclic 9-arcs of min_{q in arc} (v - q), same for (q - v), minus 1.
__device__ __forceinline__ int fast_score_lds(const uint8_t *p, int pitch, int t)
{
    const int v = p[0];
    // ring 0 = (0,+3) and ring 8 = (0,-3): every 9-arc contains one of each opposite pair
    const int q0 = p[3 * pitch], q8 = p[-3 * pitch];
    if (abs(v - q0) <= t && abs(v - q8) <= t) return 0;
    const int q4 = p[3], q12 = p[-3];
    if (abs(v - q4) <= t && abs(v - q12) <= t) return 0;
    int d[16];
    d[0] = v - q0;
    d[1] = v - p[3 * pitch + 1];
    d[2] = v - p[2 * pitch + 2];
    d[3] = v - p[pitch + 3];
    d[4] = v - q4;
    d[5] = v - p[-pitch + 3];
    d[6] = v - p[-2 * pitch + 2];
    d[7] = v - p[-3 * pitch + 1];
    d[8] = v - q8;
    d[9] = v - p[-3 * pitch - 1];
    d[10] = v - p[-2 * pitch - 2];
    d[11] = v - p[-pitch - 3];
    d[12] = v - q12;
    d[13] = v - p[pitch - 3];
    d[14] = v - p[2 * pitch - 2];
    d[15] = v - p[3 * pitch - 1];
    int lo2[16], hi2[16], lo4[16], hi4[16];
#pragma unroll
    for (int k = 0; k < 16; k++) {
        lo2[k] = min(d[k], d[(k + 1) & 15]);
        hi2[k] = max(d[k], d[(k + 1) & 15]);
    }
#pragma unroll
    for (int k = 0; k < 16; k++) {
        lo4[k] = min(lo2[k], lo2[(k + 2) & 15]);
        hi4[k] = max(hi2[k], hi2[(k + 2) & 15]);
    }
    int a = -256, b = 256;
#pragma unroll
    for (int k = 0; k < 16; k++) {
        const int lo9 = min(min(lo4[k], lo4[(k + 4) & 15]), d[(k + 8) & 15]);
        const int hi9 = max(max(hi4[k], hi4[(k + 4) & 15]), d[(k + 8) & 15]);
        a = max(a, lo9);
        b = min(b, hi9);
    }
    const int s = max(a, -b) - 1;
    return s >= t ? s : 0;
}

__global__ __launch_bounds__(256) void k_fast(const OrbLevels G, const uint8_t *__restrict__ lvl0,
                                              int stride0, unsigned long long frame0,
                                              const uint8_t *__restrict__ pyr,
                                              unsigned long long pyrFrame,
                                              const FastTile *__restrict__ tiles,
                                              uint32_t *__restrict__ cand,
                                              uint16_t *__restrict__ cellCnt, int pixBytes)
{
    extern __shared__ __align__(16) uint8_t smem[];
    __shared__ unsigned long long s_masks[4][FAST_MASK_SLOTS];

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
    const int SP = (TW + 3) & ~3;
    for (int i = tid; i < RH * nchunk; i += 256) {
        const int r = i / nchunk, c = i - r * nchunk;
        const uint4 v = *reinterpret_cast<const uint4 *>(img + (size_t)(iniY + r) * stride + XA + (c << 4));
        *reinterpret_cast<uint4 *>(s_pix + r * pitch + (c << 4)) = v;
    }
    __syncthreads();

    // ---- 2. scores of the run's domain ----
    const int xoff = X0 + 3 - XA;  // LDS column of domain column 0
    for (int r = wave; r < DH; r += 4) {
        const uint8_t *prow = s_pix + (r + 3) * pitch + xoff;
        uint8_t *srow = s_score + r * SP;
        for (int c = lane; c < TW; c += 64) srow[c] = (uint8_t)fast_score_lds(prow + c, pitch, G.minTh);
    }
    __syncthreads();

    // ---- 3. per cell: NMS, threshold choice, ordered compaction ----
    const size_t candFrame = (size_t)frame * G.totalCands + L.candBase;
    for (int cj = wave; cj < T.ncells; cj += 4) {
        const int cx0 = cj * L.wCell;
        int cx1 = cx0 + L.wCell;
        if (cx1 > TW) cx1 = TW;
        const int cdw = cx1 - cx0;
        // a cell whose iniX >= maxBorderX-6 is skipped by the reference (:805); its domain is empty
        if (cdw <= 0) {
            if (lane == 0) cnt[cj] = 0;
            continue;
        }
        const int npx = cdw * DH;
        const int nch = (npx + 63) >> 6;
        const unsigned magic = (1u << 20) / (unsigned)cdw + 1u;  // idx / cdw for idx < 2^20 / cdw
        unsigned long long anyIni = 0;
        for (int ch = 0; ch < nch; ch++) {
            const int idx = (ch << 6) + lane;
            bool surv = false, strong = false;
            if (idx < npx) {
                const int r = (int)(((unsigned)idx * magic) >> 20);
                const int c = idx - r * cdw;
                const uint8_t *sp = s_score + r * SP + cx0 + c;
                const int s = sp[0];
                if (s > 0) {
                    const bool up = r > 0, dn = r < DH - 1, lf = c > 0, rt = c < cdw - 1;
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
                    strong = surv && s >= G.iniTh;
                }
            }
            const unsigned long long mk = __ballot(surv);
            anyIni |= __ballot(strong);
            if (lane == 0 && ch < FAST_MASK_SLOTS) s_masks[wave][ch] = mk;
        }
        // lane 0 wrote s_masks, every lane of the SAME wave reads it: order the LDS accesses
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        const int thr = anyIni ? G.iniTh : G.minTh;
        uint32_t *slot = cand + candFrame + (size_t)(T.row * L.nCols + T.c0 + cj) * L.cellCap;
        int count = 0;
        for (int ch = 0; ch < nch; ch++) {
            const unsigned long long mk = s_masks[wave][ch < FAST_MASK_SLOTS ? ch : 0];
            const int idx = (ch << 6) + lane;
            bool keep = (mk >> lane) & 1ull;
            int r = 0, c = 0, s = 0;
            if (keep) {
                r = (int)(((unsigned)idx * magic) >> 20);
                c = idx - r * cdw;
                s = s_score[r * SP + cx0 + c];
                keep = s >= thr;
            }
            const unsigned long long kk = __ballot(keep);
            if (keep) {
                const int pos = count + __popcll(kk & ((1ull << lane) - 1ull));
                const int px = X0 + 3 + cx0 + c - ORB_MIN_BORDER;   // relative to (16,16), :824-825
                const int py = iniY + 3 + r - ORB_MIN_BORDER;
                slot[pos] = (uint32_t)px | ((uint32_t)py << 12) | ((uint32_t)s << 24);
            }
            count += __popcll(kk);
        }
        if (lane == 0) cnt[cj] = (uint16_t)count;
    }
}

void launch_fast(hipStream_t s, const OrbLevels &G, const uint8_t *lvl0, int stride0, size_t frame0,
                 const uint8_t *pyr, size_t pyrFrame, const FastTile *tiles, int ntiles,
                 uint32_t *cand, uint16_t *cellCnt, int B)
{
    // LDS: pixel tile + score tile of the largest run over all levels
    int pixBytes = 0, scoreBytes = 0;
    for (int l = 0; l < G.nlevels; l++) {
        const OrbLevel &L = G.lv[l];
        int tileCells = FAST_TILE_CELLS;
        while (tileCells > 1 && tileCells * L.wCell + 6 + 16 > FAST_MAX_TILE_W) tileCells--;
        const int regw = tileCells * L.wCell + 6;
        const int pitch = ((regw + 15 + 15) >> 4) << 4;
        const int rh = L.hCell + 6;
        pixBytes = std::max(pixBytes, pitch * rh);
        scoreBytes = std::max(scoreBytes, ((tileCells * L.wCell + 3) & ~3) * L.hCell);
    }
    pixBytes = (pixBytes + 15) & ~15;
    dim3 grid(ntiles, B, 1), block(256, 1, 1);
    hipLaunchKernelGGL(k_fast, grid, block, (size_t)(pixBytes + scoreBytes), s, G, lvl0, stride0,
                       (unsigned long long)frame0, pyr, (unsigned long long)pyrFrame, tiles, cand, cellCnt,
                       pixBytes);
}
