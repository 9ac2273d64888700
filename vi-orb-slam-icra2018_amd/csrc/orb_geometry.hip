// orb_geometry.hip -- host-side tables: constructor arithmetic of ORBextractor
// (ref: src/ORBextractor.cc:412-472), per-level sizes (:1132-1133), the FAST cell grid
// (:771-808), quadtree roots (:545-547) and the fixed-point tables of cv::resize INTER_LINEAR
// (OpenCV 2.4 imgwarp.cpp, restated in DESIGN.md "pyramid").
#include "orbhip_internal.h"

#include <algorithm>

#include <cmath>
#include <cstdlib>
#include <cstring>

static inline int cv_round_d(double v) { return (int)lrint(v); }  // SSE2 cvtsd2si: half to even
static inline int cv_floor_d(double v) { int i = (int)v; return i - (i > v); }
static inline int cv_ceil_d(double v) { int i = (int)v; return i + (i < v); }
static inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

int fast_tile_cells()
{
    static const int n = [] {
        // 4: the fixed-layout kernel finishes a run with one wave per cell (second pass, candidate order) -- four cells keep
        // its four waves equally busy to the end (5 cells: 1.08 ms per 1024 frames, 4: 1.04, 3: 1.18; r02's kernel preferred 5)
        int v = ORB_TUNE("FAST_TILE_CELLS", 4);
        return v < 1 ? 1 : (v > FAST_TILE_CELLS ? FAST_TILE_CELLS : v);
    }();
    return n;
}

int orb_init_tables(orbhip_ctx *c, int nfeatures, float scaleFactor, int nlevels, int iniTh, int minTh)
{
    if (nfeatures < 1 || nlevels < 1 || nlevels > ORBHIP_MAX_LEVELS || !(scaleFactor > 1.0f)) return ORBHIP_E_ARG;
    // FAST thresholds: the reference passes whatever the settings file holds to cv::FAST, which clamps to [0, 255]
    // (OpenCV 2.4 fast.cpp: threshold = min(max(threshold, 0), 255)).  Threshold 0 equals threshold 1 here: a corner of
    // score 0 never beats its neighbours' 0 in the strict non-maximum suppression.  iniThFAST < minThFAST is legal too (the
    // second run then finds nothing the first did not) -- k_fast's passes handle every combination.
    auto eff = [](int t) { return std::max(std::min(std::max(t, 0), 255), 1); };
    c->nfeatures = nfeatures;
    c->nlevels = nlevels;
    c->iniTh = eff(iniTh);
    c->minTh = eff(minTh);
    c->scaleFactor = (double)scaleFactor;  // the member is a double (include/ORBextractor.h:116)
    c->mvScaleFactor[0] = 1.0f;
    c->mvLevelSigma2[0] = 1.0f;
    for (int i = 1; i < nlevels; i++) {
        c->mvScaleFactor[i] = (float)((double)c->mvScaleFactor[i - 1] * c->scaleFactor);
        c->mvLevelSigma2[i] = c->mvScaleFactor[i] * c->mvScaleFactor[i];
    }
    for (int i = 0; i < nlevels; i++) {
        c->mvInvScaleFactor[i] = 1.0f / c->mvScaleFactor[i];
        c->mvInvLevelSigma2[i] = 1.0f / c->mvLevelSigma2[i];
    }
    const float factor = (float)(1.0 / c->scaleFactor);
    float want = (float)nfeatures * (1 - factor) / (1 - (float)pow((double)factor, (double)nlevels));
    int sum = 0;
    for (int l = 0; l < nlevels - 1; l++) {
        c->mnFeaturesPerLevel[l] = cv_round_d((double)want);
        sum += c->mnFeaturesPerLevel[l];
        want *= factor;
    }
    c->mnFeaturesPerLevel[nlevels - 1] = std::max(nfeatures - sum, 0);

    const int vmax = cv_floor_d((double)(ORB_HALF_PATCH * sqrtf(2.f) / 2 + 1));
    const int vmin = cv_ceil_d((double)(ORB_HALF_PATCH * sqrtf(2.f) / 2));
    const double hp2 = ORB_HALF_PATCH * ORB_HALF_PATCH;
    memset(c->umax, 0, sizeof(c->umax));
    for (int v = 0; v <= vmax; ++v) c->umax[v] = cv_round_d(sqrt(hp2 - v * v));
    for (int v = ORB_HALF_PATCH, v0 = 0; v >= vmin; --v) {
        while (c->umax[v0] == c->umax[v0 + 1]) ++v0;
        c->umax[v] = v0;
        ++v0;
    }
    return ORBHIP_OK;
}

void orb_level_size(const orbhip_ctx *c, int w, int h, int level, int *lw, int *lh)
{
    const float s = c->mvInvScaleFactor[level];
    *lw = cv_round_d((double)((float)w * s));
    *lh = cv_round_d((double)((float)h * s));
}

// cv::resize(src, dst, dsize, 0, 0, INTER_LINEAR), 8UC1: per destination column the two source
// columns and their 11-bit weights, per destination row the two (clamped) source rows and weights.
//   xtab[2*dx]   = sx0 | sx1 << 16          xtab[2*dx+1] = a0 | a1 << 16   (a as uint16 of int16)
//   ytab[4*dy]   = sy0, ytab[4*dy+1] = sy1, ytab[4*dy+2] = b0, ytab[4*dy+3] = b1
void orb_build_resize_tables(int sw, int sh, int dw, int dh, std::vector<int32_t> &xtab,
                             std::vector<int32_t> &ytab)
{
    const double inv_x = (double)dw / sw, inv_y = (double)dh / sh;
    const double scale_x = 1. / inv_x, scale_y = 1. / inv_y;
    auto wgt = [](float v) {
        int i = cv_round_d((double)v);
        return i < -32768 ? -32768 : (i > 32767 ? 32767 : i);
    };
    xtab.resize((size_t)dw * 2);
    ytab.resize((size_t)dh * 4);
    for (int dx = 0; dx < dw; dx++) {
        float fx = (float)((dx + 0.5) * scale_x - 0.5);
        int sx = cv_floor_d((double)fx);
        fx -= sx;
        if (sx < 0) { fx = 0; sx = 0; }
        bool edge = false;
        if (sx + 1 >= sw) {
            edge = true;  // columns past xmax read a single source pixel with weight 2048
            if (sx >= sw - 1) { fx = 0; sx = sw - 1; }
        }
        int a0 = wgt((1.f - fx) * 2048), a1 = wgt(fx * 2048);
        int sx1 = sx + 1;
        if (edge) { a0 = 2048; a1 = 0; sx1 = sx; }
        xtab[2 * dx] = (sx & 0xFFFF) | (sx1 << 16);
        xtab[2 * dx + 1] = (a0 & 0xFFFF) | (a1 << 16);
    }
    for (int dy = 0; dy < dh; dy++) {
        float fy = (float)((dy + 0.5) * scale_y - 0.5);
        int sy = cv_floor_d((double)fy);
        fy -= sy;
        int b0 = wgt((1.f - fy) * 2048), b1 = wgt(fy * 2048);
        int sy0 = sy < 0 ? 0 : (sy < sh ? sy : sh - 1);
        int sy1 = sy + 1 < 0 ? 0 : (sy + 1 < sh ? sy + 1 : sh - 1);
        ytab[4 * dy] = sy0;
        ytab[4 * dy + 1] = sy1;
        ytab[4 * dy + 2] = b0;
        ytab[4 * dy + 3] = b1;
    }
}

// Column taps of four adjacent destination pixels as byte selectors into the 8 source bytes that start at the
// first pixel's left tap: 12 ints per group = {base column, 0, 0, 0}, v_perm selectors of the four pixels
// (bytes [left tap, 0, right tap, 0]), weight pairs a0 | a1 << 16.  Returns false when a group does not fit
// (a tap more than 7 columns from the base: scale factors above 2, or a negative weight).
bool orb_build_resize_groups(const std::vector<int32_t> &xtab, const std::vector<int32_t> &ytab, int dw,
                             std::vector<int32_t> &gtab)
{
    const int ng = (dw + 3) / 4;
    gtab.assign((size_t)ng * 12, 0);
    for (size_t dy = 0; dy < ytab.size() / 4; dy++)
        if (ytab[4 * dy + 2] < 0 || ytab[4 * dy + 2] > 2048 || ytab[4 * dy + 3] < 0 || ytab[4 * dy + 3] > 2048) return false;
    for (int g = 0; g < ng; g++) {
        const int base = xtab[2 * (4 * g)] & 0xFFFF;
        gtab[(size_t)g * 12] = base;
        for (int k = 0; k < 4; k++) {
            const int dx = std::min(4 * g + k, dw - 1);
            const int sx0 = xtab[2 * dx] & 0xFFFF, sx1 = (int)((uint32_t)xtab[2 * dx] >> 16);
            const int a0 = (int16_t)(xtab[2 * dx + 1] & 0xFFFF), a1 = xtab[2 * dx + 1] >> 16;
            const int o0 = sx0 - base, o1 = sx1 - base;
            if (o0 < 0 || o0 > 7 || o1 < 0 || o1 > 7 || a0 < 0 || a1 < 0 || a0 > 2048 || a1 > 2048) return false;
            gtab[(size_t)g * 12 + 4 + k] = o0 | (0x0c << 8) | (o1 << 16) | (0x0c << 24);
            gtab[(size_t)g * 12 + 8 + k] = a0 | (a1 << 16);
        }
    }
    return true;
}

// The derived fields of a FAST run (same arithmetic as the generic kernel's prologue, k_fast.hip).
void fast_tile_geometry(const OrbLevels &G, FastTile &t)
{
    const OrbLevel &L = G.lv[t.level];
    const int maxBX = L.w - ORB_MIN_BORDER, maxBY = L.h - ORB_MIN_BORDER;
    const int iniY = ORB_MIN_BORDER + t.row * L.hCell;
    const int X0 = ORB_MIN_BORDER + t.c0 * L.wCell;
    int maxY = iniY + L.hCell + 6;
    if (maxY > maxBY) maxY = maxBY;
    int X1 = ORB_MIN_BORDER + (t.c0 + t.ncells) * L.wCell + 6;
    if (X1 > maxBX) X1 = maxBX;
    const bool rowLive = iniY < maxBY - 3;
    int DH = rowLive ? maxY - iniY - 6 : 0, TW = X1 - X0 - 6;
    if (DH <= 0 || TW <= 0) DH = TW = 0;
    const int XA = X0 & ~15;
    t.nc = t.ncells;
    t.iniY = iniY;
    t.xa = XA;
    t.RH = (DH ? maxY - iniY : 0);
    t.nchunk = (DH ? (X1 - XA + 15) >> 4 : 0);
    t.DH = DH;
    t.TW = TW;
    const int j0 = X0 + 3 - XA;
    t.j0 = j0;
    const int GPR = DH ? ((j0 + TW + 3) >> 2) - (j0 >> 2) : 1;
    t.GPR = GPR;
    t.wCell = L.wCell;
    const int S = std::max(1, 256 / GPR);
    t.seg = (DH ? (DH + S - 1) / S : 1);
    t.py0 = (iniY + 3 - ORB_MIN_BORDER);
    t.px0 = (X0 + 3 - ORB_MIN_BORDER);
    t.cellMagic = (int)(65536u / (unsigned)L.wCell + 1u);
    t.grpMagic = (int)(65536u / (unsigned)GPR + 1u);
    t.dhMagic = (int)(65536u / (unsigned)std::max(DH, 1) + 1u);
    t.cntOff = L.cellBase + t.row * L.nCols + t.c0;
    t.candOff = L.candBase + (t.row * L.nCols + t.c0) * L.cellCap;
    t.cellCap = L.cellCap;
    t.stride = L.stride;
    t.lvlOff = t.level == 0 ? 0xFFFFFFFFu : (unsigned)L.imgOff;
    t.pad = 0;
}

int orb_build_geometry(orbhip_ctx *c, int w, int h, int stride0)
{
    OrbLevels &G = c->G;
    memset(&G, 0, sizeof(G));
    G.nlevels = c->nlevels;
    G.iniTh = c->iniTh;
    G.minTh = c->minTh;
    memcpy(G.umax, c->umax, sizeof(G.umax));
    c->fastTiles.clear();
    c->nFastTilesBatch = 0;
    std::vector<FastTile> single;   // one cell per workgroup: the list used for a frame or two (more, shorter workgroups)
    c->blurTiles.clear();
    size_t pyrOff = 0;
    int cellBase = 0, candBase = 0, kpBase = 0;
    for (int l = 0; l < c->nlevels; l++) {
        OrbLevel &L = G.lv[l];
        orb_level_size(c, w, h, l, &L.w, &L.h);
        if (L.w < 1 || L.h < 1 || L.w > 4096 + 32 || L.h > 4096 + 32) return ORBHIP_E_SIZE;
        if (l == 0) {
            L.stride = stride0;
            L.imgOff = 0;
        } else {
            L.stride = (int)align_up((size_t)L.w, 64);
            L.imgOff = pyrOff;
            pyrOff += align_up((size_t)L.stride * L.h, 256);
        }
        // cell grid, :771-789 (float arithmetic as written)
        const int maxBX = L.w - ORB_EDGE_THRESHOLD + 3, maxBY = L.h - ORB_EDGE_THRESHOLD + 3;
        const float width = (float)(maxBX - ORB_MIN_BORDER), height = (float)(maxBY - ORB_MIN_BORDER);
        L.nCols = (int)(width / (float)ORB_CELL_W);
        L.nRows = (int)(height / (float)ORB_CELL_W);
        if (L.nCols < 1 || L.nRows < 1) return ORBHIP_E_SIZE;  // the reference divides by zero here
        L.wCell = (int)ceilf(width / L.nCols);
        L.hCell = (int)ceilf(height / L.nRows);
        if (L.hCell + 6 > FAST_MAX_TILE_H || L.wCell * L.hCell > 4096) return ORBHIP_E_SIZE;
        L.cellCap = ((L.wCell + 1) / 2) * ((L.hCell + 1) / 2);
        L.cellBase = cellBase;
        L.candBase = candBase;
        L.ptBase = candBase;
        L.ptCap = L.nCols * L.nRows * L.cellCap;
        cellBase += L.nCols * L.nRows;
        candBase += L.ptCap;
        // quadtree, :545-547
        L.N = c->mnFeaturesPerLevel[l];
        L.regw = maxBX - ORB_MIN_BORDER;
        L.regh = maxBY - ORB_MIN_BORDER;
        L.nIni = (int)roundf((float)L.regw / (float)L.regh);
        if (L.nIni < 1) return ORBHIP_E_SIZE;  // the reference divides by zero here
        L.hX = (float)L.regw / (float)L.nIni;
        L.kpCap = std::max(L.N + 4, 4 * L.nIni + 4);
        L.kpBase = kpBase;
        kpBase += L.kpCap;
        L.scale = c->mvScaleFactor[l];
        L.kpSize = (float)(int)(ORB_PATCH_SIZE * c->mvScaleFactor[l]);
        // FAST tiles: runs of cells of one cell-row
        int tileCells = fast_tile_cells();
        while (tileCells > 1 && tileCells * L.wCell + 6 + 16 > FAST_MAX_TILE_W) tileCells--;
        if (L.wCell + 6 + 16 > FAST_MAX_TILE_W) return ORBHIP_E_SIZE;
        // the cells of a cell-row are dealt to ceil(nCols / tileCells) runs of nearly equal length
        const int nruns = (L.nCols + tileCells - 1) / tileCells, runBase = L.nCols / nruns, runExtra = L.nCols % nruns;
        for (int i = 0; i < L.nRows; i++)
            for (int r = 0, j = 0; r < nruns; r++) {
                FastTile t{};
                t.level = (short)l;
                t.row = (short)i;
                t.c0 = (short)j;
                t.ncells = (short)(runBase + (r < runExtra ? 1 : 0));
                j += t.ncells;
                c->fastTiles.push_back(t);
            }
        for (int i = 0; i < L.nRows; i++)
            for (int j = 0; j < L.nCols; j++) {
                FastTile t{};
                t.level = (short)l;
                t.row = (short)i;
                t.c0 = (short)j;
                t.ncells = 1;
                single.push_back(t);
            }
        c->blurLevelFirst[l] = (int)c->blurTiles.size();
        for (int ty = 0; ty < (L.h + BLUR_TILE_H - 1) / BLUR_TILE_H; ty++)
            for (int tx = 0; tx < (L.w + BLUR_TILE_W - 1) / BLUR_TILE_W; tx++) {
                BlurTile t;
                t.level = (short)l;
                t.tx = (short)tx;
                t.ty = (short)ty;
                t.pad = 0;
                c->blurTiles.push_back(t);
            }
    }
    c->blurLevelFirst[G.nlevels] = (int)c->blurTiles.size();
    // levels with cells taller than the fixed-layout kernel's first instance holds go to the end of the batch list (launch_fast)
    std::stable_partition(c->fastTiles.begin(), c->fastTiles.end(),
                          [&](const FastTile &t) { return G.lv[t.level].hCell + 6 <= FAST_FIX_ROWS; });
    c->nFastTilesTall = 0;
    for (const FastTile &t : c->fastTiles)
        if (G.lv[t.level].hCell + 6 > FAST_FIX_ROWS) c->nFastTilesTall++;
    c->nFastTilesBatch = (int)c->fastTiles.size();
    c->fastTiles.insert(c->fastTiles.end(), single.begin(), single.end());   // [batch list | single-frame list]
    for (FastTile &t : c->fastTiles) fast_tile_geometry(G, t);
    G.totalCells = cellBase;
    G.totalCands = candBase;
    G.totalPts = candBase;
    G.totalKps = kpBase;
    int cap = 0;
    for (int l = 0; l < c->nlevels; l++) cap += G.lv[l].kpCap;
    G.outCap = cap;
    c->pyrFrameBytes = align_up(pyrOff, 256);
    c->lvl0FrameBytes = align_up((size_t)align_up((size_t)w, 64) * h, 256);
    G.bstride0 = (int)align_up((size_t)w, 64);
    G.boff1 = c->lvl0FrameBytes;
    c->cur_w = w;
    c->cur_h = h;
    return ORBHIP_OK;
}
