// orbhip_internal.h -- context, per-level geometry and kernel launch prototypes of liborbhip.so.
// Host-side structures only; the kernels live in k_*.hip.
#ifndef ORBHIP_INTERNAL_H
#define ORBHIP_INTERNAL_H

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <memory>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/orbhip.h"

// ---- environment switches ----
// The shipped liborbhip.so reads FOUR environment variables, each choosing between two equivalent paths that are both under
// the parity tests (ORB_SWITCH): ORBHIP_KNN2_MFMA, ORBHIP_FAST_FIX, ORBHIP_NO_GRAPH, ORBHIP_NO_CHAIN.  Everything else -- tuning
// knobs, the forced-overflow capacities with which the tests reach the fallback paths, occupancy dummies and the timing-ablation
// stops that make a kernel return early with INVALID results -- is ORB_TUNE: its default, a compile-time constant, in the shipped
// library, and an environment variable only in liborbhip_ablation.so (-DORBHIP_ABLATION; built for tests/ and tools/, never
// loaded by the drop-in classes).  A stray variable in a SLAM process can therefore not change what the library computes.
#include <stdlib.h>
static inline int orb_env_int(const char *name, int dflt)
{
    const char *e = getenv(name);
    return e ? atoi(e) : dflt;
}
#define ORB_SWITCH(name, dflt) orb_env_int("ORBHIP_" name, (dflt))
#ifdef ORBHIP_ABLATION
#define ORB_TUNE(name, dflt) orb_env_int("ORBHIP_" name, (dflt))
#define ORB_ABL_PARAM , int phases
#define ORB_ABL_ARG(p) , (p)
#define ORB_ABL_STOP(cond) do { if (cond) return; } while (0)
#define ORB_ABL_IF(cond) if (cond)
#else
#define ORB_TUNE(name, dflt) (dflt)
#define ORB_ABL_PARAM
#define ORB_ABL_ARG(p)
#define ORB_ABL_STOP(cond) do { } while (0)
#define ORB_ABL_IF(cond) if (false)
#endif

// Which kernel variants have been launched in this process (orbhip_debug_path_mask, include/orbhip.h): the tests that claim to reach
// a fallback path check the bit instead of trusting the geometry they picked.
#include <atomic>
enum OrbPath : unsigned {
    ORB_PATH_FAST_FIX = 1u << 0, ORB_PATH_FAST_GENERIC = 1u << 1, ORB_PATH_RESIZE_FIT = 1u << 2, ORB_PATH_RESIZE_TILES = 1u << 3,
    ORB_PATH_PYRAMID_CHAIN = 1u << 4, ORB_PATH_QT_LDS = 1u << 5, ORB_PATH_QT_LDSPTS = 1u << 6, ORB_PATH_QT_GLOBAL = 1u << 7,
    ORB_PATH_BOW_LANE = 1u << 8, ORB_PATH_BOW_SEQ_LDS = 1u << 9, ORB_PATH_BOW_SEQ_GLOBAL = 1u << 10, ORB_PATH_FAST_TALL = 1u << 11, ORB_PATH_DESCRIBE = 1u << 12, ORB_PATH_DESCRIBE_BLUR = 1u << 13, ORB_PATH_BLUR = 1u << 14
};
extern std::atomic<unsigned> g_orbPathMask;
static inline void orb_path(unsigned bit) { g_orbPathMask.fetch_or(bit, std::memory_order_relaxed); }

#define ORB_PATCH_SIZE 31      // ref: src/ORBextractor.cc:74
#define ORB_HALF_PATCH 15      // :75
#define ORB_EDGE_THRESHOLD 19  // :76
#define ORB_MIN_BORDER 16      // EDGE_THRESHOLD-3, :775
#define ORB_CELL_W 30          // :771

#define FAST_TILE_CELLS 8      // max cells of one cell-row handled by one FAST workgroup
int fast_tile_cells();        // cells per workgroup actually used (<= FAST_TILE_CELLS; env ORBHIP_FAST_TILE_CELLS)
#define FAST_MAX_TILE_W 320    // LDS tile width bound (pixels incl. halo, before 16-B rounding)
#define FAST_MAX_TILE_H 72     // LDS tile height bound (hCell + 6)
#define FAST_FIX_ROWS 40       // staged rows of the fixed-layout FAST kernel's first instance (k_fast.hip); levels with taller cells come last in the batch run list

// Geometry of one pyramid level, shared by host and device code (passed by value in kernel args).
struct OrbLevel {
    int w, h;           // level size (:1132-1133)
    int stride;         // bytes between rows of the device level image
    int nCols, nRows;   // cell grid (:783-786)
    int wCell, hCell;   // :787-788
    int cellCap;        // candidate slots per cell = ceil(wCell/2)*ceil(hCell/2)
    int cellBase;       // index of this level's first cell in the per-frame cell arrays
    int candBase;       // index of this level's first slot in the per-frame candidate array
    int ptBase;         // index of this level's first entry in the per-frame compact point arrays
    int ptCap;          // capacity of the compact point arrays for this level (= ncells*cellCap)
    int N;              // mnFeaturesPerLevel[level] (:437-448)
    int nIni;           // quadtree roots (:545)
    float hX;           // :547
    int regw, regh;     // maxBorder - minBorder
    int kpCap;          // node-list capacity = max(N + 4, 4*nIni + 4)
    int kpBase;         // index of this level's first entry in the per-frame level-keypoint array
    float scale;        // mvScaleFactor[level]
    float kpSize;       // (float)(int)(PATCH_SIZE*mvScaleFactor[level]) (:836)
    unsigned long long imgOff;   // byte offset of this level inside one frame's pyramid block
};

struct OrbLevels {
    int nlevels;
    int iniTh, minTh;
    int totalCells;     // cells per frame (all levels)
    int totalCands;     // candidate slots per frame
    int totalPts;       // compact point capacity per frame
    int totalKps;       // sum of kpCap
    int outCap;         // keypoints per frame in the output arrays
    int umax[16];       // :456-471
    int bstride0;       // row stride of level 0 inside the blurred-pyramid block
    unsigned long long boff1;    // offset of levels >= 1 inside one frame's blurred block
    OrbLevel lv[ORBHIP_MAX_LEVELS];
};

// One FAST workgroup's work: `ncells` consecutive cells starting at column c0 of cell-row `row`.  The fields after the first
// four follow from them and the level (orb_build_geometry fills them, fast_tile_geometry): the fixed-layout kernel reads its
// whole geometry from here with scalar loads instead of deriving it in every workgroup.
struct FastTile {
    short level, row, c0, ncells;
    // (32-bit fields: the kernel reads them with scalar loads)
    int nc;                // = ncells
    int iniY, xa;          // first staged row, first staged column (a multiple of 16) of the level image
    int RH, nchunk;        // staged rows, 16-byte chunks per staged row
    int DH, TW;            // domain rows / columns (0 rows: the reference skips this run, :797-806)
    int j0, GPR;           // staged column of domain column 0; aligned dword groups per row that touch the domain
    int wCell, seg;        // cell width; domain rows per thread of the compass pass over all cells (256 threads)
    int py0, px0;          // image coordinates relative to (16, 16) of the domain's row 0 / column 0 (:824-825)
    int cellMagic;         // 65536 / wCell + 1:  c / wCell = (c * cellMagic) >> 16 for c < 65536 / wCell
    int grpMagic;          // 65536 / GPR + 1
    int dhMagic;           // 65536 / DH + 1
    int cntOff;            // index of the run's first cell in the per-frame cell arrays
    int candOff;           // index of the run's first candidate slot in the per-frame candidate array
    int cellCap;           // candidate slots per cell
    int stride;            // row stride of the level image (level 0: the launch argument counts)
    unsigned lvlOff;       // byte offset of the level inside one frame's pyramid block; level 0: 0xFFFFFFFF
    int pad;
};
static_assert(sizeof(FastTile) == 96, "FastTile layout");
void fast_tile_geometry(const OrbLevels &G, FastTile &t);

// One blur workgroup's work: BLUR_TILE_W x BLUR_TILE_H output tile (k_blur.hip: 64 raw rows = two 32-row MFMA tiles -> 58
// output rows; four waves of 32 columns).
#define BLUR_TILE_W 128
#define BLUR_TILE_H 58
// k_resize_fit: tiles fitted to the level (tile columns / rows, column groups per tile, rows per pass, passes, tid / twg by multiplication)
struct ResizeFit {
    int ntx = 0, nty = 0, twg = 0, rpp = 0, npass = 0, rmagic = 0;
};

struct BlurTile {
    short level, tx, ty, pad;
};

// One workgroup of the chained pyramid (k_pyramid_chain, a frame or two): a CHAIN_TW x CHAIN_TH tile of level `level`,
// built from level `base` through the levels between, all of them inside LDS.
#define CHAIN_TW 64
#define CHAIN_TH 16
struct ChainTile {
    short level, base, tx, ty;
};
struct ChainLevels {                       // by value: per level l (built from l - 1)
    float winx[ORBHIP_MAX_LEVELS], winy[ORBHIP_MAX_LEVELS];      // source / destination size ratios (window hint)
    uint32_t xoff[ORBHIP_MAX_LEVELS], yoff[ORBHIP_MAX_LEVELS];   // int32 offsets of the column / row tap tables in the table block
};
struct ChainGroup {                        // one launch: the tiles of levels base + 1 .. top
    int firstTile, ntiles, bufA, bufB, xtabBytes, ytabBytes;   // LDS: buffer A | buffer B | column taps | row taps
};

// ORB vocabulary: host-side parse result and the device tables.  The device tables are indexed
// by EDGE (position in the children CSR: the children of a node are consecutive, in file order), so that a
// descent step is two memory round trips: the child range of the current node, then all its children's
// descriptors at once.  id 0 = root (no edge).
struct OrbVocabHost {
    int k = 0, L = 0, scoring = 0, weighting = 0, nnodes = 0, nwords = 0;
    std::vector<uint8_t> desc, leaf;          // by node id
    std::vector<float> weight;                // by node id
    std::vector<int32_t> word, childOff, child;
    // by edge e (child[e] = node id of the e-th edge)
    std::vector<uint8_t> edesc;               // descriptor of the child
    std::vector<int32_t> erange;              // 2 per edge: child range [first, last) of the child
    std::vector<int32_t> eword;               // word id of the child (-1 for inner nodes)
    std::vector<float> eweight;
};
struct OrbVocabDev {
    int k = 0, L = 0, scoring = 0, weighting = 0, nnodes = 0, nwords = 0;
    int rootFirst = 0, rootLast = 0;          // child range of the root
    uint8_t *desc = nullptr;                  // [nnodes - 1][32] by edge
    int32_t *erange = nullptr;                // [nnodes - 1][2]
    int32_t *eid = nullptr;                   // [nnodes - 1] node id
    int32_t *eword = nullptr;
    float *eweight = nullptr;
    unsigned long long gen = 0;               // process-wide load counter (orbhip_vocab_generation): which load these tables are
};

// Host-fed pipeline (orbhip_pipe_*): a ring of `depth` device input slots and device / pinned-host output slots, a
// copy-in and a copy-out stream beside the context's compute stream.
struct OrbPipe {
    int depth = 0, B = 0, w = 0, h = 0, stride = 0, dcap = 0;
    size_t frameBytes = 0, inBytes = 0, outBytes = 0, koff = 0, doff = 0, coff = 0;
    hipStream_t sIn = nullptr, sOut = nullptr;
    std::vector<uint8_t *> d_in, d_out;         // `depth` device slots
    std::vector<uint8_t *> h_out;               // depth + 1 page-locked result blocks (batch n -> block n % (depth + 1)): no
                                                // submit that the ring admits overwrites the block the last wait returned
    std::vector<hipEvent_t> evIn, evK, evOut;
    std::vector<int> slotB;
    long submitted = 0, waited = 0;
    // optional matching stage (orbhip_pipe_enable_bow): vocabulary transform + SearchByBoW of frame b against frame b - 1
    bool bow = false;
    int levelsup = 4, check_ori = 1;
    float nnratio = 0.7f;
    size_t m12off = 0, m21off = 0, nmoff = 0;   // inside an output slot, behind the counts
    uint8_t *d_bowScratch = nullptr;            // word | weight | node, B * dcap entries each
    int lastWaited = -1;                        // host result block of the batch the last orbhip_pipe_wait returned
};

struct orbhip_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    hipStream_t stream2 = nullptr;   // blur runs here, concurrently with the quadtree kernel
    hipEvent_t evx[3] = {nullptr, nullptr, nullptr};   // pyramid done (cross-stream hand-over) | blur start | blur end
    int blurPlace = 0;               // where the blur runs in a batch (orbhip_set_blur_placement, include/orbhip.h)
    std::string err;

    // constructor tables (E0)
    int nfeatures = 0, nlevels = 0, iniTh = 0, minTh = 0;
    double scaleFactor = 0;
    float mvScaleFactor[ORBHIP_MAX_LEVELS], mvInvScaleFactor[ORBHIP_MAX_LEVELS];
    float mvLevelSigma2[ORBHIP_MAX_LEVELS], mvInvLevelSigma2[ORBHIP_MAX_LEVELS];
    int mnFeaturesPerLevel[ORBHIP_MAX_LEVELS];
    int umax[16];

    int max_w = 0, max_h = 0, max_batch = 0;

    // geometry of the image size currently configured (rebuilt when w/h change)
    int cur_w = 0, cur_h = 0;
    OrbLevels G;
    std::vector<FastTile> fastTiles;              // runs of up to 5 cells (batches), then runs of 1 cell (a frame or two)
    int nFastTilesBatch = 0;
    int nFastTilesTall = 0;                       // ... of which the last nFastTilesTall belong to levels with cells taller than 34 rows
    std::vector<BlurTile> blurTiles;              // level by level
    bool fuseBlurOk = false;                      // every level's k_resize_blur window fits (ORBHIP_FUSE_BLUR)
    int blurLevelFirst[ORBHIP_MAX_LEVELS + 1] = {};   // first tile of every level (and the end)
    std::vector<ChainTile> chainTiles;            // chained pyramid of the single-frame path (empty = not available)
    std::vector<ChainGroup> chainGroups;
    ChainLevels chainLevels;
    ChainTile *d_chainTiles = nullptr;
    size_t cap_chainTiles = 0;
    size_t pyrFrameBytes = 0;      // bytes of one frame's levels 1..n-1 (level 0 separate)
    size_t lvl0FrameBytes = 0;

    // device buffers (sized for max_w x max_h x max_batch at create time)
    uint8_t *d_lvl0 = nullptr;     // own copy of level 0 (host API)          [B][h][stride0]
    uint8_t *d_pyr = nullptr;      // levels 1.. of every frame               [B][pyrFrameBytes]
    uint8_t *d_blur = nullptr;     // blurred levels 0..                      [B][lvl0+pyr bytes]
    uint32_t *d_cand = nullptr;    // candidate slots                         [B][totalCands]
    uint16_t *d_cellCnt = nullptr; // per-cell counts                         [B][totalCells]
    uint32_t *d_pts = nullptr;     // compact candidates                      [B][totalPts]
    uint32_t *d_pnode = nullptr;   // quadtree scratch                        [B][totalPts]
    int32_t *d_lvlCandCnt = nullptr; // candidates per (frame, level)         [B][16]
    uint32_t *d_lvlKp = nullptr;   // quadtree winners (packed)               [B][totalKps]
    int32_t *d_lvlKpCnt = nullptr; // winners per (frame, level)              [B][16]
    uint8_t *d_qtTables = nullptr; // quadtree node tables when they exceed the LDS (large per-level quotas)
    size_t cap_qtTables = 0;
    float *d_lvlAngle = nullptr;   // orientation per winner                  [B][totalKps]
    orbhip_keypoint *d_kps = nullptr; // output staging (host API)            [B][outCap]
    uint8_t *d_desc = nullptr;     //                                          [B][outCap][32]
    int32_t *d_counts = nullptr;   //                                          [B]
    FastTile *d_fastTiles = nullptr;
    BlurTile *d_blurTiles = nullptr;
    uint32_t *d_blurBands = nullptr;   // 6 x 64 x 16 bytes: the MFMA band operands of k_blur (blur_band_table)
    int32_t *d_resizeTab = nullptr; // per level: x table [dw] int2, y table [dh] int4
    size_t resizeTabOff[ORBHIP_MAX_LEVELS][3];   // column taps, row taps, 4-pixel groups (k_pyramid.hip)
    bool resizeGroups[ORBHIP_MAX_LEVELS] = {};
    ResizeFit resizeFit[ORBHIP_MAX_LEVELS];       // .ntx == 0: the level keeps k_resize<32>
    bool resizeHint[ORBHIP_MAX_LEVELS][2] = {};  // the computed source window is valid for 32-row / 8-row tiles    // the level has a group table (fast path of k_resize)
    size_t cap_lvl0 = 0, cap_pyr = 0, cap_blur = 0, cap_cand = 0, cap_cells = 0, cap_pts = 0,
           cap_kps = 0, cap_out = 0, cap_resize = 0, cap_fastTiles = 0, cap_blurTiles = 0,
           cap_pnode = 0, cap_angle = 0, cap_cnt1 = 0, cap_cnt2 = 0, cap_cnt3 = 0;

    // state of the last extract call
    bool blurValid = false;        // d_blur holds the blurred pyramid of the last call (false after a batch through k_describe_blur)
    const uint8_t *last_lvl0 = nullptr;  // device pointer to level 0 of frame 0
    int last_stride0 = 0;
    size_t last_frame0 = 0;
    int last_B = 0;

    // pinned host staging for the host API
    uint8_t *h_stage = nullptr;
    size_t h_stage_bytes = 0;
    uint8_t *h_pack = nullptr;        // page-locked twin of d_tmp for the small host-pointer calls (struct Packed, api_common.h)
    size_t h_pack_bytes = 0;
    // the host-pointer call of a frame or two as ONE hipGraph launch (copy in, the twelve kernels, copy out): captured at the
    // first call of a geometry, replayed while (w, h, B, buffers) stay the same
    uint8_t *h_in = nullptr;          // pinned input staging, rows s0 apart
    size_t h_in_bytes = 0;
    // host copy of the pyramid levels 1.. (pinned), filled beside the kernels when orbhip_set_host_pyramid is on: the
    // drop-in's mvImagePyramid (Frame.cc:817) are headers into it and into h_in (level 0)
    bool hostPyr = false;
    uint8_t *h_pyr = nullptr;
    size_t h_pyr_bytes = 0;
    int h_pyr_B = 0;                  // frames of the last call that are valid in h_pyr (0 = none)
    bool h_in_valid = false;          // h_in holds the frames of the last call
    hipEvent_t evp[2] = {nullptr, nullptr};   // pyramid built | pyramid copied out
    hipGraphExec_t g_exec = nullptr;
    hipGraph_t g_graph = nullptr;
    int g_w = 0, g_h = 0, g_B = 0;
    bool capturing = false;           // run_pipeline leaves the timing events out of a capture
    unsigned g_calls = 0;
    const void *g_key[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};   // d_lvl0, d_kps block, h_in, h_stage, h_pyr (or null) at capture time
    unsigned long allocGen = 0;       // bumped whenever a device buffer of the context is reallocated (ensure())
    unsigned long g_gen = 0;          // allocGen at capture time: part of the replay key (the kernel nodes hold d_pyr, d_cand, ...)

    // matching scratch
    void *d_match = nullptr;
    size_t d_match_bytes = 0;
    // staging block of the host-pointer entry points (grow-only; they used to hipMalloc / hipFree per call)
    void *d_tmp = nullptr;
    size_t d_tmp_bytes = 0;

    // timing
    hipEvent_t ev[8] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    bool haveStageEvents = false, haveMatchEvents = false;
    int stageTiming = 2;     // orbhip_set_stage_timing: 2 = every stage (default), 1 = the FAST launch only, 0 = none
    bool haveFastEvents = false;

    // rectification maps (orbhip_remap_set_maps): mapx then mapy, map_w * map_h floats each
    float *d_maps = nullptr;
    int map_w = 0, map_h = 0;

    // vocabulary
    OrbVocabDev voc;
    mutable std::mutex vocMutex;              // orders a load on this context against another thread's share / generation of it
    std::shared_ptr<void> vocHold;            // the device block behind `voc`, shared by every context that borrowed it
                                              // (orbhip_vocab_share): freed when the last of them lets go

    // host-fed pipeline
    OrbPipe *pipe = nullptr;
    // resident feature sets (api_sets.hip)
    void *setTable = nullptr;
    // orbhip_frame_build (api_frame.hip): the Frame constructor's device work as one captured graph
    void *frameBuild = nullptr;
    long long describeMirror = 0;     // != 0 while orbhip_frame_build enqueues: k_describe<.., MIRROR> stores its results twice

    // RCCL
    void *comm = nullptr;
    int rank = 0, nranks = 1;
};

// ---- geometry (orb_geometry.hip) ----
int orb_init_tables(orbhip_ctx *c, int nfeatures, float scaleFactor, int nlevels, int iniTh, int minTh);
void orb_level_size(const orbhip_ctx *c, int w, int h, int level, int *lw, int *lh);
// Builds c->G, tiles and resize tables for a w x h image.  Returns ORBHIP_OK or an error.
int orb_build_geometry(orbhip_ctx *c, int w, int h, int stride0);
// Host-side resize tables for level `l` (dst) from level l-1 (src).
bool orb_build_resize_groups(const std::vector<int32_t> &xtab, const std::vector<int32_t> &ytab, int dw,
                             std::vector<int32_t> &gtab);
void orb_build_resize_tables(int sw, int sh, int dw, int dh, std::vector<int32_t> &xtab,
                             std::vector<int32_t> &ytab);

// ---- kernel launchers ----
void launch_resize(hipStream_t s, const uint8_t *src, int sw, int sh, int sstride, size_t sframe,
                   uint8_t *dst, int dw, int dh, int dstride, size_t dframe, const int32_t *xtab,
                   const int32_t *ytab, const int32_t *gtab, bool hint, int B);
bool resize_hint_fits(const int32_t *xt, const int32_t *yt, int sw, int sh, int dw, int dh, int th);
bool resize_fit_plan(const int32_t *xt, const int32_t *yt, int sw, int sh, int dw, int dh, ResizeFit &out);
void launch_resize_fit(hipStream_t s, const uint8_t *src, int sw, int sh, int sstride, size_t sframe, uint8_t *dst, int dw, int dh,
                       int dstride, size_t dframe, const int32_t *ytab, const int32_t *gtab, const ResizeFit &f, int B);
bool resize_hint_pointwise(const int32_t *xt, const int32_t *yt, int sw, int sh, int dw, int dh);
// plans the chained pyramid (groups of levels per launch, their tiles and LDS sizes); false = some level cannot be chained
bool chain_plan(const OrbLevels &G, const bool *levelOk, const ChainLevels &CL, std::vector<ChainTile> &tiles,
                std::vector<ChainGroup> &groups);
void launch_pyramid_chain(hipStream_t s, const OrbLevels &G, const ChainLevels &CL, const ChainGroup &grp, const ChainTile *tiles,
                          const uint8_t *lvl0, int stride0, size_t frame0, uint8_t *pyr, size_t pyrFrame, const int32_t *tab, int B,
                          uint8_t *hostPyr);
void launch_fast(hipStream_t s, const OrbLevels &G, const uint8_t *lvl0, int stride0, size_t frame0,
                 const uint8_t *pyr, size_t pyrFrame, const FastTile *tiles, int ntiles,
                 uint32_t *cand, uint16_t *cellCnt, int B, int ntall = 0);
void launch_quadtree(hipStream_t s, const OrbLevels &G, const uint32_t *cand, const uint16_t *cellCnt,
                     uint32_t *pts, uint32_t *pnode, int32_t *lvlCandCnt, uint32_t *lvlKp,
                     int32_t *lvlKpCnt, int B, uint8_t *tableScratch);
size_t quadtree_table_scratch_bytes(const OrbLevels &G, int B);   // 0 when the node tables fit in LDS
bool resize_blur_fits(const int32_t *xt, const int32_t *yt, int sw, int sh, int dw, int dh);
void launch_resize_blur(hipStream_t s, const uint8_t *src, int sw, int sh, int sstride, size_t sframe, uint8_t *dst, int dw, int dh,
                        int dstride, size_t dframe, uint8_t *bdst, int bstride, size_t bframe, const int32_t *ytab, const int32_t *gtab,
                        const uint32_t *bands, int B);
void launch_blur(hipStream_t s, const OrbLevels &G, const uint8_t *lvl0, int stride0, size_t frame0,
                 const uint8_t *pyr, size_t pyrFrame, uint8_t *blur, size_t blurFrame,
                 const BlurTile *tiles, int ntiles, const uint32_t *bands, int B);
void blur_band_table(uint32_t out[6 * 64 * 4]);
void launch_describe(hipStream_t s, const OrbLevels &G, const uint8_t *lvl0, int stride0, size_t frame0,
                     const uint8_t *pyr, size_t pyrFrame, const uint8_t *blur, size_t blurFrame,
                     const uint32_t *lvlKp, const int32_t *lvlKpCnt, float *lvlAngle,
                     orbhip_keypoint *kps, uint8_t *desc, int32_t *counts, int cap, int B, long long mirror = 0);
bool describe_blur_available(const OrbLevels &G, int B);
void launch_describe_blur(hipStream_t s, const OrbLevels &G, const uint8_t *lvl0, int stride0, size_t frame0, const uint8_t *pyr,
                          size_t pyrFrame, const uint32_t *lvlKp, const int32_t *lvlKpCnt, float *lvlAngle, orbhip_keypoint *kps,
                          uint8_t *desc, int32_t *counts, int cap, int B);
size_t quadtree_lds_bytes(const OrbLevels &G);

void launch_knn2(hipStream_t s, const uint8_t *q, int nq, const uint8_t *db, int ndb, int32_t *best_idx,
                 int32_t *best_d, int32_t *second_d, void *scratch, size_t scratch_bytes);
size_t knn2_scratch_bytes(int nq, int ndb);
void launch_knn2_seq(hipStream_t s, const uint8_t *desc, const int32_t *counts, int cap, int B, int lag,
                     int32_t *best_idx, int32_t *best_d, int32_t *second_d);
void launch_knn2_merge(hipStream_t s, const int32_t *parts, int nshards, int nq, int32_t *best_idx, int32_t *best_d,
                       int32_t *second_d);
void launch_fill_i32(hipStream_t s, int32_t *p, int32_t v, int n);
void launch_knn2_lists(hipStream_t s, const uint8_t *q, int nq, const uint8_t *db, const int32_t *off,
                       const int32_t *cand, int32_t *best_idx, int32_t *best_d, int32_t *second_d);
void launch_bow_match(hipStream_t s, const uint8_t *desc1, const uint8_t *valid1, const int32_t *off1,
                      const int32_t *idx1, const uint8_t *desc2, const uint8_t *valid2,
                      const int32_t *off2, const int32_t *idx2, const int32_t *pairs, int npairs,
                      int th, int th_mode, float nnratio, int32_t *match12, int32_t *match21);

// XCD-aware work assignment for (tile, frame) grids -- a per-kernel default (see orb_xcd_map), off for k_fast.  Workgroups are dealt
// round-robin over the 8 XCDs by linear id (MI355X_MICROARCH.md, "Workgroup dispatch"; placement affects speed
// only).  With ORBHIP_XCD_MAP != 0 the grid's x extent is padded to a multiple of 8, workgroup x of a frame runs
// on XCD x % 8 and takes a tile from a contiguous eighth of the frame's tiles, so the 128-byte lines that
// neighbouring tiles share (halo rows, row segments straddling a line) are fetched into one XCD's L2 once
// instead of once per XCD.  Measured (profiles/r01e_xcd_map.md): fabric read traffic of k_fast / k_blur /
// k_describe falls 2-3x to about the algorithmic bytes, but every variant is SLOWER than the plain mapping
// (k_fast +2..+100 %): the re-reads are served by the Infinity Cache, the kernels are VALU/LDS- or
// latency-bound, and spreading neighbouring tiles over all eight L2s balances the load better.  Hence 0.
static inline int orb_xcd_map(int dflt = 0)
{
    static int v = -2;
    if (v == -2) v = ORB_TUNE("XCD_MAP", -1);
    return v >= 0 ? v : dflt;   // per-kernel defaults: k_blur 2 (uniform tiles), k_describe / k_resize 1, k_fast 0
}
static inline int orb_xcd_chunk();
static inline int orb_xcd_arg(int dflt = 0);
static inline int orb_xcd_chunk()
{
    static int v = -1;
    if (v < 0) v = ORB_TUNE("XCD_CHUNK", 4);
    return v;
}
static inline int orb_xcd_grid(int ntiles, int dflt = 0)
{
    const int m = orb_xcd_map(dflt);
    if (!m) return ntiles;
    const int unit = m == 3 ? 8 * orb_xcd_chunk() : 8;
    return (ntiles + unit - 1) / unit * unit;
}
static inline int orb_xcd_arg(int dflt) { return orb_xcd_map(dflt) | (orb_xcd_chunk() << 8); }
#ifdef __HIPCC__
// Minimum over the 16 lanes of a DPP row (result in every lane of the row) / over the wave (a scalar).  The row steps are
// v_min_i32 with a DPP source operand: the builtin form (min(v, update_dpp(v, v, ...))) compiles to v_mov_b32 + s_nop + v_mov_b32_dpp
// + v_min_i32 per step, and in the one-wave kernels that use these (sequential matching semantics: their time is their instruction
// count) that was a quarter of the instructions of a feature.  Every lane of the wave must be active.
__device__ __forceinline__ int orb_row_min_i(int v)
{
    asm volatile("s_nop 1\n\t"
                 "v_min_i32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
                 "s_nop 1\n\t"
                 "v_min_i32_dpp %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
                 "s_nop 1\n\t"
                 "v_min_i32_dpp %0, %0, %0 row_half_mirror row_mask:0xf bank_mask:0xf\n\t"
                 "s_nop 1\n\t"
                 "v_min_i32_dpp %0, %0, %0 row_mirror row_mask:0xf bank_mask:0xf\n\t"
                 "s_nop 1"
                 : "+v"(v));
    return v;
}
__device__ __forceinline__ int orb_wave_min_i(int v)
{
    v = orb_row_min_i(v);
    return min(min(__builtin_amdgcn_readlane(v, 0), __builtin_amdgcn_readlane(v, 16)),
               min(__builtin_amdgcn_readlane(v, 32), __builtin_amdgcn_readlane(v, 48)));
}

// mode 4 (k_describe): a whole FRAME per XCD.  Workgroups are dealt round-robin over the XCDs by linear id
// L = blockIdx.y * gridDim.x + blockIdx.x, so XCD k receives the ids L = k, k + 8, ...; the i-th of them (i = L / 8) works on
// tile i % gridDim.x of frame k + 8 * (i / gridDim.x): every workgroup that touches a frame's raw and blurred pyramids
// (1.9 MB at 640 x 480) shares one 4 MB L2, and the eight XCDs hold eight different frames.  Placement affects speed only.
__device__ __forceinline__ void xcd_frame_tile(int nframes, int &tile, int &frame)
{
    const unsigned T = gridDim.x, L = blockIdx.y * T + blockIdx.x;
    const unsigned full = ((unsigned)nframes >> 3) << 3;   // frames in complete groups of 8
    if (L < full * T) {
        // the ids below full * T: residue class k holds (full / 8) * T of them, one per (tile, frame = k + 8 q)
        const unsigned k = L & 7u, i = L >> 3, q = i / T;
        tile = (int)(i - q * T);
        frame = (int)(k + 8u * q);
    } else {
        // the last nframes % 8 frames: in id order (their workgroups spread over the XCDs)
        const unsigned t = L - full * T, f = t / T;
        frame = (int)(full + f);
        tile = (int)(t - f * T);
    }
}
// mode 1: band (x % 8 + frame) % 8 -- every XCD sees every band over 8 consecutive frames (tile cost differs
// between pyramid levels); mode 2: band x % 8; mode 3: chunks of xcdMap >> 8 tiles dealt round-robin.
__device__ __forceinline__ int xcd_tile(int xcdMap)
{
    const unsigned x = blockIdx.x, k = x & 7u, s = x >> 3, bw = gridDim.x >> 3;
    const int mode = xcdMap & 255;
    if (mode == 1) return (int)(((k + blockIdx.y) & 7u) * bw + s);
    if (mode == 2) return (int)(k * bw + s);
    if (mode == 3) {
        const unsigned c = (unsigned)xcdMap >> 8;
        return (int)(((s / c) * 8u + k) * c + s % c);
    }
    return (int)x;
}
#endif

int launch_grid_build(hipStream_t s, const orbhip_keypoint *kps, const int32_t *cnt, int cap, int B, float minX,
                      float minY, float invW, float invH, int32_t *cellOff, int32_t *cellIdx, int32_t *cellOff2 = nullptr,
                      int32_t *cellIdx2 = nullptr);
int launch_area_list(hipStream_t s, const orbhip_keypoint *kps, float minX, float minY, float invW, float invH,
                     const int32_t *cellOff, const int32_t *cellIdx, const orbhip_proj_query *queries, int nq, int slots,
                     int32_t *outCnt, int32_t *outIdx);
size_t window_best_scratch_bytes(int B, int cap);
int launch_window_best(hipStream_t s, const orbhip_keypoint *kps, const uint8_t *desc, int cap, int B, const float *uRight,
                       const float *invLevelSigma2, int nlevels, float minX, float minY, float invW, float invH,
                       const int32_t *cellOff, const int32_t *cellIdx, const orbhip_proj_query *queries,
                       const uint8_t *qdesc, const int32_t *nq, int capQ, int32_t *bestIdx, int32_t *bestDist, void *scratch);
size_t proj_scratch_bytes(int B, int capQ, int cap);
size_t proj_assign_lds(int cap);
int launch_search_by_projection(hipStream_t s, const orbhip_keypoint *kps, const uint8_t *desc, const int32_t *cnt, int cap,
                                int B, const float *uRight, const uint8_t *occupied, float minX, float minY, float invW,
                                float invH, const int32_t *cellOff, const int32_t *cellIdx, const orbhip_proj_query *queries,
                                const uint8_t *qdesc, const int32_t *nq, int capQ, int use_ratio, float nnratio,
                                int check_ori, int th_high, int32_t *match, int32_t *nmatches, void *scratch);
void launch_distinctive(hipStream_t s, const uint8_t *desc, const int32_t *off, int P, int32_t *best, int32_t *bestMedian);
void launch_tri_match(hipStream_t s, const orbhip_keypoint *kps1, const uint8_t *desc1, const uint8_t *skip1, const float *ur1,
                      const int32_t *off1, const int32_t *idx1, const orbhip_keypoint *kps2, const uint8_t *desc2,
                      const uint8_t *skip2, const float *ur2, const int32_t *off2, const int32_t *idx2, const int32_t *pairs,
                      int npairs, const float F12[9], float ex, float ey, int only_stereo, int th_low, const float *scale2,
                      const float *sigma2, int32_t *match12);
size_t init_scratch_bytes(int B, int cap1, int cap2);
size_t init_assign_lds(int cap1, int cap2);
int launch_search_for_initialization(hipStream_t s, const orbhip_keypoint *kps1, const uint8_t *desc1, const int32_t *cnt1,
                                     int cap1, const orbhip_keypoint *kps2, const uint8_t *desc2, const int32_t *cnt2, int cap2,
                                     int B, float minX, float minY, float invW, float invH, const int32_t *cellOff2,
                                     const int32_t *cellIdx2, float *prev, int windowSize, float nnratio, int check_ori,
                                     int th_low, int32_t *matches12, int32_t *nmatches, void *scratch);
int launch_undistort(hipStream_t s, const orbhip_keypoint *kps, const int32_t *cnt, int cap, int B, const float *K,
                     const float *D, int nD, const float *P, orbhip_keypoint *out, orbhip_keypoint *out2 = nullptr);
int launch_remap(hipStream_t s, const uint8_t *src, int B, int sw, int sh, int sstride, size_t sframe, const float *mapx,
                 const float *mapy, int dw, int dh, uint8_t *dst, int dstride, size_t dframe);
void orb_init_undistort_rectify_map(const double *K, const double *D, int nD, const double *R, const double *P, int w,
                                    int h, float *mapx, float *mapy);
size_t stereo_scratch_bytes(int B, int cap);
int launch_stereo(orbhip_ctx *L, orbhip_ctx *R, const orbhip_keypoint *kpsL, const uint8_t *descL, const int32_t *cntL,
                  const orbhip_keypoint *kpsR, const uint8_t *descR, const int32_t *cntR, int cap, int B, float mb,
                  float mbf, float *uRight, float *depth, int32_t *scratch, int32_t *nmatch);
int orb_vocab_parse(const uint8_t *blob, size_t nbytes, OrbVocabHost &V, std::string &err);
void launch_vocab_transform(hipStream_t s, const OrbVocabDev &V, const uint8_t *desc, int n, int levelsup,
                            int32_t *word_id, float *weight, int32_t *node_id, const int32_t *cnt = nullptr);
hipError_t launch_bow_seq(hipStream_t s, const uint8_t *desc, const orbhip_keypoint *kps, const int32_t *counts,
                          const int32_t *node, const float *weight, const uint8_t *valid, int cap, int B, int lag, int th,
                          int th_mode, float nnratio, int check_ori, int32_t *match12, int32_t *match21,
                          int32_t *nmatches);

#endif
