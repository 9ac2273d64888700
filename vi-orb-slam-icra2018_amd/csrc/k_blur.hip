// k_blur.hip -- E6: cv::GaussianBlur(level, 7x7, sigma 2, BORDER_REFLECT_101) for every level of
// every frame in one launch (ref call site: src/ORBextractor.cc:1103-1104; OpenCV 2.4 smooth.cpp
// / filter.cpp fixed-point separable filter: kernel {18,34,49,55,49,34,18}/256 applied twice,
// sum brought back by 2^16).
//
// Rounding of the column pass follows the x86-64 (SSE2) build of OpenCV 2.4 that the reference
// links: for x < w - w%4 the 32s->8u symmetric column filter accumulates in float and converts
// with cvtps2dq (round half to EVEN); the scalar tail uses (sum + 32768) >> 16 (half UP).  All
// float products/sums there are exact below 256, so the float path is evaluated as an integer
// tie-break rule -- DESIGN.md "blur".
//
// r03: both passes run on the matrix pipe as banded-Toeplitz products (the VALU is what k_fast and k_describe
// saturate; r02's dot4 / dot2 formulation cost 27 lane-instructions per pixel, this one ~6).  One 256-thread
// workgroup per 128 x 58 output tile, one wave per 32-column strip:
//   1. raw rows y0-3 .. y0+60 (reflected at the top / bottom), columns x0-16 .. x0+175, straight into LDS
//      (global_load_lds_dwordx4, no registers; two tiles per workgroup, the second one in flight while the first is
//      computed); at the left / right image edge the 3 reflected columns are patched into the halo;
//   2. row pass, v_mfma_i32_32x32x32_i8: H[row][col] = sum_c (p[row][c] - 128) * Kx[c][col] + 128 * 257, A = 32 rows x
//      32 input columns straight from LDS (ds_read_b128: the 16 consecutive bytes of a lane ARE its 16 k-slots; xor 0x80
//      turns the pixels into int8),
//      B = the band matrix of the taps (a per-lane constant), two k-steps per 32 x 32 tile, two row tiles (64 rows);
//      the result tile has its column on the lane and 16 rows in the lane's registers;
//   3. column pass, v_mfma_f32_32x32x16_f16 with the row-pass tile as the A operand in place (no LDS round trip:
//      the contraction runs over the tile's ROW index, which lives in the registers): the 16-bit sums are split
//      into byte planes, each byte becomes the binary16 pattern 0x0400 | byte = (1024 + byte) * 2^-24 -- ONE
//      v_perm per two values, the 0x04 byte rides in the accumulator's start value -- and the band weights are
//      tap * 2^8, so a product is tap * (1024 + byte) * 2^-16 and the f32 accumulator (started at minus the
//      constant part) holds EXACTLY sum_j tap_j * byte_j * 2^-16: every partial sum is a multiple of 2^-16 below 8.
//      Two accumulators (low-byte plane, high-byte plane); result = fma(high, 256, low) = S * 2^-16 exactly, and
//      v_cvt_pk_u8_f32 is the SSE2 sequence cvtps2dq + packus (round half to even, unsigned saturation) in one
//      instruction: two vector instructions per output pixel;
//   4. the 32 x 32 result tiles (output row on the lane, four adjacent columns per register group) go through an
//      LDS image of the output tile and leave with 16-byte row-coalesced stores.
// Measured and not kept (profiles/r03): the same staged tile also yielding the pixels of the NEXT pyramid level (one launch
// per level builds blurred level l and raw level l + 1, every level read once instead of twice: 1 GB less traffic per
// 1024 frames, bit-exact) -- 1.19-1.35 ms against 0.60 + 0.57 for the separate kernels: neither kernel is bound by bytes.
// Bound: nominally HBM (reads and writes one byte per pixel; halo re-reads 64/58 x 160/128); measured: the lifetime of a
// workgroup (load latency + matrix chain + store) at four workgroups per CU.
#include "orbhip_internal.h"

#include <algorithm>
#include <cmath>
#include <vector>

#define BM_W BLUR_TILE_W          // 128 output columns = 4 waves x 32
#define BM_H BLUR_TILE_H          // 58 output rows
#define BM_IN 64                  // raw rows (two 32-row tiles of the row pass)
#define BM_CW 13                  // 16-byte chunks of a raw row in LDS: 16 halo + 128 + 16 halo, 2 spare, 1 pad (see blur_dma)
#define BM_PITCH (BM_CW * 16)     // 208: 52 dwords, so that 16 rows' 16-byte reads fall into distinct bank groups
#define BM_OP 144                 // pitch of the output image: 36 dwords (32 would put every row's dword store in one bank)

typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));
typedef _Float16 v8h __attribute__((ext_vector_type(8)));
typedef float v16f __attribute__((ext_vector_type(16)));

__device__ __forceinline__ int reflect101(int p, int len)
{
    // BORDER_REFLECT_101: gfedcb|abcdefgh|gfedcba ; |excursion| <= 3 < len
    if (p < 0) p = -p;
    if (p >= len) p = 2 * len - 2 - p;
    return p;
}

// row of a 32x32 MFMA result tile that register i of a lane with l >> 5 == hh holds (cdna_hip_programming.md section 3)
__host__ __device__ __forceinline__ int tile_row(int i, int hh) { return (i & 3) + 8 * (i >> 2) + 4 * hh; }

// geometry of one tile, everything block-uniform
struct BlurGeom {
    const uint8_t *src;
    uint8_t *dst;
    int sstride, dstride, w, h, x0, y0;
};

__device__ __forceinline__ BlurGeom blur_geom(const OrbLevels &G, const BlurTile T, int frame, const uint8_t *lvl0, int stride0,
                                              unsigned long long frame0, const uint8_t *pyr, unsigned long long pyrFrame,
                                              uint8_t *blur, unsigned long long blurFrame)
{
    BlurGeom g;
    const int l = T.level;
    const OrbLevel &L = G.lv[l];
    g.w = L.w;
    g.h = L.h;
    if (l == 0) {
        g.src = lvl0 + (size_t)frame * frame0;
        g.sstride = stride0;
    } else {
        g.src = pyr + (size_t)frame * pyrFrame + L.imgOff;
        g.sstride = L.stride;
    }
    g.dst = blur + (size_t)frame * blurFrame + (l == 0 ? 0ull : G.boff1 + L.imgOff);
    g.dstride = l == 0 ? G.bstride0 : L.stride;
    g.x0 = T.tx * BM_W;
    g.y0 = T.ty * BM_H;
    return g;
}

// The raw tile travels global -> LDS without passing through registers (global_load_lds_dwordx4: a wave instruction writes
// 64 x 16 bytes to CONSECUTIVE LDS addresses, so the LDS image is chunk-linear: chunk id = 13 * row + column chunk, the
// thirteenth chunk of a row being the pad that keeps the rows' bank groups apart).  16 wave instructions of four rows each,
// dealt to the four waves (blur_dma).  A workgroup takes TWO consecutive tiles of a frame and requests both at once, each into its own LDS
// buffer: the second tile travels while the first is computed.  Order of one workgroup (no wait ever covers a store):
//     request tile 0, tile 1, the tables | wait tile 0 | compute 0 | wait tile 1 | store 0 | compute 1 | store 1
#define BM_TPW 2

// one global_load_lds_dwordx4: 16 bytes per lane from its own address to LDS byte address ldsAddr + 16 * lane.  As inline
// assembly: the builtin makes hipcc wait vmcnt(0) before every LDS access that might alias the destination (here: every
// read of the OTHER buffer), which serialises the transfers and drains them before the compute phase; the waits are ours.
// M0 carries the LDS address and is restored.
__device__ __forceinline__ void glds16(const void *gsrc, uint32_t ldsAddr)
{
    uint32_t keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(gsrc), "s"(ldsAddr)
                 : "memory");
}

// A transfer carries FOUR WHOLE ROWS of the chunk-linear image (4 x 13 chunks = 52 of the 64 lanes; 16 transfers per tile, four
// per wave): a lane's (row of the four, chunk) is then the same for every transfer, the global address of transfer t is the
// lane's first address plus 4 t rows -- a scalar offset -- and the rows need no reflection unless the tile touches the top or
// the bottom of the image (block-uniform).  (The first version dealt 64 consecutive chunks to a transfer, 13 transfers per
// tile: a division by 13, a reflection and two clamps per lane and transfer -- a quarter of the kernel's vector instructions.)
__device__ __forceinline__ void blur_dma(const BlurGeom &g, uint8_t *ldsTile, int wv, int lane)
{
    const uint32_t ldsBase = (uint32_t)(uintptr_t)ldsTile;   // LDS byte address (low half of the flat address)
    const int wAl = (g.w + 15) & ~15;   // bytes of a row that may be read with 16-byte loads
    const int rl = (lane * 5) >> 6;     // lane / 13 for lane < 64
    const int c = lane - BM_CW * rl;
    const bool live = lane < 4 * BM_CW;
    // (a chunk outside the row -- or the pad chunk -- reads other pixels of the row: columns beyond the three reflected
    // ones, patched below, only reach outputs outside the image)
    const int xoff = min(max(g.x0 - 16 + (c << 4), 0), wAl - 16);
    const int yb = g.y0 - 3;            // image row of the tile's row 0
    if (yb >= 0 && yb + BM_IN <= g.h) {
        const uint8_t *p = g.src + (size_t)(yb + rl) * g.sstride + xoff;
#pragma unroll
        for (int j4 = 0; j4 < BM_IN / 16; j4++) {
            const int t = wv + 4 * j4;   // wave-uniform
            if (live) glds16(p + (size_t)(4 * t) * g.sstride, __builtin_amdgcn_readfirstlane(ldsBase + (uint32_t)(4 * BM_PITCH) * (uint32_t)t));
        }
    } else {
#pragma unroll
        for (int j4 = 0; j4 < BM_IN / 16; j4++) {
            const int t = wv + 4 * j4;
            const int sy = reflect101(min(yb + 4 * t + rl, g.h + 2), g.h);
            if (live) glds16(g.src + (size_t)sy * g.sstride + xoff, __builtin_amdgcn_readfirstlane(ldsBase + (uint32_t)(4 * BM_PITCH) * (uint32_t)t));
        }
    }
}
static_assert(BM_PITCH == 16 * BM_CW && BM_IN % 16 == 0, "four rows of the chunk-linear image per transfer");

#define BM_RAW_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")

__global__ __launch_bounds__(256) void k_blur(const OrbLevels G, const uint8_t *__restrict__ lvl0,
                                              int stride0, unsigned long long frame0,
                                              const uint8_t *__restrict__ pyr, unsigned long long pyrFrame,
                                              uint8_t *__restrict__ blur, unsigned long long blurFrame,
                                              const BlurTile *__restrict__ tiles, const uint4 *__restrict__ bands, int xcdMap,
                                              int ntiles)
{
    // ONE shared array (a second __shared__ object can make the compiler wait for the transfers before LDS reads)
    __shared__ __align__(16) uint8_t smem[2 * BM_IN * BM_PITCH + 64 * BM_OP];
    uint8_t *const s_out0 = smem + 2 * BM_IN * BM_PITCH;         // [64][BM_OP]; rows 58..63: the unused part of the second result tile
    const int tile0 = xcd_tile(xcdMap) * BM_TPW, frame = blockIdx.y;
    if (tile0 >= ntiles) return;   // grid padded to a multiple of 8 (orbhip_internal.h, xcd_tile)
    const bool two = tile0 + 1 < ntiles;
    const int tid = threadIdx.x, lane = tid & 63, n = lane & 31, hh = lane >> 5;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);   // the wave's strip, as a scalar

    // the band operands of this lane (built on the host, launch_blur): 2 k-steps of the row pass, 4 of the column pass
    uint4 bq[6];
#pragma unroll
    for (int q = 0; q < 6; q++) bq[q] = bands[q * 64 + lane];
    // (the tile descriptors are requested AFTER the band operands and needed first -- for the addresses below --, so the wait
    // the compiler puts in front of their first use retires the band operands too: nothing it knows about is outstanding
    // once the transfers are in flight, and it adds no wait of its own that would cover them)
    asm volatile("" ::: "memory");
    const BlurTile T0 = tiles[tile0], T1 = tiles[min(tile0 + 1, ntiles - 1)];
    asm volatile("" :: "s"((int)T0.level + (int)T0.tx + (int)T0.ty), "s"((int)T1.level + (int)T1.tx + (int)T1.ty));
    const BlurGeom g0 = blur_geom(G, T0, frame, lvl0, stride0, frame0, pyr, pyrFrame, blur, blurFrame);
    const BlurGeom g1 = blur_geom(G, T1, frame, lvl0, stride0, frame0, pyr, pyrFrame, blur, blurFrame);
    blur_dma(g0, smem, wv, lane);
    if (two) blur_dma(g1, smem + BM_IN * BM_PITCH, wv, lane);
    // tile 0 has landed when at most tile 1's four transfers are outstanding (in-order return)
    if (two) {
        asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    BM_RAW_BARRIER();

#define S_RAW(r, c) raw[(r) * BM_PITCH + (c)]
    // reflected halo columns at the image edges (x = -1,-2,-3 <- 1,2,3 ; x = w,w+1,w+2 <- w-2,w-3,w-4): block-uniform
    auto edge_patch = [&](uint8_t *raw, const BlurGeom &cur) {
        const int w = cur.w, x0 = cur.x0;
        const bool edgeL = x0 == 0, edgeR = x0 + BM_W + 3 > w;
        if (edgeL || edgeR) {
            if (edgeL)
                for (int i = tid; i < BM_IN * 3; i += 256) {
                    const int r = i / 3, k = i - r * 3 + 1;
                    S_RAW(r, 16 - k) = S_RAW(r, 16 + k);
                }
            if (edgeR)
                for (int i = tid; i < BM_IN * 3; i += 256) {
                    const int r = i / 3, k = i - r * 3;
                    S_RAW(r, 16 + (w + k - x0)) = S_RAW(r, 16 + (w - 2 - k - x0));
                }
            BM_RAW_BARRIER();
        }
    };

    // stages 2 and 3: the strip of this wave, raw tile -> s_out
    auto compute = [&](const uint8_t *raw, const BlurGeom &cur) {
        const int w = cur.w, x0 = cur.x0;
        const int c0 = x0 + 32 * wv;            // first output column of this wave's strip
        if (c0 >= w) return;                    // (wave-uniform; a strip right of the image has nothing to store)
        // ---- 2. row pass: H[T][i] = 16-bit row sum at (row 32 T + tile_row(i, hh), column c0 + n), + 0x04000000 ----
        v16i hinit;
        v16f zinit;
#pragma unroll
        for (int i = 0; i < 16; i++) {
            hinit[i] = 128 * 257 + 0x04000000;
            zinit[i] = -(float)(257 * 1024) / 65536.0f;            // minus sum_j tap_j * 1024 * 2^-16
        }
        v16i H[2];
#pragma unroll
        for (int t = 0; t < 2; t++) {
            uint4 a0 = *reinterpret_cast<const uint4 *>(&S_RAW(32 * t + n, 32 * wv + 16 * hh));
            uint4 a1 = *reinterpret_cast<const uint4 *>(&S_RAW(32 * t + n, 32 * wv + 32 + 16 * hh));
            a0.x ^= 0x80808080u; a0.y ^= 0x80808080u; a0.z ^= 0x80808080u; a0.w ^= 0x80808080u;   // pixel - 128 as int8
            a1.x ^= 0x80808080u; a1.y ^= 0x80808080u; a1.z ^= 0x80808080u; a1.w ^= 0x80808080u;
            H[t] = __builtin_amdgcn_mfma_i32_32x32x32_i8(__builtin_bit_cast(v4i, a0), __builtin_bit_cast(v4i, bq[0]), hinit, 0, 0, 0);
            H[t] = __builtin_amdgcn_mfma_i32_32x32x32_i8(__builtin_bit_cast(v4i, a1), __builtin_bit_cast(v4i, bq[1]), H[t], 0, 0, 0);
        }
        // ---- 3. column pass ----
        // byte planes as binary16 pairs: slot j of k-step s of row tile t is register 8 s + j
        uint4 lo[2][2], hi[2][2];
#pragma unroll
        for (int t = 0; t < 2; t++)
#pragma unroll
            for (int s = 0; s < 2; s++) {
                uint32_t pl[4], ph[4];
#pragma unroll
                for (int d = 0; d < 4; d++) {
                    const uint32_t a = (uint32_t)H[t][8 * s + 2 * d], b = (uint32_t)H[t][8 * s + 2 * d + 1];
                    pl[d] = __builtin_amdgcn_perm(b, a, 0x07040300u);   // a.b0 | 0x04 << 8 | b.b0 << 16 | 0x04 << 24
                    ph[d] = __builtin_amdgcn_perm(b, a, 0x07050301u);   // a.b1 | 0x04 << 8 | b.b1 << 16 | 0x04 << 24
                }
                lo[t][s] = make_uint4(pl[0], pl[1], pl[2], pl[3]);
                hi[t][s] = make_uint4(ph[0], ph[1], ph[2], ph[3]);
            }
        const int wvec = w - (w & 3);
        const bool tail = wvec < w && c0 <= wvec && wvec < c0 + 32;   // this strip holds the columns of the scalar tail (wave-uniform)
        const int gT = (wvec - c0) >> 3, hT = ((wvec - c0) >> 2) & 1; // ... in register group gT of the lanes with hh == hT
#pragma unroll
        for (int t = 0; t < 2; t++) {
            // result tile t: output rows 32 t + n (centre = raw row 32 t + n + 3); tile 0 draws on both row tiles, tile 1 on
            // row tile 1 only, with the same band operands (the band only depends on raw row - output row)
            v16f zl = zinit, zh = zinit;
#pragma unroll
            for (int q = 0; q < (t == 0 ? 4 : 2); q++) {
                const int rt = t == 0 ? (q >> 1) : 1, s = q & 1;
                const v8h bw = __builtin_bit_cast(v8h, bq[2 + q]);
                zl = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(v8h, lo[rt][s]), bw, zl, 0, 0, 0);
                zh = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(v8h, hi[rt][s]), bw, zh, 0, 0, 0);
            }
            // lane (output row 32 t + n, hh), register 4 g + e: column c0 + 8 g + 4 hh + e
#pragma unroll
            for (int gg = 0; gg < 4; gg++) {
                uint32_t packed = 0;
#pragma unroll
                for (int e = 0; e < 4; e++)
                    packed = __builtin_amdgcn_cvt_pk_u8_f32(fmaf(zh[4 * gg + e], 256.0f, zl[4 * gg + e]), e, packed);   // S * 2^-16, exact
                *reinterpret_cast<uint32_t *>(&s_out0[(32 * t + n) * BM_OP + 32 * wv + 8 * gg + 4 * hh]) = packed;
            }
            if (tail) {
                // scalar tail of the reference's column filter, (S + 32768) >> 16 = floor(S * 2^-16 + 0.5): the four columns
                // from wvec on are written once more by the lanes that hold them (a real branch: one strip per level row)
#pragma unroll
                for (int gg = 0; gg < 4; gg++)
                    if (gg == gT) {
                        asm volatile("" ::: "memory");
                        uint32_t packed = 0;
#pragma unroll
                        for (int e = 0; e < 4; e++)
                            packed = __builtin_amdgcn_cvt_pk_u8_f32(floorf(fmaf(zh[4 * gg + e], 256.0f, zl[4 * gg + e]) + 0.5f), e, packed);
                        if (hh == hT) *reinterpret_cast<uint32_t *>(&s_out0[(32 * t + n) * BM_OP + 32 * wv + 8 * gg + 4 * hh]) = packed;
                    }
            }
        }
    };

    // stage 4: s_out -> blurred level (16-byte stores; rows are padded to 64 bytes: a chunk that starts inside the image may run
    // into the padding)
    auto store = [&](const BlurGeom &cur) {
        const int w = cur.w, h = cur.h, x0 = cur.x0, y0 = cur.y0;
        for (int i = tid; i < BM_H * (BM_W / 16); i += 256) {
            const int r = i >> 3, c = i & 7;
            const int y = y0 + r, x = x0 + (c << 4);
            if (y < h && x < w)
                *reinterpret_cast<uint4 *>(cur.dst + (size_t)y * cur.dstride + x) = *reinterpret_cast<const uint4 *>(&s_out0[r * BM_OP + (c << 4)]);
        }
    };
    uint8_t *const raw0 = smem, *const raw1 = smem + BM_IN * BM_PITCH;
    edge_patch(raw0, g0);
    compute(raw0, g0);
    // tile 1 has had a tile's computation to arrive; no store has been issued yet, so this wait covers loads only
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    BM_RAW_BARRIER();                     // s_out and (all waves' transfers of) tile 1 are complete
    store(g0);
    if (!two) return;
    BM_RAW_BARRIER();                     // s_out is free again
    edge_patch(raw1, g1);
    compute(raw1, g1);
    BM_RAW_BARRIER();
    store(g1);
#undef S_RAW
}

// =====================================================================================================================
// r05 review item 5: the blurred twin of a level from the kernel that BUILDS the level.  One workgroup per 128 x 58 tile of
// level l (the blur's tile): the source window of level l - 1 travels to LDS (as in k_resize), the 64 x 136 pixels of level l
// that the tile's blur reads (3 halo rows above and below, a 4-pixel group left and right) are interpolated from it into the raw
// image k_blur stages -- k_resize's arithmetic, item by item -- the tile's own 128 x 58 of them are stored as level l, and the
// row / column passes of k_blur run on the raw image.  Level l is then written once and read only by the next level's
// launch, FAST and the orientation; its blur costs no read at all.  One launch per level (the chain of k_resize); level 0 keeps
// k_blur.  ORBHIP_FUSE_BLUR=1 (experiment: see DESIGN section 7 for the measurement).
// =====================================================================================================================
#define RB_SROWS 82                // staged source rows (64 output rows x 1.2 + margin: 81 at level 4 of 640 x 480)
#define RB_SPITCH (13 * 16)        // 13 chunks: 136 output columns x 1.2 + margin + alignment
#define RB_NG 34                   // output column groups of 4 pixels: 4 halo + 128 + 4 halo
typedef unsigned short rb_us2 __attribute__((ext_vector_type(2)));

__global__ __launch_bounds__(256) void k_resize_blur(const uint8_t *__restrict__ src, int sstride, unsigned long long sframe, int sw,
                                                     int sh, uint8_t *__restrict__ dst, int dw, int dh, int dstride,
                                                     unsigned long long dframe, uint8_t *__restrict__ bdst, int bstride,
                                                     unsigned long long bframe, const int4 *__restrict__ ytab,
                                                     const int4 *__restrict__ gtab, float winx, float winy,
                                                     const uint4 *__restrict__ bands, int xcdMap, int ntiles)
{
    // [raw image of the blur 64 x 208][source window 81 x 208, later the blur's output image 64 x 144][row taps 64][group taps 34 x 3]
    __shared__ __align__(16) uint8_t smem[BM_IN * BM_PITCH + (RB_SROWS + 1) * RB_SPITCH + 64 * 16 + RB_NG * 3 * 16];
    uint8_t *const raw = smem;
    uint8_t *const s_src = smem + BM_IN * BM_PITCH;
    uint8_t *const s_out0 = s_src;
    const int4 *const s_ytab = reinterpret_cast<const int4 *>(smem + BM_IN * BM_PITCH + (RB_SROWS + 1) * RB_SPITCH);
    const int4 *const s_gtab = s_ytab + 64;
    const int t = xcd_tile(xcdMap), frame = blockIdx.y;
    if (t >= ntiles) return;
    const int tid = threadIdx.x, lane = tid & 63, n = lane & 31, hh = lane >> 5;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int tilesX = (dw + BM_W - 1) / BM_W;
    const int ty = t / tilesX, tx = t - ty * tilesX;
    const int x0 = tx * BM_W, y0 = ty * BM_H;
    const uint8_t *S = src + (size_t)frame * sframe;
    uint8_t *D = dst + (size_t)frame * dframe;

    uint4 bq[6];
#pragma unroll
    for (int q = 0; q < 6; q++) bq[q] = bands[q * 64 + lane];
    asm volatile("" ::: "memory");

    // the pixels of level l to interpolate: rows [ryA, ryB], column groups from gxA on (inside the image)
    const int ryA = max(y0 - 3, 0), ryB = min(y0 + BM_IN - 4, dh - 1);
    const int gxA = max(x0 - 4, 0), gxB = min(x0 + BM_W + 4, (dw + 3) & ~3);
    const int ngrp = (gxB - gxA) >> 2, nrowsOut = ryB - ryA + 1;
    // their source window (k_resize's window hint; launch_resize_blur checks it against the tap tables)
    const int sxmin = max((int)((float)gxA * winx) - 1, 0), sxmax = min((int)((float)min(gxB, dw) * winx) + 1, sw - 1);
    const int symin = max((int)((float)ryA * winy) - 1, 0), symax = min((int)((float)(ryB + 1) * winy) + 1, sh - 1);
    const int XA = sxmin & ~15;
    const int nch = ((sxmax - XA) >> 4) + 1, nrows = symax - symin + 1;
    {
        // LDS-DMA, four whole rows of 13 chunks per wave transfer (blur_dma)
        const int rl = (lane * 5) >> 6, ch = lane - 13 * rl;
        const uint8_t *sp = S + (size_t)(symin + rl) * sstride + XA + (ch << 4);
        const uint32_t ldsBase = (uint32_t)(uintptr_t)s_src;
        for (int rb = wv * 4; rb < nrows; rb += 16)
            if (lane < 52 && ch < nch && rb + rl < nrows)
                glds16(sp + (size_t)rb * sstride, __builtin_amdgcn_readfirstlane(ldsBase + (uint32_t)(rb * RB_SPITCH)));
        if (wv == 1) {
            glds16(ytab + min(ryA + lane, dh - 1), (uint32_t)(uintptr_t)s_ytab);
        } else if (wv >= 2) {
            const int q = 64 * (wv - 2) + lane, grp = q / 3, part = q - 3 * grp;
            if (q < RB_NG * 3) glds16(gtab + 3 * (min(gxA + 4 * grp, dw - 1) >> 2) + part, (uint32_t)(uintptr_t)(s_gtab + 64 * (wv - 2)));
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    BM_RAW_BARRIER();

#define S_RAW(r, c) raw[(r) * BM_PITCH + (c)]
    // ---- level l: k_resize's grouped path, item = (row, group of 4 columns) ----
    for (int i = tid; i < 64 * RB_NG; i += 256) {
        const int row = (int)(((unsigned)i * 1928u) >> 16), g = i - row * RB_NG;   // i / 34 for i < 2176
        if (row >= nrowsOut || g >= ngrp) continue;
        const int y = ryA + row, gx = gxA + 4 * g;
        const int4 yt = s_ytab[row];
        const int4 g0 = s_gtab[3 * g], gsel = s_gtab[3 * g + 1], gw = s_gtab[3 * g + 2];
        const int bcol = g0.x - XA, wb = bcol & ~3, shf = bcol & 3;
        const uint32_t sel[4] = {(uint32_t)gsel.x, (uint32_t)gsel.y, (uint32_t)gsel.z, (uint32_t)gsel.w};
        const uint32_t wt[4] = {(uint32_t)gw.x, (uint32_t)gw.y, (uint32_t)gw.z, (uint32_t)gw.w};
        const uint32_t b0s = ((uint32_t)yt.z & 0xFFFu) << 12, b1s = ((uint32_t)yt.w & 0xFFFu) << 12;
        const uint32_t *q0 = reinterpret_cast<const uint32_t *>(s_src + (yt.x - symin) * RB_SPITCH + wb);
        const uint32_t *q1 = reinterpret_cast<const uint32_t *>(s_src + (yt.y - symin) * RB_SPITCH + wb);
        const uint32_t a0 = q0[0], a1 = q0[1], a2 = q0[2], c0 = q1[0], c1 = q1[1], c2 = q1[2];
        const uint32_t lo0 = __builtin_amdgcn_alignbyte(a1, a0, shf), hi0 = __builtin_amdgcn_alignbyte(a2, a1, shf);
        const uint32_t lo1 = __builtin_amdgcn_alignbyte(c1, c0, shf), hi1 = __builtin_amdgcn_alignbyte(c2, c1, shf);
        uint32_t v[4];
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const rb_us2 p0 = __builtin_bit_cast(rb_us2, __builtin_amdgcn_perm(hi0, lo0, sel[k]));
            const rb_us2 p1 = __builtin_bit_cast(rb_us2, __builtin_amdgcn_perm(hi1, lo1, sel[k]));
            const rb_us2 w2 = __builtin_bit_cast(rb_us2, wt[k]);
            const uint32_t r0 = __builtin_amdgcn_udot2(p0, w2, 0u, false);
            const uint32_t r1 = __builtin_amdgcn_udot2(p1, w2, 0u, false);
            v[k] = (__umulhi(b0s, r0 & 0x7FFFF0u) + __umulhi(b1s, r1 & 0x7FFFF0u) + 2u) >> 2;   // <= 255
        }
        const uint32_t packed = v[0] | (v[1] << 8) | (v[2] << 16) | (v[3] << 24);
        *reinterpret_cast<uint32_t *>(&S_RAW(y - (y0 - 3), 16 + gx - x0)) = packed;
        if (y >= y0 && y < y0 + BM_H && gx >= x0 && gx < x0 + BM_W) {   // the tile's own pixels: level l
            uint8_t *o = D + (size_t)y * dstride + gx;
            if (gx + 3 < dw) {
                *reinterpret_cast<uint32_t *>(o) = packed;
            } else {
                for (int k = 0; k < 4 && gx + k < dw; k++) o[k] = (uint8_t)(packed >> (8 * k));
            }
        }
    }
    BM_RAW_BARRIER();
    // reflected rows above the image (y = -1, -2, -3 <- 1, 2, 3) and below it (y = h + k <- h - 2 - k): block-uniform
    if (y0 == 0 || y0 + BM_IN - 4 >= dh) {
        if (y0 == 0)
            for (int i = tid; i < 3 * BM_CW; i += 256) {
                const int r = i / BM_CW, c = i - r * BM_CW;
                reinterpret_cast<uint4 *>(raw + r * BM_PITCH)[c] = reinterpret_cast<const uint4 *>(raw + (6 - r) * BM_PITCH)[c];
            }
        if (y0 + BM_IN - 4 >= dh)
            for (int i = tid; i < 3 * BM_CW; i += 256) {
                const int k = i / BM_CW, c = i - k * BM_CW;
                const int r = dh + k - (y0 - 3), rs = dh - 2 - k - (y0 - 3);
                if (r < BM_IN && rs >= 0) reinterpret_cast<uint4 *>(raw + r * BM_PITCH)[c] = reinterpret_cast<const uint4 *>(raw + rs * BM_PITCH)[c];
            }
        BM_RAW_BARRIER();
    }
    {
        // reflected halo columns at the image edges (k_blur's edge_patch)
        const bool edgeL = x0 == 0, edgeR = x0 + BM_W + 3 > dw;
        if (edgeL || edgeR) {
            if (edgeL)
                for (int i = tid; i < BM_IN * 3; i += 256) {
                    const int r = i / 3, k = i - r * 3 + 1;
                    S_RAW(r, 16 - k) = S_RAW(r, 16 + k);
                }
            if (edgeR)
                for (int i = tid; i < BM_IN * 3; i += 256) {
                    const int r = i / 3, k = i - r * 3;
                    S_RAW(r, 16 + (dw + k - x0)) = S_RAW(r, 16 + (dw - 2 - k - x0));
                }
            BM_RAW_BARRIER();
        }
    }
    // ---- k_blur's row and column pass on the raw image (stages 2 and 3 of k_blur, same operands) ----
    {
        const int w = dw;
        const int c0 = x0 + 32 * wv;
        if (c0 < w) {
            v16i hinit;
            v16f zinit;
#pragma unroll
            for (int i = 0; i < 16; i++) {
                hinit[i] = 128 * 257 + 0x04000000;
                zinit[i] = -(float)(257 * 1024) / 65536.0f;
            }
            v16i H[2];
#pragma unroll
            for (int tt = 0; tt < 2; tt++) {
                uint4 a0 = *reinterpret_cast<const uint4 *>(&S_RAW(32 * tt + n, 32 * wv + 16 * hh));
                uint4 a1 = *reinterpret_cast<const uint4 *>(&S_RAW(32 * tt + n, 32 * wv + 32 + 16 * hh));
                a0.x ^= 0x80808080u; a0.y ^= 0x80808080u; a0.z ^= 0x80808080u; a0.w ^= 0x80808080u;
                a1.x ^= 0x80808080u; a1.y ^= 0x80808080u; a1.z ^= 0x80808080u; a1.w ^= 0x80808080u;
                H[tt] = __builtin_amdgcn_mfma_i32_32x32x32_i8(__builtin_bit_cast(v4i, a0), __builtin_bit_cast(v4i, bq[0]), hinit, 0, 0, 0);
                H[tt] = __builtin_amdgcn_mfma_i32_32x32x32_i8(__builtin_bit_cast(v4i, a1), __builtin_bit_cast(v4i, bq[1]), H[tt], 0, 0, 0);
            }
            uint4 lo[2][2], hi[2][2];
#pragma unroll
            for (int tt = 0; tt < 2; tt++)
#pragma unroll
                for (int s2 = 0; s2 < 2; s2++) {
                    uint32_t pl[4], ph[4];
#pragma unroll
                    for (int d = 0; d < 4; d++) {
                        const uint32_t a = (uint32_t)H[tt][8 * s2 + 2 * d], b = (uint32_t)H[tt][8 * s2 + 2 * d + 1];
                        pl[d] = __builtin_amdgcn_perm(b, a, 0x07040300u);
                        ph[d] = __builtin_amdgcn_perm(b, a, 0x07050301u);
                    }
                    lo[tt][s2] = make_uint4(pl[0], pl[1], pl[2], pl[3]);
                    hi[tt][s2] = make_uint4(ph[0], ph[1], ph[2], ph[3]);
                }
            const int wvec = w - (w & 3);
            const bool tail = wvec < w && c0 <= wvec && wvec < c0 + 32;
            const int gT = (wvec - c0) >> 3, hT = ((wvec - c0) >> 2) & 1;
#pragma unroll
            for (int tt = 0; tt < 2; tt++) {
                v16f zl = zinit, zh = zinit;
#pragma unroll
                for (int q = 0; q < (tt == 0 ? 4 : 2); q++) {
                    const int rt = tt == 0 ? (q >> 1) : 1, s2 = q & 1;
                    const v8h bw = __builtin_bit_cast(v8h, bq[2 + q]);
                    zl = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(v8h, lo[rt][s2]), bw, zl, 0, 0, 0);
                    zh = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(v8h, hi[rt][s2]), bw, zh, 0, 0, 0);
                }
#pragma unroll
                for (int gg = 0; gg < 4; gg++) {
                    uint32_t packed = 0;
#pragma unroll
                    for (int e = 0; e < 4; e++)
                        packed = __builtin_amdgcn_cvt_pk_u8_f32(fmaf(zh[4 * gg + e], 256.0f, zl[4 * gg + e]), e, packed);
                    // (the output image lies over the source window, which no wave reads after the barrier behind the interpolation)
                    *reinterpret_cast<uint32_t *>(&s_out0[(32 * tt + n) * BM_OP + 32 * wv + 8 * gg + 4 * hh]) = packed;
                }
                if (tail) {
#pragma unroll
                    for (int gg = 0; gg < 4; gg++)
                        if (gg == gT) {
                            asm volatile("" ::: "memory");
                            uint32_t packed = 0;
#pragma unroll
                            for (int e = 0; e < 4; e++)
                                packed = __builtin_amdgcn_cvt_pk_u8_f32(floorf(fmaf(zh[4 * gg + e], 256.0f, zl[4 * gg + e]) + 0.5f), e, packed);
                            if (hh == hT) *reinterpret_cast<uint32_t *>(&s_out0[(32 * tt + n) * BM_OP + 32 * wv + 8 * gg + 4 * hh]) = packed;
                        }
                }
            }
        }
    }
    BM_RAW_BARRIER();
    {
        uint8_t *BD = bdst + (size_t)frame * bframe;
        for (int i = tid; i < BM_H * (BM_W / 16); i += 256) {
            const int r = i >> 3, c = i & 7;
            const int y = y0 + r, x = x0 + (c << 4);
            if (y < dh && x < dw)
                *reinterpret_cast<uint4 *>(BD + (size_t)y * bstride + x) = *reinterpret_cast<const uint4 *>(&s_out0[r * BM_OP + (c << 4)]);
        }
    }
#undef S_RAW
}

// Does every tile's computed source window hold the taps of the pixels it interpolates, and fit the staging area?
bool resize_blur_fits(const int32_t *xt, const int32_t *yt, int sw, int sh, int dw, int dh)
{
    const float winx = (float)sw / (float)dw, winy = (float)sh / (float)dh;
    for (int x0 = 0; x0 < dw; x0 += BM_W) {
        const int gxA = std::max(x0 - 4, 0), gxB = std::min(x0 + BM_W + 4, (dw + 3) & ~3);
        const int c1 = std::min(gxB, dw) - 1;
        const int lo = std::max((int)((float)gxA * winx) - 1, 0), hi = std::min((int)((float)std::min(gxB, dw) * winx) + 1, sw - 1);
        if (lo > (xt[2 * gxA] & 0xFFFF) || hi < (int)((uint32_t)xt[2 * c1] >> 16)) return false;
        if (((hi - (lo & ~15)) >> 4) + 1 > 13) return false;
        for (int y0 = 0; y0 < dh; y0 += BM_H) {
            const int ryA = std::max(y0 - 3, 0), ryB = std::min(y0 + BM_IN - 4, dh - 1);
            const int rlo = std::max((int)((float)ryA * winy) - 1, 0), rhi = std::min((int)((float)(ryB + 1) * winy) + 1, sh - 1);
            if (rlo > yt[4 * ryA] || rhi < yt[4 * ryB + 1]) return false;
            if (rhi - rlo + 1 > RB_SROWS) return false;
        }
    }
    return true;
}

void launch_resize_blur(hipStream_t s, const uint8_t *src, int sw, int sh, int sstride, size_t sframe, uint8_t *dst, int dw, int dh,
                        int dstride, size_t dframe, uint8_t *bdst, int bstride, size_t bframe, const int32_t *ytab, const int32_t *gtab,
                        const uint32_t *bands, int B)
{
    const int ntiles = ((dw + BM_W - 1) / BM_W) * ((dh + BM_H - 1) / BM_H);
    dim3 grid(orb_xcd_grid(ntiles, 1), B, 1), block(256, 1, 1);
    hipLaunchKernelGGL(k_resize_blur, grid, block, 0, s, src, sstride, (unsigned long long)sframe, sw, sh, dst, dw, dh, dstride,
                       (unsigned long long)dframe, bdst, bstride, (unsigned long long)bframe, reinterpret_cast<const int4 *>(ytab),
                       reinterpret_cast<const int4 *>(gtab), (float)sw / (float)dw, (float)sh / (float)dh,
                       reinterpret_cast<const uint4 *>(bands), orb_xcd_arg(1), ntiles);
}

// cv::getGaussianKernel(7, 2, CV_32F) converted to CV_32S with scale 256 (filter.cpp
// createSeparableLinearFilter, 8U fixed-point branch).
static void gaussian_taps(int k[4])
{
    double t[7], sum = 0;
    float cf[7];
    const double scale2X = -0.5 / (2.0 * 2.0);
    for (int i = 0; i < 7; i++) {
        const double x = i - 3.0;
        t[i] = exp(scale2X * x * x);
        cf[i] = (float)t[i];
        sum += cf[i];
    }
    sum = 1. / sum;
    for (int i = 0; i < 4; i++) {
        cf[i] = (float)(cf[i] * sum);
        k[i] = (int)lrint((double)(cf[i] * 256.f));
    }
}

// binary16 pattern of tap * 256 (tap < 64: exact)
static uint16_t half_of_tap256(int tap)
{
    if (tap == 0) return 0;
    int e = 0;
    while ((tap >> (e + 1)) != 0) e++;                // tap = 1.m * 2^e
    const int exp = e + 8 + 15;                       // value tap * 2^8
    const int mant = ((tap << (10 - e)) & 0x3FF);
    return (uint16_t)((exp << 10) | mant);
}

// The band operands, per lane (n = lane & 31, hh = lane >> 5), 16 bytes each: [q][lane], q = 0, 1: k-steps of the row pass
// (int8 taps; slot j of k-step s is input column c0 - 16 + 32 s + 16 hh + j, output column c0 + n); q = 2..5: k-steps of
// the column pass (binary16 tap * 2^8; slot j of k-step (rt, s) is raw row 32 rt + tile_row(8 s + j, hh), output row n).
void blur_band_table(uint32_t out[6 * 64 * 4])
{
    int k4[4];
    gaussian_taps(k4);
    const int tap[7] = {k4[0], k4[1], k4[2], k4[3], k4[2], k4[1], k4[0]};
    uint8_t *p = reinterpret_cast<uint8_t *>(out);
    for (int lane = 0; lane < 64; lane++) {
        const int n = lane & 31, hh = lane >> 5;
        for (int s = 0; s < 2; s++)
            for (int j = 0; j < 16; j++) {
                const int t = (-16 + 32 * s + 16 * hh + j) - n + 3;
                p[((size_t)s * 64 + lane) * 16 + j] = (uint8_t)((t >= 0 && t < 7) ? tap[t] : 0);
            }
        for (int q = 0; q < 4; q++)
            for (int j = 0; j < 8; j++) {
                const int rt = q >> 1, s = q & 1;
                const int t = 32 * rt + tile_row(8 * s + j, hh) - n;
                const uint16_t hv = (t >= 0 && t < 7) ? half_of_tap256(tap[t]) : 0;
                p[((size_t)(2 + q) * 64 + lane) * 16 + 2 * j] = (uint8_t)(hv & 0xFF);
                p[((size_t)(2 + q) * 64 + lane) * 16 + 2 * j + 1] = (uint8_t)(hv >> 8);
            }
    }
}

void launch_blur(hipStream_t s, const OrbLevels &G, const uint8_t *lvl0, int stride0, size_t frame0,
                 const uint8_t *pyr, size_t pyrFrame, uint8_t *blur, size_t blurFrame,
                 const BlurTile *tiles, int ntiles, const uint32_t *bands, int B)
{
    if (ntiles <= 0) return;
    orb_path(ORB_PATH_BLUR);
    dim3 grid(orb_xcd_grid((ntiles + BM_TPW - 1) / BM_TPW, 2), B, 1), block(256, 1, 1);
    hipLaunchKernelGGL(k_blur, grid, block, 0, s, G, lvl0, stride0, (unsigned long long)frame0, pyr,
                       (unsigned long long)pyrFrame, blur, (unsigned long long)blurFrame, tiles,
                       reinterpret_cast<const uint4 *>(bands), orb_xcd_arg(2), ntiles);
}
