// k_pyramid.hip -- E2: one pyramid level from the previous one, bit-exact cv::resize INTER_LINEAR
// 8UC1 (ref call site: src/ORBextractor.cc:1141).  The 19-px border of the reference
// (copyMakeBorder :1143-1149) is never read by later stages and is not produced.
//
// One 256-thread workgroup per 128x32 output tile.  The source rows/columns the tile touches
// (about 156 x 40 pixels at scale 1.2) are staged into LDS with 16-byte row-coalesced loads; the window
// is derived from the scale factors so that the staging loads do not wait for a table read (measured: the
// kernel is bound by that dependent-load latency and by HBM, not by arithmetic).  A thread then produces 4
// horizontally adjacent output pixels from LDS and stores them as one dword: the taps of the four pixels
// are byte selectors into one realigned 8-byte window (v_perm), the horizontal sums are v_dot2, the
// vertical products mul_hi -- the tables (orb_build_resize_tables, orb_build_resize_groups) are built on the
// host.  Bound: HBM (reads 1.44 px and writes 1 px per output pixel; with halo ~1.8 px).
// Measured and dropped: building two levels per launch (level l+1 kept in LDS, written once): -3 % before the
// window hint, slower than one level per launch after it.
#include "orbhip_internal.h"

#include <algorithm>

#define RZ_TW 128
typedef unsigned short us2 __attribute__((ext_vector_type(2)));
// RZ_TH = rows of a tile (RZ_TH / 8 per thread): 32 for batches, 8 for a single frame or two (4x the workgroups)
#define RZ_MAXCH 16    // 16-byte chunks per staged source row (source span <= 240 px + alignment)
#define RZ_MAXROWS 44  // staged source rows

// 16 bytes per lane from global memory straight into LDS at (ldsAddr + 16 * lane); M0 carries the LDS address and is restored
// (assembly: the builtin makes hipcc wait vmcnt(0) at every LDS access that might alias).
__device__ __forceinline__ void rz_glds16(const void *gsrc, uint32_t ldsAddr)
{
    uint32_t keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(gsrc), "s"(ldsAddr)
                 : "memory");
}

template <int RZ_TH>
__global__ __launch_bounds__(256) void k_resize(const uint8_t *__restrict__ src, int sstride,
                                                unsigned long long sframe, uint8_t *__restrict__ dst,
                                                int dw, int dh, int dstride, unsigned long long dframe,
                                                const int2 *__restrict__ xtab,
                                                const int4 *__restrict__ ytab, const int4 *__restrict__ gtab, int sw, int sh,
                                                float winx, float winy, int xcdMap)
{
    __shared__ __align__(16) uint8_t s_src[RZ_MAXROWS + 1][RZ_MAXCH * 16];   // + 1: the 12-byte window reads of the last row
    const int tid = threadIdx.x;
    // grid = (tiles padded to a multiple of 8, frames); orbhip_internal.h, xcd_tile
    const int tilesX = (dw + RZ_TW - 1) / RZ_TW, tilesY = (dh + RZ_TH - 1) / RZ_TH;
    const int t = xcd_tile(xcdMap), frame = blockIdx.y;
    if (t >= tilesX * tilesY) return;
    const int by = t / tilesX, bx = t - by * tilesX;
    const int ox0 = bx * RZ_TW, oy0 = by * RZ_TH;
    const int ox1 = min(ox0 + RZ_TW, dw) - 1, oy1 = min(oy0 + RZ_TH, dh) - 1;   // inclusive
    const uint8_t *S = src + (size_t)frame * sframe;
    uint8_t *D = dst + (size_t)frame * dframe;

    // source window of the tile.  With a window hint (scale factors, winx / winy > 0) the bounds are computed, with a
    // margin of one pixel each side, instead of being read from the tables: the staging loads below then depend on
    // nothing that has to come from memory first (taps lie in [floor(o * scale), floor((o + 1) * scale)] for scale >= 1;
    // the host checks the hint against the tables before it passes one).
    int sxmin, sxmax, symin, symax;
    if (winx > 0.f) {
        sxmin = max((int)((float)ox0 * winx) - 1, 0);
        sxmax = min((int)((float)(ox1 + 1) * winx) + 1, sw - 1);
        symin = max((int)((float)oy0 * winy) - 1, 0);
        symax = min((int)((float)(oy1 + 1) * winy) + 1, sh - 1);
    } else {
        sxmin = xtab[ox0].x & 0xFFFF;
        sxmax = (unsigned)xtab[ox1].x >> 16;
        symin = ytab[oy0].x;
        symax = ytab[oy1].y;
    }
    const int XA = sxmin & ~15;
    const int nch = ((sxmax - XA) >> 4) + 1;
    const int nrows = symax - symin + 1;
    if (nch <= RZ_MAXCH && nrows <= RZ_MAXROWS && nrows * nch <= 512) {
        // LDS-DMA (global_load_lds_dwordx4): a wave transfer writes 64 x 16 bytes to consecutive LDS addresses = four staged
        // rows of 16 chunks, so lane = (row of the four) * 16 + chunk; no registers, no division, and the thread's tap
        // loads below are in flight beside it.  Waited for (vmcnt) before the barrier: the compiler does not know of them.
        const int lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
        const int rl = lane >> 4, ch = lane & 15;
        const uint8_t *sp = S + (size_t)(symin + rl) * sstride + XA + (ch << 4);
        const uint32_t ldsBase = (uint32_t)(uintptr_t)&s_src[0][0];
        for (int rb = wv * 4; rb < nrows; rb += 16)
            if (ch < nch && rb + rl < nrows) rz_glds16(sp + (size_t)rb * sstride, ldsBase + (uint32_t)(rb * (RZ_MAXCH * 16)));
    }
    // this thread's taps are requested before the barrier so that their latency overlaps the staging
    constexpr int NR = RZ_TH / 8;                 // output rows per thread (rows dy, dy + 8, ...)
    const int gx = ox0 + ((tid & 31) << 2);
    const int dy0 = oy0 + (tid >> 5);
    const bool colLive = gx < dw;
    int4 yt[NR];
    int2 xt[4] = {make_int2(0, 0), make_int2(0, 0), make_int2(0, 0), make_int2(0, 0)};
    const bool staged = nch <= RZ_MAXCH && nrows <= RZ_MAXROWS && nrows * nch <= 512;
    const bool grouped = staged && gtab != nullptr;
    // batches, grouped taps: the tile's 32 row entries and 32 column-group entries (2 KB) travel to LDS once per workgroup by
    // LDS-DMA (three wave transfers) instead of seven 16-byte gathers per thread, four times over (every wave wants the same
    // column groups): the tables were as much L1 traffic as the pixels
    constexpr bool TABDMA = RZ_TH == 32;
    __shared__ __align__(16) int4 s_ytab[TABDMA ? 32 : 1];
    __shared__ __align__(16) int4 s_gtab[TABDMA ? 96 : 1];
    int4 g0 = make_int4(0, 0, 0, 0), gsel = g0, gw = g0;
    if (TABDMA && grouped) {
        const int lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
        if (wv == 1) {
            if (lane < 32) rz_glds16(ytab + min(oy0 + lane, dh - 1), (uint32_t)(uintptr_t)&s_ytab[0]);
        } else if (wv >= 2) {
            const int q = 64 * (wv - 2) + lane;                 // 16-byte chunk of the 32 x 3 group entries
            const int grp = q / 3, part = q - 3 * grp;
            if (q < 96) rz_glds16(gtab + 3 * (min(ox0 + 4 * grp, dw - 1) >> 2) + part, (uint32_t)(uintptr_t)&s_gtab[64 * (wv - 2)]);
        }
    } else {
#pragma unroll
        for (int j = 0; j < NR; j++) yt[j] = ytab[min(dy0 + 8 * j, dh - 1)];
        if (grouped) {
            const int4 *gp = gtab + 3 * (min(gx, dw - 1) >> 2);
            g0 = gp[0];
            gsel = gp[1];
            gw = gp[2];
        } else {
#pragma unroll
            for (int k = 0; k < 4; k++) xt[k] = xtab[min(gx + k, dw - 1)];
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (!colLive) return;
    if (TABDMA && grouped) {
#pragma unroll
        for (int j = 0; j < NR; j++) yt[j] = s_ytab[(tid >> 5) + 8 * j];
        const int4 *gp = s_gtab + 3 * (tid & 31);
        g0 = gp[0];
        gsel = gp[1];
        gw = gp[2];
    }
    if (grouped) {
        // Fast path: the 8 source bytes from the group's base column on (three aligned LDS dwords, realigned) hold all
        // eight taps of a row; a pixel's tap pair comes out with one v_perm, the horizontal sum is one v_dot2 with
        // the weight pair, and (b * (r >> 4)) >> 16 is the high half of a 24-bit product (below).  Same integers as the
        // generic path below.
        const int bcol = g0.x - XA, wb = bcol & ~3, sh = bcol & 3;
        const uint32_t sel[4] = {(uint32_t)gsel.x, (uint32_t)gsel.y, (uint32_t)gsel.z, (uint32_t)gsel.w};
        const uint32_t wt[4] = {(uint32_t)gw.x, (uint32_t)gw.y, (uint32_t)gw.z, (uint32_t)gw.w};
#pragma unroll
        for (int j = 0; j < NR; j++) {
            const int dy = dy0 + 8 * j;
            if (dy >= dh) break;
            // (b * (r >> 4)) >> 16 = high word of (b << 12) * (r & ~15): both factors below 2^24 (b <= 2048, r <= 255 * 2048), so
            // that it is the full-rate v_mul_hi_u32_u24 and the shift of r is gone (v_mul_hi_u32 issues at a quarter of the rate)
            const uint32_t b0s = ((uint32_t)yt[j].z & 0xFFFu) << 12, b1s = ((uint32_t)yt[j].w & 0xFFFu) << 12;
            const uint32_t *q0 = reinterpret_cast<const uint32_t *>(&s_src[yt[j].x - symin][wb]);
            const uint32_t *q1 = reinterpret_cast<const uint32_t *>(&s_src[yt[j].y - symin][wb]);
            const uint32_t a0 = q0[0], a1 = q0[1], a2 = q0[2], c0 = q1[0], c1 = q1[1], c2 = q1[2];
            const uint32_t lo0 = __builtin_amdgcn_alignbyte(a1, a0, sh), hi0 = __builtin_amdgcn_alignbyte(a2, a1, sh);
            const uint32_t lo1 = __builtin_amdgcn_alignbyte(c1, c0, sh), hi1 = __builtin_amdgcn_alignbyte(c2, c1, sh);
            uint32_t v[4];
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const us2 p0 = __builtin_bit_cast(us2, __builtin_amdgcn_perm(hi0, lo0, sel[k]));
                const us2 p1 = __builtin_bit_cast(us2, __builtin_amdgcn_perm(hi1, lo1, sel[k]));
                const us2 w2 = __builtin_bit_cast(us2, wt[k]);
                const uint32_t r0 = __builtin_amdgcn_udot2(p0, w2, 0u, false);
                const uint32_t r1 = __builtin_amdgcn_udot2(p1, w2, 0u, false);
                v[k] = (__umulhi(b0s, r0 & 0x7FFFF0u) + __umulhi(b1s, r1 & 0x7FFFF0u) + 2u) >> 2;   // <= 255
            }
            const uint32_t packed = v[0] | (v[1] << 8) | (v[2] << 16) | (v[3] << 24);
            uint8_t *o = D + (size_t)dy * dstride + gx;
            if (gx + 3 < dw) {
                *reinterpret_cast<uint32_t *>(o) = packed;
            } else {
                for (int k = 0; k < 4 && gx + k < dw; k++) o[k] = (uint8_t)(packed >> (8 * k));
            }
        }
        return;
    }
#pragma unroll
    for (int j = 0; j < NR; j++) {
        const int dy = dy0 + 8 * j;
        if (dy >= dh) break;
        const int b0 = yt[j].z, b1 = yt[j].w;
        uint32_t packed = 0;
        if (staged) {
            const uint8_t *L0 = &s_src[yt[j].x - symin][0];
            const uint8_t *L1 = &s_src[yt[j].y - symin][0];
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const int sx0 = (xt[k].x & 0xFFFF) - XA, sx1 = (int)((unsigned)xt[k].x >> 16) - XA;
                const int a0 = (short)(xt[k].y & 0xFFFF), a1 = xt[k].y >> 16;
                const int r0 = L0[sx0] * a0 + L0[sx1] * a1;
                const int r1 = L1[sx0] * a0 + L1[sx1] * a1;
                const int v = (((b0 * (r0 >> 4)) >> 16) + ((b1 * (r1 >> 4)) >> 16) + 2) >> 2;
                packed |= (uint32_t)(v & 0xFF) << (8 * k);
            }
        } else {
            // generic path (scale factors far from 1.2 whose source window exceeds the LDS tile): read global
            const uint8_t *S0 = S + (size_t)yt[j].x * sstride;
            const uint8_t *S1 = S + (size_t)yt[j].y * sstride;
            for (int k = 0; k < 4; k++) {
                const int sx0 = xt[k].x & 0xFFFF, sx1 = (unsigned)xt[k].x >> 16;
                const int a0 = (short)(xt[k].y & 0xFFFF), a1 = xt[k].y >> 16;
                const int r0 = S0[sx0] * a0 + S0[sx1] * a1;
                const int r1 = S1[sx0] * a0 + S1[sx1] * a1;
                const int v = (((b0 * (r0 >> 4)) >> 16) + ((b1 * (r1 >> 4)) >> 16) + 2) >> 2;
                packed |= (uint32_t)(v & 0xFF) << (8 * k);
            }
        }
        uint8_t *o = D + (size_t)dy * dstride + gx;
        if (gx + 3 < dw) {
            *reinterpret_cast<uint32_t *>(o) = packed;
        } else {
            for (int k = 0; k < 4 && gx + k < dw; k++) o[k] = (uint8_t)(packed >> (8 * k));
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// r05: tiles FITTED to the level.  k_resize<32> lays 128 x 32 tiles over every level: 533 columns are five tiles of which the last
// holds 21, 257 columns three tiles of which the last holds ONE -- and a wave (two rows of 32 column groups) issues its
// instructions whether 32 or 5 of its groups exist: 17 % of the lanes of the batch's pyramid are such groups (33 % at level 5).
// Here the host picks, per level, the number of tile columns, the groups per tile (<= 32), from them the rows per pass
// (256 / groups) and the passes (3..6): a thread is (row of the pass, group) by one multiplication, everything else as in the
// grouped path of k_resize.  RF_* bound the staged window and the tables.
// ---------------------------------------------------------------------------------------------------------------------
#define RF_MAXROWS 62   // staged source rows: tiles of up to 48 rows
#define RF_MAXTH 48

__global__ __launch_bounds__(256) void k_resize_fit(const uint8_t *__restrict__ src, int sstride, unsigned long long sframe,
                                                    uint8_t *__restrict__ dst, int dw, int dh, int dstride,
                                                    unsigned long long dframe, const int4 *__restrict__ ytab,
                                                    const int4 *__restrict__ gtab, int sw, int sh, float winx, float winy,
                                                    int xcdMap, int ntx, int nty, int twg, int rpp, int npass, int rmagic)
{
    __shared__ __align__(16) uint8_t s_src[RF_MAXROWS + 1][RZ_MAXCH * 16];
    __shared__ __align__(16) int4 s_ytab[64];
    __shared__ __align__(16) int4 s_gtab[96];
    const int tid = threadIdx.x;
    const int t = xcd_tile(xcdMap), frame = blockIdx.y;
    if (t >= ntx * nty) return;
    const int by = t / ntx, bx = t - by * ntx;
    const int TH = rpp * npass;
    const int ox0 = bx * (twg << 2), oy0 = by * TH;
    const int ox1 = min(ox0 + (twg << 2), dw) - 1, oy1 = min(oy0 + TH, dh) - 1;   // inclusive
    const uint8_t *S = src + (size_t)frame * sframe;
    uint8_t *D = dst + (size_t)frame * dframe;
    const int sxmin = max((int)((float)ox0 * winx) - 1, 0), sxmax = min((int)((float)(ox1 + 1) * winx) + 1, sw - 1);
    const int symin = max((int)((float)oy0 * winy) - 1, 0), symax = min((int)((float)(oy1 + 1) * winy) + 1, sh - 1);
    const int XA = sxmin & ~15;
    const int nch = ((sxmax - XA) >> 4) + 1, nrows = symax - symin + 1;
    {
        const int lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
        const int rl = lane >> 4, ch = lane & 15;
        const uint8_t *sp = S + (size_t)(symin + rl) * sstride + XA + (ch << 4);
        const uint32_t ldsBase = (uint32_t)(uintptr_t)&s_src[0][0];
        for (int rb = wv * 4; rb < nrows; rb += 16)
            if (ch < nch && rb + rl < nrows) rz_glds16(sp + (size_t)rb * sstride, ldsBase + (uint32_t)(rb * (RZ_MAXCH * 16)));
        if (wv == 1) {
            if (lane < TH) rz_glds16(ytab + min(oy0 + lane, dh - 1), (uint32_t)(uintptr_t)&s_ytab[0]);
        } else if (wv >= 2) {
            const int q = 64 * (wv - 2) + lane;                 // 16-byte chunk of the twg x 3 group entries
            const int grp = q / 3, part = q - 3 * grp;
            if (q < 3 * twg) rz_glds16(gtab + 3 * (min(ox0 + 4 * grp, dw - 1) >> 2) + part, (uint32_t)(uintptr_t)&s_gtab[64 * (wv - 2)]);
        }
    }
    const int r = (int)(((unsigned)tid * (unsigned)rmagic) >> 16), g = tid - r * twg;   // tid / twg, tid % twg
    const int gx = ox0 + (g << 2);
    const bool live = r < rpp && gx < dw;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (!live) return;
    const int4 *gp = s_gtab + 3 * g;
    const int4 g0 = gp[0], gsel = gp[1], gw = gp[2];
    const int bcol = g0.x - XA, wb = bcol & ~3, shf = bcol & 3;
    const uint32_t sel[4] = {(uint32_t)gsel.x, (uint32_t)gsel.y, (uint32_t)gsel.z, (uint32_t)gsel.w};
    const uint32_t wt[4] = {(uint32_t)gw.x, (uint32_t)gw.y, (uint32_t)gw.z, (uint32_t)gw.w};
    for (int j = 0; j < npass; j++) {
        const int ry = r + rpp * j, dy = oy0 + ry;
        if (dy >= dh) break;
        const int4 ytj = s_ytab[ry];
        const uint32_t b0s = ((uint32_t)ytj.z & 0xFFFu) << 12, b1s = ((uint32_t)ytj.w & 0xFFFu) << 12;   // (k_resize: v_mul_hi_u32_u24)
        const uint32_t *q0 = reinterpret_cast<const uint32_t *>(&s_src[ytj.x - symin][wb]);
        const uint32_t *q1 = reinterpret_cast<const uint32_t *>(&s_src[ytj.y - symin][wb]);
        const uint32_t a0 = q0[0], a1 = q0[1], a2 = q0[2], c0 = q1[0], c1 = q1[1], c2 = q1[2];
        const uint32_t lo0 = __builtin_amdgcn_alignbyte(a1, a0, shf), hi0 = __builtin_amdgcn_alignbyte(a2, a1, shf);
        const uint32_t lo1 = __builtin_amdgcn_alignbyte(c1, c0, shf), hi1 = __builtin_amdgcn_alignbyte(c2, c1, shf);
        uint32_t v[4];
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const us2 p0 = __builtin_bit_cast(us2, __builtin_amdgcn_perm(hi0, lo0, sel[k]));
            const us2 p1 = __builtin_bit_cast(us2, __builtin_amdgcn_perm(hi1, lo1, sel[k]));
            const us2 w2 = __builtin_bit_cast(us2, wt[k]);
            const uint32_t r0 = __builtin_amdgcn_udot2(p0, w2, 0u, false);
            const uint32_t r1 = __builtin_amdgcn_udot2(p1, w2, 0u, false);
            v[k] = (__umulhi(b0s, r0 & 0x7FFFF0u) + __umulhi(b1s, r1 & 0x7FFFF0u) + 2u) >> 2;   // <= 255
        }
        const uint32_t packed = v[0] | (v[1] << 8) | (v[2] << 16) | (v[3] << 24);
        uint8_t *o = D + (size_t)dy * dstride + gx;
        if (gx + 3 < dw) {
            *reinterpret_cast<uint32_t *>(o) = packed;
        } else {
            for (int k = 0; k < 4 && gx + k < dw; k++) o[k] = (uint8_t)(packed >> (8 * k));
        }
    }
}

// The fitted tile geometry of a level (cost = workgroups x (prologue + passes x row), in thread-instructions), or false when no
// geometry passes the window check.
bool resize_fit_plan(const int32_t *xt, const int32_t *yt, int sw, int sh, int dw, int dh, ResizeFit &out)
{
    const int ng = (dw + 3) / 4;
    const float winx = (float)sw / (float)dw, winy = (float)sh / (float)dh;
    long best = -1;
    for (int ntx = (ng + 31) / 32; ntx <= (ng + 31) / 32 + 3; ntx++) {
        const int twg = (ng + ntx - 1) / ntx;
        if (twg < 8 || twg > 32) continue;
        const int rpp = 256 / twg, rmagic = (65536 + twg - 1) / twg;
        bool okMagic = true;
        for (int tid = 0; tid < 256; tid++) okMagic = okMagic && (int)(((unsigned)tid * (unsigned)rmagic) >> 16) == tid / twg;
        if (!okMagic) continue;
        for (int npass = 3; npass <= 6; npass++) {
            const int TH = rpp * npass;
            if (TH > RF_MAXTH) continue;
            const int nty = (dh + TH - 1) / TH;
            static const int prologue = ORB_TUNE("RESIZE_FIT_P", 100);   // thread-instructions of a workgroup's fixed part / of a pass (70)
            const long cost = (long)ntx * nty * (prologue + 70 * npass);
            if (best >= 0 && cost >= best) continue;
            // the computed windows hold the taps and fit the staging area
            bool ok = true;
            for (int bx = 0; bx < ntx && ok; bx++) {
                const int ox0 = bx * 4 * twg, ox1 = std::min(ox0 + 4 * twg, dw) - 1;
                if (ox0 >= dw) { ok = false; break; }
                const int lo = std::max((int)((float)ox0 * winx) - 1, 0), hi = std::min((int)((float)(ox1 + 1) * winx) + 1, sw - 1);
                if (lo > (xt[2 * ox0] & 0xFFFF) || hi < (int)((uint32_t)xt[2 * ox1] >> 16)) ok = false;
                if (((hi - (lo & ~15)) >> 4) + 1 > RZ_MAXCH) ok = false;
            }
            for (int by = 0; by < nty && ok; by++) {
                const int oy0 = by * TH, oy1 = std::min(oy0 + TH, dh) - 1;
                const int rlo = std::max((int)((float)oy0 * winy) - 1, 0), rhi = std::min((int)((float)(oy1 + 1) * winy) + 1, sh - 1);
                if (rlo > yt[4 * oy0] || rhi < yt[4 * oy1 + 1] || rhi - rlo + 1 > RF_MAXROWS) ok = false;
            }
            if (!ok) continue;
            best = cost;
            out.ntx = ntx; out.nty = nty; out.twg = twg; out.rpp = rpp; out.npass = npass; out.rmagic = rmagic;
        }
    }
    return best >= 0;
}

void launch_resize_fit(hipStream_t s, const uint8_t *src, int sw, int sh, int sstride, size_t sframe, uint8_t *dst, int dw, int dh,
                       int dstride, size_t dframe, const int32_t *ytab, const int32_t *gtab, const ResizeFit &f, int B)
{
    dim3 grid(orb_xcd_grid(f.ntx * f.nty, 1), B, 1), block(256, 1, 1);
    orb_path(ORB_PATH_RESIZE_FIT);
    hipLaunchKernelGGL(k_resize_fit, grid, block, 0, s, src, sstride, (unsigned long long)sframe, dst, dw, dh, dstride,
                       (unsigned long long)dframe, reinterpret_cast<const int4 *>(ytab), reinterpret_cast<const int4 *>(gtab), sw, sh,
                       (float)sw / (float)dw, (float)sh / (float)dh, orb_xcd_arg(1), f.ntx, f.nty, f.twg, f.rpp, f.npass, f.rmagic);
}

// Can every tile of the level stage the window that the kernel derives from the scale factors, and does that window
// contain the taps of the tile?  (Checked on the host against the tables; otherwise the kernel reads the bounds.)
static bool resize_window_hint_ok(const int32_t *xt, const int32_t *yt, int sw, int sh, int dw, int dh, int th, float winx,
                                  float winy)
{
    for (int ox0 = 0; ox0 < dw; ox0 += RZ_TW) {
        const int ox1 = std::min(ox0 + RZ_TW, dw) - 1;
        const int lo = std::max((int)((float)ox0 * winx) - 1, 0), hi = std::min((int)((float)(ox1 + 1) * winx) + 1, sw - 1);
        if (lo > (xt[2 * ox0] & 0xFFFF) || hi < (int)((uint32_t)xt[2 * ox1] >> 16)) return false;
        const int nch = ((hi - (lo & ~15)) >> 4) + 1;
        if (nch > RZ_MAXCH) return false;
        for (int oy0 = 0; oy0 < dh; oy0 += th) {
            const int oy1 = std::min(oy0 + th, dh) - 1;
            const int rlo = std::max((int)((float)oy0 * winy) - 1, 0), rhi = std::min((int)((float)(oy1 + 1) * winy) + 1, sh - 1);
            if (rlo > yt[4 * oy0] || rhi < yt[4 * oy1 + 1]) return false;
            const int nrows = rhi - rlo + 1;
            if (nrows > RZ_MAXROWS || nrows * nch > 512) return false;
        }
    }
    return true;
}

bool resize_hint_fits(const int32_t *xt, const int32_t *yt, int sw, int sh, int dw, int dh, int th)
{
    return resize_window_hint_ok(xt, yt, sw, sh, dw, dh, th, (float)sw / (float)dw, (float)sh / (float)dh);
}

void launch_resize(hipStream_t s, const uint8_t *src, int sw, int sh, int sstride, size_t sframe,
                   uint8_t *dst, int dw, int dh, int dstride, size_t dframe, const int32_t *xtab,
                   const int32_t *ytab, const int32_t *gtab, bool hint, int B)
{
    dim3 block(256, 1, 1);
    const int th = B >= 8 ? 32 : 8;
    orb_path(ORB_PATH_RESIZE_TILES);
    const float winx = hint ? (float)sw / (float)dw : 0.f, winy = hint ? (float)sh / (float)dh : 0.f;
    dim3 grid(orb_xcd_grid(((dw + RZ_TW - 1) / RZ_TW) * ((dh + th - 1) / th), 1), B, 1);
    static const int ldsPad = ORB_TUNE("RESIZE_LDS_PAD", 0);   // occupancy experiment (ablation build): unused dynamic LDS per workgroup
    if (th == 32)
        hipLaunchKernelGGL(k_resize<32>, grid, block, (size_t)ldsPad, s, src, sstride, (unsigned long long)sframe, dst, dw, dh, dstride,
                           (unsigned long long)dframe, reinterpret_cast<const int2 *>(xtab),
                           reinterpret_cast<const int4 *>(ytab), reinterpret_cast<const int4 *>(gtab), sw, sh, winx, winy,
                           orb_xcd_arg(1));
    else
        hipLaunchKernelGGL(k_resize<8>, grid, block, 0, s, src, sstride, (unsigned long long)sframe, dst, dw, dh, dstride,
                           (unsigned long long)dframe, reinterpret_cast<const int2 *>(xtab),
                           reinterpret_cast<const int4 *>(ytab), reinterpret_cast<const int4 *>(gtab), sw, sh, winx, winy,
                           orb_xcd_arg(1));
}

// =====================================================================================================================
// Chained pyramid for a frame or two (single-frame latency).  Seven dependent k_resize launches cost ~4.7 us each although
// the arithmetic of a whole 640x480 pyramid is a few microseconds of the chip: the chain is launch-to-launch latency.
// k_pyramid_chain builds several levels per launch: a workgroup owns a CHAIN_TW x CHAIN_TH tile of level `top` and
// recomputes, inside LDS, the regions of the levels between `base` and `top` that the tile depends on (each level is a
// deterministic function of the previous one, so the recomputed pixels are the pixels k_resize writes -- same tap tables,
// same integer formula).  Every level's tiles are independent workgroups of the same launch; the redundant arithmetic (a
// level-4 tile recomputes ~9x its own pixel count) is noise next to the launches saved.  All global reads of a
// workgroup -- the base region, the tap-table slices of every level of its chain -- are issued before the first wait.
// Host side: chain_plan() groups the levels (ORBHIP_CHAIN_DEPTH levels per launch, default 4) and sizes the LDS;
// resize_hint_pointwise() proves per level that the computed source windows contain the taps.  Batches keep k_resize.
// =====================================================================================================================
#define CH_NT 1024 // threads of a workgroup: a pixel of a chain step is a chain of dependent LDS reads (tap entry -> source
                   // bytes -> store), so the steps are latency, and many short per-thread loops hide it better than few long ones
#define CH_KX 2    // column-tap entries staged per thread (2048 over the levels of a chain)
#define CH_KY 1    // row-tap entries per thread (1024)
#define CH_KP 2    // 16-byte chunks of the base region per thread (2048 = 32 KB)

// the region of level l - 1 read by the region [x0, x1] x [y0, y1] of level l: the window hint of k_resize
__host__ __device__ inline void chain_source_region(int &x0, int &x1, int &y0, int &y1, float winx, float winy, int sw, int sh)
{
    int a = (int)((float)x0 * winx) - 1, b = (int)((float)(x1 + 1) * winx) + 1;
    x0 = a < 0 ? 0 : a;
    x1 = b > sw - 1 ? sw - 1 : b;
    a = (int)((float)y0 * winy) - 1;
    b = (int)((float)(y1 + 1) * winy) + 1;
    y0 = a < 0 ? 0 : a;
    y1 = b > sh - 1 ? sh - 1 : b;
}

__global__ __launch_bounds__(CH_NT) void k_pyramid_chain(const OrbLevels G, const ChainLevels CL, const ChainTile *__restrict__ tiles,
                                                       const uint8_t *__restrict__ lvl0, int stride0, unsigned long long frame0,
                                                       uint8_t *__restrict__ pyr, unsigned long long pyrFrame,
                                                       const int32_t *__restrict__ tab, int bufA, int bufB, int xtabBytes,
                                                       uint8_t *__restrict__ hostPyr)
{
    // hostPyr: page-locked host twin of `pyr` (orbhip_set_host_pyramid) or null.  The finished pixels of a tile are stored to it
    // as well: the level arrives on the host with the kernel, over PCIe as posted 64-byte writes, instead of through a copy
    // node of the graph that the FAST kernel ends up queued behind (measured: 12 us + 6 us of gaps per call).
    extern __shared__ __align__(16) uint8_t smem[];   // [buffer A | buffer B | column taps | row taps]
    __shared__ int s_r[ORBHIP_MAX_LEVELS][4];          // region of every level of the chain: x0, x1, y0, y1 (inclusive)
    __shared__ int s_xo[ORBHIP_MAX_LEVELS], s_yo[ORBHIP_MAX_LEVELS];   // first staged entry of the level's tap slices
    __shared__ unsigned s_xoff[ORBHIP_MAX_LEVELS], s_yoff[ORBHIP_MAX_LEVELS];
    __shared__ int s_tot[2];
    ChainTile T;
    *reinterpret_cast<uint2 *>(&T) = *reinterpret_cast<const uint2 *>(tiles + blockIdx.x);   // one 8-byte load
    const int frame = blockIdx.y, tid = threadIdx.x;
    const int top = T.level, base = T.base;
    if (tid == 0) {
        int x0 = T.tx * CHAIN_TW, y0 = T.ty * CHAIN_TH;
        int x1 = min(x0 + CHAIN_TW, G.lv[top].w) - 1, y1 = min(y0 + CHAIN_TH, G.lv[top].h) - 1;
        int sx = 0, sy = 0;
        for (int l = top; l > base; l--) {
            s_r[l][0] = x0; s_r[l][1] = x1; s_r[l][2] = y0; s_r[l][3] = y1;
            s_xo[l] = sx; s_yo[l] = sy;
            s_xoff[l] = CL.xoff[l]; s_yoff[l] = CL.yoff[l];
            sx += x1 - x0 + 1;
            sy += y1 - y0 + 1;
            chain_source_region(x0, x1, y0, y1, CL.winx[l], CL.winy[l], G.lv[l - 1].w, G.lv[l - 1].h);
        }
        s_r[base][0] = x0; s_r[base][1] = x1; s_r[base][2] = y0; s_r[base][3] = y1;
        s_tot[0] = sx; s_tot[1] = sy;
    }
    __syncthreads();
    uint8_t *bA = smem, *bB = smem + bufA;
    int2 *s_xt = reinterpret_cast<int2 *>(smem + bufA + bufB);
    int4 *s_yt = reinterpret_cast<int4 *>(smem + bufA + bufB + xtabBytes);

    // ---- stage: base region (16-byte row-coalesced chunks) and the tap slices of every level; loads first, stores after ----
    const int bx0 = s_r[base][0], bx1 = s_r[base][1], by0 = s_r[base][2], by1 = s_r[base][3];
    const int XA = bx0 & ~15, nch = ((bx1 - XA) >> 4) + 1, nrows = by1 - by0 + 1, pS = nch << 4;
    const uint8_t *S = base == 0 ? lvl0 + (size_t)frame * frame0 : pyr + (size_t)frame * pyrFrame + G.lv[base].imgOff;
    const int sstride = base == 0 ? stride0 : G.lv[base].stride;
    const int npix = nrows * nch, sumx = s_tot[0], sumy = s_tot[1];
    const float invNch = __builtin_amdgcn_rcpf((float)nch);   // i / nch = floor((i + 0.5) * invNch) for i < 2^13 (see k_fast)
    uint4 pv[CH_KP];
    int2 xv[CH_KX];
    int4 yv[CH_KY];
    const uint8_t *S0 = S + (size_t)by0 * sstride + XA;
#pragma unroll
    for (int k = 0; k < CH_KP; k++) {
        const int i = min(tid + CH_NT * k, npix - 1);
        const int r = (int)(((float)i + 0.5f) * invNch), c = i - r * nch;
        pv[k] = *reinterpret_cast<const uint4 *>(S0 + (unsigned)(r * sstride + (c << 4)));
    }
#pragma unroll
    for (int k = 0; k < CH_KX; k++) {
        int rem = min(tid + CH_NT * k, sumx - 1), l = top;
        while (l > base + 1 && rem >= s_r[l][1] - s_r[l][0] + 1) {
            rem -= s_r[l][1] - s_r[l][0] + 1;
            l--;
        }
        xv[k] = reinterpret_cast<const int2 *>(tab + s_xoff[l])[s_r[l][0] + rem];
    }
#pragma unroll
    for (int k = 0; k < CH_KY; k++) {
        int rem = min(tid + CH_NT * k, sumy - 1), l = top;
        while (l > base + 1 && rem >= s_r[l][3] - s_r[l][2] + 1) {
            rem -= s_r[l][3] - s_r[l][2] + 1;
            l--;
        }
        yv[k] = reinterpret_cast<const int4 *>(tab + s_yoff[l])[s_r[l][2] + rem];
    }
    // (the empty asm pins every loaded value here: without it the compiler sinks each load into the conditional store below
    // and the workgroup waits for them one by one)
#pragma unroll
    for (int k = 0; k < CH_KP; k++) asm volatile("" : "+v"(pv[k].x), "+v"(pv[k].y), "+v"(pv[k].z), "+v"(pv[k].w));
#pragma unroll
    for (int k = 0; k < CH_KX; k++) asm volatile("" : "+v"(xv[k].x), "+v"(xv[k].y));
#pragma unroll
    for (int k = 0; k < CH_KY; k++) asm volatile("" : "+v"(yv[k].x), "+v"(yv[k].y), "+v"(yv[k].z), "+v"(yv[k].w));
#pragma unroll
    for (int k = 0; k < CH_KP; k++) {
        const int i = tid + CH_NT * k;
        if (i < npix) {
            const int r = (int)(((float)i + 0.5f) * invNch), c = i - r * nch;
            *reinterpret_cast<uint4 *>(bA + r * pS + (c << 4)) = pv[k];
        }
    }
#pragma unroll
    for (int k = 0; k < CH_KX; k++)
        if (tid + CH_NT * k < sumx) s_xt[tid + CH_NT * k] = xv[k];
#pragma unroll
    for (int k = 0; k < CH_KY; k++)
        if (tid + CH_NT * k < sumy) s_yt[tid + CH_NT * k] = yv[k];
    __syncthreads();

    // ---- level by level inside LDS; the last level goes to memory ----
    const uint8_t *src = bA;
    uint8_t *dst = bB;
    int ox = XA, oy = by0, ps = pS;
    uint8_t *D = pyr + (size_t)frame * pyrFrame + G.lv[top].imgOff;
    uint8_t *HD = hostPyr ? hostPyr + (size_t)frame * pyrFrame + G.lv[top].imgOff : nullptr;
    const int dstride = G.lv[top].stride;
    for (int l = base + 1; l <= top; l++) {
        const int x0 = s_r[l][0], y0 = s_r[l][2];
        const int nx = s_r[l][1] - x0 + 1, ny = s_r[l][3] - y0 + 1, pd = (nx + 3) & ~3;
        const int2 *xt = s_xt + s_xo[l];
        const int4 *yt = s_yt + s_yo[l];
        const float invNx = __builtin_amdgcn_rcpf((float)nx);
        const bool last = l == top;
        // two pixels per trip: their LDS round trips overlap
        auto pixel = [&](int i) -> int {
            const int ry = (int)(((float)i + 0.5f) * invNx), rx = i - ry * nx;
            const int4 t = yt[ry];
            const int2 u = xt[rx];
            const int sx0 = (u.x & 0xFFFF) - ox, sx1 = (int)((unsigned)u.x >> 16) - ox;
            const int a0 = (short)(u.y & 0xFFFF), a1 = u.y >> 16;
            const uint8_t *L0 = src + (t.x - oy) * ps, *L1 = src + (t.y - oy) * ps;
            // the integers of k_resize's generic path
            const int r0 = L0[sx0] * a0 + L0[sx1] * a1;
            const int r1 = L1[sx0] * a0 + L1[sx1] * a1;
            return (((t.z * (r0 >> 4)) >> 16) + ((t.w * (r1 >> 4)) >> 16) + 2) >> 2;
        };
        auto put = [&](int i, int v) {
            const int ry = (int)(((float)i + 0.5f) * invNx), rx = i - ry * nx;
            if (last) {
                D[(size_t)(y0 + ry) * dstride + x0 + rx] = (uint8_t)v;
                if (HD) HD[(size_t)(y0 + ry) * dstride + x0 + rx] = (uint8_t)v;
            } else
                dst[ry * pd + rx] = (uint8_t)v;
        };
        const int n = nx * ny;
        for (int i = tid; i < n; i += 2 * CH_NT) {
            const int j = i + CH_NT;
            const int va = pixel(i), vb = pixel(min(j, n - 1));
            put(i, va);
            if (j < n) put(j, vb);
        }
        __syncthreads();
        uint8_t *nd = const_cast<uint8_t *>(src);
        src = dst;
        dst = nd;
        ox = x0;
        oy = y0;
        ps = pd;
    }
}

// Does the window that chain_source_region / k_resize derive from the scale factors contain the taps of EVERY output
// column and row taken alone?  (Then it contains the taps of any interval: the bounds and the taps are monotone.)
bool resize_hint_pointwise(const int32_t *xt, const int32_t *yt, int sw, int sh, int dw, int dh)
{
    const float winx = (float)sw / (float)dw, winy = (float)sh / (float)dh;
    int prev = -1;
    for (int o = 0; o < dw; o++) {
        const int lo = std::max((int)((float)o * winx) - 1, 0), hi = std::min((int)((float)(o + 1) * winx) + 1, sw - 1);
        const int s0 = xt[2 * o] & 0xFFFF, s1 = (int)((uint32_t)xt[2 * o] >> 16);
        if (lo > s0 || hi < s1 || s0 < prev) return false;
        prev = s0;
    }
    prev = -1;
    for (int o = 0; o < dh; o++) {
        const int lo = std::max((int)((float)o * winy) - 1, 0), hi = std::min((int)((float)(o + 1) * winy) + 1, sh - 1);
        if (lo > yt[4 * o] || hi < yt[4 * o + 1] || yt[4 * o] < prev) return false;
        prev = yt[4 * o];
    }
    return true;
}

struct ChainNeed {
    int bufA = 0, bufB = 0, sumx = 0, sumy = 0;
    bool ok = true;
};

// LDS and staging needs of the tiles of level `top` chained from level `base`
static void chain_need(const OrbLevels &G, const ChainLevels &CL, int base, int top, ChainNeed &N)
{
    const int tilesX = (G.lv[top].w + CHAIN_TW - 1) / CHAIN_TW, tilesY = (G.lv[top].h + CHAIN_TH - 1) / CHAIN_TH;
    for (int ty = 0; ty < tilesY; ty++)
        for (int tx = 0; tx < tilesX; tx++) {
            int x0 = tx * CHAIN_TW, y0 = ty * CHAIN_TH;
            int x1 = std::min(x0 + CHAIN_TW, G.lv[top].w) - 1, y1 = std::min(y0 + CHAIN_TH, G.lv[top].h) - 1;
            int sx = 0, sy = 0;
            for (int l = top; l > base; l--) {
                const int nx = x1 - x0 + 1, ny = y1 - y0 + 1, bytes = ((nx + 3) & ~3) * ny;
                if (nx * ny >= 32768) N.ok = false;
                // level l is written by step l - base: odd steps into buffer B, even steps into buffer A
                if ((l - base) & 1) N.bufB = std::max(N.bufB, bytes); else N.bufA = std::max(N.bufA, bytes);
                sx += nx;
                sy += ny;
                chain_source_region(x0, x1, y0, y1, CL.winx[l], CL.winy[l], G.lv[l - 1].w, G.lv[l - 1].h);
            }
            const int nch = ((x1 - (x0 & ~15)) >> 4) + 1, nrows = y1 - y0 + 1;
            if (nch * nrows > CH_NT * CH_KP || nch * nrows >= 8192) N.ok = false;
            N.bufA = std::max(N.bufA, nch * 16 * nrows);
            N.sumx = std::max(N.sumx, sx);
            N.sumy = std::max(N.sumy, sy);
        }
    if (N.sumx > CH_NT * CH_KX || N.sumy > CH_NT * CH_KY) N.ok = false;
}

bool chain_plan(const OrbLevels &G, const bool *levelOk, const ChainLevels &CL, std::vector<ChainTile> &tiles,
                std::vector<ChainGroup> &groups)
{
    static const int depthEnv = ORB_TUNE("CHAIN_DEPTH", 4);
    const int maxDepth = depthEnv < 1 ? 1 : depthEnv;
    const int ldsCap = 64 * 1024;
    tiles.clear();
    groups.clear();
    int base = 0;
    while (base < G.nlevels - 1) {
        int top = base;
        ChainNeed need;
        while (top + 1 < G.nlevels && top + 1 - base <= maxDepth && levelOk[top + 1]) {
            ChainNeed n = need;
            chain_need(G, CL, base, top + 1, n);
            const int lds = ((n.bufA + 15) & ~15) + ((n.bufB + 15) & ~15) + ((n.sumx * 8 + 15) & ~15) + n.sumy * 16;
            if (!n.ok || lds > ldsCap) break;
            need = n;
            top++;
        }
        if (top == base) {
            tiles.clear();
            groups.clear();
            return false;
        }
        ChainGroup g;
        g.firstTile = (int)tiles.size();
        for (int l = top; l > base; l--)   // the deepest chains first
            for (int ty = 0; ty < (G.lv[l].h + CHAIN_TH - 1) / CHAIN_TH; ty++)
                for (int tx = 0; tx < (G.lv[l].w + CHAIN_TW - 1) / CHAIN_TW; tx++) {
                    ChainTile t;
                    t.level = (short)l;
                    t.base = (short)base;
                    t.tx = (short)tx;
                    t.ty = (short)ty;
                    tiles.push_back(t);
                }
        g.ntiles = (int)tiles.size() - g.firstTile;
        g.bufA = (need.bufA + 15) & ~15;
        g.bufB = (need.bufB + 15) & ~15;
        g.xtabBytes = (need.sumx * 8 + 15) & ~15;
        g.ytabBytes = need.sumy * 16;
        groups.push_back(g);
        base = top;
    }
    return true;
}

void launch_pyramid_chain(hipStream_t s, const OrbLevels &G, const ChainLevels &CL, const ChainGroup &grp, const ChainTile *tiles,
                          const uint8_t *lvl0, int stride0, size_t frame0, uint8_t *pyr, size_t pyrFrame, const int32_t *tab, int B,
                          uint8_t *hostPyr)
{
    dim3 grid(grp.ntiles, B, 1), block(CH_NT, 1, 1);
    orb_path(ORB_PATH_PYRAMID_CHAIN);
    hipLaunchKernelGGL(k_pyramid_chain, grid, block, (size_t)(grp.bufA + grp.bufB + grp.xtabBytes + grp.ytabBytes), s, G, CL,
                       tiles + grp.firstTile, lvl0, stride0, (unsigned long long)frame0, pyr, (unsigned long long)pyrFrame, tab,
                       grp.bufA, grp.bufB, grp.xtabBytes, hostPyr);
}
