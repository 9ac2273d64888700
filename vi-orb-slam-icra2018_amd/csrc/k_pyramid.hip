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
        // at most RZ_MAXROWS * RZ_MAXCH = 384 chunks: two unconditional loads per thread, issued together
        const int n = nrows * nch;
        const int i0 = min(tid, n - 1), i1 = min(tid + 256, n - 1);
        const float invNch = 1.0f / (float)nch;   // i / nch = floor((i + 0.5) * invNch), exact for i < 2^16
        const int r0 = (int)(((float)i0 + 0.5f) * invNch), c0 = i0 - r0 * nch;
        const int r1 = (int)(((float)i1 + 0.5f) * invNch), c1 = i1 - r1 * nch;
        const uint4 v0 = *reinterpret_cast<const uint4 *>(S + (size_t)(symin + r0) * sstride + XA + (c0 << 4));
        const uint4 v1 = *reinterpret_cast<const uint4 *>(S + (size_t)(symin + r1) * sstride + XA + (c1 << 4));
        if (tid < n) *reinterpret_cast<uint4 *>(&s_src[r0][c0 << 4]) = v0;
        if (tid + 256 < n) *reinterpret_cast<uint4 *>(&s_src[r1][c1 << 4]) = v1;
    }
    // this thread's taps are requested before the barrier so that their latency overlaps the staging
    constexpr int NR = RZ_TH / 8;                 // output rows per thread (rows dy, dy + 8, ...)
    const int gx = ox0 + ((tid & 31) << 2);
    const int dy0 = oy0 + (tid >> 5);
    const bool colLive = gx < dw;
    int4 yt[NR];
    int2 xt[4] = {make_int2(0, 0), make_int2(0, 0), make_int2(0, 0), make_int2(0, 0)};
#pragma unroll
    for (int j = 0; j < NR; j++) yt[j] = ytab[min(dy0 + 8 * j, dh - 1)];
    const bool staged = nch <= RZ_MAXCH && nrows <= RZ_MAXROWS && nrows * nch <= 512;
    const bool grouped = staged && gtab != nullptr;
    int4 g0 = make_int4(0, 0, 0, 0), gsel = g0, gw = g0;
    if (grouped) {
        const int4 *gp = gtab + 3 * (min(gx, dw - 1) >> 2);
        g0 = gp[0];
        gsel = gp[1];
        gw = gp[2];
    } else {
#pragma unroll
        for (int k = 0; k < 4; k++) xt[k] = xtab[min(gx + k, dw - 1)];
    }
    __syncthreads();
    if (!colLive) return;
    if (grouped) {
        // Fast path: the 8 source bytes from the group's base column on (three aligned LDS dwords, realigned) hold all
        // eight taps of a row; a pixel's tap pair comes out with one v_perm, the horizontal sum is one v_dot2 with
        // the weight pair, and (b * (r >> 4)) >> 16 is the high half of (b << 16) * (r >> 4).  Same integers as the
        // generic path below.
        const int bcol = g0.x - XA, wb = bcol & ~3, sh = bcol & 3;
        const uint32_t sel[4] = {(uint32_t)gsel.x, (uint32_t)gsel.y, (uint32_t)gsel.z, (uint32_t)gsel.w};
        const uint32_t wt[4] = {(uint32_t)gw.x, (uint32_t)gw.y, (uint32_t)gw.z, (uint32_t)gw.w};
#pragma unroll
        for (int j = 0; j < NR; j++) {
            const int dy = dy0 + 8 * j;
            if (dy >= dh) break;
            const uint32_t b0s = (uint32_t)yt[j].z << 16, b1s = (uint32_t)yt[j].w << 16;
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
                v[k] = (__umulhi(b0s, r0 >> 4) + __umulhi(b1s, r1 >> 4) + 2u) >> 2;   // <= 255
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
    const float winx = hint ? (float)sw / (float)dw : 0.f, winy = hint ? (float)sh / (float)dh : 0.f;
    dim3 grid(orb_xcd_grid(((dw + RZ_TW - 1) / RZ_TW) * ((dh + th - 1) / th), 1), B, 1);
    if (th == 32)
        hipLaunchKernelGGL(k_resize<32>, grid, block, 0, s, src, sstride, (unsigned long long)sframe, dst, dw, dh, dstride,
                           (unsigned long long)dframe, reinterpret_cast<const int2 *>(xtab),
                           reinterpret_cast<const int4 *>(ytab), reinterpret_cast<const int4 *>(gtab), sw, sh, winx, winy,
                           orb_xcd_arg(1));
    else
        hipLaunchKernelGGL(k_resize<8>, grid, block, 0, s, src, sstride, (unsigned long long)sframe, dst, dw, dh, dstride,
                           (unsigned long long)dframe, reinterpret_cast<const int2 *>(xtab),
                           reinterpret_cast<const int4 *>(ytab), reinterpret_cast<const int4 *>(gtab), sw, sh, winx, winy,
                           orb_xcd_arg(1));
}
