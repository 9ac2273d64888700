// k_pyramid.hip -- E2: one pyramid level from the previous one, bit-exact cv::resize INTER_LINEAR
// 8UC1 (ref call site: src/ORBextractor.cc:1141).  The 19-px border of the reference
// (copyMakeBorder :1143-1149) is never read by later stages and is not produced.
//
// One 256-thread workgroup per 128x32 output tile.  The source rows/columns the tile touches
// (about 156 x 40 pixels at scale 1.2) are staged into LDS with 16-byte row-coalesced loads; a
// thread then produces 4 horizontally adjacent output pixels from LDS and stores them as one
// dword.  The column/row tap tables (source index pair + 11-bit weights) are built on the host
// (orb_build_resize_tables).  Bound: HBM (reads 1.44 px and writes 1 px per output pixel).
#include "orbhip_internal.h"

#define RZ_TW 128
// RZ_TH = rows of a tile (RZ_TH / 8 per thread): 32 for batches, 8 for a single frame or two (4x the workgroups)
#define RZ_MAXCH 16    // 16-byte chunks per staged source row (source span <= 240 px + alignment)
#define RZ_MAXROWS 44  // staged source rows

template <int RZ_TH>
__global__ __launch_bounds__(256) void k_resize(const uint8_t *__restrict__ src, int sstride,
                                                unsigned long long sframe, uint8_t *__restrict__ dst,
                                                int dw, int dh, int dstride, unsigned long long dframe,
                                                const int2 *__restrict__ xtab,
                                                const int4 *__restrict__ ytab, int xcdMap)
{
    __shared__ __align__(16) uint8_t s_src[RZ_MAXROWS][RZ_MAXCH * 16];
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

    // source window of the tile (tables are monotone)
    const int sxmin = xtab[ox0].x & 0xFFFF, sxmax = (unsigned)xtab[ox1].x >> 16;
    const int symin = ytab[oy0].x, symax = ytab[oy1].y;
    const int XA = sxmin & ~15;
    const int nch = ((sxmax - XA) >> 4) + 1;
    const int nrows = symax - symin + 1;
    if (nch <= RZ_MAXCH && nrows <= RZ_MAXROWS && nrows * nch <= 512) {
        // at most RZ_MAXROWS * RZ_MAXCH = 384 chunks: two unconditional loads per thread, issued together
        const int n = nrows * nch;
        const int i0 = min(tid, n - 1), i1 = min(tid + 256, n - 1);
        const int r0 = i0 / nch, c0 = i0 - r0 * nch, r1 = i1 / nch, c1 = i1 - r1 * nch;
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
#pragma unroll
    for (int k = 0; k < 4; k++) xt[k] = xtab[min(gx + k, dw - 1)];
    __syncthreads();
    if (!colLive) return;
    const bool staged = nch <= RZ_MAXCH && nrows <= RZ_MAXROWS && nrows * nch <= 512;
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

void launch_resize(hipStream_t s, const uint8_t *src, int sw, int sh, int sstride, size_t sframe,
                   uint8_t *dst, int dw, int dh, int dstride, size_t dframe, const int32_t *xtab,
                   const int32_t *ytab, int B)
{
    (void)sw;
    (void)sh;
    dim3 block(256, 1, 1);
    const int th = B >= 8 ? 32 : 8;
    dim3 grid(orb_xcd_grid(((dw + RZ_TW - 1) / RZ_TW) * ((dh + th - 1) / th), 1), B, 1);
    if (th == 32)
        hipLaunchKernelGGL(k_resize<32>, grid, block, 0, s, src, sstride, (unsigned long long)sframe, dst, dw, dh, dstride,
                           (unsigned long long)dframe, reinterpret_cast<const int2 *>(xtab),
                           reinterpret_cast<const int4 *>(ytab), orb_xcd_arg(1));
    else
        hipLaunchKernelGGL(k_resize<8>, grid, block, 0, s, src, sstride, (unsigned long long)sframe, dst, dw, dh, dstride,
                           (unsigned long long)dframe, reinterpret_cast<const int2 *>(xtab),
                           reinterpret_cast<const int4 *>(ytab), orb_xcd_arg(1));
}
