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

// ---- two levels per launch ------------------------------------------------------------------------------
// Level l+1 and level l+2 from level l in one pass: a workgroup owns a 128 x 16 tile of level l+2, computes
// the patch of level l+1 that tile reads (about 158 x 22 pixels) from a staged window of level l, keeps it in
// LDS, writes the part of it that it OWNS to memory, and resizes the tile from LDS.  Level l+1 is therefore
// written once and never read back, and the pyramid takes 4 launches instead of 7.  Ownership: the columns of
// level l+1 are cut at the first source column of every tile's first pixel (the tables are monotone), rows
// likewise; the first / last tile of a row or column takes the margins.  The arithmetic of each level is the
// one of k_resize (same tables), so the levels are bit-identical to the single-level kernels.
#define R2_TW 128
#define R2_TH 16
#define R2_MW 176   // bytes of a patch row of level l+1 (multiple of 16)
#define R2_MH 24    // patch rows
#define R2_SCH 14   // 16-byte chunks of a staged row of level l
#define R2_SH 32    // staged rows of level l

struct Resize2Args {
    const uint8_t *src;
    uint8_t *mid, *dst;
    int sstride, mw, mh, mstride, dw, dh, dstride;
    unsigned long long sframe, mframe, dframe;
    const int2 *xt1, *xt2;
    const int4 *yt1, *yt2;
};

__global__ __launch_bounds__(256) void k_resize2(const Resize2Args A, int xcdMap)
{
    __shared__ __align__(16) uint8_t s_src[R2_SH][R2_SCH * 16];
    __shared__ __align__(16) uint8_t s_mid[R2_MH][R2_MW];
    __shared__ int2 s_xt[R2_MW];
    __shared__ int4 s_yt[R2_MH];
    const int tid = threadIdx.x;
    const int tilesX = (A.dw + R2_TW - 1) / R2_TW, tilesY = (A.dh + R2_TH - 1) / R2_TH;
    const int t = xcd_tile(xcdMap), frame = blockIdx.y;
    if (t >= tilesX * tilesY) return;
    const int by = t / tilesX, bx = t - by * tilesX;
    const int ox0 = bx * R2_TW, oy0 = by * R2_TH;
    const int ox1 = min(ox0 + R2_TW, A.dw) - 1, oy1 = min(oy0 + R2_TH, A.dh) - 1;   // inclusive
    const uint8_t *S = A.src + (size_t)frame * A.sframe;
    uint8_t *M = A.mid + (size_t)frame * A.mframe;
    uint8_t *D = A.dst + (size_t)frame * A.dframe;

    // patch of level l+1: what the tile reads, united with what it owns
    const int mx0 = A.xt2[ox0].x & 0xFFFF, mx1 = (unsigned)A.xt2[ox1].x >> 16;
    const int my0 = A.yt2[oy0].x, my1 = A.yt2[oy1].y;
    const int ownX0 = bx == 0 ? 0 : mx0, ownX1 = bx == tilesX - 1 ? A.mw : (A.xt2[ox1 + 1].x & 0xFFFF);   // [ , )
    const int ownY0 = by == 0 ? 0 : my0, ownY1 = by == tilesY - 1 ? A.mh : A.yt2[oy1 + 1].x;
    const int rx0 = min(ownX0, mx0) & ~3, rx1 = max(ownX1 - 1, mx1);
    const int ry0 = min(ownY0, my0), ry1 = max(ownY1 - 1, my1);
    const int RW4 = (rx1 - rx0 + 4) >> 2, RH = ry1 - ry0 + 1;     // groups of 4 columns, rows
    // taps of the patch (level l -> l+1) into LDS
    for (int i = tid; i < RW4 * 4; i += 256) s_xt[i] = A.xt1[min(rx0 + i, A.mw - 1)];
    if (tid < RH) s_yt[tid] = A.yt1[ry0 + tid];
    // window of level l
    const int sxmin = A.xt1[rx0].x & 0xFFFF, sxmax = (unsigned)A.xt1[rx1].x >> 16;
    const int symin = A.yt1[ry0].x, symax = A.yt1[ry1].y;
    const int XA = sxmin & ~15;
    const int nch = ((sxmax - XA) >> 4) + 1, nrows = symax - symin + 1;
    {
        const int n = nrows * nch;     // <= R2_SH * R2_SCH = 448 (checked on the host): two loads per thread
        const int i0 = min(tid, n - 1), i1 = min(tid + 256, n - 1);
        const int r0 = i0 / nch, c0 = i0 - r0 * nch, r1 = i1 / nch, c1 = i1 - r1 * nch;
        const uint4 v0 = *reinterpret_cast<const uint4 *>(S + (size_t)(symin + r0) * A.sstride + XA + (c0 << 4));
        const uint4 v1 = *reinterpret_cast<const uint4 *>(S + (size_t)(symin + r1) * A.sstride + XA + (c1 << 4));
        if (tid < n) *reinterpret_cast<uint4 *>(&s_src[r0][c0 << 4]) = v0;
        if (tid + 256 < n) *reinterpret_cast<uint4 *>(&s_src[r1][c1 << 4]) = v1;
    }
    // the tile's own taps (level l+1 -> l+2), requested before the barrier
    const int gx = ox0 + ((tid & 31) << 2);
    const int dy0 = oy0 + (tid >> 5);
    int4 yt2[2];
    int2 xt2[4];
#pragma unroll
    for (int j = 0; j < 2; j++) yt2[j] = A.yt2[min(dy0 + 8 * j, A.dh - 1)];
#pragma unroll
    for (int k = 0; k < 4; k++) xt2[k] = A.xt2[min(gx + k, A.dw - 1)];
    __syncthreads();

    // ---- level l+1 patch: 4 pixels per item ----
    const unsigned rwMagic = 65536u / (unsigned)RW4 + 1u;   // i / RW4 for i < 65536 / RW4
    for (int i = tid; i < RH * RW4; i += 256) {
        const int row = (int)(((unsigned)i * rwMagic) >> 16), g = i - row * RW4;
        const int4 yt = s_yt[row];
        const uint8_t *L0 = &s_src[yt.x - symin][0], *L1 = &s_src[yt.y - symin][0];
        uint32_t packed = 0;
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const int2 xt = s_xt[4 * g + k];
            const int sx0 = (xt.x & 0xFFFF) - XA, sx1 = (int)((unsigned)xt.x >> 16) - XA;
            const int a0 = (short)(xt.y & 0xFFFF), a1 = xt.y >> 16;
            const int r0 = L0[sx0] * a0 + L0[sx1] * a1;
            const int r1 = L1[sx0] * a0 + L1[sx1] * a1;
            const int v = (((yt.z * (r0 >> 4)) >> 16) + ((yt.w * (r1 >> 4)) >> 16) + 2) >> 2;
            packed |= (uint32_t)(v & 0xFF) << (8 * k);
        }
        *reinterpret_cast<uint32_t *>(&s_mid[row][4 * g]) = packed;
        const int y = ry0 + row, x = rx0 + 4 * g;
        if (y >= ownY0 && y < ownY1) {
            uint8_t *o = M + (size_t)y * A.mstride + x;
            if (x >= ownX0 && x + 3 < ownX1)
                *reinterpret_cast<uint32_t *>(o) = packed;
            else
                for (int k = 0; k < 4; k++)
                    if (x + k >= ownX0 && x + k < ownX1) o[k] = (uint8_t)(packed >> (8 * k));
        }
    }
    __syncthreads();

    // ---- level l+2 tile from the patch ----
    if (gx >= A.dw) return;
#pragma unroll
    for (int j = 0; j < 2; j++) {
        const int dy = dy0 + 8 * j;
        if (dy >= A.dh) break;
        const uint8_t *L0 = &s_mid[yt2[j].x - ry0][0], *L1 = &s_mid[yt2[j].y - ry0][0];
        uint32_t packed = 0;
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const int sx0 = (xt2[k].x & 0xFFFF) - rx0, sx1 = (int)((unsigned)xt2[k].x >> 16) - rx0;
            const int a0 = (short)(xt2[k].y & 0xFFFF), a1 = xt2[k].y >> 16;
            const int r0 = L0[sx0] * a0 + L0[sx1] * a1;
            const int r1 = L1[sx0] * a0 + L1[sx1] * a1;
            const int v = (((yt2[j].z * (r0 >> 4)) >> 16) + ((yt2[j].w * (r1 >> 4)) >> 16) + 2) >> 2;
            packed |= (uint32_t)(v & 0xFF) << (8 * k);
        }
        uint8_t *o = D + (size_t)dy * A.dstride + gx;
        if (gx + 3 < A.dw)
            *reinterpret_cast<uint32_t *>(o) = packed;
        else
            for (int k = 0; k < 4 && gx + k < A.dw; k++) o[k] = (uint8_t)(packed >> (8 * k));
    }
}

// Host check: do all tiles of this level pair fit the LDS windows of k_resize2?  (Tables as built by
// orb_build_resize_tables: xt = {sx0 | sx1 << 16, weights}, yt = {sy0, sy1, b0, b1}.)
bool resize2_fits(const int32_t *xt1, const int32_t *yt1, int mw, int mh, const int32_t *xt2, const int32_t *yt2, int dw,
                  int dh)
{
    const int tilesX = (dw + R2_TW - 1) / R2_TW, tilesY = (dh + R2_TH - 1) / R2_TH;
    for (int bx = 0; bx < tilesX; bx++) {
        const int ox0 = bx * R2_TW, ox1 = std::min(ox0 + R2_TW, dw) - 1;
        const int mx0 = xt2[2 * ox0] & 0xFFFF, mx1 = (unsigned)xt2[2 * ox1] >> 16;
        const int ownX0 = bx == 0 ? 0 : mx0, ownX1 = bx == tilesX - 1 ? mw : (xt2[2 * (ox1 + 1)] & 0xFFFF);
        const int rx0 = std::min(ownX0, mx0) & ~3, rx1 = std::max(ownX1 - 1, mx1);
        if (rx1 >= mw || ((rx1 - rx0 + 4) >> 2) * 4 > R2_MW) return false;
        const int sxmin = xt1[2 * rx0] & 0xFFFF, sxmax = (unsigned)xt1[2 * rx1] >> 16;
        const int nch = ((sxmax - (sxmin & ~15)) >> 4) + 1;
        if (nch > R2_SCH) return false;
        for (int by = 0; by < tilesY; by++) {
            const int oy0 = by * R2_TH, oy1 = std::min(oy0 + R2_TH, dh) - 1;
            const int my0 = yt2[4 * oy0], my1 = yt2[4 * oy1 + 1];
            const int ownY0 = by == 0 ? 0 : my0, ownY1 = by == tilesY - 1 ? mh : yt2[4 * (oy1 + 1)];
            const int ry0 = std::min(ownY0, my0), ry1 = std::max(ownY1 - 1, my1);
            if (ry1 >= mh || ry1 - ry0 + 1 > R2_MH) return false;
            const int nrows = yt1[4 * ry1 + 1] - yt1[4 * ry0] + 1;
            if (nrows > R2_SH || nrows * nch > 512) return false;
        }
    }
    return true;
}

void launch_resize2(hipStream_t s, const uint8_t *src, int sstride, size_t sframe, uint8_t *mid, int mw, int mh, int mstride,
                    uint8_t *dst, int dw, int dh, int dstride, size_t pyrFrame, const int32_t *xt1, const int32_t *yt1,
                    const int32_t *xt2, const int32_t *yt2, int B)
{
    Resize2Args A;
    A.src = src;
    A.mid = mid;
    A.dst = dst;
    A.sstride = sstride;
    A.mw = mw;
    A.mh = mh;
    A.mstride = mstride;
    A.dw = dw;
    A.dh = dh;
    A.dstride = dstride;
    A.sframe = sframe;
    A.mframe = pyrFrame;
    A.dframe = pyrFrame;
    A.xt1 = reinterpret_cast<const int2 *>(xt1);
    A.xt2 = reinterpret_cast<const int2 *>(xt2);
    A.yt1 = reinterpret_cast<const int4 *>(yt1);
    A.yt2 = reinterpret_cast<const int4 *>(yt2);
    dim3 grid(orb_xcd_grid(((dw + R2_TW - 1) / R2_TW) * ((dh + R2_TH - 1) / R2_TH), 1), B, 1);
    hipLaunchKernelGGL(k_resize2, grid, dim3(256, 1, 1), 0, s, A, orb_xcd_arg(1));
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
