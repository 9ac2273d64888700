// k_pyramid.hip -- E2: one pyramid level from the previous one, bit-exact cv::resize INTER_LINEAR
// 8UC1 (ref call site: src/ORBextractor.cc:1141).  The 19-px border of the reference
// (copyMakeBorder :1143-1149) is never read by later stages and is not produced.
//
// One thread produces 4 horizontally adjacent output pixels and stores them as one dword; the
// column/row tap tables are built on the host (orb_build_resize_tables).  Rows of one level are
// 64-byte aligned so every store is aligned.  Bound: HBM/L2 streaming (reads 1.44 px and writes
// 1 px per output pixel).
#include "orbhip_internal.h"

__global__ __launch_bounds__(256) void k_resize(const uint8_t *__restrict__ src, int sstride,
                                                unsigned long long sframe, uint8_t *__restrict__ dst,
                                                int dw, int dh, int dstride, unsigned long long dframe,
                                                const int2 *__restrict__ xtab,
                                                const int4 *__restrict__ ytab)
{
    const int gx = (blockIdx.x * 64 + threadIdx.x) * 4;
    const int dy = blockIdx.y * 4 + threadIdx.y;
    if (gx >= dw || dy >= dh) return;
    const uint8_t *S = src + (size_t)blockIdx.z * sframe;
    uint8_t *D = dst + (size_t)blockIdx.z * dframe;
    const int4 yt = ytab[dy];
    const uint8_t *S0 = S + (size_t)yt.x * sstride;
    const uint8_t *S1 = S + (size_t)yt.y * sstride;
    const int b0 = yt.z, b1 = yt.w;
    uint32_t packed = 0;
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const int dx = gx + k;
        if (dx < dw) {
            const int2 xt = xtab[dx];
            const int sx0 = xt.x & 0xFFFF, sx1 = (unsigned)xt.x >> 16;
            const int a0 = (short)(xt.y & 0xFFFF), a1 = xt.y >> 16;
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

void launch_resize(hipStream_t s, const uint8_t *src, int sw, int sh, int sstride, size_t sframe,
                   uint8_t *dst, int dw, int dh, int dstride, size_t dframe, const int32_t *xtab,
                   const int32_t *ytab, int B)
{
    (void)sw;
    (void)sh;
    dim3 block(64, 4, 1);
    dim3 grid((dw + 255) / 256, (dh + 3) / 4, B);
    hipLaunchKernelGGL(k_resize, grid, block, 0, s, src, sstride, (unsigned long long)sframe, dst, dw, dh,
                       dstride, (unsigned long long)dframe, reinterpret_cast<const int2 *>(xtab),
                       reinterpret_cast<const int4 *>(ytab));
}
