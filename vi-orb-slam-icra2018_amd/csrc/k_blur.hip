// k_blur.hip -- E6: cv::GaussianBlur(level, 7x7, sigma 2, BORDER_REFLECT_101) for every level of
// every frame in one launch (ref call site: src/ORBextractor.cc:1103-1104; OpenCV 2.4 smooth.cpp
// / filter.cpp fixed-point separable filter: kernel {18,34,49,55,49,34,18}/256 applied twice,
// sum brought back by 2^16).
//
// Rounding of the column pass follows the x86-64 (SSE2) build of OpenCV 2.4 that the reference
// links: for x < w - w%4 the 32s->8u symmetric column filter accumulates in float and converts
// with cvtps2dq (round half to EVEN); the scalar tail uses (sum + 32768) >> 16 (half UP).  All
// float products/sums here are exact below 256, so the float path is evaluated as an integer
// tie-break rule -- DESIGN.md "blur".
//
// One 256-thread workgroup per 64x16 output tile: raw (70x22) -> LDS, row pass -> LDS int32
// (64x22), column pass -> one dword store per thread.  Bound: HBM (read + write one byte per
// pixel).
#include "orbhip_internal.h"

#define BT_W 64
#define BT_H 16

__device__ __forceinline__ int reflect101(int p, int len)
{
    // BORDER_REFLECT_101: gfedcb|abcdefgh|gfedcba ; |excursion| <= 3 < len
    if (p < 0) p = -p;
    if (p >= len) p = 2 * len - 2 - p;
    return p;
}

__global__ __launch_bounds__(256) void k_blur(const OrbLevels G, const uint8_t *__restrict__ lvl0,
                                              int stride0, unsigned long long frame0,
                                              const uint8_t *__restrict__ pyr, unsigned long long pyrFrame,
                                              uint8_t *__restrict__ blur, unsigned long long blurFrame,
                                              const BlurTile *__restrict__ tiles, int4 kq)
{
    __shared__ uint8_t s_raw[BT_H + 6][BT_W + 8];
    __shared__ int s_row[BT_H + 6][BT_W];
    const BlurTile T = tiles[blockIdx.x];
    const int frame = blockIdx.y, l = T.level;
    const OrbLevel &L = G.lv[l];
    const int w = L.w, h = L.h;
    const uint8_t *src;
    int sstride;
    if (l == 0) {
        src = lvl0 + (size_t)frame * frame0;
        sstride = stride0;
    } else {
        src = pyr + (size_t)frame * pyrFrame + L.imgOff;
        sstride = L.stride;
    }
    uint8_t *dst = blur + (size_t)frame * blurFrame + (l == 0 ? 0ull : G.boff1 + L.imgOff);
    const int dstride = l == 0 ? G.bstride0 : L.stride;
    const int x0 = T.tx * BT_W, y0 = T.ty * BT_H;
    const int tid = threadIdx.x;

    for (int i = tid; i < (BT_H + 6) * (BT_W + 6); i += 256) {
        const int r = i / (BT_W + 6), c = i - r * (BT_W + 6);
        const int sy = reflect101(min(y0 - 3 + r, h + 2), h);
        const int sx = reflect101(min(x0 - 3 + c, w + 2), w);
        s_raw[r][c] = src[(size_t)sy * sstride + sx];
    }
    __syncthreads();
    const int k0 = kq.x, k1 = kq.y, k2 = kq.z, k3 = kq.w;  // 18 34 49 55
    for (int i = tid; i < (BT_H + 6) * BT_W; i += 256) {
        const int r = i >> 6, c = i & 63;
        const uint8_t *p = &s_raw[r][c];
        s_row[r][c] = k0 * (p[0] + p[6]) + k1 * (p[1] + p[5]) + k2 * (p[2] + p[4]) + k3 * p[3];
    }
    __syncthreads();
    const int r = tid >> 4, cb = (tid & 15) << 2;
    const int y = y0 + r;
    if (y >= h) return;
    const int wvec = w - (w & 3);
    uint32_t packed = 0;
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const int c = cb + k, x = x0 + c;
        const int s = k0 * (s_row[r][c] + s_row[r + 6][c]) + k1 * (s_row[r + 1][c] + s_row[r + 5][c]) +
                      k2 * (s_row[r + 2][c] + s_row[r + 4][c]) + k3 * s_row[r + 3][c];
        int v = (s + 32768) >> 16;                                   // round half up
        if (x < wvec && (s & 0xFFFF) == 0x8000 && (v & 1)) v -= 1;   // SSE2 body: ties to even
        v = v > 255 ? 255 : v;
        packed |= (uint32_t)v << (8 * k);
    }
    uint8_t *o = dst + (size_t)y * dstride + x0 + cb;
    if (x0 + cb + 3 < w)
        *reinterpret_cast<uint32_t *>(o) = packed;
    else
        for (int k = 0; k < 4 && x0 + cb + k < w; k++) o[k] = (uint8_t)(packed >> (8 * k));
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

void launch_blur(hipStream_t s, const OrbLevels &G, const uint8_t *lvl0, int stride0, size_t frame0,
                 const uint8_t *pyr, size_t pyrFrame, uint8_t *blur, size_t blurFrame,
                 const BlurTile *tiles, int ntiles, int B)
{
    int k[4];
    gaussian_taps(k);
    dim3 grid(ntiles, B, 1), block(256, 1, 1);
    hipLaunchKernelGGL(k_blur, grid, block, 0, s, G, lvl0, stride0, (unsigned long long)frame0, pyr,
                       (unsigned long long)pyrFrame, blur, (unsigned long long)blurFrame, tiles,
                       make_int4(k[0], k[1], k[2], k[3]));
}
