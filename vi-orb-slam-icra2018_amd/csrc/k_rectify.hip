// k_rectify.hip -- SURVEY.md section 8f row 4: the steps either side of extraction.
//   k_undistort   Frame::UndistortKeyPoints (ref: src/Frame.cc:748-778) = cv::undistortPoints(mat, mat, mK,
//                 mDistCoef, cv::Mat(), mK): OpenCV 2.4 cvUndistortPoints -- per point, in double: normalise,
//                 five fixed-point iterations of the inverse distortion, x' = P x.  One thread per keypoint;
//                 every operation is individually rounded (the OpenCV build the reference links is plain
//                 x86-64: no FMA), so the result is bit-identical to the host sequence.
//   k_remap       cv::remap(src, dst, map1, map2, INTER_LINEAR) of the stereo driver (ref:
//                 Examples/Stereo/stereo_euroc.cc:136-137), CV_32FC1 maps, BORDER_CONSTANT 0: the map is taken to
//                 fixed point with 5 fractional bits (cvRound(m * 32), round half to even), the four taps are
//                 weighted with a*b*32 (a, b in 1/32) and the sum is brought back with (s + 2^14) >> 15 -- all
//                 integers.  One thread per 4 destination pixels (dword store) of up to 4 frames, so the float
//                 maps (2 x 4 bytes per pixel, shared by all frames of a batch) are read once per 4 frames; when
//                 the taps are neighbours (any rectification map) the eight taps of a source row come from one
//                 8-byte load.  Bound: HBM (1 byte read + 1 byte written per pixel).  (A variant that staged a
//                 tile's source window in LDS was measured 1.8x slower: two barriers and a min/max reduction
//                 per tile cost more than the gathers they save.)
// The maps themselves (cv::initUndistortRectifyMap, stereo_euroc.cc:96-98) are a once-per-run table built on the
// host in double (orb_init_undistort_rectify_map), like the resize tap tables.
#include "orbhip_internal.h"

struct UndistortParams {
    double k[8];
    double fx, fy, cx, cy, ifx, ify;
    double RR[9];
    int iters;
};

__global__ __launch_bounds__(256) void k_undistort(const orbhip_keypoint *__restrict__ kps,
                                                   const int32_t *__restrict__ cnt, int cap, const UndistortParams U,
                                                   orbhip_keypoint *__restrict__ out, orbhip_keypoint *__restrict__ out2)
{
    const int b = blockIdx.y, i = blockIdx.x * 256 + threadIdx.x;
    const int n = cnt ? min(cnt[b], cap) : cap;
    if (i >= n) return;
    orbhip_keypoint kp = kps[(size_t)b * cap + i];
    double x = (double)kp.x, y = (double)kp.y;
    x = __dmul_rn(__dsub_rn(x, U.cx), U.ifx);
    y = __dmul_rn(__dsub_rn(y, U.cy), U.ify);
    const double x0 = x, y0 = y;
    const double *k = U.k;
    for (int j = 0; j < U.iters; j++) {
        const double r2 = __dadd_rn(__dmul_rn(x, x), __dmul_rn(y, y));
        const double num = __dadd_rn(1.0, __dmul_rn(__dadd_rn(__dmul_rn(__dadd_rn(__dmul_rn(k[7], r2), k[6]), r2), k[5]), r2));
        const double den = __dadd_rn(1.0, __dmul_rn(__dadd_rn(__dmul_rn(__dadd_rn(__dmul_rn(k[4], r2), k[1]), r2), k[0]), r2));
        const double icdist = __ddiv_rn(num, den);
        // 2*k[2]*x*y + k[3]*(r2 + 2*x*x)   (left to right)
        const double deltaX = __dadd_rn(__dmul_rn(__dmul_rn(__dmul_rn(2.0, k[2]), x), y),
                                        __dmul_rn(k[3], __dadd_rn(r2, __dmul_rn(__dmul_rn(2.0, x), x))));
        // k[2]*(r2 + 2*y*y) + 2*k[3]*x*y
        const double deltaY = __dadd_rn(__dmul_rn(k[2], __dadd_rn(r2, __dmul_rn(__dmul_rn(2.0, y), y))),
                                        __dmul_rn(__dmul_rn(__dmul_rn(2.0, k[3]), x), y));
        x = __dmul_rn(__dsub_rn(x0, deltaX), icdist);
        y = __dmul_rn(__dsub_rn(y0, deltaY), icdist);
    }
    const double *R = U.RR;
    const double xx = __dadd_rn(__dadd_rn(__dmul_rn(R[0], x), __dmul_rn(R[1], y)), R[2]);
    const double yy = __dadd_rn(__dadd_rn(__dmul_rn(R[3], x), __dmul_rn(R[4], y)), R[5]);
    const double ww = __ddiv_rn(1.0, __dadd_rn(__dadd_rn(__dmul_rn(R[6], x), __dmul_rn(R[7], y)), R[8]));
    kp.x = (float)__dmul_rn(xx, ww);
    kp.y = (float)__dmul_rn(yy, ww);
    out[(size_t)b * cap + i] = kp;
    if (out2) out2[(size_t)b * cap + i] = kp;   // (orbhip_frame_build: the page-locked twin of the device block)
}

__device__ __forceinline__ int remap_px(const uint8_t *__restrict__ S, int sw, int sh, int sstride, float mx, float my)
{
    const int isx = __float2int_rn(__fmul_rn(mx, 32.0f)), isy = __float2int_rn(__fmul_rn(my, 32.0f));
    const int sx = max(-32768, min(32767, isx >> 5)), sy = max(-32768, min(32767, isy >> 5));
    const int fx = isx & 31, fy = isy & 31;
    int p00 = 0, p01 = 0, p10 = 0, p11 = 0;
    if ((unsigned)sx < (unsigned)(sw - 1) && (unsigned)sy < (unsigned)(sh - 1)) {
        uint16_t a, b;
        const uint8_t *p = S + (size_t)sy * sstride + sx;
        __builtin_memcpy(&a, p, 2);
        __builtin_memcpy(&b, p + sstride, 2);
        p00 = a & 255;
        p01 = a >> 8;
        p10 = b & 255;
        p11 = b >> 8;
    } else {
        const bool x0 = (unsigned)sx < (unsigned)sw, x1 = (unsigned)(sx + 1) < (unsigned)sw;
        const bool y0 = (unsigned)sy < (unsigned)sh, y1 = (unsigned)(sy + 1) < (unsigned)sh;
        if (x0 && y0) p00 = S[(size_t)sy * sstride + sx];
        if (x1 && y0) p01 = S[(size_t)sy * sstride + sx + 1];
        if (x0 && y1) p10 = S[(size_t)(sy + 1) * sstride + sx];
        if (x1 && y1) p11 = S[(size_t)(sy + 1) * sstride + sx + 1];
    }
    const int ax = 32 - fx, ay = 32 - fy;
    const int s = (p00 * ax + p01 * fx) * ay + (p10 * ax + p11 * fx) * fy;   // = sum(p * w) / 32
    return (s * 32 + (1 << 14)) >> 15;
}

// One thread produces 4 horizontally adjacent destination pixels (one dword store) of up to RM_FRAMES frames:
// the map entries (8 bytes per pixel -- four times the image bytes) are read and taken to fixed point once and
// reused for every frame of the group.  For a rectification map the four source positions are neighbours too:
// when the four x lie within 7 columns of each other on one source row pair, the eight taps of each row come
// from ONE unaligned 8-byte load (2 loads per thread and frame instead of 8); otherwise every pixel fetches its
// own taps (remap_px).
#ifndef RM_FRAMES
#define RM_FRAMES 4
#endif

__global__ __launch_bounds__(256) void k_remap(const uint8_t *__restrict__ src, int sw, int sh, int sstride,
                                               unsigned long long sframe, const float *__restrict__ mapx,
                                               const float *__restrict__ mapy, int dw, int dh,
                                               uint8_t *__restrict__ dst, int dstride, unsigned long long dframe, int B)
{
    const int x4 = (blockIdx.x * 64 + (threadIdx.x & 63)) * 4, y = blockIdx.y * 4 + (threadIdx.x >> 6);
    const int f0 = blockIdx.z * RM_FRAMES, f1 = min(f0 + RM_FRAMES, B);
    if (x4 >= dw || y >= dh) return;
    const size_t m = (size_t)y * dw + x4;
    const size_t doff = (size_t)y * dstride + x4;
    if (x4 + 3 >= dw || (dw & 3) != 0) {            // ragged right edge / unaligned map rows
        for (int f = f0; f < f1; f++)
            for (int k = 0; k < 4 && x4 + k < dw; k++)
                dst[(size_t)f * dframe + doff + k] =
                    (uint8_t)remap_px(src + (size_t)f * sframe, sw, sh, sstride, mapx[m + k], mapy[m + k]);
        return;
    }
    const float4 mx = *reinterpret_cast<const float4 *>(mapx + m), my = *reinterpret_cast<const float4 *>(mapy + m);
    const float fxs[4] = {mx.x, mx.y, mx.z, mx.w}, fys[4] = {my.x, my.y, my.z, my.w};
    int isx[4], isy[4], sx[4];
#pragma unroll
    for (int k = 0; k < 4; k++) {
        isx[k] = __float2int_rn(__fmul_rn(fxs[k], 32.0f));
        isy[k] = __float2int_rn(__fmul_rn(fys[k], 32.0f));
        sx[k] = isx[k] >> 5;
    }
    const int sy = isy[0] >> 5;
    const int lo = min(min(sx[0], sx[1]), min(sx[2], sx[3])), hi = max(max(sx[0], sx[1]), max(sx[2], sx[3]));
    const bool together = (isy[1] >> 5) == sy && (isy[2] >> 5) == sy && (isy[3] >> 5) == sy && hi - lo <= 6 && lo >= 0 &&
                          lo + 8 <= sw && sy >= 0 && sy + 1 < sh;
    const bool dword_store = ((dstride | (int)(dframe & 3)) & 3) == 0 && (((size_t)dst) & 3) == 0;
    const size_t soff = together ? (size_t)sy * sstride + lo : 0;
    if (together) {
        // all loads of the group first (independent, in flight together), then the arithmetic
        unsigned long long r0[RM_FRAMES], r1[RM_FRAMES];
#pragma unroll
        for (int j = 0; j < RM_FRAMES; j++) {
            const int f = min(f0 + j, f1 - 1);
            const uint8_t *S = src + (size_t)f * sframe + soff;
            __builtin_memcpy(&r0[j], S, 8);
            __builtin_memcpy(&r1[j], S + sstride, 8);
        }
        int wgt[4][4];
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const int fx = isx[k] & 31, fy = isy[k] & 31, ax = 32 - fx, ay = 32 - fy;
            wgt[k][0] = ax * ay;
            wgt[k][1] = fx * ay;
            wgt[k][2] = ax * fy;
            wgt[k][3] = fx * fy;
        }
#pragma unroll
        for (int j = 0; j < RM_FRAMES; j++) {
            if (f0 + j >= f1) break;
            uint32_t out = 0;
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const int sh8 = (sx[k] - lo) * 8;
                const int p00 = (int)(r0[j] >> sh8) & 255, p01 = (int)(r0[j] >> (sh8 + 8)) & 255;
                const int p10 = (int)(r1[j] >> sh8) & 255, p11 = (int)(r1[j] >> (sh8 + 8)) & 255;
                const int acc = p00 * wgt[k][0] + p01 * wgt[k][1] + p10 * wgt[k][2] + p11 * wgt[k][3];
                out |= (uint32_t)((acc * 32 + (1 << 14)) >> 15) << (8 * k);
            }
            uint8_t *D = dst + (size_t)(f0 + j) * dframe + doff;
            if (dword_store)
                *reinterpret_cast<uint32_t *>(D) = out;
            else
                for (int k = 0; k < 4; k++) D[k] = (uint8_t)(out >> (8 * k));
        }
        return;
    }
    for (int f = f0; f < f1; f++) {
        const uint8_t *S = src + (size_t)f * sframe;
        uint32_t out = 0;
#pragma unroll
        for (int k = 0; k < 4; k++) out |= (uint32_t)remap_px(S, sw, sh, sstride, fxs[k], fys[k]) << (8 * k);
        uint8_t *D = dst + (size_t)f * dframe + doff;
        if (dword_store)
            *reinterpret_cast<uint32_t *>(D) = out;
        else
            for (int k = 0; k < 4; k++) D[k] = (uint8_t)(out >> (8 * k));
    }
}

// ---- host: parameters and the once-per-run map table ----
static void undistort_params(const float *K, const float *D, int nD, const float *P, UndistortParams &U)
{
    for (int i = 0; i < 8; i++) U.k[i] = (D && i < nD) ? (double)D[i] : 0.0;
    U.fx = (double)K[0];
    U.fy = (double)K[4];
    U.cx = (double)K[2];
    U.cy = (double)K[5];
    U.ifx = 1. / U.fx;
    U.ify = 1. / U.fy;
    for (int i = 0; i < 9; i++) U.RR[i] = P ? (double)P[i] : (i % 4 == 0 ? 1.0 : 0.0);   // P * I
    U.iters = (D && nD > 0) ? 5 : 1;
}

int launch_undistort(hipStream_t s, const orbhip_keypoint *kps, const int32_t *cnt, int cap, int B, const float *K,
                     const float *D, int nD, const float *P, orbhip_keypoint *out, orbhip_keypoint *out2)
{
    UndistortParams U;
    undistort_params(K, D, nD, P, U);
    hipLaunchKernelGGL(k_undistort, dim3((cap + 255) / 256, B, 1), dim3(256, 1, 1), 0, s, kps, cnt, cap, U, out, out2);
    return ORBHIP_OK;
}

int launch_remap(hipStream_t s, const uint8_t *src, int B, int sw, int sh, int sstride, size_t sframe, const float *mapx,
                 const float *mapy, int dw, int dh, uint8_t *dst, int dstride, size_t dframe)
{
    hipLaunchKernelGGL(k_remap, dim3((dw + 255) / 256, (dh + 3) / 4, (B + RM_FRAMES - 1) / RM_FRAMES), dim3(256, 1, 1), 0, s, src,
                       sw, sh, sstride, (unsigned long long)sframe, mapx, mapy, dw, dh, dst, dstride,
                       (unsigned long long)dframe, B);
    return ORBHIP_OK;
}

// cv::initUndistortRectifyMap(K, D, R, P(0:3,0:3), size, CV_32FC1, map1, map2): OpenCV 2.4 undistort.cpp, double
// arithmetic; the row-wise running sums (_x += ir[0] ...) are kept because they determine the rounding.
void orb_init_undistort_rectify_map(const double *K, const double *D, int nD, const double *R, const double *P, int w,
                                    int h, float *mapx, float *mapy)
{
    double M[9], ir[9];
    for (int r = 0; r < 3; r++)
        for (int c = 0; c < 3; c++) {
            double acc = 0;
            for (int k = 0; k < 3; k++) acc += P[r * 3 + k] * R[k * 3 + c];
            M[r * 3 + c] = acc;
        }
    auto m = [&](int y, int x) { return M[y * 3 + x]; };
    double det = m(0, 0) * (m(1, 1) * m(2, 2) - m(1, 2) * m(2, 1)) - m(0, 1) * (m(1, 0) * m(2, 2) - m(1, 2) * m(2, 0)) +
                 m(0, 2) * (m(1, 0) * m(2, 1) - m(1, 1) * m(2, 0));
    det = 1. / det;   // cv::invert, 3x3: cofactors times the reciprocal determinant
    ir[0] = (m(1, 1) * m(2, 2) - m(1, 2) * m(2, 1)) * det;
    ir[1] = (m(0, 2) * m(2, 1) - m(0, 1) * m(2, 2)) * det;
    ir[2] = (m(0, 1) * m(1, 2) - m(0, 2) * m(1, 1)) * det;
    ir[3] = (m(1, 2) * m(2, 0) - m(1, 0) * m(2, 2)) * det;
    ir[4] = (m(0, 0) * m(2, 2) - m(0, 2) * m(2, 0)) * det;
    ir[5] = (m(0, 2) * m(1, 0) - m(0, 0) * m(1, 2)) * det;
    ir[6] = (m(1, 0) * m(2, 1) - m(1, 1) * m(2, 0)) * det;
    ir[7] = (m(0, 1) * m(2, 0) - m(0, 0) * m(2, 1)) * det;
    ir[8] = (m(0, 0) * m(1, 1) - m(0, 1) * m(1, 0)) * det;
    const double u0 = K[2], v0 = K[5], fx = K[0], fy = K[4];
    const double k1 = nD > 0 ? D[0] : 0, k2 = nD > 1 ? D[1] : 0, p1 = nD > 2 ? D[2] : 0, p2 = nD > 3 ? D[3] : 0;
    const double k3 = nD >= 5 ? D[4] : 0, k4 = nD >= 8 ? D[5] : 0, k5 = nD >= 8 ? D[6] : 0, k6 = nD >= 8 ? D[7] : 0;
    for (int i = 0; i < h; i++) {
        double _x = i * ir[1] + ir[2], _y = i * ir[4] + ir[5], _w = i * ir[7] + ir[8];
        for (int j = 0; j < w; j++, _x += ir[0], _y += ir[3], _w += ir[6]) {
            const double ww = 1. / _w, x = _x * ww, y = _y * ww;
            const double x2 = x * x, y2 = y * y;
            const double r2 = x2 + y2, _2xy = 2 * x * y;
            const double kr = (1 + ((k3 * r2 + k2) * r2 + k1) * r2) / (1 + ((k6 * r2 + k5) * r2 + k4) * r2);
            mapx[(size_t)i * w + j] = (float)(fx * (x * kr + p1 * _2xy + p2 * (r2 + 2 * x2)) + u0);
            mapy[(size_t)i * w + j] = (float)(fy * (y * kr + p1 * (r2 + 2 * y2) + p2 * _2xy) + v0);
        }
    }
}
