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
// One 256-thread workgroup per 128x32 output tile:
//   1. raw rows y0-3 .. y0+34 (reflected at the top/bottom), columns x0-16 .. x0+143, into LDS with
//      16-byte row-coalesced loads; at the left/right image edge the 3 reflected columns are
//      patched into the halo;
//   2. row pass: v_dot4_u32_u8 on byte windows cut out with v_alignbyte (7 taps = 2 dot4); the sums
//      are at most 257*255 = 65535 and two vertically adjacent rows share one LDS dword;
//   3. column pass: 7 taps = 4 v_dot2_u32_u16 on those row pairs, 4 adjacent outputs per thread
//      (one dword store).  Integer dot instructions on the vector ALU, not MFMA.
// Bound: HBM (reads and writes one byte per pixel; halo re-reads 38/32 x 160/128).
#include "orbhip_internal.h"

#define BT_W BLUR_TILE_W
#define BT_H BLUR_TILE_H
#define BT_RAWP (BT_W + 32)       // raw pitch: 16 halo bytes each side (16-byte aligned chunks)
#define BT_ROWS (BT_H + 6)       // 38: even, rows are processed in pairs

typedef unsigned short us2 __attribute__((ext_vector_type(2)));

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
                                              const BlurTile *__restrict__ tiles, int4 kq, int xcdMap, int ntiles)
{
    __shared__ __align__(16) uint8_t s_raw[BT_ROWS][BT_RAWP];
    __shared__ __align__(16) uint32_t s_pair[BT_ROWS / 2][BT_W];
    const int tileId = xcd_tile(xcdMap), frame = blockIdx.y;
    if (tileId >= ntiles) return;   // grid padded to a multiple of 8 (orbhip_internal.h, xcd_tile)
    const BlurTile T = tiles[tileId];
    const int l = T.level;
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

    // ---- 1. raw tile: LDS column j <-> image column x0 - 16 + j ----
    const int wAl = (w + 15) & ~15;   // bytes of a row that may be read with 16-byte loads
    // both 16-byte loads of a thread are issued before either is stored (unconditional loads from a clamped
    // address, zeroed afterwards when the chunk lies outside the row: a conditional load would be waited for
    // before the next one is issued)
    {
        constexpr int NCH = BT_RAWP / 16, NIT = (BT_ROWS * NCH + 255) / 256;
        uint4 v[NIT];
        bool ok[NIT];
#pragma unroll
        for (int k = 0; k < NIT; k++) {
            const int i = min(tid + k * 256, BT_ROWS * NCH - 1);
            const int r = i / NCH, c = i - r * NCH;
            const int sy = reflect101(min(y0 - 3 + r, h + 2), h);
            const int sx = x0 - 16 + (c << 4);
            ok[k] = sx >= 0 && sx < wAl;
            v[k] = *reinterpret_cast<const uint4 *>(src + (size_t)sy * sstride + min(max(sx, 0), wAl - 16));
        }
#pragma unroll
        for (int k = 0; k < NIT; k++) {
            const int i = tid + k * 256;
            if (i < BT_ROWS * NCH) {
                const int r = i / NCH, c = i - r * NCH;
                *reinterpret_cast<uint4 *>(&s_raw[r][c << 4]) = ok[k] ? v[k] : make_uint4(0, 0, 0, 0);
            }
        }
    }
    __syncthreads();
    // reflected halo columns at the image edges (x = -1,-2,-3 <- 1,2,3 ; x = w,w+1,w+2 <- w-2,w-3,w-4)
    if (x0 == 0) {
        for (int i = tid; i < BT_ROWS * 3; i += 256) {
            const int r = i / 3, k = i - r * 3 + 1;
            s_raw[r][16 - k] = s_raw[r][16 + k];
        }
    }
    if (x0 + BT_W + 3 > w) {
        for (int i = tid; i < BT_ROWS * 3; i += 256) {
            const int r = i / 3, k = i - r * 3;
            s_raw[r][16 + (w + k - x0)] = s_raw[r][16 + (w - 2 - k - x0)];
        }
    }
    __syncthreads();

    // ---- 2. row pass with v_dot4_u32_u8: an item = (pair of raw rows, 4 adjacent columns) ----
    // the two row sums of a column (each <= 257*255 = 65535) are packed into one dword:
    // s_pair[rp][x] = H[2rp][x] | H[2rp+1][x] << 16, so that the column pass can use v_dot2_u32_u16
    const uint32_t k0 = kq.x, k1 = kq.y, k2 = kq.z, k3 = kq.w;  // 18 34 49 55
    const uint32_t wlo = k0 | (k1 << 8) | (k2 << 16) | (k3 << 24);   // taps -3..0
    const uint32_t whi = k2 | (k1 << 8) | (k0 << 16);                // taps +1..+3
    for (int i = tid; i < (BT_ROWS / 2) * (BT_W / 4); i += 256) {
        const int rp = i >> 5, g = i & 31;
        uint32_t h[2][4];
#pragma unroll
        for (int q = 0; q < 2; q++) {
            const uint32_t *p = reinterpret_cast<const uint32_t *>(&s_raw[2 * rp + q][16 + (g << 2)]);
            const uint32_t A = p[-1], Bw = p[0], C = p[1];   // columns x-4..x-1 | x..x+3 | x+4..x+7
            // pixel k: taps are bytes k+1..k+7 of the 12-byte stream A|B|C
            h[q][0] = __builtin_amdgcn_udot4(__builtin_amdgcn_alignbyte(Bw, A, 1), wlo,
                                             __builtin_amdgcn_udot4(__builtin_amdgcn_alignbyte(C, Bw, 1), whi, 0u, false), false);
            h[q][1] = __builtin_amdgcn_udot4(__builtin_amdgcn_alignbyte(Bw, A, 2), wlo,
                                             __builtin_amdgcn_udot4(__builtin_amdgcn_alignbyte(C, Bw, 2), whi, 0u, false), false);
            h[q][2] = __builtin_amdgcn_udot4(__builtin_amdgcn_alignbyte(Bw, A, 3), wlo,
                                             __builtin_amdgcn_udot4(__builtin_amdgcn_alignbyte(C, Bw, 3), whi, 0u, false), false);
            h[q][3] = __builtin_amdgcn_udot4(Bw, wlo, __builtin_amdgcn_udot4(C, whi, 0u, false), false);
        }
        uint4 o;
        o.x = h[0][0] | (h[1][0] << 16);
        o.y = h[0][1] | (h[1][1] << 16);
        o.z = h[0][2] | (h[1][2] << 16);
        o.w = h[0][3] | (h[1][3] << 16);
        *reinterpret_cast<uint4 *>(&s_pair[rp][g << 2]) = o;
    }
    __syncthreads();

    // ---- 3. column pass with v_dot2_u32_u16: an item = (two output rows, 4 adjacent columns) ----
    // Output rows r (even) and r + 1 read the same four row pairs (r,r+1)(r+2,r+3)(r+4,r+5)(r+6,r+7) with the
    // constant weight pairs (k0,k1)(k2,k3)(k2,k1)(k0,0) and (0,k0)(k1,k2)(k3,k2)(k1,k0): four LDS reads serve
    // eight outputs and no weight depends on the lane.
    const int wvec = w - (w & 3);
    const us2 wa0 = {(unsigned short)k0, (unsigned short)k1}, wa1 = {(unsigned short)k2, (unsigned short)k3};
    const us2 wa2 = {(unsigned short)k2, (unsigned short)k1}, wa3 = {(unsigned short)k0, 0};
    const us2 wb0 = {0, (unsigned short)k0}, wb1 = {(unsigned short)k1, (unsigned short)k2};
    const us2 wb2 = {(unsigned short)k3, (unsigned short)k2}, wb3 = {(unsigned short)k1, (unsigned short)k0};
    for (int i = tid; i < (BT_H / 2) * (BT_W / 4); i += 256) {
        const int rp = i >> 5, g = i & 31;
        const int y = y0 + 2 * rp, xb = x0 + (g << 2);
        if (y >= h || xb >= w) continue;
        const uint4 q0 = *reinterpret_cast<const uint4 *>(&s_pair[rp][g << 2]);
        const uint4 q1 = *reinterpret_cast<const uint4 *>(&s_pair[rp + 1][g << 2]);
        const uint4 q2 = *reinterpret_cast<const uint4 *>(&s_pair[rp + 2][g << 2]);
        const uint4 q3 = *reinterpret_cast<const uint4 *>(&s_pair[rp + 3][g << 2]);
        const uint32_t c0[4] = {q0.x, q0.y, q0.z, q0.w}, c1[4] = {q1.x, q1.y, q1.z, q1.w};
        const uint32_t c2[4] = {q2.x, q2.y, q2.z, q2.w}, c3[4] = {q3.x, q3.y, q3.z, q3.w};
        uint32_t sa[4], sb[4];
#pragma unroll
        for (int k = 0; k < 4; k++) {
            sa[k] = __builtin_amdgcn_udot2(__builtin_bit_cast(us2, c0[k]), wa0, 0u, false);
            sa[k] = __builtin_amdgcn_udot2(__builtin_bit_cast(us2, c1[k]), wa1, sa[k], false);
            sa[k] = __builtin_amdgcn_udot2(__builtin_bit_cast(us2, c2[k]), wa2, sa[k], false);
            sa[k] = __builtin_amdgcn_udot2(__builtin_bit_cast(us2, c3[k]), wa3, sa[k], false);
            sb[k] = __builtin_amdgcn_udot2(__builtin_bit_cast(us2, c0[k]), wb0, 0u, false);
            sb[k] = __builtin_amdgcn_udot2(__builtin_bit_cast(us2, c1[k]), wb1, sb[k], false);
            sb[k] = __builtin_amdgcn_udot2(__builtin_bit_cast(us2, c2[k]), wb2, sb[k], false);
            sb[k] = __builtin_amdgcn_udot2(__builtin_bit_cast(us2, c3[k]), wb3, sb[k], false);
        }
        // SSE2 body (x < wvec): round half to even = (s + 0x7FFF + bit16(s)) >> 16;
        // scalar tail: round half up = (s + 0x8000) >> 16
        uint32_t packedA = 0, packedB = 0;
        if (xb + 3 < wvec) {   // every pixel of the group is in the body (all but the last group of a row)
            // exactly what the SSE2 body does: the sum as a float (exact below 2^24) times 2^-16, converted with round
            // half to even and packed with unsigned saturation -- v_cvt_pk_u8_f32 does the last two in one instruction
#pragma unroll
            for (int k = 0; k < 4; k++) {
                packedA = __builtin_amdgcn_cvt_pk_u8_f32((float)sa[k] * (1.0f / 65536.0f), k, packedA);
                packedB = __builtin_amdgcn_cvt_pk_u8_f32((float)sb[k] * (1.0f / 65536.0f), k, packedB);
            }
        } else {
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const bool even = xb + k < wvec;
                const uint32_t ba = even ? 0x7FFFu + ((sa[k] >> 16) & 1u) : 0x8000u;
                const uint32_t bb = even ? 0x7FFFu + ((sb[k] >> 16) & 1u) : 0x8000u;
                packedA |= min((sa[k] + ba) >> 16, 255u) << (8 * k);
                packedB |= min((sb[k] + bb) >> 16, 255u) << (8 * k);
            }
        }
        uint8_t *o = dst + (size_t)y * dstride + xb;
        if (xb + 3 < w) {
            *reinterpret_cast<uint32_t *>(o) = packedA;
            if (y + 1 < h) *reinterpret_cast<uint32_t *>(o + dstride) = packedB;
        } else {
            for (int k = 0; k < 4 && xb + k < w; k++) {
                o[k] = (uint8_t)(packedA >> (8 * k));
                if (y + 1 < h) o[dstride + k] = (uint8_t)(packedB >> (8 * k));
            }
        }
    }
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
    dim3 grid(orb_xcd_grid(ntiles, 2), B, 1), block(256, 1, 1);
    hipLaunchKernelGGL(k_blur, grid, block, 0, s, G, lvl0, stride0, (unsigned long long)frame0, pyr,
                       (unsigned long long)pyrFrame, blur, (unsigned long long)blurFrame, tiles,
                       make_int4(k[0], k[1], k[2], k[3]), orb_xcd_arg(2), ntiles);
}
