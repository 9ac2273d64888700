// k_describe.hip -- E5 + E7 + E8 + output assembly: one 64-lane wave per retained keypoint
// computes the intensity-centroid angle on the un-blurred level (ref: src/ORBextractor.cc:79-106),
// the steered 256-bit BRIEF descriptor on the blurred level (:109-149), scales the coordinates
// (:1113-1119) and writes the cv::KeyPoint record and the descriptor row at the keypoint's final
// position (levels concatenated in order 0..n-1, :1094-1122).
//
// Float reproducibility (SURVEY.md "hard parts"):
//  * fastAtan2 (OpenCV 2.4 mathfuncs.cpp) is evaluated with explicitly rounded float ops
//    (__fmul_rn/__fadd_rn/__fdiv_rn: no FMA contraction).
//  * cos/sin of the angle: the reference calls libm's cosf/sinf (src/ORBextractor.cc:115).  The
//    device evaluates the same published algorithm (glibc >= 2.28 s_sincosf: double-precision
//    range reduction by pi/2 and two minimax polynomials) as a fixed sequence of IEEE double
//    multiplies/adds.  That sequence was checked against glibc 2.35 cosf/sinf for EVERY float in
//    [0, 2*pi*1.01] (1 087 050 388 inputs, 0 mismatches; tests/test_trig.py re-checks a sample and
//    all angles k/64 degrees), so a = cos(angle), b = sin(angle) are bit-identical to the host.
//  * x*b + y*a is two products and one sum, each rounded (no contraction) -- Appendix A7.
#include "orbhip_internal.h"
#include "orb_pattern_data.h"
#include "orb_trig.h"

__constant__ __attribute__((aligned(16))) signed char c_pattern[1024] = {ORB_PATTERN_VALUES};

__device__ __forceinline__ float fast_atan2_dev(float y, float x)
{
    const float rad2deg = (float)(180.0 / 3.1415926535897932384626433832795);
    const float p1 = 0.9997878412794807f * rad2deg;
    const float p3 = -0.3258083974640975f * rad2deg;
    const float p5 = 0.1555786518463281f * rad2deg;
    const float p7 = -0.04432655554792128f * rad2deg;
    const float eps = (float)2.2204460492503131e-16;
    const float ax = fabsf(x), ay = fabsf(y);
    float a, c, c2, t;
    if (ax >= ay) {
        c = __fdiv_rn(ay, __fadd_rn(ax, eps));
        c2 = __fmul_rn(c, c);
        t = __fadd_rn(__fmul_rn(p7, c2), p5);
        t = __fadd_rn(__fmul_rn(t, c2), p3);
        t = __fadd_rn(__fmul_rn(t, c2), p1);
        a = __fmul_rn(t, c);
    } else {
        c = __fdiv_rn(ax, __fadd_rn(ay, eps));
        c2 = __fmul_rn(c, c);
        t = __fadd_rn(__fmul_rn(p7, c2), p5);
        t = __fadd_rn(__fmul_rn(t, c2), p3);
        t = __fadd_rn(__fmul_rn(t, c2), p1);
        a = __fsub_rn(90.f, __fmul_rn(t, c));
    }
    if (x < 0) a = __fsub_rn(180.f, a);
    if (y < 0) a = __fsub_rn(360.f, a);
    return a;
}

// Sum over the 64 lanes, result wave-uniform.  DPP inside each row of 16 lanes (quad swaps, half
// mirror, mirror), then the four row sums through v_readlane: no LDS round trips.
__device__ __forceinline__ int wave_sum(int v)
{
    v += __builtin_amdgcn_update_dpp(0, v, 0xB1, 0xF, 0xF, true);    // quad_perm [1,0,3,2]
    v += __builtin_amdgcn_update_dpp(0, v, 0x4E, 0xF, 0xF, true);    // quad_perm [2,3,0,1]
    v += __builtin_amdgcn_update_dpp(0, v, 0x141, 0xF, 0xF, true);   // row_half_mirror
    v += __builtin_amdgcn_update_dpp(0, v, 0x140, 0xF, 0xF, true);   // row_mirror
    return __builtin_amdgcn_readlane(v, 0) + __builtin_amdgcn_readlane(v, 16) + __builtin_amdgcn_readlane(v, 32) +
           __builtin_amdgcn_readlane(v, 48);
}

__global__ __launch_bounds__(256) void k_describe(const OrbLevels G, const uint8_t *__restrict__ lvl0,
                                                  int stride0, unsigned long long frame0,
                                                  const uint8_t *__restrict__ pyr,
                                                  unsigned long long pyrFrame,
                                                  const uint8_t *__restrict__ blur,
                                                  unsigned long long blurFrame,
                                                  const uint32_t *__restrict__ lvlKp,
                                                  const int32_t *__restrict__ lvlKpCnt,
                                                  float *__restrict__ lvlAngle,
                                                  orbhip_keypoint *__restrict__ kps,
                                                  uint8_t *__restrict__ desc, int32_t *__restrict__ counts,
                                                  int cap, int xcdMap)
{
    const int blk = xcd_tile(xcdMap), frame = blockIdx.y;   // consecutive keypoints of a level share an XCD's L2
    const int lane = threadIdx.x & 63;
    const int g = blk * 4 + (threadIdx.x >> 6);  // slot in the per-frame level-keypoint array
    if (g >= G.totalKps) return;
    const int32_t *cnts = lvlKpCnt + frame * ORBHIP_MAX_LEVELS;
    // level of this slot and output offset of the level
    int l = 0, off = 0, total = 0;
    for (int k = 0; k < G.nlevels; k++) {
        const int c = cnts[k];
        if (g >= G.lv[k].kpBase) {
            l = k;
            off = total;
        }
        total += c;
    }
    if (g == 0 && lane == 0) counts[frame] = total;
    const OrbLevel &L = G.lv[l];
    const int i = g - L.kpBase;
    if (i >= cnts[l]) return;
    const uint32_t pk = lvlKp[(size_t)frame * G.totalKps + g];
    const int cx = (int)(pk & 0xFFFu) + ORB_MIN_BORDER;         // :843-844
    const int cy = (int)((pk >> 12) & 0xFFFu) + ORB_MIN_BORDER;
    const int score = (int)(pk >> 24);

    // ---- E5: IC_Angle on the un-blurred level ----
    const uint8_t *img;
    int stride;
    if (l == 0) {
        img = lvl0 + (size_t)frame * frame0;
        stride = stride0;
    } else {
        img = pyr + (size_t)frame * pyrFrame + L.imgOff;
        stride = L.stride;
    }
    // The disc rows are read as aligned dwords: lane -> (row of a group of 7, dword 0..8 of the row),
    // 5 trips cover the 31 rows; a row needs at most 31 + 3 bytes = 9 dwords.  (A byte gather costs the
    // texture-address unit 16 cycles per wave instruction; this is 5 coalesced instructions, not 16.)
    int m10 = 0, m01 = 0;
    {
        const int rsub = lane / 9, di = lane - rsub * 9;
        const int xs = ((cx - ORB_HALF_PATCH) & ~3) + 4 * di;     // image column of byte 0 of this lane's dword
        const int u0 = xs - cx;
#pragma unroll
        for (int it = 0; it < 5; it++) {
            const int v = -ORB_HALF_PATCH + it * 7 + rsub;
            if (lane < 63 && v <= ORB_HALF_PATCH) {
                const int d = G.umax[v < 0 ? -v : v];
                const uint32_t wd = *reinterpret_cast<const uint32_t *>(img + (size_t)(cy + v) * stride + xs);
                int rs = 0, ru = 0;
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    const int u = u0 + k;
                    const int val = (u >= -d && u <= d) ? (int)((wd >> (8 * k)) & 0xFF) : 0;
                    rs += val;
                    ru += u * val;
                }
                m10 += ru;
                m01 += v * rs;
            }
        }
    }
    m10 = wave_sum(m10);
    m01 = wave_sum(m01);
    const float angle = fast_atan2_dev((float)m01, (float)m10);

    // ---- E7: steered BRIEF on the blurred level ----
    const uint8_t *bimg = blur + (size_t)frame * blurFrame + (l == 0 ? 0ull : G.boff1 + L.imgOff);
    const int bstride = l == 0 ? G.bstride0 : L.stride;
    const uint8_t *bc = bimg + (size_t)cy * bstride + cx;
    const float factorPI = (float)(3.1415926535897932384626433832795 / 180.f);
    const float rad = __fmul_rn(angle, factorPI);
    float a, b;
    orb_sincosf(rad, &b, &a);
    const int o = off + i;
    unsigned long long words[4];
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const int t = 64 * j + lane;
        const int pw = reinterpret_cast<const int *>(c_pattern)[t];
        const float x0 = (float)(signed char)(pw & 0xFF), y0 = (float)(signed char)((pw >> 8) & 0xFF);
        const float x1 = (float)(signed char)((pw >> 16) & 0xFF), y1 = (float)(signed char)((pw >> 24) & 0xFF);
        const int r0 = __float2int_rn(__fadd_rn(__fmul_rn(x0, b), __fmul_rn(y0, a)));
        const int c0 = __float2int_rn(__fsub_rn(__fmul_rn(x0, a), __fmul_rn(y0, b)));
        const int r1 = __float2int_rn(__fadd_rn(__fmul_rn(x1, b), __fmul_rn(y1, a)));
        const int c1 = __float2int_rn(__fsub_rn(__fmul_rn(x1, a), __fmul_rn(y1, b)));
        const int t0 = bc[r0 * bstride + c0], t1 = bc[r1 * bstride + c1];
        words[j] = __ballot(t0 < t1);
    }
    lvlAngle[(size_t)frame * G.totalKps + g] = angle;
    if (o < cap) {
        if (lane < 4) {
            // bit (64j + lane) of the descriptor = test 64j+lane, LSB first inside each byte (:128-145)
            const unsigned long long wsel = lane == 0 ? words[0] : (lane == 1 ? words[1] : (lane == 2 ? words[2] : words[3]));
            reinterpret_cast<unsigned long long *>(desc + ((size_t)frame * cap + o) * 32)[lane] = wsel;
        }
        if (lane == 0) {
            orbhip_keypoint kp;
            kp.x = __fmul_rn((float)cx, L.scale);   // level 0: scale == 1.0f, identical to "no scaling"
            kp.y = __fmul_rn((float)cy, L.scale);
            kp.size = L.kpSize;
            kp.angle = angle;
            kp.response = (float)score;
            kp.octave = l;
            kp.class_id = -1;
            kps[(size_t)frame * cap + o] = kp;
        }
    }
}

void launch_describe(hipStream_t s, const OrbLevels &G, const uint8_t *lvl0, int stride0, size_t frame0,
                     const uint8_t *pyr, size_t pyrFrame, const uint8_t *blur, size_t blurFrame,
                     const uint32_t *lvlKp, const int32_t *lvlKpCnt, float *lvlAngle,
                     orbhip_keypoint *kps, uint8_t *desc, int32_t *counts, int cap, int B)
{
    dim3 grid(orb_xcd_grid((G.totalKps + 3) / 4), B, 1), block(256, 1, 1);
    hipLaunchKernelGGL(k_describe, grid, block, 0, s, G, lvl0, stride0, (unsigned long long)frame0, pyr,
                       (unsigned long long)pyrFrame, blur, (unsigned long long)blurFrame, lvlKp, lvlKpCnt,
                       lvlAngle, kps, desc, counts, cap, orb_xcd_arg());
}
