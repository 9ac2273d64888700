// k_describe.hip -- E5 + E7 + E8 + output assembly: per retained keypoint the intensity-centroid angle
// on the un-blurred level (ref: src/ORBextractor.cc:79-106),
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

#include <cstdlib>

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

// One 256-thread workgroup per DS_KP slots (16 for batches) of the per-frame level-keypoint array:
//   0. thread per slot: level, position, output index (levels concatenated in order, :1094-1122);
//   A. wave per keypoint (each wave walks 16 slots): the disc rows as aligned dwords, lane -> (row of a group
//      of 7, dword 0..8 of the row), 5 trips cover the 31 rows; valid bytes are cut out with a mask and summed
//      with v_sad_u8 / v_dot4 (sum of bytes, sum of k * byte), then two DPP wave sums;
//   B. thread per keypoint: fastAtan2, cos / sin in double, the cv::KeyPoint record.  The scalar float / double
//      sequence is evaluated once per keypoint by one lane instead of once per wave by 64 lanes;
//   C. wave per keypoint again: the lane's four test pairs are decoded once (16 floats), then per keypoint 4 x
//      (rotate, round, two byte gathers from the blurred level, compare, ballot).
// DS_KP = slots per workgroup: 16 for batches (see DESCRIBE_DEFAULT_MAP below), 8 for a single frame or two (more
// workgroups, shorter serial chains per wave: latency); 32 remains for A/B runs (ORBHIP_DESCRIBE_KPW)
#define DS_R 18                       // pattern radius <= 18.385, so a rotated, rounded coordinate is at most 18
#define DS_ROWS (2 * DS_R + 1)        // 37 patch rows
#define DS_PDW 10                     // dwords per staged row: 37 bytes + up to 3 bytes of alignment
#define DS_TRIPS ((DS_ROWS * DS_PDW + 63) / 64)   // 6

// Default placement: a whole frame per XCD (xcd_frame_tile) with 16 keypoint slots per workgroup, so that the workgroups
// resident on one XCD at a time (224 of them) work on three or four frames and the raw + blurred pyramids of those frames
// (1.9 MB each) are served by that XCD's 4 MB L2: L2 hit rate 51 % -> ~70 %, fabric reads 5.1 GB -> ~2.4 GB per 1024-frame
// launch, 0.79 -> 0.74 ms (profiles/r02a/describe.md).  Fewer slots per workgroup cut the traffic further (8: 1.95 GB =
// every pyramid byte once) but cost more in per-workgroup set-up than they gain: the kernel is bound by the number of
// 128-byte lines its row gathers touch (~88 per keypoint), not by where they come from.
#define DESCRIBE_DEFAULT_MAP 4
typedef const __attribute__((address_space(3))) uint8_t *lds_u8p;

// Angle phase, batches: the whole 31 x 31 disc in ONE load instruction -- lane (row = lane >> 1, half = lane & 1) takes the 16 bytes
// from column cx - 15 + 16 half of disc row `row` (byte-aligned dwordx4; 32 adjacent bytes per row) -- and per-lane constant byte
// weights instead of per-keypoint masks: the load starts at the disc's own first column, so which bytes of a lane lie inside the
// disc, and their u, never change.  wu = u + 16 inside the disc (1..31), 0 outside; wm = 1 inside, 0 outside:
//   S = sum_in p = dot4(p, wm),  m10 = sum_in u p = dot4(p, wu) - 16 S,  m01 = v S        (all unsigned dot4; :79-106 exactly).
#define ANGLE_UMAX_VALUES 15, 15, 15, 15, 14, 14, 14, 13, 13, 12, 11, 10, 9, 8, 6, 3   // umax of HALF_PATCH_SIZE 15 (:456-471)
struct AngleWt {
    unsigned wu[64][4], wm[64][4];   // [lane][dword of the lane's 16 bytes]
};
constexpr AngleWt make_angle_wt()
{
    constexpr int um[16] = {ANGLE_UMAX_VALUES};
    AngleWt w{};
    for (int lane = 0; lane < 64; lane++)
        for (int k = 0; k < 4; k++) {
            unsigned a = 0, m = 0;
            for (int j = 0; j < 4; j++) {
                const int row = lane >> 1, u = 16 * (lane & 1) + 4 * k + j - ORB_HALF_PATCH, v = row - ORB_HALF_PATCH;
                const int au = u < 0 ? -u : u, av = v < 0 ? -v : v;
                const bool in = row <= 2 * ORB_HALF_PATCH && au <= ORB_HALF_PATCH && au <= um[av < 16 ? av : 15];
                a |= (unsigned)(in ? u + 16 : 0) << (8 * j);
                m |= (unsigned)(in ? 1 : 0) << (8 * j);
            }
            w.wu[lane][k] = a;
            w.wm[lane][k] = m;
        }
    return w;
}
__constant__ __attribute__((aligned(16))) AngleWt c_angle_wt = make_angle_wt();
struct __attribute__((packed, aligned(1))) UnalignedU4 {
    uint4 v;
};

// MIRROR: every result (count, keypoint records, descriptors) is stored a second time `mirror` bytes further on -- the page-locked
// twin of the device block the results go to (orbhip_frame_build: the kernels behind this one read the device copy, the host reads
// the twin without a copy command in between).  Instantiated for the single-frame kernel only.
template <int DS_KP, bool AX4, bool MIRROR = false>
__global__ __launch_bounds__(256, 8) void k_describe(const OrbLevels G, const uint8_t *__restrict__ lvl0,
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
                                                  int cap, int xcdMap, long long mirror ORB_ABL_PARAM)
{
    __shared__ int s_pos[DS_KP];      // cx | cy << 12 | level << 24, -1 = empty slot
    __shared__ int s_out[DS_KP];      // output index
    __shared__ int s_m10[DS_KP], s_m01[DS_KP];
    __shared__ float s_a[DS_KP], s_b[DS_KP];
    // level geometry of the slot, looked up once in phase 0 (indexing the by-value OrbLevels argument with a
    // per-lane level costs a dependent memory round trip every time)
    __shared__ unsigned s_ioff[DS_KP], s_boff[DS_KP];
    __shared__ int s_istride[DS_KP], s_bstride[DS_KP];
    // batches: three patch buffers per wave, filled by LDS-DMA (the next two keypoints' patches travel while one is used)
    constexpr bool PDMA = DS_KP >= 16;
    __shared__ __align__(16) uint32_t s_patch[4][PDMA ? 3 : 1][DS_TRIPS * 64];
    int blk = xcd_tile(xcdMap), frame = blockIdx.y;
    if ((xcdMap & 255) == 4) xcd_frame_tile((int)gridDim.y, blk, frame);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int g0 = blk * DS_KP;
    if (g0 >= G.totalKps) return;
    const int32_t *cnts = lvlKpCnt + frame * ORBHIP_MAX_LEVELS;

    // ---- 0. slot -> (level, position, output index) ----
    int my_score = 0;
    if (tid < DS_KP) {
        const int g = g0 + tid;
        int pos = -1, o = 0;
        if (g < G.totalKps) {
            int l = 0, off = 0, total = 0;
            for (int k = 0; k < G.nlevels; k++) {
                if (g >= G.lv[k].kpBase) {
                    l = k;
                    off = total;
                }
                total += cnts[k];
            }
            if (g == 0) {
                counts[frame] = total;
                if (MIRROR) *reinterpret_cast<int32_t *>(reinterpret_cast<char *>(counts + frame) + mirror) = total;
            }
            const int i = g - G.lv[l].kpBase;
            if (i < cnts[l]) {
                const uint32_t pk = lvlKp[(size_t)frame * G.totalKps + g];
                const int cx = (int)(pk & 0xFFFu) + ORB_MIN_BORDER;         // :843-844
                const int cy = (int)((pk >> 12) & 0xFFFu) + ORB_MIN_BORDER;
                my_score = (int)(pk >> 24);
                pos = cx | (cy << 12) | (l << 24);
                o = off + i;
            }
        }
        s_pos[tid] = pos;
        s_out[tid] = o;
        const int l = pos >= 0 ? pos >> 24 : 0;
        s_ioff[tid] = l == 0 ? 0u : (unsigned)G.lv[l].imgOff;
        s_istride[tid] = l == 0 ? stride0 : G.lv[l].stride;
        s_boff[tid] = l == 0 ? 0u : (unsigned)(G.boff1 + G.lv[l].imgOff);
        s_bstride[tid] = l == 0 ? G.bstride0 : G.lv[l].stride;
    }
    __syncthreads();
    ORB_ABL_STOP(phases < 1);   // timing ablation only (liborbhip_ablation.so, ORBHIP_DESCRIBE_PHASES): results are then invalid

    // ---- A. E5: IC_Angle moments on the un-blurred level ----
    if constexpr (AX4) {
        const uint4 wu = *reinterpret_cast<const uint4 *>(&c_angle_wt.wu[lane][0]);
        const uint4 wm = *reinterpret_cast<const uint4 *>(&c_angle_wt.wm[lane][0]);
        const int rowc = min(lane >> 1, 2 * ORB_HALF_PATCH), half16 = 16 * (lane & 1), vrow = (lane >> 1) - ORB_HALF_PATCH;
        auto load1 = [&](int kp, uint4 &wd) {
            const int pos = __builtin_amdgcn_readfirstlane(s_pos[kp]);
            if (pos >= 0) {
                const int cx = pos & 0xFFF, cy = (pos >> 12) & 0xFFF, l = pos >> 24;
                const uint8_t *img = l == 0 ? lvl0 + (size_t)frame * frame0 : pyr + (size_t)frame * pyrFrame + (unsigned)__builtin_amdgcn_readfirstlane((int)s_ioff[kp]);
                const int stride = __builtin_amdgcn_readfirstlane(s_istride[kp]);
                const uint8_t *p = img + (size_t)(cy - ORB_HALF_PATCH) * stride + (cx - ORB_HALF_PATCH);
                wd = reinterpret_cast<const UnalignedU4 *>(p + (unsigned)(__mul24(rowc, stride) + half16))->v;
            }
        };
        const int kp0 = wave;
        constexpr int NQ = DS_KP / 4;
        uint4 ring[NQ];   // every keypoint of the wave requested at once
#pragma unroll
        for (int q = 0; q < NQ; q++) load1(kp0 + 4 * q, ring[q]);
#pragma unroll
        for (int q = 0; q < NQ; q++) {
            const int kp = kp0 + 4 * q;
            const int pos = __builtin_amdgcn_readfirstlane(s_pos[kp]);
            if (pos >= 0) {   // wave-uniform
                const uint4 c = ring[q];
                unsigned S = __builtin_amdgcn_udot4(c.x, wm.x, 0u, false);
                S = __builtin_amdgcn_udot4(c.y, wm.y, S, false);
                S = __builtin_amdgcn_udot4(c.z, wm.z, S, false);
                S = __builtin_amdgcn_udot4(c.w, wm.w, S, false);
                unsigned U = __builtin_amdgcn_udot4(c.x, wu.x, 0u, false);
                U = __builtin_amdgcn_udot4(c.y, wu.y, U, false);
                U = __builtin_amdgcn_udot4(c.z, wu.z, U, false);
                U = __builtin_amdgcn_udot4(c.w, wu.w, U, false);
                const int m10 = wave_sum((int)U - 16 * (int)S);
                const int m01 = wave_sum(__mul24(vrow, (int)S));
                if (lane == 0) {
                    s_m10[kp] = m10;
                    s_m01[kp] = m01;
                }
            }
        }
    } else {
    // software pipeline: the five row dwords of the wave's next keypoint are in flight while the current one is
    // reduced
    {
        const int rsub = lane / 9, di = lane - rsub * 9;
        // Loads are unconditional inside one wave-uniform region (lanes without a dword -- lane 63, rows beyond
        // +15 -- read a valid neighbouring address and are masked in the arithmetic): a conditional load per
        // trip would compile to five exec-masked regions with a full wait between them.
        const int lrs = min(lane, 62) / 9, ldi = min(lane, 62) - lrs * 9;
        // A keypoint's position, level and strides are the same for the whole wave: taken as SCALARS (v_readfirstlane of the
        // LDS broadcast), so that the image address is scalar arithmetic and a load is (scalar base) + (32-bit lane offset):
        // one v_mad per load instead of a 64-bit multiply-add chain per lane (a sixth of the kernel's vector instructions).
        int lrow[5];   // this lane's row of the disc in trip it, counted from the top row (clamped to the last one)
#pragma unroll
        for (int it = 0; it < 5; it++) lrow[it] = min(it * 7 + lrs, 2 * ORB_HALF_PATCH);
        auto load5 = [&](int kp, uint32_t wd[5]) {
            const int pos = __builtin_amdgcn_readfirstlane(s_pos[kp]);
            if (pos >= 0) {
                const int cx = pos & 0xFFF, cy = (pos >> 12) & 0xFFF, l = pos >> 24;
                const uint8_t *img = l == 0 ? lvl0 + (size_t)frame * frame0 : pyr + (size_t)frame * pyrFrame + (unsigned)__builtin_amdgcn_readfirstlane((int)s_ioff[kp]);
                const int stride = __builtin_amdgcn_readfirstlane(s_istride[kp]);
                const uint8_t *p = img + (size_t)(cy - ORB_HALF_PATCH) * stride + ((cx - ORB_HALF_PATCH) & ~3);
#pragma unroll
                for (int it = 0; it < 5; it++)
                    wd[it] = *reinterpret_cast<const uint32_t *>(p + (unsigned)(__mul24(lrow[it], stride) + 4 * ldi));
            }
        };
        int dmax[5];   // half width of the disc row this lane reads in trip it (umax of |v|), fixed per lane
#pragma unroll
        for (int it = 0; it < 5; it++) {
            const int v = -ORB_HALF_PATCH + it * 7 + rsub;
            dmax[it] = G.umax[min(v < 0 ? -v : v, ORB_HALF_PATCH)];
        }
        const int kp0 = wave;   // wave w takes the slots w, w + 4, w + 8, ...: at any time the four waves work on four ADJACENT slots
                                // (neighbours in the quadtree's list are mostly neighbours in the image: their rows share cache lines,
                                // -3.5 % time; sorting a level's keypoints by the Morton code of their position first adds nothing)
        // ring of three row sets, prefetch distance two: the loads of keypoints q + 1 and q + 2 are in flight while q is
        // reduced; the keypoint loop is unrolled so that the ring is indexed statically
        constexpr int NQ = DS_KP / 4;
        constexpr int RA = NQ <= 4 ? 4 : 3;    // ring entries: with four or fewer keypoints per wave all their rows are requested at once
        uint32_t ring[RA][5];
#pragma unroll
        for (int q = 0; q < RA - 1 && q < NQ; q++) load5(kp0 + 4 * q, ring[q]);
#pragma unroll
        for (int q = 0; q < NQ; q++) {
            const int kp = kp0 + 4 * q;
            if (q + RA - 1 < NQ) load5(kp + 4 * (RA - 1), ring[(q + RA - 1) % RA]);
            const uint32_t *cur = ring[q % RA];
            const int pos = __builtin_amdgcn_readfirstlane(s_pos[kp]);
            if (pos >= 0) {   // wave-uniform
                const int cx = pos & 0xFFF;
                const int u0 = ((cx - ORB_HALF_PATCH) & ~3) + 4 * di - cx;
                int m10 = 0, m01 = 0;
#pragma unroll
                for (int it = 0; it < 5; it++) {
                    const int v = -ORB_HALF_PATCH + it * 7 + rsub;
                    const int d = dmax[it];
                    // bytes k with -d <= u0 + k <= d
                    const int lo = max(0, -d - u0), hi = min(3, d - u0);
                    if (lo <= hi && lane < 63 && v <= ORB_HALF_PATCH) {
                        const uint32_t mask = (0xFFFFFFFFu >> (8 * (3 - hi))) & (0xFFFFFFFFu << (8 * lo));
                        const uint32_t wm = cur[it] & mask;
                        const int rs = (int)__builtin_amdgcn_sad_u8(wm, 0u, 0u);                 // sum of the 4 bytes
                        const int rk = (int)__builtin_amdgcn_udot4(wm, 0x03020100u, 0u, false);  // sum of k * byte
                        m10 += __mul24(u0, rs) + rk;
                        m01 += __mul24(v, rs);
                    }
                }
                m10 = wave_sum(m10);
                m01 = wave_sum(m01);
                if (lane == 0) {
                    s_m10[kp] = m10;
                    s_m01[kp] = m01;
                }
            }
        }
    }
    }
    __syncthreads();
    ORB_ABL_STOP(phases < 2);

    // ---- B. angle, cos / sin, keypoint record: one thread per keypoint ----
    if (tid < DS_KP) {
        const int pos = s_pos[tid];
        if (pos >= 0) {
            const int cx = pos & 0xFFF, cy = (pos >> 12) & 0xFFF, l = pos >> 24;
            const OrbLevel &L = G.lv[l];
            const float angle = fast_atan2_dev((float)s_m01[tid], (float)s_m10[tid]);
            const float factorPI = (float)(3.1415926535897932384626433832795 / 180.f);
            const float rad = __fmul_rn(angle, factorPI);
            float a, b;
            orb_sincosf(rad, &b, &a);
            s_a[tid] = a;
            s_b[tid] = b;
            lvlAngle[(size_t)frame * G.totalKps + g0 + tid] = angle;
            const int o = s_out[tid];
            if (o < cap) {
                orbhip_keypoint kp;
                kp.x = __fmul_rn((float)cx, L.scale);   // level 0: scale == 1.0f, identical to "no scaling"
                kp.y = __fmul_rn((float)cy, L.scale);
                kp.size = L.kpSize;
                kp.angle = angle;
                kp.response = (float)my_score;
                kp.octave = l;
                kp.class_id = -1;
                kps[(size_t)frame * cap + o] = kp;
                if (MIRROR) *reinterpret_cast<orbhip_keypoint *>(reinterpret_cast<char *>(kps + ((size_t)frame * cap + o)) + mirror) = kp;
            }
        }
    }
    __syncthreads();
    ORB_ABL_STOP(phases < 3);

    // ---- C. E7: steered BRIEF on the blurred level ----
    // The 37 x 37 neighbourhood of the keypoint (the rotated pattern stays within 18 pixels) is staged per wave in
    // LDS with row-coalesced dword loads (10 dwords per row, 6 trips), software-pipelined like phase A; the 512
    // test pixels are then LDS byte reads.  A byte gather straight from memory touches up to 64 cache lines per
    // wave instruction; the staged form touches each line once.
    float px0[4], py0[4], px1[4], py1[4];
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const int pw = reinterpret_cast<const int *>(c_pattern)[64 * j + lane];
        px0[j] = (float)(signed char)(pw & 0xFF);
        py0[j] = (float)(signed char)((pw >> 8) & 0xFF);
        px1[j] = (float)(signed char)((pw >> 16) & 0xFF);
        py1[j] = (float)(signed char)((pw >> 24) & 0xFF);
    }
    int prow[DS_TRIPS], pdw[DS_TRIPS];     // this lane's (row, dword) of every trip; the tail lanes repeat the last dword
#pragma unroll
    for (int it = 0; it < DS_TRIPS; it++) {
        const int idx = min(it * 64 + lane, DS_ROWS * DS_PDW - 1);
        prow[it] = (int)(((unsigned)idx * 6554u) >> 16);   // idx / 10 for idx < 384
        pdw[it] = idx - prow[it] * DS_PDW;
    }
    // the patch of slot kp: scalar base + this lane's 32-bit offsets (one region, unconditional loads, see phase A)
    auto patch_base = [&](int kp, int pos, int &bstride) -> const uint8_t * {
        const int cx = pos & 0xFFF, cy = (pos >> 12) & 0xFFF;
        const uint8_t *bimg = blur + (size_t)frame * blurFrame + (unsigned)__builtin_amdgcn_readfirstlane((int)s_boff[kp]);
        bstride = __builtin_amdgcn_readfirstlane(s_bstride[kp]);
        return bimg + (size_t)(cy - DS_R) * bstride + ((cx - DS_R) & ~3);
    };
    auto load7 = [&](int kp, uint32_t wd[DS_TRIPS]) {
        const int pos = __builtin_amdgcn_readfirstlane(s_pos[kp]);
        if (pos >= 0) {
            int bstride;
            const uint8_t *p = patch_base(kp, pos, bstride);
#pragma unroll
            for (int it = 0; it < DS_TRIPS; it++)
                wd[it] = *reinterpret_cast<const uint32_t *>(p + (unsigned)(__mul24(prow[it], bstride) + 4 * pdw[it]));
        }
    };
    // LDS-DMA form (batches): global_load_lds_dword writes lane i's dword to LDS address M0 + 4 i, i.e. trip `it` fills the
    // dwords 64 it .. 64 it + 63 of the buffer -- the patch image itself (row-major, 10 dwords per row; the tail lanes of the last
    // trip land beyond it).  No registers, no ds_write; returns whether the transfers were issued (wave-uniform).
    auto dma7 = [&](int kp, uint32_t *buf) -> bool {
        const int pos = __builtin_amdgcn_readfirstlane(s_pos[kp]);
        if (pos < 0) return false;
        int bstride;
        const uint8_t *p = patch_base(kp, pos, bstride);
        const uint32_t ldsAddr = (uint32_t)(uintptr_t)buf;
#pragma unroll
        for (int it = 0; it < DS_TRIPS; it++) {
            const uint8_t *src = p + (unsigned)(__mul24(prow[it], bstride) + 4 * pdw[it]);
            uint32_t keep;
            asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %1, off\n\ts_mov_b32 m0, %0"
                         : "=&s"(keep)
                         : "v"(src), "s"(__builtin_amdgcn_readfirstlane((int)(ldsAddr + 256u * (uint32_t)it)))
                         : "memory");
        }
        return true;
    };
    const int kp0 = wave;   // (as in phase A)
    constexpr int NQ = DS_KP / 4;
    uint32_t ring[PDMA ? 1 : 3][DS_TRIPS];       // register form: two keypoints' patches in flight behind the one in LDS
    bool sent[NQ + 2];
#pragma unroll
    for (int q = 0; q < NQ + 2; q++) sent[q] = false;
    if (PDMA) {
        sent[0] = dma7(kp0, s_patch[wave][0]);
        if (NQ > 1) sent[1] = dma7(kp0 + 4, s_patch[wave][1]);
    } else {
        load7(kp0, ring[0]);
        if (NQ > 1) load7(kp0 + 4, ring[(PDMA ? 0 : 1)]);
    }
#pragma unroll
    for (int q = 0; q < NQ; q++) {
        const int kp = kp0 + 4 * q;
        const int pos = __builtin_amdgcn_readfirstlane(s_pos[kp]);
        uint32_t *patch = s_patch[wave][PDMA ? q % 3 : 0];
        if (PDMA) {
            // this keypoint's transfers have landed when only the next keypoint's six (if any were issued) are outstanding
            if (sent[q + 1]) asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        } else {
            const uint32_t *cur = ring[PDMA ? 0 : q % 3];
            if (pos >= 0) {
#pragma unroll
                for (int it = 0; it < DS_TRIPS; it++)
                    if (it * 64 + lane < DS_ROWS * DS_PDW) patch[it * 64 + lane] = cur[it];
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        if (q + 2 < NQ) {
            if (PDMA) sent[q + 2] = dma7(kp + 8, s_patch[wave][(q + 2) % 3]);
            else load7(kp + 8, ring[PDMA ? 0 : (q + 2) % 3]);
        }
        const uint8_t *patchB = reinterpret_cast<const uint8_t *>(patch);
        const int o = __builtin_amdgcn_readfirstlane(s_out[kp]);
        if (pos >= 0 && o < cap) {   // wave-uniform
            const int cx = pos & 0xFFF;
            // A test pixel's LDS address = keypoint's address in the patch + 40 * row + column, with row / column = the rotated
            // coordinates rounded to nearest-even (cvRound, :119-122).  x + 1.5 * 2^23 has the rounded integer in its low mantissa
            // bits (|x| <= 18.4), so the address is one 24-bit multiply-add of the two sums' bit patterns (the multiplicand is the low
            // 24 bits, 0x400000 + row) plus one scalar that also takes the constants back out.  v_add_f32 issues at twice the rate of
            // v_rndne_f32 / v_cvt_i32_f32 (profiles/r03/valu_rates.txt), and there are two instructions fewer per pixel.
            const float RMAGIC = 12582912.f;   // 0x4B400000
            const uint32_t bcAddr = (uint32_t)(uintptr_t)patchB + (uint32_t)(DS_R * (DS_PDW * 4) + DS_R + ((cx - DS_R) & 3)) -
                                    (0x400000u * (DS_PDW * 4) + 0x4B400000u);
            const float a = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, s_a[kp])));
            const float b = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, s_b[kp])));
            unsigned long long words[4];
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const int r0 = __float_as_int(__fadd_rn(__fadd_rn(__fmul_rn(px0[j], b), __fmul_rn(py0[j], a)), RMAGIC));
                const int c0 = __float_as_int(__fadd_rn(__fsub_rn(__fmul_rn(px0[j], a), __fmul_rn(py0[j], b)), RMAGIC));
                const int r1 = __float_as_int(__fadd_rn(__fadd_rn(__fmul_rn(px1[j], b), __fmul_rn(py1[j], a)), RMAGIC));
                const int c1 = __float_as_int(__fadd_rn(__fsub_rn(__fmul_rn(px1[j], a), __fmul_rn(py1[j], b)), RMAGIC));
                const int t0 = *(lds_u8p)(bcAddr + (uint32_t)(__mul24(r0, DS_PDW * 4) + c0));
                const int t1 = *(lds_u8p)(bcAddr + (uint32_t)(__mul24(r1, DS_PDW * 4) + c1));
                words[j] = __ballot(t0 < t1);
            }
            if (lane < 4) {
                // bit (64j + lane) of the descriptor = test 64j+lane, LSB first inside each byte (:128-145)
                const unsigned long long wsel = lane == 0 ? words[0] : (lane == 1 ? words[1] : (lane == 2 ? words[2] : words[3]));
                reinterpret_cast<unsigned long long *>(desc + ((size_t)frame * cap + o) * 32)[lane] = wsel;
                if (MIRROR) reinterpret_cast<unsigned long long *>(desc + ((size_t)frame * cap + o) * 32 + mirror)[lane] = wsel;
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
}

// =====================================================================================================================
// r06: the blur INSIDE the describe kernel.  The only consumer of the blurred pyramid is the 37 x 37 patch
// around a retained keypoint (ref: src/ORBextractor.cc:1103-1107 blurs the whole level, :109-149 reads 512 pixels of it), so a
// batch never builds that pyramid: a wave stages the 43 x 43 RAW neighbourhood of its keypoint (rows cy - 21 .. cy + 21, 48 bytes
// from column cx - 23, by three byte-unaligned global_load_lds_dwordx4 -- the next keypoint's travel while this one is computed),
// runs k_blur's two passes on it -- the same arithmetic on the 16 x 16 MFMA shapes, a keypoint being 3 x 3 tiles of them:
//   row pass     v_mfma_i32_16x16x64_i8: H[row][col] = sum_c (p[row][c] - 128) Kx[c][col] + 128 * 257; A = 16 rows x 64 bytes
//                straight from LDS (ds_read_b128; the lanes of the upper half of K multiply zeros of the band), B = the taps'
//                band, a per-lane constant; the result tile has its column on the lane and four rows in the lane's registers;
//   column pass  v_mfma_f32_16x16x32_f16 with two row-pass tiles (rows 0-15 and 16-31 of a block) as the B operand IN PLACE:
//                the contraction runs over the tiles' row index, which lives in the registers; k-slot j of lane quarter q is
//                row 4q + j (j < 4) or 16 + 4q + j - 4, and the band operand A is laid out for exactly that order; byte planes
//                as binary16 0x2400 | byte = (1024 + byte) * 2^-16, weights tap * 2^8 for the high plane and tap for the low one, both
//                into one accumulator started at minus the constant part, high plane first: exact wherever the result is not
//                saturated anyway (see the kernel; k_blur.hip keeps the planes at 2^-24, two accumulators and an fma);
//   rounding     v_cvt_pk_u8_f32 = round half to even + saturation (the SSE2 column filter of OpenCV 2.4 for x < w - w % 4);
//                keypoints whose patch reaches the scalar tail's columns take floor(v + 0.5) there (wave-uniform branch);
// -- one strip of 16 blurred columns at a time (12 + 12 registers of tiles and planes live) into the wave's patch buffer (37 x 37,
// transposed: [column][row], pitch 40), and runs the 256 tests on it.  32 keypoint slots per workgroup, eight per wave.
// Image borders (BORDER_REFLECT_101 in LEVEL coordinates): rows by reflecting the row index of the transfer; the two columns
// a keypoint 19-20 pixels from the left / right edge reaches beyond it are patched into the staged rows (rare, wave-uniform).
// Phases 0, A (IC angle from the raw level, one 16-byte load per lane) and B are those of k_describe<16, true>.
// k_blur stays for a frame or two and for orbhip_debug_get_blurred_level.
// Measured on the way (profiles/r06/describe_blur.md; 1024 frames, the kernel's two launches incl. the quadtree half beside them):
//   first form (16 slots, all nine tiles live, patch over the raw rows, 96 registers)              1.09 ms  (k_blur 0.63 + k_describe 0.74 before)
//   + the corner block skipped                                                                      1.08
//   + the strip's six column products issued before their roundings                                 1.07
//   + 32 slots per workgroup, first neighbourhood requested behind the disc loads                   1.01
//   + one strip at a time into a patch buffer of its own (74 registers)                             0.97
//   + both byte planes into ONE accumulator (planes at 2^-16, a band per plane: no combining fma)   0.955
// and not kept: v_pk_mul / add / fma_f32 for the tests and the roundings (they issue at half the rate of the scalar forms: the
// same time, 16 more registers: 1.17 at four workgroups per CU); five / six / four workgroups per CU by register bound (1.08 /
// 1.15 with spills / 1.08); the moments from the staged rows instead of loads of their own, in rounds of two keypoints per wave
// with the angles of a round computed between two workgroup barriers (removes ~40 cache lines per keypoint, worth 0.09 ms
// by ablation; the barriers and the lost overlap cost 0.17: 1.05); unused LDS capping the workgroups per CU at 4 / 3 / 2:
// 0.99 / 1.16 / 1.52 for the kernel alone (0.85 at 5); every wave on its own (DF_AUTONOMOUS, no workgroup barrier): 0.972 / 0.969;
// a keypoint's descriptor stored one iteration late (so that the wait at the top of the next iteration does not cover it): 0.975 / 0.975;
// the column pass on v_mfma_f32_16x16x16_f16 (K = 16 = one row-pass tile: its two plane registers ARE the operand, no copies into four
// consecutive registers; four chained products per block, two for rows 32 .. 36): 0.955 / 0.953, bit-exact.
// =====================================================================================================================
#define DF_ROWS 48                    // staged raw rows: 43 needed, the rest complete the three 16-row tiles
#define DF_PITCH 48                   // bytes per staged row: columns cx - 23 .. cx + 24 (12 dwords: the A operand's 16 rows fall on distinct banks)
#define DF_A 2                        // byte of column cx - 21 (the first one the blur reads) in a staged row
#define DF_RAW (DF_ROWS * DF_PITCH)   // 2304
#define DF_BUF 2368                   // + 64: the upper-K lanes of the last rows' A operand read (and ignore) bytes beyond the image
#ifndef DF_WG_PER_CU
#define DF_WG_PER_CU 6                // 74 registers per lane, 25 KB of LDS per workgroup (measured: 4 -> 5 workgroups per CU -13 %, 5 -> 6 none)
#endif
#ifndef DF_KP
#define DF_KP 32                      // keypoint slots per workgroup (a multiple of 16)
#endif
#ifndef DF_AUTONOMOUS
#define DF_AUTONOMOUS 0                // 1: every wave decodes its own slots and computes its own angles, no workgroup barrier (measured: no change)
#endif
#ifndef DF_SKIP_CORNER
#define DF_SKIP_CORNER 1
#endif
#define DF_OBUF 1536                  // 37 columns x DF_OP
#define DF_OP 40                      // pitch of the blurred patch, [column 0..36][row 0..36 (+3)]
typedef int dfv4i __attribute__((ext_vector_type(4)));
typedef float dfv4f __attribute__((ext_vector_type(4)));
typedef _Float16 dfv8h __attribute__((ext_vector_type(8)));

struct DfBands {
    unsigned row[64][4];   // B operand of the row pass: lane (n = l & 15, q = l >> 4), byte j: tap[16 q + j - n - DF_A]
    unsigned col[64][4];   // A operand of the column pass, HIGH byte plane: lane (m = l & 15, q), half j: tap[row(q, j) - m] * 256 as binary16
    unsigned colLo[64][4]; // ... LOW byte plane: tap[row(q, j) - m]
};
constexpr unsigned df_half_bits(int v)   // binary16 pattern of a small positive integer (exact below 2048 * 2^k)
{
    if (v == 0) return 0;
    int e = 0, m = v;
    while (m >= 2048) { m >>= 1; e++; }
    while (m < 1024) { m <<= 1; e--; }
    return (unsigned)(((e + 25) << 10) | (m & 1023));   // m in [1024, 2048): value m * 2^e = 1.f * 2^(e + 10)
}
constexpr DfBands make_df_bands()
{
    constexpr int tap[7] = {18, 34, 49, 55, 49, 34, 18};
    DfBands b{};
    for (int l = 0; l < 64; l++) {
        const int n = l & 15, q = l >> 4;
        for (int j = 0; j < 16; j++) {
            const int i = 16 * q + j - n - DF_A;
            const unsigned v = (i >= 0 && i <= 6) ? (unsigned)tap[i] : 0u;
            b.row[l][j >> 2] |= v << (8 * (j & 3));
        }
        for (int j = 0; j < 8; j++) {
            const int r = j < 4 ? 4 * q + j : 16 + 4 * q + (j - 4), i = r - n;
            const unsigned v = (i >= 0 && i <= 6) ? df_half_bits(tap[i] * 256) : 0u, vlo = (i >= 0 && i <= 6) ? df_half_bits(tap[i]) : 0u;
            b.col[l][j >> 1] |= v << (16 * (j & 1));
            b.colLo[l][j >> 1] |= vlo << (16 * (j & 1));
        }
    }
    return b;
}
__constant__ __attribute__((aligned(16))) DfBands c_df_bands = make_df_bands();
static_assert(df_half_bits(18 * 256) == 0x6C80 && df_half_bits(55 * 256) == 0x72E0 && df_half_bits(18) == 0x4C80 && df_half_bits(55) == 0x52E0, "binary16 of tap * 256, tap");

__device__ __forceinline__ void df_glds16(const void *gsrc, uint32_t ldsAddr)
{
    uint32_t keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(gsrc), "s"(ldsAddr)
                 : "memory");
}

__global__ __launch_bounds__(256, DF_WG_PER_CU) void k_describe_blur(const OrbLevels G, const uint8_t *__restrict__ lvl0, int stride0,
                                                          unsigned long long frame0, const uint8_t *__restrict__ pyr,
                                                          unsigned long long pyrFrame, const uint32_t *__restrict__ lvlKp,
                                                          const int32_t *__restrict__ lvlKpCnt, float *__restrict__ lvlAngle,
                                                          orbhip_keypoint *__restrict__ kps, uint8_t *__restrict__ desc,
                                                          int32_t *__restrict__ counts, int cap, int xcdMap ORB_ABL_PARAM)
{
    constexpr int DS_KP = DF_KP, NQ = DS_KP / 4;   // slots per workgroup, keypoints per wave
    __shared__ int s_pos[DS_KP], s_out[DS_KP], s_m10[DS_KP], s_m01[DS_KP];
    __shared__ float s_a[DS_KP], s_b[DS_KP];
    __shared__ unsigned s_ioff[DS_KP];
    __shared__ int s_istride[DS_KP], s_wh[DS_KP];      // level width | height << 16
    __shared__ __align__(16) uint8_t s_raw[4][2][DF_BUF];
    __shared__ __align__(16) uint8_t s_blurred[4][DF_OBUF];   // the keypoint's blurred 37 x 37 patch, [column][row]
    int blk = xcd_tile(xcdMap), frame = blockIdx.y;
    if ((xcdMap & 255) == 4) xcd_frame_tile((int)gridDim.y, blk, frame);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int g0 = blk * DS_KP;
    if (g0 >= G.totalKps) return;
    const int32_t *cnts = lvlKpCnt + frame * ORBHIP_MAX_LEVELS;

    // ---- 0. slot -> (level, position, output index): as k_describe ----
    // (DF_AUTONOMOUS=1, an experiment kept as a compile-time option: lanes 0 .. 7 of EVERY wave decode the wave's own eight slots
    // (wave + 4 q) and compute their angles, so that no workgroup barrier is left and a wave that waits for its loads or walks the
    // scalar angle sequence holds no other -- 0.972 ms against 0.969 with the three barriers: they were not what the kernel waits for.)
    int my_score = 0;
    const int mySlot = DF_AUTONOMOUS ? wave + 4 * lane : tid;
    if (DF_AUTONOMOUS ? lane < NQ : tid < DS_KP) {
        const int g = g0 + mySlot;
        int pos = -1, o = 0;
        if (g < G.totalKps) {
            int l = 0, off = 0, total = 0;
            for (int k = 0; k < G.nlevels; k++) {
                if (g >= G.lv[k].kpBase) {
                    l = k;
                    off = total;
                }
                total += cnts[k];
            }
            if (g == 0) counts[frame] = total;
            const int i = g - G.lv[l].kpBase;
            if (i < cnts[l]) {
                const uint32_t pk = lvlKp[(size_t)frame * G.totalKps + g];
                const int cx = (int)(pk & 0xFFFu) + ORB_MIN_BORDER;
                const int cy = (int)((pk >> 12) & 0xFFFu) + ORB_MIN_BORDER;
                my_score = (int)(pk >> 24);
                pos = cx | (cy << 12) | (l << 24);
                o = off + i;
            }
        }
        s_pos[mySlot] = pos;
        s_out[mySlot] = o;
        const int l = pos >= 0 ? pos >> 24 : 0;
        s_ioff[mySlot] = l == 0 ? 0u : (unsigned)G.lv[l].imgOff;
        s_istride[mySlot] = l == 0 ? stride0 : G.lv[l].stride;
        s_wh[mySlot] = G.lv[l].w | (G.lv[l].h << 16);
    }
    if (DF_AUTONOMOUS) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    } else
        __syncthreads();
    ORB_ABL_STOP(phases < 1);

    // the raw neighbourhood of slot kp -> buf: lane (row of 21, chunk of 3); rows 43.. repeat row 42
    const int drow = (lane * 43) >> 7, dchunk = lane - 3 * drow;   // lane / 3, lane % 3 (lane 63: row 21, not sent)
    const int stageRow[3] = {drow, min(21 + drow, 42), 42};         // the staged row of this lane in each of the three transfers
    auto stage = [&](int kp, uint8_t *buf) -> bool {
        const int pos = __builtin_amdgcn_readfirstlane(s_pos[kp]);
        if (pos < 0 || __builtin_amdgcn_readfirstlane(s_out[kp]) >= cap) return false;
        const int cx = pos & 0xFFF, cy = (pos >> 12) & 0xFFF, l = pos >> 24;
        const uint8_t *img = l == 0 ? lvl0 + (size_t)frame * frame0
                                    : pyr + (size_t)frame * pyrFrame + (unsigned)__builtin_amdgcn_readfirstlane((int)s_ioff[kp]);
        const int stride = __builtin_amdgcn_readfirstlane(s_istride[kp]);
        const int wh = __builtin_amdgcn_readfirstlane(s_wh[kp]), w = wh & 0xFFFF, h = wh >> 16;
        // columns: chunk c starts at cx - 23 + 16 c, kept inside [0, wAl - 16] (what may be read; a chunk that was moved is put right
        // by fix_columns below)
        const int wAl = (w + 15) & ~15;
        const int xc = min(max(cx - 23 + 16 * dchunk, 0), wAl - 16);
        const uint32_t ldsBase = (uint32_t)(uintptr_t)buf;
        if (cy >= 21 && cy + 21 < h) {   // wave-uniform: no row to reflect -- a lane's three rows are fixed distances below the first one
            const uint8_t *src = img + (size_t)(unsigned)(__mul24(cy - 21, stride) + xc);
#pragma unroll
            for (int t = 0; t < 3; t++)
                if (lane < (t < 2 ? 63 : 18))
                    df_glds16(src + (unsigned)__mul24(stageRow[t], stride), __builtin_amdgcn_readfirstlane((int)(ldsBase + 1008u * (uint32_t)t)));
            return true;
        }
#pragma unroll
        for (int t = 0; t < 3; t++) {
            int y = cy - 21 + stageRow[t];
            y = y < 0 ? -y : y;                      // BORDER_REFLECT_101 of the level's rows (|excursion| <= 2)
            y = y >= h ? 2 * h - 2 - y : y;
            const uint8_t *src = img + (size_t)(unsigned)(__mul24(y, stride) + xc);
            if (lane < (t < 2 ? 63 : 18)) df_glds16(src, __builtin_amdgcn_readfirstlane((int)(ldsBase + 1008u * (uint32_t)t)));
        }
        return true;
    };
    // A keypoint closer than 23 pixels to the left edge or 25 to the (16-byte padded) right edge, or one whose blur reaches beyond
    // column w - 1: every byte the blur reads (DF_A .. DF_A + 42) is fetched again from where its (reflected) column landed.
    auto fix_columns = [&](int kp, uint8_t *buf) {
        const int pos = __builtin_amdgcn_readfirstlane(s_pos[kp]);
        const int cx = pos & 0xFFF;
        const int w = __builtin_amdgcn_readfirstlane(s_wh[kp]) & 0xFFFF, wAl = (w + 15) & ~15;
        if (cx - 23 >= 0 && cx + 25 <= wAl && cx + 21 < w) return;   // wave-uniform
        uint8_t *row = buf + lane * DF_PITCH;
        if (lane < DF_ROWS) {
            uint32_t nw[12];   // the row's new dwords (fully unrolled: registers, no scratch); in-order LDS: every read precedes the writes
#pragma unroll
            for (int d = 0; d < 12; d++) {
                uint32_t x = 0;
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    const int j = 4 * d + k - DF_A;   // 0 .. 42: column cx - 21 + j
                    int at = 4 * d + k;
                    if (j >= 0 && j < 43) {
                        int col = cx - 21 + j;
                        col = col < 0 ? -col : col;
                        col = col >= w ? 2 * w - 2 - col : col;
                        // the chunk that holds it: the last one whose (moved) first column is not beyond it
                        int c = 2, first = min(max(cx - 23 + 32, 0), wAl - 16);
                        if (col < first) { c = 1; first = min(max(cx - 23 + 16, 0), wAl - 16); }
                        if (col < first) { c = 0; first = min(max(cx - 23, 0), wAl - 16); }
                        at = 16 * c + (col - first);
                    }
                    x |= (uint32_t)row[at] << (8 * k);
                }
                nw[d] = x;
            }
#pragma unroll
            for (int d = 0; d < 12; d++) reinterpret_cast<uint32_t *>(row)[d] = nw[d];
        }
    };

    const int kp0 = wave;

    // ---- A. IC_Angle moments on the un-blurred level: k_describe<16, true> ----
    bool sent0 = false;
    {
        const uint4 wu = *reinterpret_cast<const uint4 *>(&c_angle_wt.wu[lane][0]);
        const uint4 wm = *reinterpret_cast<const uint4 *>(&c_angle_wt.wm[lane][0]);
        const int rowc = min(lane >> 1, 2 * ORB_HALF_PATCH), half16 = 16 * (lane & 1), vrow = (lane >> 1) - ORB_HALF_PATCH;
#pragma unroll
        for (int qb = 0; qb < NQ; qb += 4) {
        uint4 ring[4];
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const int kp = kp0 + 4 * (qb + q);
            const int pos = __builtin_amdgcn_readfirstlane(s_pos[kp]);
            if (pos >= 0) {
                const int cx = pos & 0xFFF, cy = (pos >> 12) & 0xFFF, l = pos >> 24;
                const uint8_t *img = l == 0 ? lvl0 + (size_t)frame * frame0
                                            : pyr + (size_t)frame * pyrFrame + (unsigned)__builtin_amdgcn_readfirstlane((int)s_ioff[kp]);
                const int stride = __builtin_amdgcn_readfirstlane(s_istride[kp]);
                const uint8_t *p = img + (size_t)(cy - ORB_HALF_PATCH) * stride + (cx - ORB_HALF_PATCH);
                bool ld = true;
                ORB_ABL_IF(phases == 5) ld = false;   // timing ablation only: what do the disc loads cost?
                if (ld) ring[q] = reinterpret_cast<const UnalignedU4 *>(p + (unsigned)(__mul24(rowc, stride) + half16))->v;
                else ring[q] = make_uint4(cx, cy, lane, q);
            }
        }
        // the first keypoint's neighbourhood: requested BEHIND the disc loads (loads return in order: the moments do not wait for it),
        // in flight during the rest of phase A and phase B
        if (qb == 0) sent0 = stage(kp0, s_raw[wave][0]);
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const int kp = kp0 + 4 * (qb + q);
            const int pos = __builtin_amdgcn_readfirstlane(s_pos[kp]);
            if (pos >= 0) {
                const uint4 c = ring[q];
                unsigned S = __builtin_amdgcn_udot4(c.x, wm.x, 0u, false);
                S = __builtin_amdgcn_udot4(c.y, wm.y, S, false);
                S = __builtin_amdgcn_udot4(c.z, wm.z, S, false);
                S = __builtin_amdgcn_udot4(c.w, wm.w, S, false);
                unsigned U = __builtin_amdgcn_udot4(c.x, wu.x, 0u, false);
                U = __builtin_amdgcn_udot4(c.y, wu.y, U, false);
                U = __builtin_amdgcn_udot4(c.z, wu.z, U, false);
                U = __builtin_amdgcn_udot4(c.w, wu.w, U, false);
                const int m10 = wave_sum((int)U - 16 * (int)S);
                const int m01 = wave_sum(__mul24(vrow, (int)S));
                if (lane == 0) {
                    s_m10[kp] = m10;
                    s_m01[kp] = m01;
                }
            }
        }
        }
    }
    if (DF_AUTONOMOUS) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    } else
        __syncthreads();
    ORB_ABL_STOP(phases < 2);

    // ---- B. angle, cos / sin, keypoint record: one thread per keypoint (k_describe) ----
    if (DF_AUTONOMOUS ? lane < NQ : tid < DS_KP) {
        const int pos = s_pos[mySlot];
        if (pos >= 0) {
            const int cx = pos & 0xFFF, cy = (pos >> 12) & 0xFFF, l = pos >> 24;
            const OrbLevel &L = G.lv[l];
            const float angle = fast_atan2_dev((float)s_m01[mySlot], (float)s_m10[mySlot]);
            const float factorPI = (float)(3.1415926535897932384626433832795 / 180.f);
            const float rad = __fmul_rn(angle, factorPI);
            float a, b;
            orb_sincosf(rad, &b, &a);
            s_a[mySlot] = a;
            s_b[mySlot] = b;
            lvlAngle[(size_t)frame * G.totalKps + g0 + mySlot] = angle;
            const int o = s_out[mySlot];
            if (o < cap) {
                orbhip_keypoint kp;
                kp.x = __fmul_rn((float)cx, L.scale);
                kp.y = __fmul_rn((float)cy, L.scale);
                kp.size = L.kpSize;
                kp.angle = angle;
                kp.response = (float)my_score;
                kp.octave = l;
                kp.class_id = -1;
                kps[(size_t)frame * cap + o] = kp;
            }
        }
    }
    if (DF_AUTONOMOUS) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    } else
        __syncthreads();
    ORB_ABL_STOP(phases < 3);

    // ---- C. per keypoint: blur of the staged neighbourhood, then the 256 tests on it ----
    float px0[4], py0[4], px1[4], py1[4];
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const int pw = reinterpret_cast<const int *>(c_pattern)[64 * j + lane];
        px0[j] = (float)(signed char)(pw & 0xFF);
        py0[j] = (float)(signed char)((pw >> 8) & 0xFF);
        px1[j] = (float)(signed char)((pw >> 16) & 0xFF);
        py1[j] = (float)(signed char)((pw >> 24) & 0xFF);
    }
    const uint4 bandRow = *reinterpret_cast<const uint4 *>(&c_df_bands.row[lane][0]);
    const uint4 bandCol = *reinterpret_cast<const uint4 *>(&c_df_bands.col[lane][0]);
    const uint4 bandColLo = *reinterpret_cast<const uint4 *>(&c_df_bands.colLo[lane][0]);
    const int n16 = lane & 15, q4 = lane >> 4;
    dfv4i hinit;
    dfv4f zinit;
#pragma unroll
    for (int i = 0; i < 4; i++) {
        // Byte planes as binary16 0x2400 | byte = (1024 + byte) * 2^-16 (the 0x24 rides in the row pass's start value); the high plane is
        // weighted tap * 2^8, the low plane tap, and BOTH accumulate into one register set started at minus the constant part
        // (257 * 1024 * (2^-8 + 2^-16)), high plane first: after it every partial sum is a multiple of 2^-8 below 2^11 (19 bits); during the
        // low plane they are multiples of 2^-16 that only grow -- exact below 256 (24 bits), and one that reaches 256 ends above 255.5,
        // i.e. saturated whatever its last bit.  The result is S * 2^-16 itself: no combine, one start value per block (k_blur keeps two
        // accumulators and an fma: its planes sit at 2^-24 because its 32 x 32 tiles leave no registers for a second band).
        hinit[i] = 128 * 257 + 0x24000000;
        zinit[i] = -(float)(257 * 1024) * (1.0f / 256.0f + 1.0f / 65536.0f);
    }
    bool sentCur = sent0;
#pragma unroll 1
    for (int q = 0; q < NQ; q++) {
        const int kp = kp0 + 4 * q;
        uint8_t *buf = s_raw[wave][q & 1], *obuf = s_blurred[wave];
        // this keypoint's rows have landed; the next one's start now, into the buffer whose tests ended an iteration ago
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        const bool sentNext = q + 1 < NQ ? stage(kp + 4, s_raw[wave][(q + 1) & 1]) : false;
        if (sentCur) {   // wave-uniform
            const int pos = __builtin_amdgcn_readfirstlane(s_pos[kp]);
            const int o = __builtin_amdgcn_readfirstlane(s_out[kp]);
            const int cx = pos & 0xFFF;
            const int w = __builtin_amdgcn_readfirstlane(s_wh[kp]) & 0xFFFF;
            fix_columns(kp, buf);
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            // One strip of 16 blurred columns at a time (three row-pass tiles, their planes, three column blocks: 12 + 12 registers live
            // instead of 36 + 36): row pass -- tile rt = rows 16 rt .., lanes of K's upper half re-read the lower half's bytes --,
            // byte planes as binary16 pairs, column pass + rounding -- block mb = blurred rows 16 mb .. -- into the wave's patch buffer
            // ([column][row], pitch DF_OP).
            const int wvec = w - (w & 3);
            const bool tail = cx + 18 >= wvec;   // wave-uniform: some of the patch's columns belong to the reference's scalar tail
            const uint8_t *arow = buf + n16 * DF_PITCH + 16 * (q4 & 1);
            uint8_t *obase = obuf + n16 * DF_OP + 4 * q4;
#pragma unroll
            for (int cs = 0; cs < 3; cs++) {
                dfv4i H[3];
#pragma unroll
                for (int rt = 0; rt < 3; rt++) {
                    uint4 a = *reinterpret_cast<const uint4 *>(arow + rt * 16 * DF_PITCH + 16 * cs);
                    a.x ^= 0x80808080u; a.y ^= 0x80808080u; a.z ^= 0x80808080u; a.w ^= 0x80808080u;
                    H[rt] = __builtin_amdgcn_mfma_i32_16x16x64_i8(__builtin_bit_cast(dfv4i, a), __builtin_bit_cast(dfv4i, bandRow), hinit, 0, 0, 0);
                }
                uint32_t lo[3][2], hi[3][2];
#pragma unroll
                for (int rt = 0; rt < 3; rt++)
#pragma unroll
                    for (int d = 0; d < 2; d++) {
                        const uint32_t x = (uint32_t)H[rt][2 * d], y = (uint32_t)H[rt][2 * d + 1];
                        lo[rt][d] = __builtin_amdgcn_perm(y, x, 0x07040300u);
                        hi[rt][d] = __builtin_amdgcn_perm(y, x, 0x07050301u);
                    }
                dfv4f z[3];
#pragma unroll
                for (int mb = 0; mb < 3; mb++) {
                    // rows 32 .. 36 x columns 32 .. 36: 14 or more pixels from the centre both ways -- beyond the pattern's radius
                    // (18.4 after rotation, each coordinate rounded): never read
                    if (DF_SKIP_CORNER && mb == 2 && cs == 2) continue;
                    const int m2 = mb < 2 ? mb + 1 : 2;   // (what follows tile 2 is multiplied by zeros of the band)
                    const uint4 bl = make_uint4(lo[mb][0], lo[mb][1], lo[m2][0], lo[m2][1]);
                    const uint4 bh = make_uint4(hi[mb][0], hi[mb][1], hi[m2][0], hi[m2][1]);
                    z[mb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(dfv8h, bandCol), __builtin_bit_cast(dfv8h, bh), zinit, 0, 0, 0);
                    z[mb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(dfv8h, bandColLo), __builtin_bit_cast(dfv8h, bl), z[mb], 0, 0, 0);
                }
#pragma unroll
                for (int mb = 0; mb < 3; mb++) {
                    if (DF_SKIP_CORNER && mb == 2 && cs == 2) continue;
                    float v[4];   // S * 2^-16
#pragma unroll
                    for (int e = 0; e < 4; e++) v[e] = z[mb][e];
                    if (tail) {
                        const bool up = cx - 18 + 16 * cs + n16 >= wvec;   // this lane's column: (S + 32768) >> 16
#pragma unroll
                        for (int e = 0; e < 4; e++) v[e] = up ? floorf(v[e] + 0.5f) : v[e];
                    }
                    uint32_t packed = 0;
#pragma unroll
                    for (int e = 0; e < 4; e++) packed = __builtin_amdgcn_cvt_pk_u8_f32(v[e], e, packed);
                    if (16 * cs + n16 < 37 && 16 * mb + 4 * q4 < 37)
                        *reinterpret_cast<uint32_t *>(obase + 16 * cs * DF_OP + 16 * mb) = packed;
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            bool tests = true;
            ORB_ABL_IF(phases < 4) tests = false;
            if (tests) {
            // the 256 tests: pixel (row r, column c) of the patch at buf + (18 + c) * DF_OP + 18 + r; r, c = the rotated coordinates
            // rounded to nearest-even through the 1.5 * 2^23 trick of k_describe
            const float RMAGIC = 12582912.f;   // 0x4B400000
            const uint32_t bcAddr = (uint32_t)(uintptr_t)obuf + (uint32_t)(DS_R * DF_OP + DS_R) - (0x400000u * DF_OP + 0x4B400000u);
            const float a = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, s_a[kp])));
            const float b = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, s_b[kp])));
            unsigned long long words[4];
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const int r0 = __float_as_int(__fadd_rn(__fadd_rn(__fmul_rn(px0[j], b), __fmul_rn(py0[j], a)), RMAGIC));
                const int c0 = __float_as_int(__fadd_rn(__fsub_rn(__fmul_rn(px0[j], a), __fmul_rn(py0[j], b)), RMAGIC));
                const int r1 = __float_as_int(__fadd_rn(__fadd_rn(__fmul_rn(px1[j], b), __fmul_rn(py1[j], a)), RMAGIC));
                const int c1 = __float_as_int(__fadd_rn(__fsub_rn(__fmul_rn(px1[j], a), __fmul_rn(py1[j], b)), RMAGIC));
                const int t0 = *(lds_u8p)(bcAddr + (uint32_t)(__mul24(c0, DF_OP) + r0));
                const int t1 = *(lds_u8p)(bcAddr + (uint32_t)(__mul24(c1, DF_OP) + r1));
                words[j] = __ballot(t0 < t1);
            }
            if (lane < 4) {
                const unsigned long long wsel = lane == 0 ? words[0] : (lane == 1 ? words[1] : (lane == 2 ? words[2] : words[3]));
                reinterpret_cast<unsigned long long *>(desc + ((size_t)frame * cap + o) * 32)[lane] = wsel;
            }
            }
        }
        sentCur = sentNext;
    }
}

// Can a batch of this geometry take k_describe_blur (the blurred pyramid is then never built)?  The kernel is written for the umax
// table of a 31-pixel patch (always what orb_init_tables computes) and for 16 slots per workgroup.  ORBHIP_DESCRIBE_FUSED=0
// (liborbhip_ablation.so): k_blur + k_describe for batches too.
bool describe_blur_available(const OrbLevels &G, int B)
{
    static const int fusedEnv = ORB_TUNE("DESCRIBE_FUSED", 1);
    static const int kpwEnv = ORB_TUNE("DESCRIBE_KPW", 0);
    static const int ax4Env = ORB_TUNE("DESCRIBE_AX4", 1);
    static const int umaxWant[16] = {ANGLE_UMAX_VALUES};
    bool ok = fusedEnv != 0 && B >= 8 && ax4Env != 0 && (kpwEnv == 0 || kpwEnv == 16);
    for (int v = 0; v < 16; v++) ok = ok && G.umax[v] == umaxWant[v];
    long long px = 0;
    for (int l = 0; l < G.nlevels; l++) {
        ok = ok && G.lv[l].w >= 48 && G.lv[l].h >= 40 && G.lv[l].w < 65536 && G.lv[l].h < 32768;
        px += (long long)G.lv[l].w * G.lv[l].h;
    }
    // Blurring 37 x 37 pixels per keypoint beats blurring the pyramid while the keypoints are few for the pixels: measured
    // (profiles/r06/describe_blur.md) +10 % of the whole step at 640 x 480 / 1000 features (1.1 keypoints per 1000 pyramid pixels),
    // +6 % at 1241 x 376 / 2000 (1.4), -10 % at 640 x 480 / 4000 (4.2); the extraction alone (tools/time_extract.py): -2.9 % of its
    // time at 1.6, +2.1 % at 2.1, +5.6 % at 2.6, a tie at 2.1 on 1241 x 376.  The switch sits at 2.0 (ORBHIP_DESCRIBE_FUSED=2: always).
    if (fusedEnv != 2) ok = ok && (long long)G.totalKps * 500 <= px;
    return ok;
}

void launch_describe_blur(hipStream_t s, const OrbLevels &G, const uint8_t *lvl0, int stride0, size_t frame0, const uint8_t *pyr,
                          size_t pyrFrame, const uint32_t *lvlKp, const int32_t *lvlKpCnt, float *lvlAngle, orbhip_keypoint *kps,
                          uint8_t *desc, int32_t *counts, int cap, int B)
{
    static const int dmap = ORB_TUNE("DESCRIBE_MAP", -1);
    const int mapArg = dmap >= 0 ? (dmap | (orb_xcd_chunk() << 8)) : orb_xcd_arg(DESCRIBE_DEFAULT_MAP);
    static const int phases = ORB_TUNE("DESCRIBE_PHASES", 4);
    (void)phases;
    const int nblk = (G.totalKps + DF_KP - 1) / DF_KP;
    dim3 grid((mapArg & 255) ? (nblk + 7) / 8 * 8 : nblk, B, 1), block(256, 1, 1);
    orb_path(ORB_PATH_DESCRIBE_BLUR);
    static const int padLds = ORB_TUNE("DESCRIBE_PADLDS", 0);   // occupancy experiment only: unused dynamic LDS caps the workgroups per CU
    hipLaunchKernelGGL(k_describe_blur, grid, block, (size_t)padLds, s, G, lvl0, stride0, (unsigned long long)frame0, pyr, (unsigned long long)pyrFrame,
                       lvlKp, lvlKpCnt, lvlAngle, kps, desc, counts, cap, mapArg ORB_ABL_ARG(phases));
}

void launch_describe(hipStream_t s, const OrbLevels &G, const uint8_t *lvl0, int stride0, size_t frame0,
                     const uint8_t *pyr, size_t pyrFrame, const uint8_t *blur, size_t blurFrame,
                     const uint32_t *lvlKp, const int32_t *lvlKpCnt, float *lvlAngle,
                     orbhip_keypoint *kps, uint8_t *desc, int32_t *counts, int cap, int B, long long mirror)
{
    orb_path(ORB_PATH_DESCRIBE);
    static const int kpwEnv = ORB_TUNE("DESCRIBE_KPW", 0);
    const int kpw = mirror ? 8 : kpwEnv == 8 || kpwEnv == 16 || kpwEnv == 32 ? kpwEnv : (B >= 8 ? 16 : 8);   // (mirror: single frames only)
    // workgroup -> (slot block, frame): ORBHIP_DESCRIBE_MAP overrides this kernel's mapping alone (A/B runs)
    static const int dmap = ORB_TUNE("DESCRIBE_MAP", -1);
    const int mapArg = dmap >= 0 ? (dmap | (orb_xcd_chunk() << 8)) : orb_xcd_arg(DESCRIBE_DEFAULT_MAP);
    static const int phases = ORB_TUNE("DESCRIBE_PHASES", 3);
    (void)phases;
    // occupancy experiment only: unused dynamic LDS caps the workgroups per CU
    static const int padLds = ORB_TUNE("DESCRIBE_PADLDS", 0);
    const int nblk = (G.totalKps + kpw - 1) / kpw;
    dim3 grid((mapArg & 255) ? (nblk + 7) / 8 * 8 : nblk, B, 1), block(256, 1, 1);
    // the one-load form of the angle phase is written for the umax table of a 31-pixel patch (always what orb_init_tables computes)
    static const int ax4Env = ORB_TUNE("DESCRIBE_AX4", 1);
    static const int umaxWant[16] = {ANGLE_UMAX_VALUES};
    bool ax4 = ax4Env != 0 && kpw == 16;
    for (int v = 0; v < 16; v++) ax4 = ax4 && G.umax[v] == umaxWant[v];
#define ORB_LAUNCH_DESCRIBE(KERN)                                                                                               \
    hipLaunchKernelGGL(KERN, grid, block, (size_t)padLds, s, G, lvl0, stride0, (unsigned long long)frame0, pyr,                     \
                       (unsigned long long)pyrFrame, blur, (unsigned long long)blurFrame, lvlKp, lvlKpCnt, lvlAngle, kps, desc, \
                       counts, cap, mapArg, mirror ORB_ABL_ARG(phases))
    if (kpw == 32)
        ORB_LAUNCH_DESCRIBE((k_describe<32, false>));
    else if (kpw == 16 && ax4)
        ORB_LAUNCH_DESCRIBE((k_describe<16, true>));
    else if (kpw == 16)
        ORB_LAUNCH_DESCRIBE((k_describe<16, false>));
    else if (mirror)
        ORB_LAUNCH_DESCRIBE((k_describe<8, false, true>));
    else
        ORB_LAUNCH_DESCRIBE((k_describe<8, false>));
#undef ORB_LAUNCH_DESCRIBE
}

