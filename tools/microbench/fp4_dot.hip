// fp4_dot.hip -- does v_mfma_scale_f32_32x32x64_f8f6f4 with FP4 (e2m1) operands give EXACT integer keys?
// A row bit -> nibble 0b0001 (0.5) with block scale 2^7; query bit -> 0b0010 (+1, clear) / 0b1010 (-1, set) with block scale 2^6;
// C = row index + 2^21.  Expected D = 4096 * (|b| - 2 |a & b|) + row + 2^21 for the 64 bits of the step, exactly, as an f32.
// Also times chains of the instruction: cycles per instruction and SIMD (s_memtime) and the clock held (s_memrealtime = 100 MHz).
//   hipcc --offload-arch=gfx950 -O3 -o fp4_dot fp4_dot.hip && ./fp4_dot
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
typedef int v8i __attribute__((ext_vector_type(8)));
typedef int v4i __attribute__((ext_vector_type(4)));
typedef float v16f __attribute__((ext_vector_type(16)));

__device__ __forceinline__ v4i nib(uint32_t x)
{
    v4i r;
    r.x = (int)(x & 0x11111111u);
    r.y = (int)((x >> 1) & 0x11111111u);
    r.z = (int)((x >> 2) & 0x11111111u);
    r.w = (int)((x >> 3) & 0x11111111u);
    return r;
}

// rows[32][2], qs[32][2] words; out[32 rows][32 queries]
__global__ void k_check(const uint32_t *rows, const uint32_t *qs, float *out, int bias)
{
    const int lane = threadIdx.x, c = lane & 31, h = lane >> 5;
    v4i a = nib(rows[c * 2 + h]);
    v4i q = nib(qs[c * 2 + h]);
    q = (q << 3) | 0x22222222;
    v8i A = {a.x, a.y, a.z, a.w, 0, 0, 0, 0}, B = {q.x, q.y, q.z, q.w, 0, 0, 0, 0};
    v16f acc;
    for (int i = 0; i < 16; i++) acc[i] = (float)((i & 3) + 8 * (i >> 2) + 4 * h + bias);
    acc = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(A, B, acc, 4, 4, 0, 134, 0, 133);
    for (int i = 0; i < 16; i++) out[((i & 3) + 8 * (i >> 2) + 4 * h) * 32 + c] = acc[i];
}

// MODE 0: FP4 32x32x64 scaled; 1: int8 32x32x32; 2: FP8 (e4m3) 32x32x64 scaled; 3: FP4 16x16x128 scaled.  NACC independent accumulators.
typedef float v4f __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));
template <int MODE, int NACC>
__global__ void k_time(const uint32_t *rows, float *out, long long *cyc, int iters)
{
    const int lane = threadIdx.x & 63;
    v4i a = nib(rows[lane]);
    v4i q = (nib(rows[lane + 64]) << 3) | 0x22222222;
    v8i A = {a.x, a.y, a.z, a.w, a.x, a.y, a.z, a.w}, B = {q.x, q.y, q.z, q.w, q.x, q.y, q.z, q.w};
    v16f acc[NACC];
    v16i iacc[NACC];
    v4f sacc[NACC];
    for (int n = 0; n < NACC; n++)
        for (int i = 0; i < 16; i++) {
            acc[n][i] = (float)i;
            iacc[n][i] = i;
            sacc[n][i & 3] = (float)i;
        }
    const long long r0 = __builtin_amdgcn_s_memrealtime();
    const long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int n = 0; n < NACC; n++) {
            if (MODE == 0) acc[n] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(A, B, acc[n], 4, 4, 0, 127, 0, 127);
            if (MODE == 1) iacc[n] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, q, iacc[n], 0, 0, 0);
            if (MODE == 2) acc[n] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(A, B, acc[n], 0, 0, 0, 127, 0, 127);
            if (MODE == 3) sacc[n] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(A, B, sacc[n], 4, 4, 0, 127, 0, 127);
        }
    }
    const long long t1 = __builtin_amdgcn_s_memtime();
    const long long r1 = __builtin_amdgcn_s_memrealtime();
    float s = 0;
    for (int n = 0; n < NACC; n++)
        for (int i = 0; i < 16; i++) s += acc[n][i] + (float)iacc[n][i] + sacc[n][i & 3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) {
        cyc[blockIdx.x] = t1 - t0;
        cyc[1024 + blockIdx.x] = r1 - r0;
    }
}

// The inner loop of k_knn2_mfma, piece by piece: V0 eight matrix instructions per 32-row tile (2 accumulators x 4 k-steps), operands in
// registers; V1 + the A fragment of every k-step from LDS (ds_read_b128); V2 + the accumulators' start values from LDS; V3 + the fold
// (minimum of the 16 keys, update of the running pair).  Cycles per matrix instruction and SIMD with 4 waves per SIMD.
template <int V>
__global__ __launch_bounds__(256, 4) void k_loop(const uint32_t *rows, int *out, long long *cyc, int iters)
{
    __shared__ v4i sA[8][64];
    __shared__ v4i sT[4][2];
    const int lane = threadIdx.x & 63, c = lane & 31, h = lane >> 5;
    for (int i = threadIdx.x; i < 8 * 64; i += 256) sA[i >> 6][i & 63] = nib(rows[i & 63] + i);
    if (threadIdx.x < 8) sT[threadIdx.x >> 1][threadIdx.x & 1] = v4i{1 << 21, (1 << 21) + 1, (1 << 21) + 2, (1 << 21) + 3};
    __syncthreads();
    v4i Bq[2][4];
    for (int t = 0; t < 2; t++)
        for (int k = 0; k < 4; k++) Bq[t][k] = (nib(rows[(lane + 7 * t + 3 * k) & 63]) << 3) | 0x22222222;
    int m1[2] = {0x4C000000, 0x4C000000}, m2[2] = {0x4C000000, 0x4C000000};
    v4i a_reg = nib(rows[lane]);
    const long long r0 = __builtin_amdgcn_s_memrealtime();
    const long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; it++) {
        v16f acc[2];
        for (int g = 0; g < 4; g++) {
            v4i T4 = V >= 2 ? sT[g][h] : v4i{it, it + 1, it + 2, it + 3};
            for (int e = 0; e < 4; e++)
                for (int t = 0; t < 2; t++) acc[t][4 * g + e] = V >= 2 ? __int_as_float(T4[e]) : (float)T4[e];
        }
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const v4i a = V >= 1 ? sA[2 * k + h][c + 32 * (it & 1)] : a_reg;
            const v8i A8 = {a.x, a.y, a.z, a.w, 0, 0, 0, 0};
#pragma unroll
            for (int t = 0; t < 2; t++) {
                const v8i B8 = {Bq[t][k].x, Bq[t][k].y, Bq[t][k].z, Bq[t][k].w, 0, 0, 0, 0};
                acc[t] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(A8, B8, acc[t], 4, 4, 0, 134, 0, 133);
            }
        }
        if (V >= 3) {
#pragma unroll
            for (int t = 0; t < 2; t++) {
                int x[16];
                for (int r = 0; r < 16; r++) x[r] = __float_as_int(acc[t][r]);
                const int a0 = min(min(x[0], x[1]), x[2]), a1 = min(min(x[3], x[4]), x[5]), a2 = min(min(x[6], x[7]), x[8]);
                const int a3 = min(min(x[9], x[10]), x[11]), a4 = min(min(x[12], x[13]), x[14]);
                const int key = min(min(min(a0, a1), a2), min(min(a3, a4), x[15]));
                m2[t] = min(max(m1[t], key), m2[t]);
                m1[t] = min(m1[t], key);
            }
        } else {
            m1[0] ^= __float_as_int(acc[0][it & 15]);
            m1[1] ^= __float_as_int(acc[1][it & 15]);
        }
    }
    const long long t1 = __builtin_amdgcn_s_memtime();
    const long long r1 = __builtin_amdgcn_s_memrealtime();
    out[blockIdx.x * 256 + threadIdx.x] = m1[0] + m1[1] + m2[0] + m2[1];
    if (threadIdx.x == 0) {
        cyc[blockIdx.x] = t1 - t0;
        cyc[1024 + blockIdx.x] = r1 - r0;
    }
}

template <int V>
static void loop_one(const char *name, const uint32_t *dr, float *dout, long long *dc)
{
    long long hc[2048];
    const int iters = 4000;
    for (int rep = 0; rep < 2; rep++) hipLaunchKernelGGL((k_loop<V>), dim3(1024), dim3(256), 0, 0, dr, (int *)dout, dc, iters);
    hipDeviceSynchronize();
    hipMemcpy(hc, dc, sizeof hc, hipMemcpyDeviceToHost);
    double m = 0, r = 0;
    for (int i = 0; i < 1024; i++) {
        m += (double)hc[i];
        r += (double)hc[1024 + i];
    }
    // 4 workgroups of 4 waves per CU = 4 waves per SIMD, 8 instructions per wave and iteration
    printf("%-44s %.1f cycles per matrix instruction and SIMD, clock %.2f GHz\n", name, m / 1024 / (8.0 * iters * 4), m / r * 0.1);
}

template <int MODE, int NACC>
static void time_one(const char *name, const uint32_t *dr, float *dout, long long *dc)
{
    long long hc[2048];
    const int iters = 20000 / NACC;
    for (int wpb = 1; wpb <= 2; wpb++) {
        for (int rep = 0; rep < 2; rep++) hipLaunchKernelGGL((k_time<MODE, NACC>), dim3(1024), dim3(256 * wpb), 0, 0, dr, dout, dc, iters);
        hipDeviceSynchronize();
        hipMemcpy(hc, dc, sizeof hc, hipMemcpyDeviceToHost);
        double m = 0, r = 0;
        for (int i = 0; i < 1024; i++) {
            m += (double)hc[i];
            r += (double)hc[1024 + i];
        }
        printf("%-28s %d accumulators, %d waves per SIMD: %.1f cycles per instruction and SIMD, clock %.2f GHz\n", name, NACC, wpb,
               m / 1024 / ((double)NACC * iters * wpb), m / r * 0.1);
    }
}

int main()
{
    uint32_t hr[64], hq[64];
    uint32_t *dr, *dq;
    float *dout;
    hipMalloc(&dr, 4096);
    hipMalloc(&dq, 4096);
    hipMalloc(&dout, 4 << 20);
    const int bias = 1 << 21;
    long bad = 0, total = 0;
    srand(5);
    for (int trial = 0; trial < 200; trial++) {
        for (int i = 0; i < 64; i++) {
            hr[i] = (uint32_t)rand() ^ ((uint32_t)rand() << 16);
            hq[i] = (uint32_t)rand() ^ ((uint32_t)rand() << 16);
            if (trial == 0) { hr[i] = 0xFFFFFFFFu; hq[i] = 0; }            // extreme: +64 * 4096
            if (trial == 1) { hr[i] = 0xFFFFFFFFu; hq[i] = 0xFFFFFFFFu; }  // extreme: -64 * 4096
        }
        hipMemcpy(dr, hr, sizeof hr, hipMemcpyHostToDevice);
        hipMemcpy(dq, hq, sizeof hq, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(k_check, dim3(1), dim3(64), 0, 0, dr, dq, dout, bias);
        float ho[1024];
        hipMemcpy(ho, dout, sizeof ho, hipMemcpyDeviceToHost);
        for (int r = 0; r < 32; r++)
            for (int c = 0; c < 32; c++) {
                int s = 0;
                for (int w = 0; w < 2; w++) s += __builtin_popcount(hr[r * 2 + w]) - 2 * __builtin_popcount(hr[r * 2 + w] & hq[c * 2 + w]);
                const float want = (float)(4096 * s + r + bias);
                total++;
                if (ho[r * 32 + c] != want) {
                    if (bad < 5) printf("trial %d row %d query %d: got %.1f want %.1f\n", trial, r, c, ho[r * 32 + c], want);
                    bad++;
                }
            }
    }
    printf("exactness: %ld of %ld values differ\n", bad, total);
    long long *dc;
    hipMalloc(&dc, 2048 * sizeof(long long));
    loop_one<0>("loop: 8 instructions, operands in registers", dr, dout, dc);
    loop_one<1>("loop: + A fragments from LDS", dr, dout, dc);
    loop_one<2>("loop: + start values from LDS", dr, dout, dc);
    loop_one<3>("loop: + fold of the 16 keys", dr, dout, dc);
    time_one<1, 2>("i8 32x32x32", dr, dout, dc);
    time_one<0, 2>("fp4 32x32x64 scaled", dr, dout, dc);
    time_one<0, 4>("fp4 32x32x64 scaled", dr, dout, dc);
    time_one<2, 2>("fp8 32x32x64 scaled", dr, dout, dc);
    time_one<3, 4>("fp4 16x16x128 scaled", dr, dout, dc);
    return bad ? 1 : 0;
}
