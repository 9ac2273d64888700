// ta_cost.hip -- cycles the texture-address unit spends per wave load instruction, by shape: what k_describe (92 % TA busy),
// k_vocab_transform (87 %) and k_blur (66 %) should request.  Every pattern reads a 16 KB buffer that stays in L1, one wave per
// SIMD issuing independent loads back to back (8 in flight), so the time per instruction is the address / data path, not memory.
//   hipcc --offload-arch=gfx950 -O3 -o ta_cost ta_cost.hip && ./ta_cost
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstring>
typedef unsigned int u32;
struct __attribute__((packed, aligned(1))) U4u { uint4 v; };
struct __attribute__((packed, aligned(1))) U1u { u32 v; };

enum { DW_LINEAR, DW_ROWS40, DW_ROWS40_STRIDE, DX4_LINEAR, DX4_UNALIGNED, DX4_ROWS_2LANES, DX4_ROWS_2LANES_UNAL, DX4_SCATTER, DX4_ROWS_4LANES, DX2_LINEAR, U8_ROWS, DMA_DW_ROWS40, DMA_DX4_ROWS4, DMA_DX4_ROWS3_UNAL, DX4_ROWS_2LANES_OFF4, DX4_ROWS_2LANES_OFF8, DX3_ROWS_3LANES_OFF4, DX4_ROWS_3LANES_AL, NPAT };
static const char *names[NPAT] = {
    "dword, 64 consecutive dwords (256 B)",
    "dword, rows of 10 dwords, rows 64 B apart (describe C, narrow image)",
    "dword, rows of 10 dwords, rows 704 B apart (describe C)",
    "dwordx4, 64 consecutive chunks (1 KB)",
    "dwordx4, consecutive chunks at a byte offset of 5",
    "dwordx4, 2 lanes per row (32 B), rows 704 B apart, aligned",
    "dwordx4, 2 lanes per row, rows 704 B apart, byte offset 5 (describe A)",
    "dwordx4, every lane its own 128-B line (vocabulary, thread per descriptor)",
    "dwordx4, 4 lanes per row (64 B), rows 704 B apart, aligned",
    "dwordx2, 64 consecutive (512 B)",
    "ubyte, 37 lanes per row",
    "LDS-DMA dword, rows of 10 dwords, rows 704 B apart (describe C as shipped)",
    "LDS-DMA dwordx4, 4 lanes per row (64 B), rows 704 B apart, aligned",
    "LDS-DMA dwordx4, 3 lanes per row (48 B), rows 704 B apart, byte offset 5",
    "dwordx4, 2 lanes per row, rows 704 B apart, byte offset 4 (dword aligned)",
    "dwordx4, 2 lanes per row, rows 704 B apart, byte offset 8",
    "dwordx3, 3 lanes per row (36 B), rows 704 B apart, byte offset 4",
    "dwordx4, 3 lanes per row (48 B), rows 704 B apart, aligned",
};

template <int PAT>
__global__ __launch_bounds__(64) void k(const uint8_t *buf, u32 *out, long long *cyc, int iters)
{
    const int lane = threadIdx.x;
    size_t off;
    switch (PAT) {
    case DW_LINEAR: off = 4 * lane; break;
    case DW_ROWS40: off = (lane / 10) * 64 + 4 * (lane % 10); break;
    case DW_ROWS40_STRIDE: off = (lane / 10) * 704 + 4 * (lane % 10); break;
    case DX4_LINEAR: off = 16 * lane; break;
    case DX4_UNALIGNED: off = 16 * lane + 5; break;
    case DX4_ROWS_2LANES: off = (lane >> 1) * 704 % 16000 + 16 * (lane & 1); break;
    case DX4_ROWS_2LANES_UNAL: off = (lane >> 1) * 704 % 16000 + 16 * (lane & 1) + 5; break;
    case DX4_SCATTER: off = (lane * 128 * 37) % 16384; break;
    case DX4_ROWS_4LANES: off = (lane >> 2) * 704 % 16000 + 16 * (lane & 3); break;
    case DX2_LINEAR: off = 8 * lane; break;
    case DX4_ROWS_2LANES_OFF4: off = (lane >> 1) * 704 % 16000 + 16 * (lane & 1) + 4; break;
    case DX4_ROWS_2LANES_OFF8: off = (lane >> 1) * 704 % 16000 + 16 * (lane & 1) + 8; break;
    case DX3_ROWS_3LANES_OFF4: off = (lane / 3) * 704 % 16000 + 12 * (lane % 3) + 4; break;
    case DX4_ROWS_3LANES_AL: off = (lane / 3) * 704 % 16000 + 16 * (lane % 3); break;
    case DMA_DW_ROWS40: off = (lane / 10) * 704 + 4 * (lane % 10); break;
    case DMA_DX4_ROWS4: off = (lane >> 2) * 704 % 16000 + 16 * (lane & 3); break;
    case DMA_DX4_ROWS3_UNAL: off = (lane / 3) * 704 % 16000 + 16 * (lane % 3) + 5; break;
    default: off = (lane / 37) * 704 + lane % 37; break;
    }
    const uint8_t *p = buf + off;
    u32 acc = 0;
    const long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; it++) {
        // 8 independent loads per trip at addresses that differ by a multiple of 16 (same pattern), results folded afterwards
        if (PAT == DW_LINEAR || PAT == DW_ROWS40 || PAT == DW_ROWS40_STRIDE) {
            u32 v[8];
#pragma unroll
            for (int j = 0; j < 8; j++) {
                const u32 *q = reinterpret_cast<const u32 *>(p + ((it + j) & 7) * 16);
                asm volatile("global_load_dword %0, %1, off" : "=v"(v[j]) : "v"(q));
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
            for (int j = 0; j < 8; j++) acc ^= v[j];
        } else if (PAT == U8_ROWS) {
            u32 v[8];
#pragma unroll
            for (int j = 0; j < 8; j++) {
                const uint8_t *q = p + ((it + j) & 7) * 16;
                asm volatile("global_load_ubyte %0, %1, off" : "=v"(v[j]) : "v"(q));
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
            for (int j = 0; j < 8; j++) acc ^= v[j];
        } else if (PAT == DMA_DW_ROWS40 || PAT == DMA_DX4_ROWS4 || PAT == DMA_DX4_ROWS3_UNAL) {
            __shared__ __attribute__((aligned(16))) uint8_t s_dst[8][1024];
#pragma unroll
            for (int j = 0; j < 8; j++) {
                const uint8_t *q = p + ((it + j) & 7) * 16;
                const u32 lds = (u32)(uintptr_t)&s_dst[j][0];
                if (PAT == DMA_DW_ROWS40)
                    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dword %0, off" ::"v"(q), "s"(lds) : "memory", "m0");
                else
                    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(q), "s"(lds) : "memory", "m0");
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            acc ^= s_dst[it & 7][lane];
        } else if (PAT == DX3_ROWS_3LANES_OFF4) {
            typedef unsigned v3u __attribute__((ext_vector_type(3)));
            v3u v[8];
#pragma unroll
            for (int j = 0; j < 8; j++) {
                const uint8_t *q = p + ((it + j) & 7) * 16;
                asm volatile("global_load_dwordx3 %0, %1, off" : "=v"(v[j]) : "v"(q));
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
            for (int j = 0; j < 8; j++) acc ^= v[j].x ^ v[j].y ^ v[j].z;
        } else if (PAT == DX2_LINEAR) {
            uint2 v[8];
#pragma unroll
            for (int j = 0; j < 8; j++) {
                const uint2 *q = reinterpret_cast<const uint2 *>(p + ((it + j) & 7) * 16);
                asm volatile("global_load_dwordx2 %0, %1, off" : "=v"(v[j]) : "v"(q));
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
            for (int j = 0; j < 8; j++) acc ^= v[j].x ^ v[j].y;
        } else {
            uint4 v[8];
#pragma unroll
            for (int j = 0; j < 8; j++) {
                const U4u *q = reinterpret_cast<const U4u *>(p + ((it + j) & 7) * 16);
                asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(v[j]) : "v"(q));
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
            for (int j = 0; j < 8; j++) acc ^= v[j].x ^ v[j].y ^ v[j].z ^ v[j].w;
        }
    }
    const long long t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * 64 + lane] = acc;
    if (lane == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int PAT>
static void run(const uint8_t *buf, u32 *out, long long *cyc, int wavesPerCu)
{
    const int iters = 2000, blocks = 256 * wavesPerCu;
    for (int r = 0; r < 2; r++) hipLaunchKernelGGL(k<PAT>, dim3(blocks), dim3(64), 0, 0, buf, out, cyc, iters);
    hipDeviceSynchronize();
    static long long h[4096];
    hipMemcpy(h, cyc, blocks * sizeof(long long), hipMemcpyDeviceToHost);
    double m = 0;
    for (int i = 0; i < blocks; i++) m += (double)h[i];
    m /= blocks;
    // one TA per CU: wavesPerCu waves share it
    printf("%-75s %2d waves per CU: %6.1f cycles per instruction and CU\n", names[PAT], wavesPerCu, m / (8.0 * iters * wavesPerCu));
}

int main()
{
    uint8_t *buf;
    u32 *out;
    long long *cyc;
    hipMalloc(&buf, 1 << 16);
    hipMemset(buf, 1, 1 << 16);
    hipMalloc(&out, 4096 * 64 * 4);
    hipMalloc(&cyc, 4096 * 8);
    for (int w = 4; w <= 8; w += 4) {
        run<DW_LINEAR>(buf, out, cyc, w);
        run<DW_ROWS40>(buf, out, cyc, w);
        run<DW_ROWS40_STRIDE>(buf, out, cyc, w);
        run<DX2_LINEAR>(buf, out, cyc, w);
        run<DX4_LINEAR>(buf, out, cyc, w);
        run<DX4_UNALIGNED>(buf, out, cyc, w);
        run<DX4_ROWS_2LANES>(buf, out, cyc, w);
        run<DX4_ROWS_2LANES_UNAL>(buf, out, cyc, w);
        run<DX4_ROWS_4LANES>(buf, out, cyc, w);
        run<DX4_SCATTER>(buf, out, cyc, w);
        run<U8_ROWS>(buf, out, cyc, w);
        run<DMA_DW_ROWS40>(buf, out, cyc, w);
        run<DMA_DX4_ROWS4>(buf, out, cyc, w);
        run<DMA_DX4_ROWS3_UNAL>(buf, out, cyc, w);
        run<DX4_ROWS_2LANES_OFF4>(buf, out, cyc, w);
        run<DX4_ROWS_2LANES_OFF8>(buf, out, cyc, w);
        run<DX3_ROWS_3LANES_OFF4>(buf, out, cyc, w);
        run<DX4_ROWS_3LANES_AL>(buf, out, cyc, w);
    }
    return 0;
}
