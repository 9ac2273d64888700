// valu_rate.hip -- issue rate of individual gfx950 vector instructions (wave64): every lane of every SIMD runs a long
// unrolled block of one instruction on 8 independent register chains; the result is printed as wave-instructions per
// cycle per SIMD (1/4 = one wave64 instruction every four cycles = "full rate" on a 16-lane SIMD).
// Build: hipcc -O3 --offload-arch=gfx950 -o valu_rate tools/microbench/valu_rate.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)
#define DEF_KERNEL(NAME, ASMSTR)                                                                   \
    __global__ __launch_bounds__(256) void NAME(unsigned *out, int iters, unsigned seed)          \
    {                                                                                              \
        unsigned r[8], a = threadIdx.x * 2654435761u + seed, b = a ^ 0x9e3779b9u, c = a + 77u;     \
        for (int k = 0; k < 8; k++) r[k] = a + k * 0x01010101u;                                    \
        for (int it = 0; it < iters; it++) {                                                       \
            _Pragma("unroll") for (int u = 0; u < 8; u++)                                          \
            {                                                                                      \
                asm volatile(ASMSTR : "+v"(r[0]) : "v"(b), "v"(c));                                \
                asm volatile(ASMSTR : "+v"(r[1]) : "v"(b), "v"(c));                                \
                asm volatile(ASMSTR : "+v"(r[2]) : "v"(b), "v"(c));                                \
                asm volatile(ASMSTR : "+v"(r[3]) : "v"(b), "v"(c));                                \
                asm volatile(ASMSTR : "+v"(r[4]) : "v"(b), "v"(c));                                \
                asm volatile(ASMSTR : "+v"(r[5]) : "v"(b), "v"(c));                                \
                asm volatile(ASMSTR : "+v"(r[6]) : "v"(b), "v"(c));                                \
                asm volatile(ASMSTR : "+v"(r[7]) : "v"(b), "v"(c));                                \
            }                                                                                      \
        }                                                                                          \
        unsigned s = 0;                                                                            \
        for (int k = 0; k < 8; k++) s ^= r[k];                                                     \
        if (s == 0x12345678u) out[threadIdx.x] = s;                                                \
    }

// %0 = accumulator (in/out), %1, %2 = other operands
DEF_KERNEL(k_add, "v_add_u32 %0, %0, %1")
DEF_KERNEL(k_xor, "v_xor_b32 %0, %0, %1")
DEF_KERNEL(k_add3, "v_add3_u32 %0, %0, %1, %2")
DEF_KERNEL(k_lshl_or, "v_lshl_or_b32 %0, %0, 3, %1")
DEF_KERNEL(k_and_or, "v_and_or_b32 %0, %0, %1, %2")
DEF_KERNEL(k_bfe, "v_bfe_u32 %0, %0, 3, 9")
DEF_KERNEL(k_perm, "v_perm_b32 %0, %0, %1, %2")
DEF_KERNEL(k_alignbyte, "v_alignbyte_b32 %0, %0, %1, 1")
DEF_KERNEL(k_min3_i32, "v_min3_i32 %0, %0, %1, %2")
DEF_KERNEL(k_max3_i32, "v_max3_i32 %0, %0, %1, %2")
DEF_KERNEL(k_min_i32, "v_min_i32 %0, %0, %1")
DEF_KERNEL(k_mad_i24, "v_mad_i32_i24 %0, %0, %1, %2")
DEF_KERNEL(k_mul_i24, "v_mul_i32_i24 %0, %0, %1")
DEF_KERNEL(k_mul_lo, "v_mul_lo_u32 %0, %0, %1")
DEF_KERNEL(k_mul_hi, "v_mul_hi_u32 %0, %0, %1")
DEF_KERNEL(k_pk_min_u16, "v_pk_min_u16 %0, %0, %1")
DEF_KERNEL(k_pk_sub_u16, "v_pk_sub_u16 %0, %0, %1")
DEF_KERNEL(k_pk_min3_f16, "v_pk_minimum3_f16 %0, %0, %1, %2")
DEF_KERNEL(k_pk_max3_f16, "v_pk_maximum3_f16 %0, %0, %1, %2")
DEF_KERNEL(k_dot2_u16, "v_dot2_u32_u16 %0, %1, %2, %0")
DEF_KERNEL(k_dot4_u8, "v_dot4_u32_u8 %0, %1, %2, %0")
DEF_KERNEL(k_sad_u8, "v_sad_u8 %0, %1, %2, %0")
DEF_KERNEL(k_mul_f32, "v_mul_f32 %0, %0, %1")
DEF_KERNEL(k_fma_f32, "v_fma_f32 %0, %0, %1, %2")
DEF_KERNEL(k_cvt_i32_f32, "v_cvt_i32_f32 %0, %0")
DEF_KERNEL(k_cvt_f32_u32, "v_cvt_f32_u32 %0, %0")
DEF_KERNEL(k_rndne, "v_rndne_f32 %0, %0")
DEF_KERNEL(k_cndmask, "v_cndmask_b32 %0, %0, %1, vcc")
DEF_KERNEL(k_bcnt, "v_bcnt_u32_b32 %0, %0, %1")
DEF_KERNEL(k_ffbl, "v_ffbl_b32 %0, %0")
DEF_KERNEL(k_mbcnt, "v_mbcnt_lo_u32_b32 %0, %1, %0")
DEF_KERNEL(k_dpp_add, "v_add_u32_dpp %0, %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf")
DEF_KERNEL(k_sdwa_add, "v_add_u32_sdwa %0, %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:DWORD")
DEF_KERNEL(k_bitop3, "v_bitop3_b32 %0, %0, %1, %2 bitop3:0x36")
DEF_KERNEL(k_cmp, "v_cmp_lt_u32 vcc, %0, %1")
DEF_KERNEL(k_lerp, "v_lerp_u8 %0, %0, %1, %2")
DEF_KERNEL(k_not, "v_not_b32 %0, %0")
DEF_KERNEL(k_or, "v_or_b32 %0, %0, %1")
DEF_KERNEL(k_and, "v_and_b32 %0, %0, %1")
DEF_KERNEL(k_or3, "v_or3_b32 %0, %0, %1, %2")
DEF_KERNEL(k_lshrrev, "v_lshrrev_b32 %0, 3, %0")
DEF_KERNEL(k_pk_max_i16, "v_pk_max_i16 %0, %0, %1")
DEF_KERNEL(k_msad, "v_msad_u8 %0, %1, %2, %0")
DEF_KERNEL(k_sub, "v_sub_u32 %0, %0, %1")
DEF_KERNEL(k_mov, "v_mov_b32 %0, %1")
DEF_KERNEL(k_max_u16, "v_max_u16 %0, %0, %1")
DEF_KERNEL(k_max_i32, "v_max_i32 %0, %0, %1")
DEF_KERNEL(k_max_f32, "v_max_f32 %0, %0, %1")
DEF_KERNEL(k_min3_f32, "v_min3_f32 %0, %0, %1, %2")
DEF_KERNEL(k_mad_u32_u24, "v_mad_u32_u24 %0, %0, %1, %2")

struct Entry {
    const char *name;
    void (*fn)(unsigned *, int, unsigned);
};

int main()
{
    Entry tab[] = {{"v_lerp_u8", k_lerp}, {"v_not_b32", k_not}, {"v_or_b32", k_or}, {"v_and_b32", k_and}, {"v_or3_b32", k_or3}, {"v_lshrrev_b32", k_lshrrev}, {"v_pk_max_i16", k_pk_max_i16}, {"v_msad_u8", k_msad}, {"v_sub_u32", k_sub}, {"v_mov_b32", k_mov}, {"v_max_u16", k_max_u16}, {"v_max_i32", k_max_i32}, {"v_max_f32", k_max_f32}, {"v_min3_f32", k_min3_f32}, {"v_mad_u32_u24", k_mad_u32_u24},
                   {"v_add_u32", k_add}, {"v_xor_b32", k_xor}, {"v_add3_u32", k_add3}, {"v_lshl_or_b32", k_lshl_or},
                   {"v_and_or_b32", k_and_or}, {"v_bfe_u32", k_bfe}, {"v_perm_b32", k_perm}, {"v_alignbyte_b32", k_alignbyte},
                   {"v_min3_i32", k_min3_i32}, {"v_max3_i32", k_max3_i32}, {"v_min_i32", k_min_i32},
                   {"v_mad_i32_i24", k_mad_i24}, {"v_mul_i32_i24", k_mul_i24}, {"v_mul_lo_u32", k_mul_lo},
                   {"v_mul_hi_u32", k_mul_hi}, {"v_pk_min_u16", k_pk_min_u16},
                   {"v_pk_sub_u16", k_pk_sub_u16}, {"v_pk_minimum3_f16", k_pk_min3_f16}, {"v_pk_maximum3_f16", k_pk_max3_f16},
                   {"v_dot2_u32_u16", k_dot2_u16}, {"v_dot4_u32_u8", k_dot4_u8}, {"v_sad_u8", k_sad_u8},
                   {"v_mul_f32", k_mul_f32}, {"v_fma_f32", k_fma_f32}, {"v_cvt_i32_f32", k_cvt_i32_f32},
                   {"v_cvt_f32_u32", k_cvt_f32_u32}, {"v_rndne_f32", k_rndne}, {"v_cndmask_b32", k_cndmask},
                   {"v_bcnt_u32_b32", k_bcnt}, {"v_ffbl_b32", k_ffbl}, {"v_mbcnt_lo", k_mbcnt}, {"v_add_u32_dpp", k_dpp_add},
                   {"v_add_u32_sdwa", k_sdwa_add}, {"v_bitop3_b32", k_bitop3}, {"v_cmp_lt_u32", k_cmp}};
    hipDeviceProp_t prop;
    hipGetDeviceProperties(&prop, 0);
    const int cus = prop.multiProcessorCount;
    const double ghz = prop.clockRate * 1e-6;
    unsigned *d;
    hipMalloc(&d, 4096);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const int iters = 2000, blocks = cus * 8;   // 8 blocks x 4 waves per CU = 8 waves per SIMD
    printf("device: %s, %d CUs, clock %.3f GHz (reported)\n| instruction | wave-instr / cycle / SIMD | cycles per wave64 instr |\n|---|---|---|\n",
           prop.name, cus, ghz);
    for (auto &e : tab) {
        hipLaunchKernelGGL(e.fn, dim3(blocks), dim3(256), 0, 0, d, 10, 1u);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        hipLaunchKernelGGL(e.fn, dim3(blocks), dim3(256), 0, 0, d, iters, 1u);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms = 0;
        hipEventElapsedTime(&ms, e0, e1);
        const double instr_per_simd = (double)iters * 64.0 * (blocks * 4.0) / (cus * 4.0);   // wave-instr per SIMD
        const double cycles = ms * 1e-3 * ghz * 1e9;
        printf("| %s | %.3f | %.2f |\n", e.name, instr_per_simd / cycles, cycles / instr_per_simd);
    }
    return 0;
}
