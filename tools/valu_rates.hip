// valu_rates.hip -- issue rate of the vector instructions the hot kernels are made of, measured on the device (gfx950).
// Every kernel runs ITERS x 32 independent instructions of one opcode per wave (eight register chains, so that neither
// dependencies nor the loop's three scalar instructions matter) on a grid that puts W waves on every SIMD; the table gives
// SIMD cycles per wave-instruction = launch time x clock x (4 SIMDs x CUs) / (waves x instructions per wave).  The clock is
// measured with s_memtime around the v_add_u32 loop of one wave (shader cycles) against the same loop's wall time.
//   build: hipcc -O2 --offload-arch=gfx950 -o valu_rates tools/valu_rates.hip     run: ./valu_rates [iters]
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

#define CHECK(x)                                                                          \
    do {                                                                                  \
        hipError_t e_ = (x);                                                              \
        if (e_ != hipSuccess) {                                                           \
            fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_));     \
            exit(1);                                                                      \
        }                                                                                 \
    } while (0)

// eight independent instructions; F is the per-register instruction text with %N operand numbers: d = %0..%7, k = %8, m = %9
#define OP8(F) F(0) F(1) F(2) F(3) F(4) F(5) F(6) F(7)
#define KERNEL32(NAME, F)                                                                                        \
    __global__ __launch_bounds__(256) void NAME(uint32_t *out, int iters, uint32_t k, uint32_t m)                \
    {                                                                                                            \
        uint32_t a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, \
                 a7 = a0 + 7;                                                                                    \
        for (int i = 0; i < iters; i++) {                                                                        \
            asm volatile(OP8(F) OP8(F) OP8(F) OP8(F)                                                             \
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)        \
                         : "v"(k), "v"(m)                                                                        \
                         : "vcc", "s20", "s21");                                                                 \
        }                                                                                                        \
        if ((a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7) == 0x12345u) out[threadIdx.x] = a0;                          \
    }
// the same with 64-bit registers
#define KERNEL64(NAME, F)                                                                                        \
    __global__ __launch_bounds__(256) void NAME(uint32_t *out, int iters, uint32_t k32, uint32_t m32)            \
    {                                                                                                            \
        unsigned long long a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5,    \
                           a6 = a0 + 6, a7 = a0 + 7, k = k32, m = m32;                                           \
        for (int i = 0; i < iters; i++) {                                                                        \
            asm volatile(OP8(F) OP8(F) OP8(F) OP8(F)                                                             \
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)        \
                         : "v"(k), "v"(m)                                                                        \
                         : "vcc", "s20", "s21");                                                                 \
        }                                                                                                        \
        if ((a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7) == 0x12345u) out[threadIdx.x] = (uint32_t)a0;                \
    }

#define F_ADD(n) "v_add_u32 %" #n ", %" #n ", %8\n\t"
#define F_AND(n) "v_and_b32 %" #n ", %" #n ", %8\n\t"
#define F_LSHL(n) "v_lshlrev_b32 %" #n ", 1, %" #n "\n\t"
#define F_ADD3(n) "v_add3_u32 %" #n ", %" #n ", %8, %9\n\t"
#define F_LSHLOR(n) "v_lshl_or_b32 %" #n ", %" #n ", 3, %8\n\t"
#define F_BFE(n) "v_bfe_u32 %" #n ", %" #n ", 3, 9\n\t"
#define F_MULLO(n) "v_mul_lo_u32 %" #n ", %" #n ", %8\n\t"
#define F_MULHI(n) "v_mul_hi_u32 %" #n ", %" #n ", %8\n\t"
#define F_MUL24(n) "v_mul_u32_u24 %" #n ", %" #n ", %8\n\t"
#define F_MULHI24(n) "v_mul_hi_u32_u24 %" #n ", %" #n ", %8\n\t"
#define F_MAD24(n) "v_mad_u32_u24 %" #n ", %" #n ", %8, %9\n\t"
#define F_MADI24(n) "v_mad_i32_i24 %" #n ", %" #n ", %8, %9\n\t"
#define F_FMA(n) "v_fma_f32 %" #n ", %" #n ", %8, %9\n\t"
#define F_FMUL(n) "v_mul_f32 %" #n ", %" #n ", %8\n\t"
#define F_FADD(n) "v_add_f32 %" #n ", %" #n ", %8\n\t"
#define F_RNDNE(n) "v_rndne_f32 %" #n ", %" #n "\n\t"
#define F_CVTI(n) "v_cvt_i32_f32 %" #n ", %" #n "\n\t"
#define F_CVTF(n) "v_cvt_f32_u32 %" #n ", %" #n "\n\t"
#define F_PERM(n) "v_perm_b32 %" #n ", %" #n ", %8, %9\n\t"
#define F_ALIGNB(n) "v_alignbyte_b32 %" #n ", %" #n ", %8, 1\n\t"
#define F_DOT2(n) "v_dot2_u32_u16 %" #n ", %" #n ", %8, %9\n\t"
#define F_DOT4(n) "v_dot4_u32_u8 %" #n ", %" #n ", %8, %9\n\t"
#define F_LERP(n) "v_lerp_u8 %" #n ", %" #n ", %8, %9\n\t"
#define F_SAD(n) "v_sad_u8 %" #n ", %" #n ", %8, %9\n\t"
#define F_MSAD(n) "v_msad_u8 %" #n ", %" #n ", %8, %9\n\t"
#define F_MIN3(n) "v_min3_u32 %" #n ", %" #n ", %8, %9\n\t"
#define F_PKMIN3(n) "v_pk_minimum3_f16 %" #n ", %" #n ", %8, %9\n\t"
#define F_PKADD16(n) "v_pk_add_u16 %" #n ", %" #n ", %8\n\t"
#define F_PKSUB16(n) "v_pk_sub_i16 %" #n ", %" #n ", %8\n\t"
#define F_PKMAX16(n) "v_pk_max_u16 %" #n ", %" #n ", %8\n\t"
#define F_PKMAD16(n) "v_pk_mad_u16 %" #n ", %" #n ", %8, %9\n\t"
#define F_BITOP3(n) "v_bitop3_b32 %" #n ", %" #n ", %8, %9 bitop3:0x96\n\t"
#define F_BCNT(n) "v_bcnt_u32_b32 %" #n ", %" #n ", %8\n\t"
#define F_CNDMASK(n) "v_cndmask_b32 %" #n ", %" #n ", %8, vcc\n\t"
#define F_CMP(n) "v_cmp_lt_u32 vcc, %" #n ", %8\n\t"
#define F_CMPX(n) "v_cmp_lt_u32 s[20:21], %" #n ", %8\n\t"
#define F_DPP(n) "v_add_u32_dpp %" #n ", %" #n ", %8 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
#define F_SDWA(n) "v_add_u32_sdwa %" #n ", %" #n ", %8 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:DWORD\n\t"
#define F_READLANE(n) "v_readlane_b32 s20, %" #n ", 3\n\t"
#define F_READFIRST(n) "v_readfirstlane_b32 s20, %" #n "\n\t"
#define F_MOV(n) "v_mov_b32 %" #n ", %8\n\t"
#define F_LSHLADD64(n) "v_lshl_add_u64 %" #n ", %" #n ", 0, %8\n\t"
#define F_PKMULF(n) "v_pk_mul_f32 %" #n ", %" #n ", %8\n\t"
#define F_PKFMA(n) "v_pk_fma_f32 %" #n ", %" #n ", %8, %9\n\t"
#define F_PKADDF(n) "v_pk_add_f32 %" #n ", %" #n ", %8\n\t"
#define F_MULF64(n) "v_mul_f64 %" #n ", %" #n ", %8\n\t"
#define F_FMAF64(n) "v_fma_f64 %" #n ", %" #n ", %8, %9\n\t"
#define F_ADDF64(n) "v_add_f64 %" #n ", %" #n ", %8\n\t"
#define F_LSHL64(n) "v_lshlrev_b64 %" #n ", 1, %" #n "\n\t"

#define F_OR(n) "v_or_b32 %" #n ", %" #n ", %8\n\t"
#define F_XOR(n) "v_xor_b32 %" #n ", %" #n ", %8\n\t"
#define F_SUB(n) "v_sub_u32 %" #n ", %" #n ", %8\n\t"
#define F_MAXU(n) "v_max_u32 %" #n ", %" #n ", %8\n\t"
#define F_MINI(n) "v_min_i32 %" #n ", %" #n ", %8\n\t"
#define F_LSHR(n) "v_lshrrev_b32 %" #n ", 1, %" #n "\n\t"
#define F_ASHR(n) "v_ashrrev_i32 %" #n ", 1, %" #n "\n\t"
#define F_ANDOR(n) "v_and_or_b32 %" #n ", %" #n ", %8, %9\n\t"
#define F_OR3(n) "v_or3_b32 %" #n ", %" #n ", %8, %9\n\t"
#define F_XAD(n) "v_xad_u32 %" #n ", %" #n ", %8, %9\n\t"
#define F_LSHLADD(n) "v_lshl_add_u32 %" #n ", %" #n ", 2, %8\n\t"
#define F_ADDLSHL(n) "v_add_lshl_u32 %" #n ", %" #n ", %8, 2\n\t"
#define F_MAXF(n) "v_max_f32 %" #n ", %" #n ", %8\n\t"
#define F_MINF(n) "v_min_f32 %" #n ", %" #n ", %8\n\t"
#define F_MAX3F(n) "v_max3_f32 %" #n ", %" #n ", %8, %9\n\t"
#define F_SUBF(n) "v_sub_f32 %" #n ", %" #n ", %8\n\t"
#define F_FMAC(n) "v_fmac_f32 %" #n ", %8, %9\n\t"
#define F_PKADDH(n) "v_pk_add_f16 %" #n ", %" #n ", %8\n\t"
#define F_PKMAXH(n) "v_pk_max_f16 %" #n ", %" #n ", %8\n\t"
#define F_PKMINH(n) "v_pk_min_f16 %" #n ", %" #n ", %8\n\t"
#define F_PKFMAH(n) "v_pk_fma_f16 %" #n ", %" #n ", %8, %9\n\t"
#define F_PKMINU16(n) "v_pk_min_u16 %" #n ", %" #n ", %8\n\t"
#define F_PKLSHL16(n) "v_pk_lshlrev_b16 %" #n ", 1, %" #n "\n\t"
#define F_MBCNT(n) "v_mbcnt_lo_u32_b32 %" #n ", %8, %" #n "\n\t"
#define F_FFBL(n) "v_ffbl_b32 %" #n ", %" #n "\n\t"
#define F_CMPF(n) "v_cmp_lt_f32 vcc, %" #n ", %8\n\t"
#define F_MED3(n) "v_med3_i32 %" #n ", %" #n ", %8, %9\n\t"
#define F_CVTPKU8(n) "v_cvt_pk_u8_f32 %" #n ", %" #n ", 1, %8\n\t"
#define F_CVTUB0(n) "v_cvt_f32_ubyte0 %" #n ", %" #n "\n\t"
#define F_MQSAD(n) "v_sad_u16 %" #n ", %" #n ", %8, %9\n\t"
KERNEL32(k_add, F_ADD)
KERNEL32(k_or, F_OR)
KERNEL32(k_xor, F_XOR)
KERNEL32(k_sub, F_SUB)
KERNEL32(k_maxu, F_MAXU)
KERNEL32(k_mini, F_MINI)
KERNEL32(k_lshr, F_LSHR)
KERNEL32(k_ashr, F_ASHR)
KERNEL32(k_andor, F_ANDOR)
KERNEL32(k_or3, F_OR3)
KERNEL32(k_xad, F_XAD)
KERNEL32(k_lshladd, F_LSHLADD)
KERNEL32(k_addlshl, F_ADDLSHL)
KERNEL32(k_maxf, F_MAXF)
KERNEL32(k_minf, F_MINF)
KERNEL32(k_max3f, F_MAX3F)
KERNEL32(k_subf, F_SUBF)
KERNEL32(k_fmac, F_FMAC)
KERNEL32(k_pkaddh, F_PKADDH)
KERNEL32(k_pkmaxh, F_PKMAXH)
KERNEL32(k_pkminh, F_PKMINH)
KERNEL32(k_pkfmah, F_PKFMAH)
KERNEL32(k_pkminu16, F_PKMINU16)
KERNEL32(k_pklshl16, F_PKLSHL16)
KERNEL32(k_mbcnt, F_MBCNT)
KERNEL32(k_ffbl, F_FFBL)
KERNEL32(k_cmpf, F_CMPF)
KERNEL32(k_med3, F_MED3)
KERNEL32(k_cvtpku8, F_CVTPKU8)
KERNEL32(k_cvtub0, F_CVTUB0)
KERNEL32(k_sadu16, F_MQSAD)
KERNEL32(k_and, F_AND)
KERNEL32(k_lshl, F_LSHL)
KERNEL32(k_add3, F_ADD3)
KERNEL32(k_lshlor, F_LSHLOR)
KERNEL32(k_bfe, F_BFE)
KERNEL32(k_mullo, F_MULLO)
KERNEL32(k_mulhi, F_MULHI)
KERNEL32(k_mul24, F_MUL24)
KERNEL32(k_mulhi24, F_MULHI24)
KERNEL32(k_mad24, F_MAD24)
KERNEL32(k_madi24, F_MADI24)
KERNEL32(k_fma, F_FMA)
KERNEL32(k_fmul, F_FMUL)
KERNEL32(k_fadd, F_FADD)
KERNEL32(k_rndne, F_RNDNE)
KERNEL32(k_cvti, F_CVTI)
KERNEL32(k_cvtf, F_CVTF)
KERNEL32(k_perm, F_PERM)
KERNEL32(k_alignb, F_ALIGNB)
KERNEL32(k_dot2, F_DOT2)
KERNEL32(k_dot4, F_DOT4)
KERNEL32(k_lerp, F_LERP)
KERNEL32(k_sad, F_SAD)
KERNEL32(k_msad, F_MSAD)
KERNEL32(k_min3, F_MIN3)
KERNEL32(k_pkmin3, F_PKMIN3)
KERNEL32(k_pkadd16, F_PKADD16)
KERNEL32(k_pksub16, F_PKSUB16)
KERNEL32(k_pkmax16, F_PKMAX16)
KERNEL32(k_pkmad16, F_PKMAD16)
KERNEL32(k_bitop3, F_BITOP3)
KERNEL32(k_bcnt, F_BCNT)
KERNEL32(k_cndmask, F_CNDMASK)
KERNEL32(k_cmp, F_CMP)
KERNEL32(k_cmp_sgpr, F_CMPX)
KERNEL32(k_dpp, F_DPP)
KERNEL32(k_sdwa, F_SDWA)
KERNEL32(k_readlane, F_READLANE)
KERNEL32(k_readfirst, F_READFIRST)
KERNEL32(k_mov, F_MOV)
KERNEL64(k_lshladd64, F_LSHLADD64)
KERNEL64(k_pkmulf, F_PKMULF)
KERNEL64(k_pkfma, F_PKFMA)
KERNEL64(k_pkaddf, F_PKADDF)
KERNEL64(k_mulf64, F_MULF64)
KERNEL64(k_fmaf64, F_FMAF64)
KERNEL64(k_addf64, F_ADDF64)
KERNEL64(k_lshl64, F_LSHL64)

// one wave: shader cycles (s_memtime) of the v_add_u32 loop, to convert wall time into cycles
__global__ void k_clock(unsigned long long *cyc, int iters, uint32_t k)
{
    uint32_t a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7, m = 0;
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < iters; i++) {
        asm volatile(OP8(F_ADD) OP8(F_ADD) OP8(F_ADD) OP8(F_ADD)
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)
                     : "v"(k), "v"(m));
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    if (threadIdx.x == 0) cyc[0] = t1 - t0 + ((a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7) == 0x12345u);
}

typedef void (*kern_t)(uint32_t *, int, uint32_t, uint32_t);
struct Row {
    const char *name;
    kern_t fn;
};

static double time_ms(kern_t fn, int grid, int block, uint32_t *out, int iters, size_t lds)
{
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    if (lds) CHECK(hipFuncSetAttribute((const void *)fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(fn, dim3(grid), dim3(block), lds, 0, out, iters / 8, 3u, 5u);   // warm up
    CHECK(hipGetLastError());
    CHECK(hipDeviceSynchronize());
    float best = 1e30f;
    for (int r = 0; r < 3; r++) {
        CHECK(hipEventRecord(e0, 0));
        hipLaunchKernelGGL(fn, dim3(grid), dim3(block), lds, 0, out, iters, 3u, 5u);
        CHECK(hipEventRecord(e1, 0));
        CHECK(hipEventSynchronize(e1));
        float ms;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best) best = ms;
    }
    CHECK(hipEventDestroy(e0));
    CHECK(hipEventDestroy(e1));
    return best;
}

int main(int argc, char **argv)
{
    const int iters = argc > 1 ? atoi(argv[1]) : 4000;
    hipDeviceProp_t p;
    CHECK(hipGetDeviceProperties(&p, 0));
    const int cus = p.multiProcessorCount;
    uint32_t *out;
    CHECK(hipMalloc(&out, 4096));
    unsigned long long *cyc;
    CHECK(hipHostMalloc(&cyc, 8));
    printf("device %s, %d CUs, %d iterations x 32 instructions per wave\n", p.gcnArchName, cus, iters);

    // one wave alone on the device: cycles per instruction and the shader clock under that (light) load
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    hipLaunchKernelGGL(k_clock, dim3(1), dim3(64), 0, 0, cyc, iters, 3u);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0, 0));
    hipLaunchKernelGGL(k_clock, dim3(1), dim3(64), 0, 0, cyc, iters * 8, 3u);
    CHECK(hipEventRecord(e1, 0));
    CHECK(hipEventSynchronize(e1));
    float ms1;
    CHECK(hipEventElapsedTime(&ms1, e0, e1));
    printf("one wave alone, v_add_u32: %.3f counter ticks per instruction; counter %.1f MHz (wall %.3f ms)\n",
           (double)cyc[0] / (32.0 * iters * 8), (double)cyc[0] / (ms1 * 1e3), ms1);

    const Row rows[] = {
        {"v_add_u32", k_add}, {"v_and_b32", k_and}, {"v_lshlrev_b32", k_lshl}, {"v_add3_u32", k_add3}, {"v_lshl_or_b32", k_lshlor},
        {"v_bfe_u32", k_bfe}, {"v_mov_b32", k_mov}, {"v_mul_lo_u32", k_mullo}, {"v_mul_hi_u32", k_mulhi}, {"v_mul_u32_u24", k_mul24},
        {"v_mul_hi_u32_u24", k_mulhi24}, {"v_mad_u32_u24", k_mad24}, {"v_mad_i32_i24", k_madi24}, {"v_fma_f32", k_fma},
        {"v_mul_f32", k_fmul}, {"v_add_f32", k_fadd}, {"v_rndne_f32", k_rndne}, {"v_cvt_i32_f32", k_cvti}, {"v_cvt_f32_u32", k_cvtf},
        {"v_perm_b32", k_perm}, {"v_alignbyte_b32", k_alignb}, {"v_dot2_u32_u16", k_dot2}, {"v_dot4_u32_u8", k_dot4}, {"v_lerp_u8", k_lerp},
        {"v_sad_u8", k_sad}, {"v_msad_u8", k_msad}, {"v_min3_u32", k_min3}, {"v_pk_minimum3_f16", k_pkmin3}, {"v_pk_add_u16", k_pkadd16},
        {"v_pk_sub_i16", k_pksub16}, {"v_pk_max_u16", k_pkmax16}, {"v_pk_mad_u16", k_pkmad16}, {"v_bitop3_b32", k_bitop3},
        {"v_bcnt_u32_b32", k_bcnt}, {"v_cndmask_b32 (vcc)", k_cndmask}, {"v_cmp_lt_u32 vcc", k_cmp}, {"v_cmp_lt_u32 sgpr pair", k_cmp_sgpr},
        {"v_add_u32_dpp", k_dpp}, {"v_add_u32_sdwa", k_sdwa}, {"v_readlane_b32", k_readlane}, {"v_readfirstlane_b32", k_readfirst},
        {"v_or_b32", k_or}, {"v_xor_b32", k_xor}, {"v_sub_u32", k_sub}, {"v_max_u32", k_maxu}, {"v_min_i32", k_mini}, {"v_lshrrev_b32", k_lshr},
        {"v_ashrrev_i32", k_ashr}, {"v_and_or_b32", k_andor}, {"v_or3_b32", k_or3}, {"v_xad_u32", k_xad}, {"v_lshl_add_u32", k_lshladd},
        {"v_add_lshl_u32", k_addlshl}, {"v_max_f32", k_maxf}, {"v_min_f32", k_minf}, {"v_max3_f32", k_max3f}, {"v_sub_f32", k_subf},
        {"v_fmac_f32", k_fmac}, {"v_pk_add_f16", k_pkaddh}, {"v_pk_max_f16", k_pkmaxh}, {"v_pk_min_f16", k_pkminh}, {"v_pk_fma_f16", k_pkfmah},
        {"v_pk_min_u16", k_pkminu16}, {"v_pk_lshlrev_b16", k_pklshl16}, {"v_mbcnt_lo_u32_b32", k_mbcnt}, {"v_ffbl_b32", k_ffbl},
        {"v_cmp_lt_f32 vcc", k_cmpf}, {"v_med3_i32", k_med3}, {"v_cvt_pk_u8_f32", k_cvtpku8}, {"v_cvt_f32_ubyte0", k_cvtub0}, {"v_sad_u16", k_sadu16},
        {"v_lshl_add_u64", k_lshladd64}, {"v_lshlrev_b64", k_lshl64}, {"v_pk_mul_f32", k_pkmulf},
        {"v_pk_add_f32", k_pkaddf}, {"v_pk_fma_f32", k_pkfma}, {"v_mul_f64", k_mulf64}, {"v_add_f64", k_addf64}, {"v_fma_f64", k_fmaf64},
    };
    // W waves per SIMD: workgroups of 256 threads (one wave per SIMD each), W workgroups per CU (dynamic LDS caps the residency:
    // 160 KB / W each), grid = CUs x W so that everything is resident at once
    const int Ws[] = {1, 2, 4, 8};
    printf("\n%-26s", "SIMD cycles / instruction");
    for (int w : Ws) printf("  %d wave%s/SIMD", w, w > 1 ? "s" : " ");
    printf("   (at the clock of the v_add_u32 run with the same residency, see the last line)\n");
    // reference clock per residency: assume nothing -- print the raw ns per instruction per SIMD as well
    std::vector<double> ns_add;
    for (const Row &r : rows) {
        printf("%-26s", r.name);
        for (size_t wi = 0; wi < sizeof(Ws) / sizeof(Ws[0]); wi++) {
            const int W = Ws[wi];
            const size_t lds = W == 8 ? 0 : (size_t)(160 * 1024 / W) - 2048;
            const double ms = time_ms(r.fn, cus * W, 256, out, iters, lds);
            const double ns_per_inst = ms * 1e6 / ((double)W * 32.0 * iters);   // per SIMD: W waves x 32 x iters instructions
            if (&r == &rows[0]) ns_add.push_back(ns_per_inst);
            printf("  %7.3f ns   ", ns_per_inst);
        }
        printf("\n");
    }
    printf("\n(ns per wave-instruction per SIMD; at 2.4 GHz one cycle is 0.417 ns)\n");
    return 0;
}
