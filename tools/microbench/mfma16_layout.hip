// Operand / result lane maps of the two 16x16 MFMA shapes the fused describe kernel uses (k_describe.hip, r06), checked with exact
// integer data: v_mfma_i32_16x16x64_i8 and v_mfma_f32_16x16x32_f16.  Also: does global_load_lds_dwordx4 take a byte-unaligned
// global address?  Build: hipcc -O2 --offload-arch=gfx950 mfma16_layout.hip -o mfma16_layout ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <vector>

typedef int v4i __attribute__((ext_vector_type(4)));
typedef float v4f __attribute__((ext_vector_type(4)));
typedef _Float16 v8h __attribute__((ext_vector_type(8)));

__global__ void k_i8(const int8_t *A, const int8_t *B, int32_t *D)   // A [16][64], B [64][16] row-major, D [16][16]
{
    const int l = threadIdx.x, r = l & 15, q = l >> 4;
    int8_t a[16], b[16];
    for (int j = 0; j < 16; j++) {
        a[j] = A[r * 64 + 16 * q + j];       // hypothesis: lane holds A[row l&15][k = 16 (l>>4) + j]
        b[j] = B[(16 * q + j) * 16 + r];     //             B[k = 16 (l>>4) + j][col l&15]
    }
    v4i av, bv, c = {0, 0, 0, 0};
    __builtin_memcpy(&av, a, 16);
    __builtin_memcpy(&bv, b, 16);
    c = __builtin_amdgcn_mfma_i32_16x16x64_i8(av, bv, c, 0, 0, 0);
    for (int i = 0; i < 4; i++) D[(4 * q + i) * 16 + r] = c[i];   // hypothesis: col = l&15, row = 4 (l>>4) + reg
}

__global__ void k_f16(const float *A, const float *B, float *D)   // A [16][32], B [32][16], small integers
{
    const int l = threadIdx.x, r = l & 15, q = l >> 4;
    v8h av, bv;
    for (int j = 0; j < 8; j++) {
        av[j] = (_Float16)A[r * 32 + 8 * q + j];
        bv[j] = (_Float16)B[(8 * q + j) * 16 + r];
    }
    v4f c = {0, 0, 0, 0};
    c = __builtin_amdgcn_mfma_f32_16x16x32_f16(av, bv, c, 0, 0, 0);
    for (int i = 0; i < 4; i++) D[(4 * q + i) * 16 + r] = c[i];
}

__global__ void k_dma(const uint8_t *src, int misalign, uint8_t *out)
{
    __shared__ __align__(16) uint8_t lds[1024];
    const int l = threadIdx.x;
    const uint8_t *p = src + misalign + 16 * l;
    uint32_t keep;
    const uint32_t base = (uint32_t)(uintptr_t)lds;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0\n\ts_waitcnt vmcnt(0)"
                 : "=&s"(keep) : "v"(p), "s"(base) : "memory");
    __syncthreads();
    for (int j = 0; j < 16; j++) out[16 * l + j] = lds[16 * l + j];
}

int main()
{
    std::vector<int8_t> A(16 * 64), B(64 * 16);
    for (size_t i = 0; i < A.size(); i++) A[i] = (int8_t)((i * 37 + 11) % 255 - 127);
    for (size_t i = 0; i < B.size(); i++) B[i] = (int8_t)((i * 91 + 5) % 251 - 125);
    int8_t *dA, *dB; int32_t *dD;
    hipMalloc(&dA, A.size()); hipMalloc(&dB, B.size()); hipMalloc(&dD, 256 * 4);
    hipMemcpy(dA, A.data(), A.size(), hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), B.size(), hipMemcpyHostToDevice);
    k_i8<<<1, 64>>>(dA, dB, dD);
    std::vector<int32_t> D(256);
    hipMemcpy(D.data(), dD, 1024, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int m = 0; m < 16; m++) for (int n = 0; n < 16; n++) { int s = 0; for (int k = 0; k < 64; k++) s += A[m * 64 + k] * B[k * 16 + n]; bad += s != D[m * 16 + n]; }
    printf("mfma_i32_16x16x64_i8 lane maps (A[l&15][16(l>>4)+j], B[16(l>>4)+j][l&15], D col l&15 row 4(l>>4)+reg): %s (%d wrong)\n", bad ? "WRONG" : "ok", bad);
    std::vector<float> fA(16 * 32), fB(32 * 16), fD(256);
    for (size_t i = 0; i < fA.size(); i++) fA[i] = (float)((int)(i * 13 + 3) % 31 - 15);
    for (size_t i = 0; i < fB.size(); i++) fB[i] = (float)((int)(i * 29 + 7) % 23 - 11);
    float *gA, *gB, *gD;
    hipMalloc(&gA, fA.size() * 4); hipMalloc(&gB, fB.size() * 4); hipMalloc(&gD, 1024);
    hipMemcpy(gA, fA.data(), fA.size() * 4, hipMemcpyHostToDevice); hipMemcpy(gB, fB.data(), fB.size() * 4, hipMemcpyHostToDevice);
    k_f16<<<1, 64>>>(gA, gB, gD);
    hipMemcpy(fD.data(), gD, 1024, hipMemcpyDeviceToHost);
    bad = 0;
    for (int m = 0; m < 16; m++) for (int n = 0; n < 16; n++) { float s = 0; for (int k = 0; k < 32; k++) s += fA[m * 32 + k] * fB[k * 16 + n]; bad += s != fD[m * 16 + n]; }
    printf("mfma_f32_16x16x32_f16 lane maps (A[l&15][8(l>>4)+j], B[8(l>>4)+j][l&15]): %s (%d wrong)\n", bad ? "WRONG" : "ok", bad);
    std::vector<uint8_t> S(4096), O(1024);
    for (size_t i = 0; i < S.size(); i++) S[i] = (uint8_t)(i * 7 + 1);
    uint8_t *dS, *dO;
    hipMalloc(&dS, 4096); hipMalloc(&dO, 1024);
    hipMemcpy(dS, S.data(), 4096, hipMemcpyHostToDevice);
    for (int mis : {0, 4, 1, 7, 13}) {
        hipMemset(dO, 0, 1024);
        k_dma<<<1, 64>>>(dS, mis, dO);
        hipError_t e = hipDeviceSynchronize();
        hipMemcpy(O.data(), dO, 1024, hipMemcpyDeviceToHost);
        bad = 0;
        for (int i = 0; i < 1024; i++) bad += O[i] != S[mis + i];
        printf("global_load_lds_dwordx4, source misaligned by %2d bytes: %s (%d wrong bytes, %s)\n", mis, bad ? "WRONG" : "ok", bad, hipGetErrorString(e));
    }
    return 0;
}
