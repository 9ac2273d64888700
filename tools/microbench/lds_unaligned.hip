// Does gfx950 serve misaligned ds_read_b32 / ds_read_b64 / ds_read_u16 (byte-granular LDS addresses), and at what cost?
// Build: hipcc -O3 --offload-arch=gfx950 lds_unaligned.hip -o lds_unaligned ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

__global__ void k_check(uint32_t *out)
{
    __shared__ __align__(16) uint8_t lds[4096];
    for (int i = threadIdx.x; i < 4096; i += blockDim.x) lds[i] = (uint8_t)(i * 7 + 3);
    __syncthreads();
    const uint32_t base = (uint32_t)(uintptr_t)lds;   // LDS byte address
    const uint32_t a = base + 64 + threadIdx.x * 13;   // all alignments
    uint32_t r32, r16;
    uint64_t r64;
    asm volatile("ds_read_b32 %0, %1\n s_waitcnt lgkmcnt(0)" : "=v"(r32) : "v"(a));
    asm volatile("ds_read_u16 %0, %1\n s_waitcnt lgkmcnt(0)" : "=v"(r16) : "v"(a));
    asm volatile("ds_read_b64 %0, %1\n s_waitcnt lgkmcnt(0)" : "=v"(r64) : "v"(a));
    const int o = 64 + threadIdx.x * 13;
    uint32_t e32 = 0, e16 = 0;
    uint64_t e64 = 0;
    for (int k = 0; k < 4; k++) e32 |= (uint32_t)lds[o + k] << (8 * k);
    for (int k = 0; k < 2; k++) e16 |= (uint32_t)lds[o + k] << (8 * k);
    for (int k = 0; k < 8; k++) e64 |= (uint64_t)lds[o + k] << (8 * k);
    out[threadIdx.x * 4 + 0] = (r32 == e32);
    out[threadIdx.x * 4 + 1] = (r16 == e16);
    out[threadIdx.x * 4 + 2] = (r64 == e64);
    out[threadIdx.x * 4 + 3] = o & 7;
}

template <int MODE>
__global__ void k_rate(uint32_t *out, int iters, int misalign)
{
    __shared__ __align__(16) uint8_t lds[16384];
    for (int i = threadIdx.x; i < 16384; i += blockDim.x) lds[i] = (uint8_t)i;
    __syncthreads();
    const uint32_t base = (uint32_t)(uintptr_t)lds;
    uint32_t a = base + ((threadIdx.x * 8) & 8191) + misalign;
    uint32_t acc = 0;
    for (int it = 0; it < iters; it++) {
        uint32_t r0, r1;
        uint64_t q;
        if (MODE == 0) {
            asm volatile("ds_read_u8 %0, %2\n ds_read_u8 %1, %2 offset:1\n s_waitcnt lgkmcnt(0)" : "=v"(r0), "=v"(r1) : "v"(a));
            acc += r0 + r1;
        } else if (MODE == 1) {
            asm volatile("ds_read_b32 %0, %1\n s_waitcnt lgkmcnt(0)" : "=v"(r0) : "v"(a));
            acc += r0;
        } else {
            asm volatile("ds_read_b64 %0, %1\n s_waitcnt lgkmcnt(0)" : "=v"(q) : "v"(a));
            acc += (uint32_t)q + (uint32_t)(q >> 32);
        }
        a = base + ((a - base + 176) & 8191);
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}

int main()
{
    uint32_t *d;
    hipMalloc(&d, 1 << 22);
    k_check<<<1, 256>>>(d);
    std::vector<uint32_t> h(1024);
    hipMemcpy(h.data(), d, 4096, hipMemcpyDeviceToHost);
    int ok32[8] = {0}, ok16[8] = {0}, ok64[8] = {0}, n[8] = {0};
    for (int t = 0; t < 256; t++) {
        const int al = h[t * 4 + 3];
        n[al]++;
        ok32[al] += h[t * 4];
        ok16[al] += h[t * 4 + 1];
        ok64[al] += h[t * 4 + 2];
    }
    for (int al = 0; al < 8; al++) printf("addr%%8=%d: b32 %d/%d  u16 %d/%d  b64 %d/%d\n", al, ok32[al], n[al], ok16[al], n[al], ok64[al], n[al]);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (int mode = 0; mode < 3; mode++)
        for (int mis = 0; mis < 4; mis++) {
            float ms = 0;
            for (int rep = 0; rep < 2; rep++) {
                hipEventRecord(e0);
                if (mode == 0) k_rate<0><<<2048, 256>>>(d, 2000, mis);
                if (mode == 1) k_rate<1><<<2048, 256>>>(d, 2000, mis);
                if (mode == 2) k_rate<2><<<2048, 256>>>(d, 2000, mis);
                hipEventRecord(e1);
                hipEventSynchronize(e1);
                hipEventElapsedTime(&ms, e0, e1);
            }
            printf("mode %s misalign %d: %.3f ms\n", mode == 0 ? "2 x u8" : mode == 1 ? "b32" : "b64", mis, ms);
        }
    return 0;
}
