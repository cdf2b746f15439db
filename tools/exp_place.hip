// experiment: does a latency-bound kernel of ~1024 wavefronts get one wavefront per SIMD?  Same work per wavefront (a serial chain of dependent xor3),
// different workgroup shapes and LDS footprints; time = the slowest SIMD's queue.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
__global__ void k(uint32_t *out, uint32_t seed, int iters, uint32_t *where)
{
    extern __shared__ uint32_t lds[];
    uint32_t a[8];
    for (int i = 0; i < 8; i++) a[i] = seed * (threadIdx.x + 1) + i * 977;
    const uint32_t c = seed | 1;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int i = 0; i < 8; i++) a[i] = __builtin_amdgcn_bitop3_b32(a[i], a[(i + 1) & 7], c, 0x96);
    }
    uint32_t s = 0;
    for (int i = 0; i < 8; i++) s ^= a[i];
    if (threadIdx.x == 0 && lds) lds[0] = s;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (where && (threadIdx.x & 63) == 0) {
        uint32_t hwid;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
        uint32_t xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        where[(blockIdx.x * blockDim.x + threadIdx.x) >> 6] = hwid | (xcc << 28);
    }
}
int main()
{
    uint32_t *d, *w; hipMalloc(&d, 8192 * 64 * 4); hipMalloc(&w, 8192 * 4);
    struct { int blocks, threads; size_t lds; } cfg[] = { {1024, 64, 0}, {512, 128, 0}, {256, 256, 0}, {256, 256, 100 * 1024}, {1024, 64, 36 * 1024}, {1024, 64, 40 * 1024}, {512, 64, 0}, {256, 64, 0},
                                                        {128, 256, 0}, {128, 256, 100 * 1024}, {2048, 64, 0}, {512, 256, 0}, {512, 256, 70 * 1024} };
    for (auto &c : cfg) {
        hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        const int iters = 8192;
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipLaunchKernelGGL(k, dim3(c.blocks), dim3(c.threads), c.lds, 0, d, 12345u, 16, w);
        hipDeviceSynchronize();
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL(k, dim3(c.blocks), dim3(c.threads), c.lds, 0, d, 12345u, iters, (uint32_t *)nullptr);
        hipEventRecord(e1, 0); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        // occupancy histogram from the first launch: waves per (xcc, se, cu, simd)
        const int nw = c.blocks * c.threads / 64;
        static uint32_t h[8192]; hipMemcpy(h, w, nw * 4, hipMemcpyDeviceToHost);
        static int cnt[1 << 16]; for (auto &x : cnt) x = 0;
        for (int i = 0; i < nw; i++) {
            const uint32_t v = h[i];
            const uint32_t simd = (v >> 4) & 3, cu = (v >> 8) & 15, sh = (v >> 12) & 1, se = (v >> 13) & 7, xcc = v >> 28;
            cnt[(xcc << 10) | (se << 7) | (sh << 6) | (cu << 2) | simd]++;
        }
        int hist[9] = {0}, used = 0; for (auto x : cnt) if (x) { hist[x > 8 ? 8 : x]++; used++; }
        printf("%5d blocks x %3d threads, %6zu B LDS: %7.3f ms = %5.2f cycles per instruction of a wavefront;  SIMDs used %4d, with 1/2/3/4+ wavefronts: %d %d %d %d\n",
               c.blocks, c.threads, c.lds, ms, ms * 1e-3 * 2.4e9 / ((double)iters * 8), used, hist[1], hist[2], hist[3], hist[4] + hist[5] + hist[6] + hist[7] + hist[8]);
    }
    return 0;
}
