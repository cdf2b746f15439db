// experiment: issue rate of the bitwise instructions Keccak is made of (gfx950), at 1 / 2 / 4 / 8 wavefronts per SIMD
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
template <int OP> __global__ __launch_bounds__(64) void k(uint32_t *out, uint32_t seed, int iters)
{
    uint32_t a[8];
    for (int i = 0; i < 8; i++) a[i] = seed * (threadIdx.x + 1) + i * 977;
    const uint32_t c = seed | 1, d = seed * 77u;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int i = 0; i < 8; i++) {
            if (OP == 0) a[i] = a[i] + c;
            if (OP == 1) a[i] = a[i] ^ c;
            if (OP == 2) a[i] = __builtin_amdgcn_bitop3_b32(a[i], a[(i + 1) & 7], c, 0x96);
            if (OP == 3) a[i] = __builtin_amdgcn_bitop3_b32(a[i], a[(i + 1) & 7], a[(i + 3) & 7], 0xD2);
            if (OP == 4) a[i] = __builtin_amdgcn_alignbit(a[i], a[(i + 1) & 7], 13);
            if (OP == 5) a[i] = (a[i] & c) | d;            // v_and_or_b32
            if (OP == 6) { float f = __uint_as_float(a[i]); f = f * 1.0001f + 0.5f; a[i] = __float_as_uint(f); }   // v_fma_f32
        }
    }
    uint32_t s = 0;
    for (int i = 0; i < 8; i++) s ^= a[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int OP> void run(const char *name, uint32_t *d)
{
    for (int wps = 1; wps <= 8; wps *= 2) {
        const int blocks = 1024 * wps, iters = 8192;
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(64), 0, 0, d, 12345u, 16);
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(64), 0, 0, d, 12345u, iters);
        hipEventRecord(e1, 0); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const double winstr = (double)blocks * iters * 8;
        printf("%-22s %d waves/SIMD %8.3f ms  %7.1f G wave-instr/s  (%.2f cycles per instruction per SIMD at 2.4 GHz)\n", name, wps, ms, winstr / ms / 1e6, 1024 * 2.4e9 / (winstr / (ms * 1e-3)));
    }
}
int main()
{
    uint32_t *d; hipMalloc(&d, 8192 * 64 * 4);
    run<0>("v_add_u32", d); run<1>("v_xor_b32", d); run<2>("v_bitop3 (xor3)", d); run<3>("v_bitop3 (chi)", d); run<4>("v_alignbit_b32", d); run<5>("v_and_or_b32", d); run<6>("v_fma_f32", d);
    return 0;
}
