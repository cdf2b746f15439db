// experiment: issue rate of the integer instructions the modular arithmetic is built from (gfx950)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
extern "C" __device__ uint32_t llvm_mul_u24(uint32_t, uint32_t) __asm("llvm.amdgcn.mul.u24.i32");
extern "C" __device__ uint32_t llvm_mulhi_u24(uint32_t, uint32_t) __asm("llvm.amdgcn.mulhi.u24");
template <int OP> __global__ __launch_bounds__(256) void k(uint32_t *out, uint32_t seed, int iters)
{
    uint32_t a[8];
    for (int i = 0; i < 8; i++) a[i] = seed * (threadIdx.x + 1) + i * 977;
    const uint32_t c = seed | 1;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int i = 0; i < 8; i++) {
            if (OP == 0) a[i] = a[i] + c;
            if (OP == 1) a[i] = a[i] * c;                         // v_mul_lo_u32
            if (OP == 2) a[i] = __umulhi(a[i], c);                // v_mul_hi_u32
            if (OP == 3) a[i] = llvm_mul_u24(a[i], c);
            if (OP == 4) a[i] = llvm_mulhi_u24(a[i], c) + a[i];
            if (OP == 5) { uint64_t p = (uint64_t)a[i] * c + a[i]; a[i] = (uint32_t)p ^ (uint32_t)(p >> 32); }   // v_mad_u64_u32
            if (OP == 6) a[i] = min(a[i], a[i] - c);
            if (OP == 7) a[i] = __builtin_amdgcn_alignbit(a[i], c, 24);
            if (OP == 8) a[i] = (uint32_t)((float)a[i] * 1.0001f);   // cvt + mul + cvt
        }
    }
    uint32_t s = 0;
    for (int i = 0; i < 8; i++) s ^= a[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int OP> void run(const char *name, uint32_t *d)
{
    const int blocks = 256 * 8, iters = 4096;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, d, 12345u, 16);
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, d, 12345u, iters);
    hipEventRecord(e1, 0); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double ops = (double)blocks * 256 * iters * 8;
    printf("%-28s %8.3f ms  %7.2f T lane-ops/s\n", name, ms, ops / ms / 1e9);
}
int main()
{
    uint32_t *d; hipMalloc(&d, 256 * 8 * 256 * 4);
    run<0>("v_add_u32", d); run<1>("v_mul_lo_u32", d); run<2>("v_mul_hi_u32", d); run<3>("v_mul_u32_u24", d);
    run<4>("v_mul_hi_u32_u24 (+add)", d); run<5>("v_mad_u64_u32 (+xor)", d); run<6>("sub+min", d); run<7>("v_alignbit", d); run<8>("cvt+fmul+cvt", d);
    return 0;
}
