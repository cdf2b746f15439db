// experiment: what do the NTT kernel's global access patterns cost, with no arithmetic?
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <vector>
#define CHK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("%s: %s\n",#x,hipGetErrorString(e)); return 1;}}while(0)
__host__ __device__ constexpr int brv4(int r) { return ((r & 1) << 3) | ((r & 2) << 1) | ((r & 4) >> 1) | ((r & 8) >> 3); }

// K1: the kernel's pattern (n=256): lane (pw,lp) loads in[base + r*16 + lp], stores out[base + brv4(r)*16 + brv4(lp)]
__global__ __launch_bounds__(256) void k_pattern(const int32_t* __restrict__ in, int32_t* __restrict__ out, size_t batch) {
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, pw = lane >> 4, lp = lane & 15, blp = brv4(lp);
    const size_t groups = batch / 4;
    for (size_t g = (size_t)blockIdx.x * 4 + wid; g < groups; g += (size_t)gridDim.x * 4) {
        const size_t base = (g * 4 + pw) * 256;
        int32_t x[16];
#pragma unroll
        for (int r = 0; r < 16; r++) x[r] = in[base + r * 16 + lp];
#pragma unroll
        for (int r = 0; r < 16; r++) out[base + brv4(r) * 16 + blp] = x[r] + 1;
    }
}
// K3: same data per wave (4 KB) but 16 B per lane, fully contiguous per instruction
__global__ __launch_bounds__(256) void k_vec4(const int32_t* __restrict__ in, int32_t* __restrict__ out, size_t batch) {
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const size_t groups = batch / 4;
    for (size_t g = (size_t)blockIdx.x * 4 + wid; g < groups; g += (size_t)gridDim.x * 4) {
        const int4* src = reinterpret_cast<const int4*>(in + g * 1024);
        int4* dst = reinterpret_cast<int4*>(out + g * 1024);
        int4 v[4];
#pragma unroll
        for (int c = 0; c < 4; c++) v[c] = src[c * 64 + lane];
#pragma unroll
        for (int c = 0; c < 4; c++) { v[c].x += 1; dst[c * 64 + lane] = v[c]; }
    }
}
// K4: pattern loads, vec4 stores ; K5: vec4 loads, pattern stores
__global__ __launch_bounds__(256) void k_pl_vs(const int32_t* __restrict__ in, int32_t* __restrict__ out, size_t batch) {
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, pw = lane >> 4, lp = lane & 15;
    const size_t groups = batch / 4;
    for (size_t g = (size_t)blockIdx.x * 4 + wid; g < groups; g += (size_t)gridDim.x * 4) {
        const size_t base = (g * 4 + pw) * 256;
        int32_t x[16];
#pragma unroll
        for (int r = 0; r < 16; r++) x[r] = in[base + r * 16 + lp];
        int4* dst = reinterpret_cast<int4*>(out + g * 1024);
#pragma unroll
        for (int c = 0; c < 4; c++) dst[c * 64 + lane] = make_int4(x[4*c], x[4*c+1], x[4*c+2], x[4*c+3] + 1);
    }
}
__global__ __launch_bounds__(256) void k_vl_ps(const int32_t* __restrict__ in, int32_t* __restrict__ out, size_t batch) {
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, pw = lane >> 4, lp = lane & 15, blp = brv4(lp);
    const size_t groups = batch / 4;
    for (size_t g = (size_t)blockIdx.x * 4 + wid; g < groups; g += (size_t)gridDim.x * 4) {
        const size_t base = (g * 4 + pw) * 256;
        const int4* src = reinterpret_cast<const int4*>(in + g * 1024);
        int4 v[4];
#pragma unroll
        for (int c = 0; c < 4; c++) v[c] = src[c * 64 + lane];
        int32_t x[16] = {v[0].x,v[0].y,v[0].z,v[0].w,v[1].x,v[1].y,v[1].z,v[1].w,v[2].x,v[2].y,v[2].z,v[2].w,v[3].x,v[3].y,v[3].z,v[3].w};
#pragma unroll
        for (int r = 0; r < 16; r++) out[base + brv4(r) * 16 + blp] = x[r] + 1;
    }
}
template <typename K> float timeit(K kern, int blocks, const int32_t* in, int32_t* out, size_t batch) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 3; i++) hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, in, out, batch);
    hipEventRecord(e0, 0);
    for (int i = 0; i < 20; i++) hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, in, out, batch);
    hipEventRecord(e1, 0); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); return ms / 20;
}
int main() {
    for (size_t batch : {(size_t)65536, (size_t)180224, (size_t)524288}) {
        int32_t *in, *out; size_t bytes = batch * 1024;
        CHK(hipMalloc(&in, bytes)); CHK(hipMalloc(&out, bytes)); CHK(hipMemset(in, 1, bytes));
        for (int blocks : {1024, 2048, 4096, (int)(batch / 16)}) {
            float a = timeit(k_pattern, blocks, in, out, batch), b = timeit(k_vec4, blocks, in, out, batch);
            float c = timeit(k_pl_vs, blocks, in, out, batch), d = timeit(k_vl_ps, blocks, in, out, batch);
            printf("batch %7zu blocks %6d  pattern %7.0f  vec4 %7.0f  patL+vecS %7.0f  vecL+patS %7.0f GB/s\n", batch, blocks,
                   2 * bytes / a / 1e6, 2 * bytes / b / 1e6, 2 * bytes / c / 1e6, 2 * bytes / d / 1e6);
        }
        hipFree(in); hipFree(out);
    }
    return 0;
}
