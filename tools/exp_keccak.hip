// measurement: issue rate of the one-lane Keccak permutation (keccak_dev.h) against wavefronts per SIMD; registers only, no LDS.
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -o tools/exp_keccak tools/exp_keccak.hip ; run on the GPU box
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include "../libsafecrypto_amd/csrc/keccak_dev.h"

template <bool STD>
__global__ __launch_bounds__(256) void perm_kernel(uint64_t *io, int perms)
{
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    uint64_t s[25];
    for (int i = 0; i < 25; i++) s[i] = io[t] + (uint64_t)i * 0x9E3779B97F4A7C15ull;
    for (int p = 0; p < perms; p++) sc::keccak_permute_regs<STD>(s);
    uint64_t x = 0;
    for (int i = 0; i < 25; i++) x ^= s[i];
    io[t] = x;
}

// the same count of permutations through the sponge object (state in LDS between permutations, word-by-word absorption of register data)
template <int NT>
__global__ __launch_bounds__(NT) void sponge_kernel(uint64_t *io, int perms, int rate)
{
    __shared__ uint64_t st[25 * NT];
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    sc::Sponge<NT> sp; sp.init(st, threadIdx.x, rate);
    uint64_t v = io[t];
    for (int p = 0; p < perms; p++)
        for (int w = 0; w < rate / 8; w++) { sp.absorb_u64(v); v = v * 6364136223846793005ull + 1442695040888963407ull; }
    sp.finalize();
    uint64_t x = 0;
    for (int i = 0; i < 8; i++) x = (x << 8) | sp.squeeze_byte();
    io[t] = x;
}

int main()
{
    for (int nt : { 64, 256 }) for (int wps : { 1, 2, 4 }) {
        const int perms = 64; const size_t lanes = (size_t)1024 * 64 * wps;
        uint64_t *d; hipMalloc(&d, lanes * 8); hipMemset(d, 1, lanes * 8);
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        float best = 1e9f;
        for (int r = 0; r < 4; r++) {
            hipEventRecord(e0);
            if (nt == 64) sponge_kernel<64><<<lanes / 64, 64>>>(d, perms, 72); else sponge_kernel<256><<<lanes / 256, 256>>>(d, perms, 72);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1); if (r && ms < best) best = ms;
        }
        printf("sponge, %3d-thread workgroups, %d waves/SIMD: %8.3f ms  %7.1f G perm/s  (%5.2f us per permutation of a wavefront)\n", nt, wps, best, (double)lanes * (perms + 1) / best / 1e6, best * 1e3 / (perms + 1));
        hipFree(d);
    }

    const int perms = 64;
    for (int wps : { 1, 2, 3, 4, 6, 8 }) {
        const size_t lanes = (size_t)1024 * 64 * wps;
        uint64_t *d; hipMalloc(&d, lanes * 8); hipMemset(d, 1, lanes * 8);
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        perm_kernel<false><<<lanes / 256, 256>>>(d, perms);
        hipDeviceSynchronize();
        float best = 1e9f;
        for (int r = 0; r < 5; r++) {
            hipEventRecord(e0); perm_kernel<false><<<lanes / 256, 256>>>(d, perms); hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
        }
        const double nperm = (double)lanes * perms, winstr = nperm / 64 * 24 * 190;
        printf("%d waves/SIMD: %8.3f ms  %7.1f G perm/s  ~%6.1f G wave-instr/s (at 190 per round)  %5.2f cycles per instruction per SIMD\n", wps, best, nperm / best / 1e6,
               winstr / best / 1e6, 1024.0 * 2.4e9 * best * 1e-3 / winstr);
        hipFree(d);
    }
    return 0;
}
