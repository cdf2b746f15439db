// wave_emu.h - a CPU stand-in for the handful of HIP constructs a wave-cooperative kernel is written in, so that the TEXT of a kernel (a header under
// libsafecrypto_amd/csrc/) can be compiled with g++ and run lane by lane: one host thread per lane, 64 lanes a wavefront, every cross-lane operation
// (shuffle, ballot, readlane, the wavefront's LDS fence) a barrier over the wavefront's threads, __syncthreads a barrier over the workgroup's.  TEST
// INFRASTRUCTURE ONLY (tests/test_emu_*.py): it exists because kernels sometimes have to be written while no GPU can be reached, and because the GPU
// pool offers no AddressSanitizer - under g++ -fsanitize=address,undefined (and thread) the same text gets both.  It proves the kernel's LOGIC (indexing,
// layouts, arithmetic, fences between lanes of a wavefront: a missing fence is a data race here), not its code generation.  Cross-lane operations must be
// reached by all lanes of a wavefront together, as on the device.
#pragma once
#include <pthread.h>
#include <stdint.h>
#include <string.h>
#include <algorithm>
#include <functional>
#include <thread>
#include <type_traits>
#include <vector>

#define __global__
#define __device__
#define __host__
#define __forceinline__ inline
#define __launch_bounds__(...)
#define __shared__

struct EmuDim3 { unsigned x = 0, y = 0, z = 0; };
static thread_local EmuDim3 threadIdx;
static EmuDim3 blockIdx, blockDim, gridDim;            // workgroups run one after the other

struct alignas(16) uint4 { uint32_t x, y, z, w; };
static inline uint4 make_uint4(uint32_t x, uint32_t y, uint32_t z, uint32_t w) { return uint4{ x, y, z, w }; }

namespace emu {
struct Wave { pthread_barrier_t bar; uint64_t buf[64]; };
struct Block { pthread_barrier_t bar; std::vector<Wave> waves; };
static Block *cur_block;
static thread_local Wave *my_wave;
static thread_local int my_lane;

static inline void wave_sync() { pthread_barrier_wait(&my_wave->bar); }
static inline uint64_t exchange(uint64_t v, int src)
{
    my_wave->buf[my_lane] = v;
    wave_sync();
    const uint64_t r = my_wave->buf[src & 63];
    wave_sync();
    return r;
}

// run `body` as grid x block threads (block a multiple of 64)
static inline void launch(unsigned grid, unsigned block, const std::function<void()> &body)
{
    gridDim.x = grid; blockDim.x = block;
    for (unsigned b = 0; b < grid; b++) {
        Block blk;
        blk.waves.resize(block / 64);
        pthread_barrier_init(&blk.bar, nullptr, block);
        for (auto &w : blk.waves) pthread_barrier_init(&w.bar, nullptr, 64);
        cur_block = &blk; blockIdx.x = b;
        std::vector<std::thread> th;
        for (unsigned t = 0; t < block; t++)
            th.emplace_back([&, t]() { threadIdx.x = t; my_lane = (int)(t & 63); my_wave = &blk.waves[t >> 6]; body(); });
        for (auto &x : th) x.join();
        for (auto &w : blk.waves) pthread_barrier_destroy(&w.bar);
        pthread_barrier_destroy(&blk.bar);
    }
}
}  // namespace emu

static inline int __shfl_up(int v, int o) { return (int)(uint32_t)emu::exchange((uint32_t)v, emu::my_lane >= o ? emu::my_lane - o : emu::my_lane); }
static inline int __shfl_xor(int v, int m) { return (int)(uint32_t)emu::exchange((uint32_t)v, emu::my_lane ^ m); }
static inline int __shfl(int v, int src) { return (int)(uint32_t)emu::exchange((uint32_t)v, src); }
static inline int emu_readlane(int v, int l) { return (int)(uint32_t)emu::exchange((uint32_t)v, l); }
#define __builtin_amdgcn_readlane(v, l) emu_readlane((v), (l))
static inline uint64_t __ballot(bool p)
{
    emu::my_wave->buf[emu::my_lane] = p ? 1 : 0;
    emu::wave_sync();
    uint64_t r = 0;
    for (int i = 0; i < 64; i++) r |= emu::my_wave->buf[i] << i;
    emu::wave_sync();
    return r;
}
static inline void wave_lds_fence() { emu::wave_sync(); }               // csrc/ntt_core.h: a wavefront-scope fence + wave barrier
static inline void __syncthreads() { pthread_barrier_wait(&emu::cur_block->bar); }
static inline uint32_t atomicAdd(uint32_t *p, uint32_t v) { return __atomic_fetch_add(p, v, __ATOMIC_RELAXED); }
static inline uint32_t atomicOr(uint32_t *p, uint32_t v) { return __atomic_fetch_or(p, v, __ATOMIC_RELAXED); }
static inline uint32_t __umulhi(uint32_t a, uint32_t b) { return (uint32_t)(((uint64_t)a * b) >> 32); }
