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
#include <stdlib.h>
#include <stdio.h>
#include <math.h>
#include <string.h>
#include <algorithm>
#include <condition_variable>
#include <deque>
#include <mutex>
#include <functional>
#include <thread>
#include <type_traits>
#include <vector>

#define __global__
#define __device__
#define __host__
#define __forceinline__ inline
#define __launch_bounds__(...)
// EMU_SHARED: what `__shared__` becomes.  Empty (the default) suits a kernel whose LDS is one `extern __shared__` array the harness defines; `static` suits
// kernels that declare their LDS arrays inside the function (one instance per instantiation, shared by the workgroup's threads; workgroups run one after the other)
#ifndef EMU_SHARED
#define EMU_SHARED
#endif
#define __shared__ EMU_SHARED
#define __restrict__

struct EmuDim3 { unsigned x = 0, y = 0, z = 0; };
// (inline variables: ONE copy however many translation units of a library include this header - kernels' inline helpers are merged across them by the linker)
inline thread_local EmuDim3 threadIdx;
inline EmuDim3 blockIdx, blockDim, gridDim;            // workgroups run one after the other

struct alignas(16) uint4 { uint32_t x, y, z, w; };
static inline uint4 make_uint4(uint32_t x, uint32_t y, uint32_t z, uint32_t w) { return uint4{ x, y, z, w }; }
struct alignas(16) int4 { int32_t x, y, z, w; };
static inline int4 make_int4(int32_t x, int32_t y, int32_t z, int32_t w) { return int4{ x, y, z, w }; }
struct alignas(8) uint2 { uint32_t x, y; };
static inline uint2 make_uint2(uint32_t x, uint32_t y) { return uint2{ x, y }; }
struct alignas(8) int2 { int32_t x, y; };
static inline int2 make_int2(int32_t x, int32_t y) { return int2{ x, y }; }
struct alignas(16) double2 { double x, y; };
static inline double2 make_double2(double x, double y) { return double2{ x, y }; }
struct alignas(16) ulonglong2 { unsigned long long x, y; };
struct alignas(16) float4 { float x, y, z, w; };
static inline float4 make_float4(float x, float y, float z, float w) { return float4{ x, y, z, w }; }

namespace emu {
// A barrier out of a mutex and a condition variable rather than pthread_barrier_t: ThreadSanitizer models the former exactly, while behind a
// pthread_barrier_wait it no longer reported a race that followed (found by removing a wavefront fence from csrc/ntt_core.h on purpose and watching the
// result go wrong without a report; tests/test_emu_ntt.py keeps that experiment as a test of the harness).
struct Barrier {
    std::mutex m; std::condition_variable cv; unsigned n = 0, count = 0, gen = 0;
    void init(unsigned threads) { n = threads; count = 0; gen = 0; }
    void wait()
    {
        std::unique_lock<std::mutex> l(m);
        const unsigned g = gen;
        if (++count == n) { count = 0; gen++; cv.notify_all(); }
        else cv.wait(l, [&] { return gen != g; });
    }
    void leave()            // a lane whose kernel body has returned no longer takes part (the device masks it out)
    {
        std::unique_lock<std::mutex> l(m);
        if (n) n--;
        if (n && count == n) { count = 0; gen++; cv.notify_all(); }
    }
};
struct Wave { Barrier bar; Barrier pair[32]; uint64_t buf[64]; uint32_t pbuf[64]; };
struct Block { Barrier bar; std::deque<Wave> waves; int or_acc = 0; };
inline Block *cur_block;
inline thread_local Wave *my_wave;
inline thread_local int my_lane;

static inline void wave_sync() { my_wave->bar.wait(); }
static inline uint64_t exchange(uint64_t v, int src)
{
    my_wave->buf[my_lane] = v;
    wave_sync();
    const uint64_t r = my_wave->buf[src & 63];
    wave_sync();
    return r;
}

// run `body` as grid x block threads (block a multiple of 64).  wave_serial: the wavefronts of a workgroup one after the other - ONLY for kernels without a
// workgroup barrier, whose wavefronts share nothing (the NTT kernel).  ThreadSanitizer needs it: it has 256 thread slots, and with a whole 256-lane
// workgroup alive beside the host's own threads the slots are recycled and races go unreported (found by removing a fence on purpose: tests/test_emu_ntt.py).
inline bool wave_serial = false;
static inline void launch(unsigned grid, unsigned block, const std::function<void()> &body)
{
    gridDim.x = grid; blockDim.x = block; blockDim.y = blockDim.z = 1;
    if (!gridDim.y) gridDim.y = 1;
    if (!gridDim.z) gridDim.z = 1;
    for (unsigned b = 0; b < grid; b++) {
        Block blk;
        blk.waves.resize((block + 63) / 64);
        blk.bar.init(block);
        for (unsigned w = 0; w < blk.waves.size(); w++) {                 // (a last wavefront with fewer than 64 lanes: its barriers count the lanes it has)
            const unsigned lanes = std::min(64u, block - 64 * w);
            blk.waves[w].bar.init(lanes);
            for (unsigned pr = 0; pr < 32; pr++) blk.waves[w].pair[pr].init(2 * pr + 1 < lanes ? 2 : 1);
        }
        cur_block = &blk; blockIdx.x = b;
        for (unsigned t0 = 0; t0 < block; t0 += wave_serial ? 64 : block) {
            std::vector<std::thread> th;
            for (unsigned t = t0; t < (wave_serial ? std::min(t0 + 64, block) : block); t++)
                th.emplace_back([&, t]() { threadIdx.x = t; my_lane = (int)(t & 63); my_wave = &blk.waves[t >> 6]; body(); my_wave->pair[my_lane >> 1].leave(); my_wave->bar.leave(); blk.bar.leave(); });
            for (auto &x : th) x.join();
        }
    }
}
}  // namespace emu

static inline int __shfl_up(int v, int o) { return (int)(uint32_t)emu::exchange((uint32_t)v, emu::my_lane >= o ? emu::my_lane - o : emu::my_lane); }
static inline int __shfl_xor(int v, int m) { return (int)(uint32_t)emu::exchange((uint32_t)v, emu::my_lane ^ m); }
static inline int __shfl(int v, int src) { return (int)(uint32_t)emu::exchange((uint32_t)v, src); }
static inline int emu_readlane(int v, int l) { return (int)(uint32_t)emu::exchange((uint32_t)v, l); }
#define __builtin_amdgcn_readlane(v, l) emu_readlane((v), (l))
// DPP moves.  quad_perm controls whose every lane reads from its own PAIR of lanes (the two-lane Keccak's exchanges: [1,0,3,2], [0,0,2,2], [1,1,3,3]) are a
// barrier over the pair, not the wavefront - pairs run different numbers of permutations (a rejected sample in ExpandMask), as they do under the
// device's exec mask.  Any other quad_perm, row_half_mirror (0x141) and row_mirror (0x140) - the wave reductions' steps - are reached by the whole wavefront.
static inline int emu_update_dpp(int old, int src, int ctrl, int row_mask, int bank_mask, bool bound_ctrl)
{
    if (row_mask != 0xF || bank_mask != 0xF) { fprintf(stderr, "emu: DPP row / bank mask %x %x not emulated\n", row_mask, bank_mask); abort(); }
    const int lane = emu::my_lane, q = lane & 3;
    if (ctrl <= 0xFF) {
        const int sel = (ctrl >> (2 * q)) & 3;
        if ((sel >> 1) == (q >> 1)) {
            bool pair_local = true;
            for (int i = 0; i < 4; i++) pair_local &= (((ctrl >> (2 * i)) & 3) >> 1) == (i >> 1);
            if (pair_local) {
                emu::Barrier &b = emu::my_wave->pair[lane >> 1];
                emu::my_wave->pbuf[lane] = (uint32_t)src;
                b.wait();
                const int r = (int)emu::my_wave->pbuf[(lane & ~3) | sel];
                b.wait();
                return r;
            }
        }
        return (int)(uint32_t)emu::exchange((uint32_t)src, (lane & ~3) | sel);
    }
    if (ctrl == 0x141) return (int)(uint32_t)emu::exchange((uint32_t)src, (lane & ~7) | (7 - (lane & 7)));
    if (ctrl == 0x140) return (int)(uint32_t)emu::exchange((uint32_t)src, (lane & ~15) | (15 - (lane & 15)));
    if (ctrl >= 0x101 && ctrl <= 0x12F && (ctrl & 15)) {                          // row_shl / row_shr / row_ror by n: lane i of a row reads lane i + n / i - n / (i - n) mod 16
        const int n = ctrl & 15, kind = (ctrl >> 4) & 3, i = lane & 15;
        const int from = kind == 0 ? i + n : i - n;
        const bool inside = kind == 2 || (from >= 0 && from < 16);
        const int r = (int)(uint32_t)emu::exchange((uint32_t)src, (lane & ~15) | (from & 15));
        return inside ? r : (bound_ctrl ? 0 : old);
    }
    fprintf(stderr, "emu: DPP control 0x%x not emulated\n", ctrl); abort();
}
// The two lanes of a pair run in lockstep on the device; where a kernel relies on that for data the pair shares in LDS (one lane reads the half of a sponge
// state the other has just written back), the harness inserts this into the copy it compiles (tests/emu_harness.py `rewrites`)
static inline void emu_pair_sync() { emu::my_wave->pair[emu::my_lane >> 1].wait(); }
#define __builtin_amdgcn_update_dpp(old, src, ctrl, rm, bm, bc) emu_update_dpp((old), (src), (ctrl), (rm), (bm), (bc))
// V_BITOP3_B32: bit (a << 2 | b << 1 | c) of the truth table, per bit position
static inline uint32_t emu_bitop3(uint32_t a, uint32_t b, uint32_t c, unsigned tt)
{
    uint32_t r = 0;
    for (unsigned i = 0; i < 8; i++)
        if (tt & (1u << i)) r |= ((i & 4) ? a : ~a) & ((i & 2) ? b : ~b) & ((i & 1) ? c : ~c);
    return r;
}
#define __builtin_amdgcn_bitop3_b32(a, b, c, tt) emu_bitop3((a), (b), (c), (tt))
static inline int __clz(int x) { return x ? __builtin_clz((unsigned)x) : 32; }
// V_PERM_B32: byte i of the result is picked by byte i of the selector from the eight bytes { s0 (4-7), s1 (0-3) }; 8-11: sign bits (not used by the engine: abort),
// 12: 0x00, 13-15: 0xFF
static inline uint32_t emu_perm(uint32_t s0, uint32_t s1, uint32_t sel)
{
    const uint64_t src = ((uint64_t)s0 << 32) | s1;
    uint32_t r = 0;
    for (int i = 0; i < 4; i++) {
        const uint32_t k = (sel >> (8 * i)) & 0xFFu;
        uint32_t b;
        if (k < 8) b = (uint32_t)(src >> (8 * k)) & 0xFFu; else if (k == 12) b = 0; else if (k >= 13) b = 0xFFu; else { fprintf(stderr, "emu: v_perm selector %u not emulated\n", k); abort(); }
        r |= b << (8 * i);
    }
    return r;
}
#define __builtin_amdgcn_perm(a, b, sel) emu_perm((a), (b), (sel))
#define __builtin_amdgcn_sched_barrier(x) ((void)0)
#define __builtin_amdgcn_s_setprio(x) ((void)0)
namespace emu { alignas(16) inline unsigned char dyn_lds[160 * 1024]; }     // the workgroup's dynamic LDS (`extern __shared__`, rewritten by the harness's build step)

static inline uint64_t __ballot(bool p)
{
    emu::my_wave->buf[emu::my_lane] = p ? 1 : 0;
    emu::wave_sync();
    uint64_t r = 0;
    for (int i = 0; i < 64; i++) r |= emu::my_wave->buf[i] << i;
    emu::wave_sync();
    return r;
}
static inline int __any(int p) { return __ballot(p != 0) != 0; }
static inline int __all(int p) { return __ballot(p != 0) == ~0ull; }
// row_ror:n of a DPP operand (lane i of a row of 16 reads lane (i - n) mod 16): the harness's stand-in for csrc/falcon.hip's v_add_f32_dpp chain (inline assembly)
static inline float emu_row_ror(float v, int n) { uint32_t b; memcpy(&b, &v, 4); b = (uint32_t)emu::exchange(b, (emu::my_lane & ~15) | ((emu::my_lane - n) & 15)); float r; memcpy(&r, &b, 4); return r; }
#ifndef EMU_HAVE_NTT_CORE              // a harness that includes csrc/ntt_core.h gets sc::wave_lds_fence from there, over the two builtins below
static inline void wave_lds_fence() { emu::wave_sync(); }               // csrc/ntt_core.h: a wavefront-scope fence + wave barrier
#endif
#define __builtin_amdgcn_fence(...) ((void)0)
#define __builtin_amdgcn_wave_barrier() emu::wave_sync()
#define __builtin_amdgcn_readfirstlane(v) emu_readlane((v), 0)
static inline void __syncthreads() { emu::cur_block->bar.wait(); }
static inline int __syncthreads_or(int p)           // a barrier that also returns whether any thread of the workgroup brought a non-zero value
{
    if (p) __atomic_store_n(&emu::cur_block->or_acc, 1, __ATOMIC_RELAXED);
    emu::cur_block->bar.wait();
    const int r = __atomic_load_n(&emu::cur_block->or_acc, __ATOMIC_RELAXED);
    emu::cur_block->bar.wait();
    if (threadIdx.x == 0) __atomic_store_n(&emu::cur_block->or_acc, 0, __ATOMIC_RELAXED);
    emu::cur_block->bar.wait();
    return r;
}
static inline unsigned long long atomicAdd(unsigned long long *p, unsigned long long v) { return __atomic_fetch_add(p, v, __ATOMIC_RELAXED); }
#define __builtin_amdgcn_s_memtime() 0ull
static inline uint32_t atomicAdd(uint32_t *p, uint32_t v) { return __atomic_fetch_add(p, v, __ATOMIC_RELAXED); }
static inline uint32_t atomicOr(uint32_t *p, uint32_t v) { return __atomic_fetch_or(p, v, __ATOMIC_RELAXED); }
static inline uint32_t __umulhi(uint32_t a, uint32_t b) { return (uint32_t)(((uint64_t)a * b) >> 32); }
static inline uint64_t __umul64hi(uint64_t a, uint64_t b) { return (uint64_t)(((unsigned __int128)a * b) >> 64); }
static inline int __mul24(int a, int b) { return (int)(uint32_t)((int64_t)((int32_t)((uint32_t)a << 8) >> 8) * (int64_t)((int32_t)((uint32_t)b << 8) >> 8)); }
static inline int __float_as_int(float f) { int i; memcpy(&i, &f, 4); return i; }
static inline float __int_as_float(int i) { float f; memcpy(&f, &i, 4); return f; }
static inline unsigned __float_as_uint(float f) { unsigned i; memcpy(&i, &f, 4); return i; }
static inline float __uint_as_float(unsigned i) { float f; memcpy(&f, &i, 4); return f; }
static inline int32_t __mulhi(int32_t a, int32_t b) { return (int32_t)(((int64_t)a * b) >> 32); }
static inline uint32_t emu_alignbit(uint32_t hi, uint32_t lo, uint32_t sh) { return (uint32_t)((((uint64_t)hi << 32) | lo) >> (sh & 31)); }
#define __builtin_amdgcn_alignbit(hi, lo, sh) emu_alignbit((hi), (lo), (sh))
using std::min;
using std::max;
// csrc/modarith.h declares the 24-bit multiplies as functions bound to LLVM intrinsic names; the harness's hip/hip_runtime.h stand-in removes the binding
// (tests/emu/shim) and these are the functions (v_mul_u32_u24, v_mul_hi_u32_u24, v_mul_i32_i24: operands taken modulo 2^24, the signed one sign-extended from bit 23)
extern "C" inline uint32_t sc_llvm_mul_u24(uint32_t a, uint32_t b) { return (uint32_t)((uint64_t)(a & 0xFFFFFFu) * (b & 0xFFFFFFu)); }
extern "C" inline uint32_t sc_llvm_mulhi_u24(uint32_t a, uint32_t b) { return (uint32_t)(((uint64_t)(a & 0xFFFFFFu) * (b & 0xFFFFFFu)) >> 32); }
extern "C" inline int32_t sc_llvm_mul_i24(int32_t a, int32_t b)
{
    const int64_t x = (int64_t)((int32_t)((uint32_t)a << 8) >> 8), y = (int64_t)((int32_t)((uint32_t)b << 8) >> 8);
    return (int32_t)(uint32_t)(uint64_t)(x * y);
}
