// sc_hip.hip - kernels and C ABI (include/sc_hip.h) of the MI355X lattice-arithmetic engine.
// gfx950 only; there is no host fallback: without a device every compute entry fails.
#include "../../include/sc_hip.h"
#include "ntt_core.h"

#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <atomic>
#include <string>
#include <type_traits>
#include <vector>
#include "cdt_guide.h"

using namespace sc;

#include "ntt_kernel.h"

enum { EW_POINTWISE = 0, EW_NORMALIZE = 1, EW_CENTER = 2, EW_ADD = 3, EW_SUB = 4 };

// Elementwise ops on int32 arrays, 16 B per lane, grid-stride.  r2 = 2^32 mod q (Shoup pair
// r2/r2p) turns the Montgomery product a*b*2^-32 back into a*b.
template <int OPE>
__global__ __launch_bounds__(256) void ew_kernel(const int32_t *__restrict__ a, const int32_t *__restrict__ b, int32_t *__restrict__ out,
                                                 size_t len, ModK k, uint32_t r2, uint32_t r2p)
{
    const size_t nv = len / 4;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < nv; i += stride) {
        const int4 va = reinterpret_cast<const int4 *>(a)[i];
        int4 vb = make_int4(0, 0, 0, 0);
        if (OPE == EW_POINTWISE || OPE == EW_ADD || OPE == EW_SUB) vb = reinterpret_cast<const int4 *>(b)[i];
        int xa[4] = {va.x, va.y, va.z, va.w}, xb[4] = {vb.x, vb.y, vb.z, vb.w}, o[4];
#pragma unroll
        for (int e = 0; e < 4; e++) {
            if (OPE == EW_ADD) o[e] = (int)((uint32_t)xa[e] + (uint32_t)xb[e]);          // (any int32 operands: the sum wraps, as the reference's add_32 does on this target)
            else if (OPE == EW_SUB) o[e] = (int)((uint32_t)xa[e] - (uint32_t)xb[e]);
            else {
                uint32_t v = reduce_i32(xa[e], k);
                if (OPE == EW_POINTWISE) {
                    v = montmul(v, reduce_i32(xb[e], k), k);                 // a b 2^-32, [0,q)
                    v = Mod<MODE_U32>::mulc(v, r2, r2p, k);                   // * 2^32
                }
                v = csub(v, k.q);
                if (OPE == EW_CENTER) o[e] = (v > k.half) ? (int32_t)v - (int32_t)k.q : (int32_t)v;
                else o[e] = (int32_t)v;
            }
        }
        reinterpret_cast<int4 *>(out)[i] = make_int4(o[0], o[1], o[2], o[3]);
    }
    // tail (len not a multiple of 4)
    if (blockIdx.x == 0 && threadIdx.x < (len & 3)) {
        const size_t i = nv * 4 + threadIdx.x;
        int32_t o;
        if (OPE == EW_ADD) o = (int32_t)((uint32_t)a[i] + (uint32_t)b[i]);
        else if (OPE == EW_SUB) o = (int32_t)((uint32_t)a[i] - (uint32_t)b[i]);
        else {
            uint32_t v = reduce_i32(a[i], k);
            if (OPE == EW_POINTWISE) {
                v = montmul(v, reduce_i32(b[i], k), k);
                v = Mod<MODE_U32>::mulc(v, r2, r2p, k);
            }
            v = csub(v, k.q);
            o = (OPE == EW_CENTER && v > k.half) ? (int32_t)v - (int32_t)k.q : (int32_t)v;
        }
        out[i] = o;
    }
}

// CDT sampler.  The reference's halving search (gaussian_cdf.c:536-553) returns, for a monotone table, the largest index b
// with cdf[b] < x (0 if there is none); the sign is the LSB of the same word (:658).  The same value comes from one guide read
// and S independent table reads (cdt_guide.h); table and guide in LDS, four samples per lane per iteration (16-byte loads and stores).
template <int S>
__device__ __forceinline__ int32_t cdt_one(uint64_t u, const uint64_t *tab, const uint16_t *guide, const CdtGuide &g, int32_t centre)
{
    const uint32_t r = cdt_lookup<S>(u, tab, guide, g);
    return ((u & 1) ? (int32_t)r : -(int32_t)r) + centre;
}

// 32-bit precision (gaussian_cdf_sample_32, gaussian_cdf.c:765-774: one prng_32 word per sample, "table entry < word" on 32 bits, the sign from the word's
// bit 0) rides on the 64-bit search: the handle's table holds entry << 32 | 0xFFFFFFFF and a word x searches as x << 32 | 0xFFFFFFFE | (x & 1) - the
// same order relation (entry < x on the words; an entry equal to x is not below it) and the same sign bit.
__device__ __forceinline__ uint64_t widen32(uint32_t x) { return ((uint64_t)x << 32) | 0xFFFFFFFEull | (uint64_t)(x & 1u); }

// HALF: no usable guide - the halving search itself; otherwise S probes behind the guide (S = 0: the guide's own count, as in cdt_kernel<0>)
template <int S, bool HALF>
__global__ __launch_bounds__(256) void cdt32_kernel(const uint64_t *__restrict__ cdf, const uint16_t *__restrict__ gd, int size, CdtGuide g, const uint32_t *__restrict__ x,
                                                    int32_t *__restrict__ out, size_t count, int32_t centre)
{
    extern __shared__ __attribute__((aligned(16))) uint64_t tab[];
    uint16_t *guide = reinterpret_cast<uint16_t *>(tab + size + CDT_PAD);
    for (int i = threadIdx.x; i < size + CDT_PAD; i += blockDim.x) tab[i] = cdf[i];
    if (!HALF) for (uint32_t i = threadIdx.x; i < (g.total + 1) / 2; i += blockDim.x) reinterpret_cast<uint32_t *>(guide)[i] = reinterpret_cast<const uint32_t *>(gd)[i];
    __syncthreads();
    auto one = [&](uint32_t w) -> int32_t {
        const uint64_t u = widen32(w);
        if constexpr (!HALF) return cdt_one<S>(u, tab, guide, g, centre);
        uint32_t a = 0;
        for (uint32_t st = (uint32_t)size >> 1; st > 0; st >>= 1) { const uint32_t b = a + st; if (tab[b] < u) a = b; }
        return ((u & 1) ? (int32_t)a : -(int32_t)a) + centre;
    };
    const size_t stride = (size_t)gridDim.x * blockDim.x, quads = count / 4;
    if (((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(out)) & 15) == 0) {
        for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < quads; i += stride) {
            const uint4 w = reinterpret_cast<const uint4 *>(x)[i];
            reinterpret_cast<int4 *>(out)[i] = make_int4(one(w.x), one(w.y), one(w.z), one(w.w));
        }
        if (blockIdx.x == 0 && threadIdx.x < (count & 3)) { const size_t i = quads * 4 + threadIdx.x; out[i] = one(x[i]); }
    } else {
        for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += stride) out[i] = one(x[i]);
    }
}

template <int S>
__global__ __launch_bounds__(256) void cdt_kernel(const uint64_t *__restrict__ cdf, const uint16_t *__restrict__ gd, int size, CdtGuide g, const uint64_t *__restrict__ x,
                                                  int32_t *__restrict__ out, size_t count, int32_t centre)
{
    extern __shared__ __attribute__((aligned(16))) uint64_t tab[];
    uint16_t *guide = reinterpret_cast<uint16_t *>(tab + size + CDT_PAD);
    for (int i = threadIdx.x; i < size + CDT_PAD; i += blockDim.x) tab[i] = cdf[i];
    for (uint32_t i = threadIdx.x; i < (g.total + 1) / 2; i += blockDim.x) reinterpret_cast<uint32_t *>(guide)[i] = reinterpret_cast<const uint32_t *>(gd)[i];
    __syncthreads();
    const size_t stride = (size_t)gridDim.x * blockDim.x, quads = count / 4;
    const bool vec = ((reinterpret_cast<uintptr_t>(x) & 15) == 0) && ((reinterpret_cast<uintptr_t>(out) & 15) == 0);
    if (vec) {
        for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < quads; i += stride) {
            const ulonglong2 u = reinterpret_cast<const ulonglong2 *>(x)[2 * i], w = reinterpret_cast<const ulonglong2 *>(x)[2 * i + 1];
            reinterpret_cast<int4 *>(out)[i] = make_int4(cdt_one<S>(u.x, tab, guide, g, centre), cdt_one<S>(u.y, tab, guide, g, centre),
                                                         cdt_one<S>(w.x, tab, guide, g, centre), cdt_one<S>(w.y, tab, guide, g, centre));
        }
        if (blockIdx.x == 0 && threadIdx.x < (count & 3)) { const size_t i = quads * 4 + threadIdx.x; out[i] = cdt_one<S>(x[i], tab, guide, g, centre); }
    } else {
        for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += stride) out[i] = cdt_one<S>(x[i], tab, guide, g, centre);
    }
}

// 128- / 192-bit precision (gaussian_cdf_sample_128 / _192 over binary_search_128 / _192, gaussian_cdf.c:134-166,468-509): W = 2 / 3 prng_64 words a sample, limb 0
// first; the sign is limb 0's bit 0; the search goes to b when "x >= entry b" by compare_ge_prec (:110-132) - whose loop over the limbs ends with the verdict of
// the MOST SIGNIFICANT limb alone (retval = !(a < b) | (equal & retval): equality already makes the first term 1, so the limbs below never carry over).  So the
// search is the 64-bit one on the top limbs with ">=" for "<" - and x >= e is e < x + 1: it rides on the guide of the 64-bit form with x + 1, the one word
// without a successor (2^64 - 1, at or above every entry) answered directly: every step of the halving search is taken, the result is size - 1.
template <int S, bool HALF>
__global__ __launch_bounds__(256) void cdt_hp_kernel(const uint64_t *__restrict__ cdf, const uint16_t *__restrict__ gd, int size, CdtGuide g, const uint64_t *__restrict__ x, int W,
                                                     int32_t *__restrict__ out, size_t count, int32_t centre)
{
    extern __shared__ __attribute__((aligned(16))) uint64_t tab[];
    uint16_t *guide = reinterpret_cast<uint16_t *>(tab + size + CDT_PAD);
    for (int i = threadIdx.x; i < size + CDT_PAD; i += blockDim.x) tab[i] = cdf[i];
    if (!HALF) for (uint32_t i = threadIdx.x; i < (g.total + 1) / 2; i += blockDim.x) reinterpret_cast<uint32_t *>(guide)[i] = reinterpret_cast<const uint32_t *>(gd)[i];
    __syncthreads();
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += stride) {
        const uint64_t low = x[i * (size_t)W], top = x[i * (size_t)W + (size_t)(W - 1)];
        const uint64_t u = top + 1;
        uint32_t r;
        if constexpr (!HALF) r = cdt_lookup<S>(u, tab, guide, g);
        else { r = 0; for (uint32_t st = (uint32_t)size >> 1; st > 0; st >>= 1) { const uint32_t b = r + st; if (tab[b] < u) r = b; } }
        if (top == ~0ull) r = (uint32_t)size - 1u;
        out[i] = ((low & 1) ? (int32_t)r : -(int32_t)r) + centre;
    }
}

// no usable guide (a table some bin of which holds more than 7 entries): the halving search itself
__global__ __launch_bounds__(256) void cdt_halving_kernel(const uint64_t *__restrict__ cdf, int size, const uint64_t *__restrict__ x, int32_t *__restrict__ out, size_t count, int32_t centre)
{
    extern __shared__ __attribute__((aligned(16))) uint64_t tab[];
    for (int i = threadIdx.x; i < size; i += blockDim.x) tab[i] = cdf[i];
    __syncthreads();
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += stride) {
        const uint64_t u = x[i];
        uint32_t a = 0;
        for (uint32_t st = (uint32_t)size >> 1; st > 0; st >>= 1) { const uint32_t b = a + st; if (tab[b] < u) a = b; }
        out[i] = ((u & 1) ? (int32_t)a : -(int32_t)a) + centre;
    }
}

// ------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------

#include "engine_internal.h"
#include <atomic>

static thread_local std::string g_err;
thread_local bool sc::t_skip_host_pin = false;
thread_local const sc::HostPins *sc::t_driver_pins = nullptr;
extern "C" void sc_hip_bench_set_host_pin(int on) { sc::HostPins::mode() = on; }
static std::atomic<int> g_cus{256};                  // (read by launches on any host thread, written by sc_hip_ctx_create)
// streaming I/O of the NTT kernels (ntt_kernel<..., STREAM>): taken when an operation's operands exceed the threshold (default: the Infinity Cache's 256 MiB)
static bool g_stream_io = true;
static size_t g_stream_threshold = (size_t)256 << 20;
extern "C" void sc_hip_bench_set_ntt_stream(int mode)
{
    // measurement / test hook: 0 never, 1 by working set (default), 2 always (whatever the size: the parity tests run every transform through both forms)
    g_stream_io = mode != 0;
    g_stream_threshold = mode == 2 ? 0 : (size_t)256 << 20;
}
int sc::fail(const std::string &m) { g_err = m; return SC_HIP_FAILURE; }
using sc::fail;

static uint64_t mulmod(uint64_t a, uint64_t b, uint64_t q) { return (uint64_t)(((unsigned __int128)a * b) % q); }
static uint64_t powmod(uint64_t a, uint64_t e, uint64_t q)
{
    uint64_t r = 1; a %= q;
    while (e) { if (e & 1) r = mulmod(r, a, q); a = mulmod(a, a, q); e >>= 1; }
    return r;
}
static uint32_t brv(uint32_t x, int bits) { uint32_t r = 0; for (int b = 0; b < bits; b++) r |= ((x >> b) & 1u) << (bits - 1 - b); return r; }
static bool is_prime(uint64_t q) { if (q < 2) return false; for (uint64_t d = 2; d * d <= q; d++) if (q % d == 0) return false; return true; }

extern "C" const char *sc_hip_last_error(void) { return g_err.c_str(); }

extern "C" int sc_hip_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

extern "C" int sc_hip_ctx_create(int device, sc_hip_ctx_t **out)
{
    if (!out) return fail("ctx_create: null output");
    int n = 0;
    HIPCHK(hipGetDeviceCount(&n));
    if (device < 0 || device >= n) return fail("ctx_create: no such HIP device (the engine has no CPU fallback)");
    HIPCHK(hipSetDevice(device));
    hipDeviceProp_t prop;
    HIPCHK(hipGetDeviceProperties(&prop, device));
    sc_hip_ctx *c = new sc_hip_ctx;
    c->device = device;
    c->cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    g_cus.store(c->cus, std::memory_order_relaxed);
    *out = c;
    return SC_HIP_SUCCESS;
}

extern "C" void sc_hip_ctx_destroy(sc_hip_ctx_t *ctx) { delete ctx; }

extern "C" int sc_hip_synchronize(sc_hip_ctx_t *ctx, void *stream)
{
    if (!ctx) return fail("synchronize: null ctx");
    HIPCHK(hipSetDevice(ctx->device));
    HIPCHK(hipStreamSynchronize((hipStream_t)stream));
    return SC_HIP_SUCCESS;
}

// Shoup companion floor(w * 2^bits / q)
static uint32_t shoup(uint64_t w, uint64_t q, int bits) { return (uint32_t)((((unsigned __int128)w) << bits) / q); }

extern "C" int sc_hip_plan_create(sc_hip_ctx_t *ctx, int32_t q, int n, sc_hip_plan_t **out)
{
    if (!ctx || !out) return fail("plan_create: null argument");
    if (n != 256 && n != 512 && n != 1024) return fail("plan_create: n must be 256, 512 or 1024");
    if (q < 3 || (uint32_t)q >= (1u << 30) || !is_prime((uint64_t)q) || ((uint64_t)q - 1) % (2 * (uint64_t)n) != 0)
        return fail("plan_create: q must be a prime < 2^30 with q == 1 (mod 2n)");
    HIPCHK(hipSetDevice(ctx->device));
    const uint64_t Q = (uint64_t)q;
    // the reference's root: smallest m >= 2 with m^n == -1 (roots_of_unity.c:67-88)
    uint64_t g = 0;
    for (uint64_t m = 2; m < Q - 1; m++) if (powmod(m, (uint64_t)n, Q) == Q - 1) { g = m; break; }
    if (!g) return fail("plan_create: no primitive 2n-th root of unity");

    sc_hip_plan *p = new sc_hip_plan;
    p->ctx = ctx; p->q = q; p->n = n; p->g = (int32_t)g;
    p->mv_ws = nullptr; p->mv_cap = 0; p->mv_event = nullptr; p->d_tab_s = nullptr; p->smad_ok = p->s32_ok = false;
    p->d_sink = nullptr; p->d_tab[0] = p->d_tab[1] = nullptr;
    if (hipMalloc((void **)&p->d_sink, 4096) != hipSuccess) { delete p; return fail("plan_create: device allocation failed"); }
    p->g_exec = nullptr; p->g_a = p->g_b = nullptr; p->g_out = nullptr; p->g_batch = 0; p->g_steps = 0;
    p->L = (n == 256) ? 8 : (n == 512) ? 9 : 10;
    p->w.resize(n); p->r.resize(n);
    p->w[0] = 1;
    for (int i = 1; i < n; i++) p->w[i] = (int32_t)mulmod((uint64_t)p->w[i - 1], g, Q);
    p->r[0] = (int32_t)((Q - 1) / (uint64_t)n);              // -n^-1 mod q (roots_of_unity.c:90-104)
    for (int i = 1; i < n; i++) p->r[i] = (int32_t)mulmod((uint64_t)p->r[i - 1], g, Q);

    const int L = p->L;
    std::vector<uint64_t> zf(n), zi(n);
    for (int kx = 0; kx < n; kx++) { zf[kx] = powmod(g, brv((uint32_t)kx, L), Q); zi[kx] = powmod(zf[kx], Q - 2, Q); }
    const uint64_t ninv = powmod((uint64_t)n, Q - 2, Q);
    const uint64_t R = (1ull << 32) % Q;
    uint32_t qinv = 1;                                         // q^-1 mod 2^32 by Newton
    for (int i = 0; i < 5; i++) qinv *= 2u - (uint32_t)q * qinv;

    for (int mi = 0; mi < 2; mi++) {
        const int bits = mi == 0 ? 24 : 32;
        ModK &k = p->k[mi];
        k.q = (uint32_t)q; k.q2 = 2u * (uint32_t)q; k.qinv = qinv;
        k.m32 = (uint32_t)((1ull << 32) / Q);
        k.one_p = shoup(1, Q, bits);
        k.half = (uint32_t)((q - 1) >> 1);
        std::vector<uint32_t> h(4 * (size_t)n);
        for (int kx = 0; kx < n; kx++) {
            h[kx] = (uint32_t)zf[kx];             h[n + kx] = shoup(zf[kx], Q, bits);
            h[2 * n + kx] = (uint32_t)zi[kx];     h[3 * n + kx] = shoup(zi[kx], Q, bits);
        }
        p->d_tab[mi] = nullptr;
        if (hipMalloc(&p->d_tab[mi], h.size() * 4) != hipSuccess ||
            sc::copy_in(p->d_tab[mi], h.data(), h.size() * 4) != hipSuccess) {
            sc_hip_plan_destroy(p);
            return fail("plan_create: cannot upload twiddle tables");
        }
        const uint64_t s0 = ninv, s1 = mulmod(ninv, zi[1], Q);
        const uint64_t m0 = mulmod(ninv, R, Q), m1 = mulmod(m0, zi[1], Q);
        p->c_inv[mi][0] = (uint32_t)s0;  p->c_inv[mi][1] = shoup(s0, Q, bits);
        p->c_inv[mi][2] = (uint32_t)s1;  p->c_inv[mi][3] = shoup(s1, Q, bits);
        p->c_mont[mi][0] = (uint32_t)m0; p->c_mont[mi][1] = shoup(m0, Q, bits);
        p->c_mont[mi][2] = (uint32_t)m1; p->c_mont[mi][3] = shoup(m1, Q, bits);
    }
    p->r2 = (uint32_t)R; p->r2p = shoup(R, Q, 32);
    // lazy forward butterflies: values reach 2q(L+1); they must stay below BETA and the
    // Montgomery product bound 4q(L+2) < 2^32 must hold
    p->lazy24 = (2ull * Q * (uint64_t)(L + 2) < (1ull << 24));
    p->lazy32 = (Q < (1ull << 25));
    // measured on MI355X (profiles/microbench_r01b.txt): v_mul_lo/hi_u32 issue at full rate on gfx950,
    // so the 4-instruction 32-bit Shoup product beats the 6-instruction 24-bit one for every q
    p->mode = MODE_U32;
    // q < 2^14 (7681, 12289): the 3-instruction multiply-add form with growing inverse sums (modarith.h MODE_MAD, ntt_core.h)
    // (the pointwise product multiplies a lazy forward value < 2q(L+1) by an operand < q in 32 bits: e.g. q = 15361, n = 512 does not qualify)
    p->mad_ok = Q < (1ull << 14) && 2ull * Q * (Q - 1) * (uint64_t)(L + 1) < (1ull << 32);
    if (p->mad_ok) p->mode = MODE_MAD;
    // 36 q^2 < 2^31 and n = 256 (q = 7681): signed lazy residues, no butterfly bias, two-instruction input reductions (MODE_SMAD)
    // n = 512, 1024 and q^2 L < 2^31 (q = 7681, 12289): the same with two extra reductions (bounds in modarith.h)
    const bool smad = (n == 256 && 36ull * Q * Q < (1ull << 31)) || (n > 256 && Q < (1ull << 14) && Q * Q * (uint64_t)L < (1ull << 31));
    // q < 2^25 (2^24 for n = 512, 1024, whose first inverse round is not reduced): signed lazy residues on the 32-bit multiplier (MODE_S32)
    p->s32_ok = Q < (1ull << 25) && (n == 256 || Q < (1ull << 24));
    if (smad || p->s32_ok) {
        std::vector<uint32_t> h(4 * (size_t)n, 0u);
        auto centre = [&](uint64_t w) -> int64_t { return w > Q / 2 ? (int64_t)w - (int64_t)Q : (int64_t)w; };
        auto sshoup = [&](int64_t w) -> uint32_t {                 // floor(w 2^32 / q) as a two's-complement 32-bit value
            __int128 num = (__int128)w * ((__int128)1 << 32);                  // (not w << 32: w may be negative)
            __int128 fl = num / (__int128)Q;
            if (num % (__int128)Q != 0 && num < 0) fl -= 1;
            return (uint32_t)(int32_t)fl;
        };
        for (int kx = 0; kx < n; kx++) {
            const int64_t f = centre(zf[kx]), iv = centre(zi[kx]);
            h[kx] = (uint32_t)(int32_t)f;          h[n + kx] = sshoup(f);
            h[2 * n + kx] = (uint32_t)(int32_t)iv; h[3 * n + kx] = sshoup(iv);
        }
        if (hipMalloc(&p->d_tab_s, h.size() * 4) != hipSuccess || sc::copy_in(p->d_tab_s, h.data(), h.size() * 4) != hipSuccess) {
            sc_hip_plan_destroy(p);
            return fail("plan_create: cannot upload the signed twiddle tables");
        }
        const int64_t cs0 = centre(ninv), cs1 = centre(mulmod(ninv, zi[1], Q));
        p->c_inv_s[0] = (uint32_t)(int32_t)cs0; p->c_inv_s[1] = sshoup(cs0); p->c_inv_s[2] = (uint32_t)(int32_t)cs1; p->c_inv_s[3] = sshoup(cs1);
        const uint64_t mm0 = mulmod(ninv, R, Q);
        const int64_t cm0 = centre(mm0), cm1 = centre(mulmod(mm0, zi[1], Q));
        p->c_mont_s[0] = (uint32_t)(int32_t)cm0; p->c_mont_s[1] = sshoup(cm0); p->c_mont_s[2] = (uint32_t)(int32_t)cm1; p->c_mont_s[3] = sshoup(cm1);
        p->smad_ok = smad;
        p->mode = smad ? MODE_SMAD : MODE_S32;
    }
    *out = p;
    return SC_HIP_SUCCESS;
}

extern "C" void sc_hip_plan_destroy(sc_hip_plan_t *p)
{
    if (!p) return;
    for (int i = 0; i < 2; i++) if (p->d_tab[i]) (void)sc::wipe_free(p->d_tab[i]);
    if (p->d_tab_s) (void)sc::wipe_free(p->d_tab_s);
    if (p->d_sink) (void)hipFree(p->d_sink);
    if (p->g_exec) (void)hipGraphExecDestroy((hipGraphExec_t)p->g_exec);
    if (p->mv_event) { (void)hipEventSynchronize((hipEvent_t)p->mv_event); (void)hipEventDestroy((hipEvent_t)p->mv_event); }
    if (p->mv_ws) (void)sc::wipe_free(p->mv_ws);
    delete p;
}
extern "C" int32_t sc_hip_plan_q(const sc_hip_plan_t *p) { return p ? p->q : 0; }
extern "C" int sc_hip_plan_n(const sc_hip_plan_t *p) { return p ? p->n : 0; }
extern "C" int32_t sc_hip_plan_root(const sc_hip_plan_t *p) { return p ? p->g : 0; }
extern "C" int sc_hip_plan_tables(const sc_hip_plan_t *p, int32_t *h_w, int32_t *h_r)
{
    if (!p || !h_w || !h_r) return fail("plan_tables: null argument");
    memcpy(h_w, p->w.data(), sizeof(int32_t) * (size_t)p->n);
    memcpy(h_r, p->r.data(), sizeof(int32_t) * (size_t)p->n);
    return SC_HIP_SUCCESS;
}
extern "C" int sc_hip_plan_set_arith(sc_hip_plan_t *p, int mode)
{
    if (!p) return fail("plan_set_arith: null plan");
    if (mode == 0) mode = p->smad_ok ? MODE_SMAD : p->s32_ok ? MODE_S32 : p->mad_ok ? MODE_MAD : MODE_U32;
    if (mode == MODE_S32 && !p->s32_ok) return fail("plan_set_arith: the signed 32-bit path needs q < 2^25 (2^24 for n = 512, 1024)");
    if (mode == MODE_SMAD && !p->smad_ok) return fail("plan_set_arith: the signed small-modulus path needs 36 q^2 < 2^31 (n = 256) or q^2 log2(n) < 2^31 and q < 2^14 (n = 512, 1024)");
    if (mode == MODE_U24 && (uint64_t)p->q >= (1ull << 23)) return fail("plan_set_arith: 24-bit path needs q < 2^23");
    if (mode == MODE_MAD && !p->mad_ok) return fail("plan_set_arith: the multiply-add path needs q < 2^14 and 2 q (q - 1) (log2 n + 1) < 2^32");
    if (mode != MODE_U24 && mode != MODE_U32 && mode != MODE_MAD && mode != MODE_SMAD && mode != MODE_S32) return fail("plan_set_arith: mode must be 0 .. 5");
    if (mode != p->mode && p->g_exec) { (void)hipGraphExecDestroy((hipGraphExec_t)p->g_exec); p->g_exec = nullptr; }     // a captured loop holds the old kernel
    p->mode = mode;
    return SC_HIP_SUCCESS;
}

// grid = what is resident at once (occupancy x CUs), so that every wave loops over many groups
// and the run-ahead pipeline stays primed; never more blocks than there are groups of 4 waves
template <int L, int MODE, bool LAZY, int OP, bool STREAM = false>
static hipError_t launch_ntt_t(const NttArgs &a, int blocks, hipStream_t s)
{
    // (atomic: plans are shared between host threads - the multi-device driver launches from one thread per worker - and two first launches may both fill it in, with the same value)
    static std::atomic<int> per_cu_cache{0};
    int per_cu = per_cu_cache.load(std::memory_order_relaxed);
    if (per_cu == 0) {
        int nb = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, ntt_kernel<L, MODE, LAZY, OP, STREAM>, 256, 0) != hipSuccess || nb < 1) nb = 2;
        per_cu = nb;
        per_cu_cache.store(nb, std::memory_order_relaxed);
    }
    const long long cap = (long long)per_cu * g_cus.load(std::memory_order_relaxed);
    if (blocks > cap) blocks = (int)cap;
    if (blocks <= 0) return hipSuccess;                       // warm-up call: the occupancy query above is all that was wanted
    hipLaunchKernelGGL((ntt_kernel<L, MODE, LAZY, OP, STREAM>), dim3(blocks), dim3(256), 0, s, a);
    return hipGetLastError();
}
template <int L, int OP>
static hipError_t launch_ntt_l(const NttArgs &a, int mode, bool lazy, int blocks, hipStream_t s, bool stream_io)
{
    // the streaming I/O form exists for the two signed modes (every modulus of the reference's parameter sets runs on one of them)
    if constexpr (OP == OP_MULNTT2) {                                  // no streaming form: its two operands are broadcast and its callers' batches are cache-sized
        if (mode == MODE_SMAD) return launch_ntt_t<L, MODE_SMAD, true, OP>(a, blocks, s);
        if (mode == MODE_S32) return launch_ntt_t<L, MODE_S32, true, OP>(a, blocks, s);
    } else {
    if (mode == MODE_SMAD) return stream_io ? launch_ntt_t<L, MODE_SMAD, true, OP, true>(a, blocks, s) : launch_ntt_t<L, MODE_SMAD, true, OP>(a, blocks, s);
    if (mode == MODE_S32) return stream_io ? launch_ntt_t<L, MODE_S32, true, OP, true>(a, blocks, s) : launch_ntt_t<L, MODE_S32, true, OP>(a, blocks, s);
    }
    if constexpr (OP == OP_MULNTT2) return hipErrorInvalidValue;      // (signed modes only: sc::polymul_ntt_bcast2 falls back before it gets here)
    else if constexpr (OP == OP_INV) {       // the inverse has no lazy variant: one instantiation per mode
        if (mode == MODE_MAD) return launch_ntt_t<L, MODE_MAD, false, OP>(a, blocks, s);
        return mode == MODE_U24 ? launch_ntt_t<L, MODE_U24, false, OP>(a, blocks, s) : launch_ntt_t<L, MODE_U32, false, OP>(a, blocks, s);
    } else {
        if (mode == MODE_MAD) return launch_ntt_t<L, MODE_MAD, true, OP>(a, blocks, s);          // q < 2^14: always lazy
        if (mode == MODE_U24) return lazy ? launch_ntt_t<L, MODE_U24, true, OP>(a, blocks, s) : launch_ntt_t<L, MODE_U24, false, OP>(a, blocks, s);
        return lazy ? launch_ntt_t<L, MODE_U32, true, OP>(a, blocks, s) : launch_ntt_t<L, MODE_U32, false, OP>(a, blocks, s);
    }
}
template <int OP>
static hipError_t launch_ntt_op(int L, const NttArgs &a, int mode, bool lazy, int blocks, hipStream_t s, bool stream_io)
{
    switch (L) {
    case 8: return launch_ntt_l<8, OP>(a, mode, lazy, blocks, s, stream_io);
    case 9: return launch_ntt_l<9, OP>(a, mode, lazy, blocks, s, stream_io);
    default: return launch_ntt_l<10, OP>(a, mode, lazy, blocks, s, stream_io);
    }
}

static int run_ntt(sc_hip_plan_t *p, int op, size_t batch, const int32_t *d0, const int32_t *d1, int32_t *dout, void *stream, bool bcast = false, bool warm_only = false,
                   const uint32_t *batch_dev = nullptr)
{
    if (!p) return fail("ntt: null plan");
    if (batch == 0 && !warm_only) return SC_HIP_SUCCESS;
    if (!d0 || !dout || ((op == OP_MULNTT || op == OP_MUL) && !d1)) return fail("ntt: null device pointer");
    HIPCHK(hipSetDevice(p->ctx->device));
    const int mi = p->mode == MODE_U24 ? 0 : 1;
    const int n = p->n;
    NttArgs a;
    a.in0 = d0; a.in1 = d1; a.out = dout; a.sink = p->d_sink; a.in1_bcast = bcast ? 1u : 0u; a.batch_dev = batch_dev;
    const bool sgn = p->mode == MODE_SMAD || p->mode == MODE_S32;
    a.zf = sgn ? p->d_tab_s : p->d_tab[mi]; a.zfp = a.zf + n; a.zi = a.zf + 2 * n; a.zip = a.zf + 3 * n;
    a.batch = batch; a.k = p->k[mi];
    const uint32_t *c = p->mode == MODE_SMAD ? (p->q < 8192 ? p->c_inv[mi] : p->c_inv_s) : p->mode == MODE_S32 ? (op == OP_INV ? p->c_inv_s : p->c_mont_s)
                        : (op == OP_INV || p->mode == MODE_MAD) ? p->c_inv[mi] : p->c_mont[mi];     // the small-modulus products leave no 2^-32
    a.c0 = c[0]; a.c0p = c[1]; a.c1 = c[2]; a.c1p = c[3];
    const bool lazy = mi == 0 ? p->lazy24 : p->lazy32;
    const int pw = 1024 / n;                                  // polynomials per wave
    // operands that cannot stay in the Infinity Cache between two uses are streamed: 16 bytes per lane, non-temporal (see ntt_kernel); whole groups only
    const size_t ws_bytes = batch * (size_t)n * 4 * ((op == OP_MULNTT || op == OP_MUL) && !bcast ? 3 : 2);
    const bool stream_io = !warm_only && sgn && !bcast && g_stream_io && ws_bytes > g_stream_threshold && batch >= (size_t)pw &&
                           (((uintptr_t)d0 | (uintptr_t)dout) & 15) == 0;
    auto launch = [&](const NttArgs &aa, size_t nb, bool st) -> hipError_t {
        const size_t groups = (nb + pw - 1) / pw;
        size_t blocks = (groups + 3) / 4;
        if (blocks > (size_t)1 << 30) blocks = (size_t)1 << 30;
        if (warm_only) blocks = 0;                                // only the per-instantiation occupancy query runs (launch_ntt_t)
        switch (op) {
        case OP_FWD: return launch_ntt_op<OP_FWD>(p->L, aa, p->mode, lazy, (int)blocks, (hipStream_t)stream, st);
        case OP_INV: return launch_ntt_op<OP_INV>(p->L, aa, p->mode, lazy, (int)blocks, (hipStream_t)stream, st);
        case OP_MULNTT: return launch_ntt_op<OP_MULNTT>(p->L, aa, p->mode, lazy, (int)blocks, (hipStream_t)stream, st);
        default: return launch_ntt_op<OP_MUL>(p->L, aa, p->mode, lazy, (int)blocks, (hipStream_t)stream, st);
        }
    };
    hipError_t e;
    if (stream_io) {
        const size_t whole = batch / pw * pw, tail = batch - whole;
        a.batch = whole;
        e = launch(a, whole, true);
        if (e == hipSuccess && tail) {                            // the ragged last group: the plain form, on its own
            NttArgs t = a;
            t.in0 = d0 + whole * n; t.out = dout + whole * n; t.batch = tail;
            if (d1) t.in1 = d1 + whole * n;
            e = launch(t, tail, false);
        }
    } else e = launch(a, batch, false);
    if (e != hipSuccess) return fail(std::string("ntt kernel launch: ") + hipGetErrorString(e));
    return SC_HIP_SUCCESS;
}

extern "C" int sc_hip_ntt_fwd(sc_hip_plan_t *p, size_t batch, const int32_t *d_in, int32_t *d_out, void *stream)
{ return run_ntt(p, OP_FWD, batch, d_in, nullptr, d_out, stream); }
extern "C" int sc_hip_ntt_inv(sc_hip_plan_t *p, size_t batch, const int32_t *d_in, int32_t *d_out, void *stream)
{ return run_ntt(p, OP_INV, batch, d_in, nullptr, d_out, stream); }
extern "C" int sc_hip_polymul_ntt(sc_hip_plan_t *p, size_t batch, const int32_t *d_a, const int32_t *d_bhat, int32_t *d_out, void *stream)
{ return run_ntt(p, OP_MULNTT, batch, d_a, d_bhat, d_out, stream); }
// internal: out1[b] = INTT(NTT(a[b]) o b1hat), out2[b] = INTT(NTT(a[b]) o b2hat) with ONE forward transform (ring-TESLA: v1 = a1 y, v2 = a2 y, ring_tesla.c:842-850,
// 1005-1012): a sixth of the two products' transforms and one of their two reads of a.  The signed modes only (every reference modulus runs on one of them);
// any other plan takes the two single products.
int sc::polymul_ntt_bcast2(sc_hip_plan *p, size_t batch, const int32_t *d_a, const int32_t *d_b1hat, const int32_t *d_b2hat, int32_t *d_out1, int32_t *d_out2, void *stream,
                           const uint32_t *batch_dev)
{
    if (!p) return fail("ntt: null plan");
    if (p->mode != MODE_SMAD && p->mode != MODE_S32) {
        if (sc::polymul_ntt_bcast(p, batch, d_a, d_b1hat, d_out1, stream, batch_dev)) return SC_HIP_FAILURE;
        return sc::polymul_ntt_bcast(p, batch, d_a, d_b2hat, d_out2, stream, batch_dev);
    }
    if (batch == 0) return SC_HIP_SUCCESS;
    if (!d_a || !d_b1hat || !d_b2hat || !d_out1 || !d_out2) return fail("ntt: null device pointer");
    HIPCHK(hipSetDevice(p->ctx->device));
    const int n = p->n;
    NttArgs a;
    a.in0 = d_a; a.in1 = d_b1hat; a.in2 = d_b2hat; a.out = d_out1; a.out2 = d_out2; a.sink = p->d_sink; a.in1_bcast = 1u; a.batch_dev = batch_dev;
    a.zf = p->d_tab_s; a.zfp = a.zf + n; a.zi = a.zf + 2 * n; a.zip = a.zf + 3 * n;
    a.batch = batch; a.k = p->k[1];
    const uint32_t *c = p->mode == MODE_SMAD ? (p->q < 8192 ? p->c_inv[1] : p->c_inv_s) : p->c_mont_s;      // as run_ntt for OP_MULNTT
    a.c0 = c[0]; a.c0p = c[1]; a.c1 = c[2]; a.c1p = c[3];
    const int pw = 1024 / n;
    size_t blocks = ((batch + pw - 1) / pw + 3) / 4;
    if (blocks > (size_t)1 << 30) blocks = (size_t)1 << 30;
    const hipError_t e = launch_ntt_op<OP_MULNTT2>(p->L, a, p->mode, true, (int)blocks, (hipStream_t)stream, false);
    if (e != hipSuccess) return fail(std::string("ntt kernel launch: ") + hipGetErrorString(e));
    return SC_HIP_SUCCESS;
}
extern "C" int sc_hip_polymul_ntt_pair(sc_hip_plan_t *p, size_t batch, const int32_t *d_a, const int32_t *d_b1hat, const int32_t *d_b2hat, int32_t *d_out1, int32_t *d_out2, void *stream)
{ return sc::polymul_ntt_bcast2(p, batch, d_a, d_b1hat, d_b2hat, d_out1, d_out2, stream); }
// internal: out[b] = INTT(NTT(a[b]) o bhat) with ONE NTT-domain operand for the whole batch (bliss_b.c:1378-1384: bhat = public key a)
int sc::polymul_ntt_bcast(sc_hip_plan *p, size_t batch, const int32_t *d_a, const int32_t *d_bhat, int32_t *d_out, void *stream, const uint32_t *batch_dev)
{ return run_ntt(p, OP_MULNTT, batch, d_a, d_bhat, d_out, stream, true, false, batch_dev); }
extern "C" int sc_hip_polymul(sc_hip_plan_t *p, size_t batch, const int32_t *d_a, const int32_t *d_b, int32_t *d_out, void *stream)
{ return run_ntt(p, OP_MUL, batch, d_a, d_b, d_out, stream); }

template <int OPE>
static int run_ew(sc_hip_ctx *ctx, const ModK &k, uint32_t r2, uint32_t r2p, size_t len, const int32_t *a, const int32_t *b, int32_t *out, void *stream)
{
    if (len == 0) return SC_HIP_SUCCESS;
    if (!a || !out || ((OPE == EW_POINTWISE || OPE == EW_ADD || OPE == EW_SUB) && !b)) return fail("elementwise: null device pointer");
    if ((((uintptr_t)a | (uintptr_t)out | (uintptr_t)b) & 15) != 0) return fail("elementwise: device pointers must be 16-byte aligned");
    HIPCHK(hipSetDevice(ctx->device));
    size_t blocks = (len / 4 + 255) / 256;
    if (blocks == 0) blocks = 1;
    const size_t cap = (size_t)ctx->cus * 8;
    if (blocks > cap) blocks = cap;
    hipLaunchKernelGGL((ew_kernel<OPE>), dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, a, b, out, len, k, r2, r2p);
    HIPCHK(hipGetLastError());
    return SC_HIP_SUCCESS;
}

extern "C" int sc_hip_pointwise(sc_hip_plan_t *p, size_t len, const int32_t *d_a, const int32_t *d_b, int32_t *d_out, void *stream)
{ if (!p) return fail("pointwise: null plan"); return run_ew<EW_POINTWISE>(p->ctx, p->k[1], p->r2, p->r2p, len, d_a, d_b, d_out, stream); }
extern "C" int sc_hip_normalize(sc_hip_plan_t *p, size_t len, const int32_t *d_in, int32_t *d_out, void *stream)
{ if (!p) return fail("normalize: null plan"); return run_ew<EW_NORMALIZE>(p->ctx, p->k[1], 0, 0, len, d_in, nullptr, d_out, stream); }
extern "C" int sc_hip_center(sc_hip_plan_t *p, size_t len, const int32_t *d_in, int32_t *d_out, void *stream)
{ if (!p) return fail("center: null plan"); return run_ew<EW_CENTER>(p->ctx, p->k[1], 0, 0, len, d_in, nullptr, d_out, stream); }
extern "C" int sc_hip_poly_add(sc_hip_ctx_t *ctx, size_t len, const int32_t *d_a, const int32_t *d_b, int32_t *d_out, void *stream)
{ if (!ctx) return fail("poly_add: null ctx"); ModK k = {}; return run_ew<EW_ADD>(ctx, k, 0, 0, len, d_a, d_b, d_out, stream); }
extern "C" int sc_hip_poly_sub(sc_hip_ctx_t *ctx, size_t len, const int32_t *d_a, const int32_t *d_b, int32_t *d_out, void *stream)
{ if (!ctx) return fail("poly_sub: null ctx"); ModK k = {}; return run_ew<EW_SUB>(ctx, k, 0, 0, len, d_a, d_b, d_out, stream); }

// sc_hip_matvec lives in dilithium.hip (it shares the module-LWE kernels)

// ---- CDT -----------------------------------------------------------------------------------

extern "C" int sc_hip_cdt_table_size(float tail, float sigma)
{
    if (!(tail > 0.f) || !(sigma > 0.f)) return 0;
    // size = 2^ceil(log2((size_t)(tail*sigma)))  (gaussian_cdf.c:562, sc_math.c sc_ceil_log2)
    const float prod = tail * sigma;
    const uint32_t v = (uint32_t)prod;
    int bits = 0;
    while ((1ull << bits) < v) bits++;
    return 1 << bits;
}

// long-double recurrence of gaussian_cdf.c:588-607 (sigma*sigma is a FLOAT product there) for the standard deviation the table is tabulated at
static void cdt_tabulate(float sigma, uint64_t *cdf, int size)
{
    const long double two_sqrtpi = 1.128379167095512573896158903121545172L;
    const long double sqrt1_2 = 0.707106781186547524400844362104849039L;
    const long double d = two_sqrtpi * sqrt1_2 * 18446744073709551616.0L / (long double)sigma;
    const float sigma_sq = sigma * sigma;
    const long double e = -0.5L / (long double)sigma_sq;
    long double s = 0.5L * d;
    int i;
    cdf[0] = 0;
    for (i = 1; i < size - 1; i++) {
        cdf[i] = (uint64_t)s;
        if (cdf[i] == 0) break;
        s += d * expl(e * ((long double)(i * i)));
    }
    for (; i < size; i++) cdf[i] = 0xFFFFFFFFFFFFFFFFull;
}
extern "C" int sc_hip_cdt_build_table(float tail, float sigma, uint64_t *cdf, int size)
{
    if (!cdf || size != sc_hip_cdt_table_size(tail, sigma) || size < 2) return fail("cdt_build_table: bad size");
    cdt_tabulate(sigma, cdf, size);
    return SC_HIP_SUCCESS;
}
// the table of a sampler created with BLINDING_SAMPLES (gaussian_cdf.c:586-589): as many entries as the plain sampler's, tabulated at sigma / sqrt(2) (the
// blinding filler subtracts a second sample from every coefficient, sampling.c:168-191) - the scaling is a FLOAT *= long double
extern "C" int sc_hip_cdt_build_table_blinding(float tail, float sigma, uint64_t *cdf, int size)
{
    if (!cdf || size != sc_hip_cdt_table_size(tail, sigma) || size < 2) return fail("cdt_build_table_blinding: bad size");
    const float scaled = (float)((long double)sigma * 0.707106781186547524400844362104849039L);
    cdt_tabulate(scaled, cdf, size);
    return SC_HIP_SUCCESS;
}

// The guide of cdt_guide.h for a monotone table: the first bin layout (coarsest first) no bin of which holds more than 3 switch points,
// else the one with the smallest maximum; the result is checked against the halving search at every switch point and its neighbours
// before it is accepted (g->S = 0: none accepted).
static uint32_t cdt_halving(const std::vector<uint64_t> &cdf, uint64_t x)
{
    uint32_t a = 0;
    for (uint32_t st = (uint32_t)cdf.size() >> 1; st > 0; st >>= 1) { const uint32_t b = a + st; if (cdf[b] < x) a = b; }
    return a;
}
void sc::cdt_build_guide(const std::vector<uint64_t> &cdf, CdtGuide *out, std::vector<uint16_t> *guide, std::vector<uint64_t> *probe)
{
    // the closing run of equal entries below 2^64 - 1
    size_t last = cdf.size();
    while (last > 1 && cdf[last - 1] == ~0ULL) last--;                    // cdf[last - 1]: the largest entry below 2^64 - 1 (index 0 if there is none)
    size_t first = last - 1;
    while (first > 1 && cdf[first - 1] == cdf[last - 1]) first--;
    const bool run = last - 1 > first && first >= 1;
    const uint64_t sat = run ? cdf[first] : ~0ULL;
    const size_t live = run ? first + 1 : cdf.size();                      // entries the guide counts
    static const int layouts[][3] = { { 8, 4, 3 }, { 10, 5, 4 }, { 11, 5, 5 }, { 12, 6, 5 }, { 12, 6, 6 }, { 13, 7, 6 } };       // LB, K, MB
    const size_t size = cdf.size();
    CdtGuide best; memset(&best, 0, sizeof best);
    std::vector<uint16_t> bg;
    for (const auto &L : layouts) {
        CdtGuide g;
        g.lb_shift = 32u - (uint32_t)L[0]; g.K = (uint32_t)L[1]; g.MB = (uint32_t)L[2]; g.tail_thr = ~0u << (32 - L[1]);
        g.lin = 1u << L[0]; g.total = g.lin + ((64u - g.K) << g.MB); g.S = 0;
        std::vector<uint32_t> cnt(g.total + 1, 0);
        for (size_t b = 1; b < live; b++) {
            if (cdf[b] == ~0ULL) continue;                                         // never below any x
            const uint64_t xb = cdf[b] + 1;
            cnt[cdt_guide_bin((uint32_t)(xb >> 32), (uint32_t)xb, g)]++;
        }
        std::vector<uint16_t> gd((g.total + 2) & ~1u, 0);
        uint32_t acc = 0, smax = 0;
        for (uint32_t i = 0; i < g.total; i++) { gd[i] = (uint16_t)acc; acc += cnt[i]; if (cnt[i] > smax) smax = cnt[i]; }
        g.S = smax ? smax : 1;
        g.sat_hi = (uint32_t)(sat >> 32); g.sat_lo = (uint32_t)sat; g.jump = run ? (uint32_t)(last - 1) : 0u;
        if (best.S == 0 || g.S < best.S) { best = g; bg = gd; }
        if (best.S <= 3) break;
    }
    if (best.S > 7) best.S = 0;
    // acceptance test on the host: every switch point and both neighbours, the extremes of x
    if (best.S) {
        auto fast = [&](uint64_t x) {
            const uint32_t b0 = bg[cdt_guide_bin((uint32_t)(x >> 32), (uint32_t)x, best)];
            uint32_t r = b0;
            for (uint32_t j = 1; j <= best.S; j++) { const size_t k = (size_t)b0 + j; r += (k < live && cdf[k] < x) ? 1u : 0u; }
            return (x > sat) ? best.jump : r;
        };
        bool ok = fast(0) == cdt_halving(cdf, 0) && fast(~0ULL) == cdt_halving(cdf, ~0ULL) && fast(1) == cdt_halving(cdf, 1);
        uint64_t lcg = 0x9E3779B97F4A7C15ULL;
        for (size_t b = 0; b < size && ok; b++) {
            for (int d = -1; d <= 1; d++) { const uint64_t x = cdf[b] + (uint64_t)(int64_t)d; if (fast(x) != cdt_halving(cdf, x)) ok = false; }
            lcg = lcg * 6364136223846793005ULL + 1442695040888963407ULL;
            if (fast(lcg) != cdt_halving(cdf, lcg) || fast(~(lcg >> (b & 63))) != cdt_halving(cdf, ~(lcg >> (b & 63)))) ok = false;
        }
        if (!ok) best.S = 0;
    }
    if (!best.S) { bg.assign(2, 0); best.total = 1; }
    *out = best; *guide = bg;
    // what the kernels read: the table itself for the halving search, cut behind the run's first entry for the probes; CDT_PAD entries of 2^64 - 1 behind it
    probe->assign(cdf.begin(), cdf.end());
    if (best.S) for (size_t b = live; b < size; b++) (*probe)[b] = ~0ULL;
    probe->resize(size + CDT_PAD, ~0ULL);
}

// test hook (include/sc_hip_bench.h), host only: the guide, the bin table and the probe table sc_hip_cdt_create_from_table would put on the device for this table
extern "C" int sc_hip_cdt_guide_host(const uint64_t *h_cdf, int size, uint32_t *guide_params, uint16_t *bins, size_t bins_cap, uint64_t *probe, size_t probe_cap, size_t *bins_len)
{
    if (!h_cdf || !guide_params || !bins || !probe || !bins_len) return fail("cdt_guide_host: null argument");
    if (size < 2 || size > 4096 || (size & (size - 1))) return fail("cdt_guide_host: table must have 2..4096 entries, a power of two");
    std::vector<uint64_t> cdf(h_cdf, h_cdf + size), padded;
    std::vector<uint16_t> guide;
    CdtGuide g;
    sc::cdt_build_guide(cdf, &g, &guide, &padded);
    if (guide.size() > bins_cap || padded.size() > probe_cap) return fail("cdt_guide_host: output too small");
    static_assert(sizeof(CdtGuide) == 10 * sizeof(uint32_t), "CdtGuide is ten words");
    memcpy(guide_params, &g, sizeof g);
    memcpy(bins, guide.data(), guide.size() * 2);
    memcpy(probe, padded.data(), padded.size() * 8);
    *bins_len = guide.size();
    return SC_HIP_SUCCESS;
}

extern "C" int sc_hip_cdt_create_from_table(sc_hip_ctx_t *ctx, const uint64_t *h_cdf, int size, sc_hip_cdt_t **out)
{
    if (!ctx || !out || !h_cdf) return fail("cdt_create_from_table: null argument");
    if (size < 2 || size > 4096 || (size & (size - 1))) return fail("cdt_create: table must have 2..4096 entries, a power of two (LDS residency)");
    for (int i = 1; i < size; i++) if (h_cdf[i] < h_cdf[i - 1]) return fail("cdt_create_from_table: table is not monotone");
    sc_hip_cdt *c = new sc_hip_cdt;
    c->ctx = ctx; c->tail = 0.f; c->sigma = 0.f; c->size = size;
    c->cdf.assign(h_cdf, h_cdf + size);
    c->d_cdf = nullptr; c->d_guide = nullptr;
    std::vector<uint16_t> guide;
    std::vector<uint64_t> padded;
    sc::cdt_build_guide(c->cdf, &c->g, &guide, &padded);
    if (hipSetDevice(ctx->device) != hipSuccess || hipMalloc(&c->d_cdf, sizeof(uint64_t) * padded.size()) != hipSuccess ||
        sc::copy_in(c->d_cdf, padded.data(), sizeof(uint64_t) * padded.size()) != hipSuccess ||
        hipMalloc((void **)&c->d_guide, guide.size() * 2) != hipSuccess || sc::copy_in(c->d_guide, guide.data(), guide.size() * 2) != hipSuccess) {
        if (c->d_cdf) (void)sc::wipe_free(c->d_cdf);
        if (c->d_guide) (void)sc::wipe_free(c->d_guide);
        delete c;
        return fail("cdt_create: cannot upload the table");
    }
    *out = c;
    return SC_HIP_SUCCESS;
}

extern "C" int sc_hip_cdt_create(sc_hip_ctx_t *ctx, float tail, float sigma, sc_hip_cdt_t **out)
{
    if (!ctx || !out) return fail("cdt_create: null argument");
    const int size = sc_hip_cdt_table_size(tail, sigma);
    if (size < 2 || size > 4096) return fail("cdt_create: tail*sigma must give a table of 2..4096 entries (LDS residency)");
    std::vector<uint64_t> t((size_t)size);
    if (sc_hip_cdt_build_table(tail, sigma, t.data(), size) != SC_HIP_SUCCESS) return SC_HIP_FAILURE;
    if (sc_hip_cdt_create_from_table(ctx, t.data(), size, out) != SC_HIP_SUCCESS) return SC_HIP_FAILURE;
    (*out)->tail = tail; (*out)->sigma = sigma;
    return SC_HIP_SUCCESS;
}
extern "C" void sc_hip_cdt_destroy(sc_hip_cdt_t *c) { if (!c) return; if (c->d_cdf) (void)sc::wipe_free(c->d_cdf); if (c->d_guide) (void)sc::wipe_free(c->d_guide); delete c; }
extern "C" int sc_hip_cdt_size(const sc_hip_cdt_t *c) { return c ? c->size : 0; }
size_t sc::cdt_lds_bytes(const sc_hip_cdt *c) { return sizeof(uint64_t) * ((size_t)c->size + CDT_PAD) + (((size_t)c->g.total + 2) & ~(size_t)1) * 2; }
extern "C" int sc_hip_cdt_table(const sc_hip_cdt_t *c, uint64_t *h)
{
    if (!c || !h) return fail("cdt_table: null argument");
    memcpy(h, c->cdf.data(), sizeof(uint64_t) * (size_t)c->size);
    return SC_HIP_SUCCESS;
}
extern "C" int sc_hip_cdt_sample(sc_hip_cdt_t *c, size_t count, const uint64_t *d_uniform, int32_t centre, int32_t *d_out, void *stream)
{
    if (!c) return fail("cdt_sample: null sampler");
    if (c->w32) return fail("cdt_sample: the sampler was created at 32-bit precision (sc_hip_cdt32_sample takes its 32-bit words)");
    if (c->hp_words) return fail("cdt_sample: the sampler was created at 128- / 192-bit precision (sc_hip_cdt_hp_sample takes its 2 / 3 words a sample)");
    if (count == 0) return SC_HIP_SUCCESS;
    if (!d_uniform || !d_out) return fail("cdt_sample: null device pointer");
    HIPCHK(hipSetDevice(c->ctx->device));
    size_t blocks = (count + 256 * 32 - 1) / (256 * 32);       // >= 32 samples per lane amortise the table load
    if (blocks == 0) blocks = 1;
    const size_t lds = sc::cdt_lds_bytes(c);
    const size_t per_cu = lds ? (size_t)(160 * 1024) / lds : 8;
    const size_t cap = (size_t)c->ctx->cus * (per_cu > 8 ? 8 : (per_cu ? per_cu : 1));
    if (blocks > cap) blocks = cap;
    hipStream_t s = (hipStream_t)stream;
    const dim3 gr((unsigned)blocks), bl(256);
    switch (c->g.S) {
    case 0: hipLaunchKernelGGL(cdt_halving_kernel, gr, bl, sizeof(uint64_t) * (size_t)c->size, s, c->d_cdf, c->size, d_uniform, d_out, count, centre); break;
    case 1: hipLaunchKernelGGL(cdt_kernel<1>, gr, bl, lds, s, c->d_cdf, c->d_guide, c->size, c->g, d_uniform, d_out, count, centre); break;
    case 2: hipLaunchKernelGGL(cdt_kernel<2>, gr, bl, lds, s, c->d_cdf, c->d_guide, c->size, c->g, d_uniform, d_out, count, centre); break;
    case 3: hipLaunchKernelGGL(cdt_kernel<3>, gr, bl, lds, s, c->d_cdf, c->d_guide, c->size, c->g, d_uniform, d_out, count, centre); break;
    default: hipLaunchKernelGGL(cdt_kernel<0>, gr, bl, lds, s, c->d_cdf, c->d_guide, c->size, c->g, d_uniform, d_out, count, centre); break;
    }
    HIPCHK(hipGetLastError());
    return SC_HIP_SUCCESS;
}

// ---- the sampler at 128- / 192-bit precision over a caller-supplied table (create_sampler(CDF_GAUSSIAN_SAMPLING, SAMPLING_128BIT / _192BIT, ...): gaussian_cdf_sample_128 /
// _192, src/utils/sampling/gaussian_cdf.c:110-184,468-509).  The table itself is MPFR arithmetic (gauss_cdf_create_high_precision, :186-330) and stays the caller's. ----
extern "C" int sc_hip_cdt_hp_create_from_table(sc_hip_ctx_t *ctx, int precision_bits, const uint64_t *h_cdf, int size, sc_hip_cdt_t **out)
{
    if (!ctx || !h_cdf || !out) return fail("cdt_hp_create_from_table: null argument");
    if (precision_bits == 256) return fail("cdt_hp_create_from_table: the reference's 256-bit sampler reads an uninitialised limb (gaussian_cdf.c:519-522): it has no defined result to reproduce");
    if (precision_bits != 128 && precision_bits != 192) return fail("cdt_hp_create_from_table: precision must be 128 or 192 bits (64: sc_hip_cdt_create_from_table, 32: sc_hip_cdt32_*)");
    if (size < 2 || size > 4096 || (size & (size - 1))) return fail("cdt_hp_create_from_table: size must be a power of two in 2..4096");
    const int W = precision_bits / 64;
    std::vector<uint64_t> top((size_t)size);
    for (int i = 0; i < size; i++) top[(size_t)i] = h_cdf[(size_t)i * W + (W - 1)];
    for (int i = 1; i < size; i++) if (top[(size_t)i] < top[(size_t)i - 1]) return fail("cdt_hp_create_from_table: the table's most significant limbs are not monotone");
    if (sc_hip_cdt_create_from_table(ctx, top.data(), size, out) != SC_HIP_SUCCESS) return SC_HIP_FAILURE;
    (*out)->hp_words = W;
    (*out)->hp_low.resize((size_t)size * (W - 1));
    for (int i = 0; i < size; i++) for (int j = 0; j < W - 1; j++) (*out)->hp_low[(size_t)i * (W - 1) + j] = h_cdf[(size_t)i * W + j];
    return SC_HIP_SUCCESS;
}
extern "C" int sc_hip_cdt_hp_precision(const sc_hip_cdt_t *c) { return c ? (c->hp_words ? 64 * c->hp_words : c->w32 ? 32 : 64) : 0; }
extern "C" int sc_hip_cdt_hp_table(const sc_hip_cdt_t *c, uint64_t *h)
{
    if (!c || !h) return fail("cdt_hp_table: null argument");
    if (!c->hp_words) return fail("cdt_hp_table: not a 128- / 192-bit sampler");
    const int W = c->hp_words;
    for (int i = 0; i < c->size; i++) {
        for (int j = 0; j < W - 1; j++) h[(size_t)i * W + j] = c->hp_low[(size_t)i * (W - 1) + j];
        h[(size_t)i * W + (W - 1)] = c->cdf[(size_t)i];
    }
    return SC_HIP_SUCCESS;
}
extern "C" int sc_hip_cdt_hp_sample(sc_hip_cdt_t *c, size_t count, const uint64_t *d_uniform, int32_t centre, int32_t *d_out, void *stream)
{
    if (!c) return fail("cdt_hp_sample: null sampler");
    if (!c->hp_words) return fail("cdt_hp_sample: the sampler was not created by sc_hip_cdt_hp_create_from_table");
    if (count == 0) return SC_HIP_SUCCESS;
    if (!d_uniform || !d_out) return fail("cdt_hp_sample: null device pointer");
    HIPCHK(hipSetDevice(c->ctx->device));
    size_t blocks = (count + 256 * 32 - 1) / (256 * 32);
    if (blocks == 0) blocks = 1;
    const size_t lds = sc::cdt_lds_bytes(c);
    const size_t per_cu = lds ? (size_t)(160 * 1024) / lds : 8;
    const size_t cap = (size_t)c->ctx->cus * (per_cu > 8 ? 8 : (per_cu ? per_cu : 1));
    if (blocks > cap) blocks = cap;
    hipStream_t s = (hipStream_t)stream;
    const dim3 gr((unsigned)blocks), bl(256);
    switch (c->g.S) {
    case 0: hipLaunchKernelGGL((cdt_hp_kernel<0, true>), gr, bl, lds, s, c->d_cdf, c->d_guide, c->size, c->g, d_uniform, c->hp_words, d_out, count, centre); break;
    case 1: hipLaunchKernelGGL((cdt_hp_kernel<1, false>), gr, bl, lds, s, c->d_cdf, c->d_guide, c->size, c->g, d_uniform, c->hp_words, d_out, count, centre); break;
    case 2: hipLaunchKernelGGL((cdt_hp_kernel<2, false>), gr, bl, lds, s, c->d_cdf, c->d_guide, c->size, c->g, d_uniform, c->hp_words, d_out, count, centre); break;
    case 3: hipLaunchKernelGGL((cdt_hp_kernel<3, false>), gr, bl, lds, s, c->d_cdf, c->d_guide, c->size, c->g, d_uniform, c->hp_words, d_out, count, centre); break;
    default: hipLaunchKernelGGL((cdt_hp_kernel<0, false>), gr, bl, lds, s, c->d_cdf, c->d_guide, c->size, c->g, d_uniform, c->hp_words, d_out, count, centre); break;
    }
    HIPCHK(hipGetLastError());
    return SC_HIP_SUCCESS;
}

// ---- the sampler at 32-bit precision (create_sampler(CDF_GAUSSIAN_SAMPLING, SAMPLING_32BIT, ...): gaussian_cdf_create_32 / gaussian_cdf_sample_32,
// src/utils/sampling/gaussian_cdf.c:683-774) ----
extern "C" int sc_hip_cdt32_build_table(float tail, float sigma, uint32_t *cdf, int size)
{
    if (!cdf || size != sc_hip_cdt_table_size(tail, sigma) || size < 2) return fail("cdt32_build_table: bad size");
    // the recurrence of gaussian_cdf.c:719-735 with its types: s, d, e are FLOAT (float); the constants and expl() are wider and every assignment rounds
    const float d = (float)(1.12837916709551257390 * 0.70710678118654752440 * 4294967296.0 / sigma);      // M_2_SQRTPI * M_SQRT1_2 * 2^32 / sigma: double arithmetic
    const float e = (float)(-0.5L / (sigma * sigma));
    float s = (float)(0.5L * d);
    int i;
    cdf[0] = 0;
    for (i = 1; i < size - 1; i++) {
        cdf[i] = (uint32_t)s;
        if (cdf[i] == 0) break;
        s = (float)(s + d * expl(e * ((float)(i * i))));
    }
    for (; i < size; i++) cdf[i] = 0xFFFFFFFFu;
    return SC_HIP_SUCCESS;
}
extern "C" int sc_hip_cdt32_create_from_table(sc_hip_ctx_t *ctx, const uint32_t *h_cdf, int size, sc_hip_cdt_t **out)
{
    if (!ctx || !h_cdf || !out) return fail("cdt32_create_from_table: null argument");
    if (size < 2 || size > 4096 || (size & (size - 1))) return fail("cdt32_create_from_table: size must be a power of two in 2..4096");
    std::vector<uint64_t> t((size_t)size);
    for (int i = 0; i < size; i++) t[(size_t)i] = ((uint64_t)h_cdf[i] << 32) | 0xFFFFFFFFull;
    t[0] = 0;                                                          // (never compared: the search starts above it)
    if (sc_hip_cdt_create_from_table(ctx, t.data(), size, out) != SC_HIP_SUCCESS) return SC_HIP_FAILURE;
    (*out)->w32 = true;
    return SC_HIP_SUCCESS;
}
extern "C" int sc_hip_cdt32_create(sc_hip_ctx_t *ctx, float tail, float sigma, sc_hip_cdt_t **out)
{
    if (!ctx || !out) return fail("cdt32_create: null argument");
    const int size = sc_hip_cdt_table_size(tail, sigma);
    if (size < 2 || size > 4096) return fail("cdt32_create: tail*sigma must give a table of 2..4096 entries (LDS residency)");
    std::vector<uint32_t> t((size_t)size);
    if (sc_hip_cdt32_build_table(tail, sigma, t.data(), size) != SC_HIP_SUCCESS) return SC_HIP_FAILURE;
    if (sc_hip_cdt32_create_from_table(ctx, t.data(), size, out) != SC_HIP_SUCCESS) return SC_HIP_FAILURE;
    (*out)->tail = tail; (*out)->sigma = sigma;
    return SC_HIP_SUCCESS;
}
extern "C" int sc_hip_cdt32_table(const sc_hip_cdt_t *c, uint32_t *h)
{
    if (!c || !h) return fail("cdt32_table: null argument");
    if (!c->w32) return fail("cdt32_table: a 64-bit sampler");
    for (int i = 0; i < c->size; i++) h[i] = i ? (uint32_t)(c->cdf[(size_t)i] >> 32) : 0u;
    return SC_HIP_SUCCESS;
}
extern "C" int sc_hip_cdt32_sample(sc_hip_cdt_t *c, size_t count, const uint32_t *d_uniform, int32_t centre, int32_t *d_out, void *stream)
{
    if (!c) return fail("cdt32_sample: null sampler");
    if (!c->w32) return fail("cdt32_sample: the sampler was created at 64-bit precision (sc_hip_cdt_sample takes its 64-bit words)");
    if (count == 0) return SC_HIP_SUCCESS;
    if (!d_uniform || !d_out) return fail("cdt32_sample: null device pointer");
    HIPCHK(hipSetDevice(c->ctx->device));
    size_t blocks = (count + 256 * 32 - 1) / (256 * 32);
    if (blocks == 0) blocks = 1;
    const size_t lds = sc::cdt_lds_bytes(c);
    const size_t per_cu = lds ? (size_t)(160 * 1024) / lds : 8;
    const size_t cap = (size_t)c->ctx->cus * (per_cu > 8 ? 8 : (per_cu ? per_cu : 1));
    if (blocks > cap) blocks = cap;
    hipStream_t s = (hipStream_t)stream;
    const dim3 gr((unsigned)blocks), bl(256);
    switch (c->g.S) {
    case 0: hipLaunchKernelGGL((cdt32_kernel<0, true>), gr, bl, lds, s, c->d_cdf, c->d_guide, c->size, c->g, d_uniform, d_out, count, centre); break;
    case 1: hipLaunchKernelGGL((cdt32_kernel<1, false>), gr, bl, lds, s, c->d_cdf, c->d_guide, c->size, c->g, d_uniform, d_out, count, centre); break;
    case 2: hipLaunchKernelGGL((cdt32_kernel<2, false>), gr, bl, lds, s, c->d_cdf, c->d_guide, c->size, c->g, d_uniform, d_out, count, centre); break;
    case 3: hipLaunchKernelGGL((cdt32_kernel<3, false>), gr, bl, lds, s, c->d_cdf, c->d_guide, c->size, c->g, d_uniform, d_out, count, centre); break;
    default: hipLaunchKernelGGL((cdt32_kernel<0, false>), gr, bl, lds, s, c->d_cdf, c->d_guide, c->size, c->g, d_uniform, d_out, count, centre); break;
    }
    HIPCHK(hipGetLastError());
    return SC_HIP_SUCCESS;
}

// ---- host-buffer convenience ---------------------------------------------------------------

// Host-buffer forms of the transforms.  The caller's buffers are pinned for the call (hipHostRegister) and the batch moves in chunks through a
// ring of device buffers on three streams, so the upload of chunk c+1, the kernel of chunk c and the download of chunk c-1 overlap: the rate
// is PCIe's (8n bytes up and 4n down per product), not copy + kernel + copy in sequence.  If the buffers cannot be pinned the same loop runs
// with the runtime's staged pageable copies.
static int host_roundtrip(sc_hip_plan_t *p, int op, size_t batch, const int32_t *h0, const int32_t *h1, int32_t *hout)
{
    if (!p) return fail("host op: null plan");
    if (batch == 0) return SC_HIP_SUCCESS;
    if (!h0 || !hout) return fail("host op: null host pointer");
    HIPCHK(hipSetDevice(p->ctx->device));
    constexpr int NB = 3;
    const size_t pb = (size_t)p->n * sizeof(int32_t);
    size_t chunk = ((size_t)8 << 20) / pb;                         // 8 MiB per operand per chunk
    if (chunk > batch) chunk = batch;
    const size_t nchunks = (batch + chunk - 1) / chunk, bytes = batch * pb;
    const int nops = h1 ? 2 : 1;
    // whole interior pages of the caller's buffers are registered for the call, overlapping buffers once (sc::HostPins); a buffer the multi-device
    // driver pinned for its workers or one the caller allocated pinned moves direct as it is; everything else in pieces the runtime stages
    sc::HostPins own;
    const sc::HostPins *use = sc::t_driver_pins;                   // a worker of the multi-device driver: the driver's registrations, none of its own
    // (a read-only attribute query, in the driver's workers too: the driver's registration of a buffer the caller allocated pinned fails silently, and such a
    // buffer would otherwise move in staged 1 MiB pieces - ADVICE r03)
    const bool cp[3] = { sc::caller_allocated_pinned(h0), h1 && sc::caller_allocated_pinned(h1), sc::caller_allocated_pinned(hout) };
    if (!use) {
        if (!cp[0]) own.add(h0, bytes);
        if (h1 && !cp[1]) own.add(h1, bytes);
        if (!cp[2]) own.add(hout, bytes);
        own.commit();
        use = &own;
    }
    const sc::HostPins &pins = *use;
    hipStream_t st[NB] = { nullptr, nullptr, nullptr };
    int32_t *dev = nullptr;
    int rc = SC_HIP_SUCCESS;
    const size_t ring = (size_t)(nchunks < NB ? nchunks : NB);
    if (hipMalloc(&dev, ring * nops * chunk * pb) != hipSuccess) rc = fail("host op: device allocation failed");
    for (size_t i = 0; i < ring && rc == SC_HIP_SUCCESS; i++) if (hipStreamCreateWithFlags(&st[i], hipStreamNonBlocking) != hipSuccess) rc = fail("host op: stream creation failed");
    for (size_t c = 0; c < nchunks && rc == SC_HIP_SUCCESS; c++) {
        const size_t slot = c % ring, off = c * chunk, cnt = (off + chunk <= batch) ? chunk : batch - off;
        int32_t *d0 = dev + slot * nops * chunk * (size_t)p->n, *d1 = h1 ? d0 + chunk * (size_t)p->n : nullptr;
        hipStream_t s = st[slot];
        if (sc::copy_in_pinned(d0, h0 + off * (size_t)p->n, cnt * pb, s, &pins, cp[0]) != hipSuccess ||
            (h1 && sc::copy_in_pinned(d1, h1 + off * (size_t)p->n, cnt * pb, s, &pins, cp[1]) != hipSuccess)) { rc = fail("host op: H2D copy failed"); break; }
        rc = run_ntt(p, op, cnt, d0, d1, d0, s);
        if (rc == SC_HIP_SUCCESS && sc::copy_out_pinned(hout + off * (size_t)p->n, d0, cnt * pb, s, &pins, cp[2]) != hipSuccess) rc = fail("host op: D2H copy failed");
    }
    for (size_t i = 0; i < ring; i++) if (st[i]) { if (hipStreamSynchronize(st[i]) != hipSuccess && rc == SC_HIP_SUCCESS) rc = fail("host op: stream failed"); (void)hipStreamDestroy(st[i]); }
    if (dev) (void)sc::wipe_free(dev);
    return rc;                                                      // (the registrations end with `pins`)
}
extern "C" int sc_hip_ntt_fwd_host(sc_hip_plan_t *p, size_t batch, const int32_t *h_in, int32_t *h_out)
{ return host_roundtrip(p, OP_FWD, batch, h_in, nullptr, h_out); }
extern "C" int sc_hip_ntt_inv_host(sc_hip_plan_t *p, size_t batch, const int32_t *h_in, int32_t *h_out)
{ return host_roundtrip(p, OP_INV, batch, h_in, nullptr, h_out); }
extern "C" int sc_hip_polymul_ntt_host(sc_hip_plan_t *p, size_t batch, const int32_t *h_a, const int32_t *h_bhat, int32_t *h_out)
{ if (!h_bhat) return fail("polymul_ntt_host: null host pointer"); return host_roundtrip(p, OP_MULNTT, batch, h_a, h_bhat, h_out); }
extern "C" int sc_hip_cdt_sample_host(sc_hip_cdt_t *c, size_t count, const uint64_t *h_uniform, int32_t centre, int32_t *h_out)
{
    if (!c) return fail("cdt_sample_host: null sampler");
    if (count == 0) return SC_HIP_SUCCESS;
    if (!h_uniform || !h_out) return fail("cdt_sample_host: null host pointer");
    HIPCHK(hipSetDevice(c->ctx->device));
    uint64_t *dx = nullptr; int32_t *dv = nullptr;
    HIPCHK(hipMalloc(&dx, count * 8));
    int rc = SC_HIP_SUCCESS;
    if (hipMalloc(&dv, count * 4) != hipSuccess || sc::copy_in(dx, h_uniform, count * 8) != hipSuccess) rc = fail("cdt_sample_host: H2D failed");
    if (rc == SC_HIP_SUCCESS) rc = sc_hip_cdt_sample(c, count, dx, centre, dv, nullptr);
    if (rc == SC_HIP_SUCCESS && sc::copy_out(h_out, dv, count * 4) != hipSuccess) rc = fail("cdt_sample_host: D2H failed");
    (void)sc::wipe_free(dx);
    if (dv) (void)sc::wipe_free(dv);
    return rc;
}

extern "C" int sc_hip_cdt_hp_sample_host(sc_hip_cdt_t *c, size_t count, const uint64_t *h_uniform, int32_t centre, int32_t *h_out)
{
    if (!c) return fail("cdt_hp_sample_host: null sampler");
    if (!c->hp_words) return fail("cdt_hp_sample_host: the sampler was not created by sc_hip_cdt_hp_create_from_table");
    if (count == 0) return SC_HIP_SUCCESS;
    if (!h_uniform || !h_out) return fail("cdt_hp_sample_host: null host pointer");
    HIPCHK(hipSetDevice(c->ctx->device));
    const size_t xb = count * 8 * (size_t)c->hp_words;
    uint64_t *dx = nullptr; int32_t *dv = nullptr;
    HIPCHK(hipMalloc(&dx, xb));
    int rc = SC_HIP_SUCCESS;
    if (hipMalloc(&dv, count * 4) != hipSuccess || sc::copy_in(dx, h_uniform, xb) != hipSuccess) rc = fail("cdt_hp_sample_host: H2D failed");
    if (rc == SC_HIP_SUCCESS) rc = sc_hip_cdt_hp_sample(c, count, dx, centre, dv, nullptr);
    if (rc == SC_HIP_SUCCESS && sc::copy_out(h_out, dv, count * 4) != hipSuccess) rc = fail("cdt_hp_sample_host: D2H failed");
    (void)sc::wipe_free(dx);
    if (dv) (void)sc::wipe_free(dv);
    return rc;
}

static int host_ew(sc_hip_plan_t *p, int which, size_t len, const int32_t *h0, const int32_t *h1, int32_t *hout)
{
    if (!p) return fail("host op: null plan");
    if (len == 0) return SC_HIP_SUCCESS;
    if (!h0 || !hout || (which == EW_POINTWISE && !h1)) return fail("host op: null host pointer");
    HIPCHK(hipSetDevice(p->ctx->device));
    const size_t bytes = len * sizeof(int32_t);
    int32_t *d0 = nullptr, *d1 = nullptr;
    HIPCHK(hipMalloc(&d0, bytes));
    int rc = SC_HIP_SUCCESS;
    if (sc::copy_in(d0, h0, bytes) != hipSuccess) rc = fail("host op: H2D copy failed");
    if (rc == SC_HIP_SUCCESS && which == EW_POINTWISE) {
        if (hipMalloc(&d1, bytes) != hipSuccess || sc::copy_in(d1, h1, bytes) != hipSuccess) rc = fail("host op: H2D copy failed");
    }
    if (rc == SC_HIP_SUCCESS) {
        if (which == EW_POINTWISE) rc = sc_hip_pointwise(p, len, d0, d1, d0, nullptr);
        else if (which == EW_NORMALIZE) rc = sc_hip_normalize(p, len, d0, d0, nullptr);
        else rc = sc_hip_center(p, len, d0, d0, nullptr);
    }
    if (rc == SC_HIP_SUCCESS && sc::copy_out(hout, d0, bytes) != hipSuccess) rc = fail("host op: D2H copy failed");
    (void)sc::wipe_free(d0);
    if (d1) (void)sc::wipe_free(d1);
    return rc;
}
extern "C" int sc_hip_pointwise_host(sc_hip_plan_t *p, size_t len, const int32_t *h_a, const int32_t *h_b, int32_t *h_out)
{ return host_ew(p, EW_POINTWISE, len, h_a, h_b, h_out); }
extern "C" int sc_hip_normalize_host(sc_hip_plan_t *p, size_t len, const int32_t *h_in, int32_t *h_out)
{ return host_ew(p, EW_NORMALIZE, len, h_in, nullptr, h_out); }
extern "C" int sc_hip_center_host(sc_hip_plan_t *p, size_t len, const int32_t *h_in, int32_t *h_out)
{ return host_ew(p, EW_CENTER, len, h_in, nullptr, h_out); }

// ---- a launch-bound loop as one HIP graph ------------------------------------------------------

// `steps` back-to-back fused products (the same call as sc_hip_polymul_ntt, on the same buffers); 64 steps or more are captured once into a HIP graph and
// replayed.  The executable graph is cached in the plan for
// the (buffers, batch, steps) it was captured with.  `stream` must be a real stream (stream capture does not work on the null stream); if anything
// in the capture fails the plain loop runs instead - same result either way.
extern "C" int sc_hip_polymul_ntt_steps(sc_hip_plan_t *p, size_t batch, const int32_t *d_a, const int32_t *d_bhat, int32_t *d_out, int steps, void *stream)
{
    if (!p) return fail("polymul_ntt_steps: null plan");
    if (steps <= 0) return SC_HIP_SUCCESS;
    HIPCHK(hipSetDevice(p->ctx->device));
    hipStream_t s = (hipStream_t)stream;
    if (d_out == d_a || d_out == d_bhat) return fail("polymul_ntt_steps: the output must not alias an input (the steps would feed each other)");
    // A replay pays a fixed start-up (the runtime turns the graph's nodes into packets at launch time): measured at batch 65536, 20 steps take 762 us
    // replayed against 664 us as plain launches, 200 steps 6196 us either way (tools/exp_ntt_stream.hip, profiles/r04_ntt_stream_variants.txt) - so short
    // loops run as plain launches (the kernel is 31 us, a launch costs the host 2 us: the queue stays ahead) and only long ones are replayed
    const bool use_graph = s && steps >= 64;
    const bool same = p->g_exec && p->g_a == d_a && p->g_b == d_bhat && p->g_out == d_out && p->g_batch == batch && p->g_steps == steps;
    if (use_graph && !same) {
        if (p->g_exec) { (void)hipGraphExecDestroy((hipGraphExec_t)p->g_exec); p->g_exec = nullptr; }
        if (run_ntt(p, OP_MULNTT, 0, d_a, d_bhat, d_out, stream, false, true) != SC_HIP_SUCCESS) return SC_HIP_FAILURE;   // first-use initialisation (occupancy query) outside the capture, no launch
        hipGraph_t graph = nullptr;
        if (hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal) == hipSuccess) {
            int rc = SC_HIP_SUCCESS;
            for (int i = 0; i < steps && rc == SC_HIP_SUCCESS; i++) rc = run_ntt(p, OP_MULNTT, batch, d_a, d_bhat, d_out, stream);
            const hipError_t e = hipStreamEndCapture(s, &graph);
            hipGraphExec_t exec = nullptr;
            if (rc == SC_HIP_SUCCESS && e == hipSuccess && graph && hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0) == hipSuccess) {
                p->g_exec = exec; p->g_a = d_a; p->g_b = d_bhat; p->g_out = d_out; p->g_batch = batch; p->g_steps = steps;
            }
            if (graph) (void)hipGraphDestroy(graph);
        }
        (void)hipGetLastError();
    }
    if (use_graph && p->g_exec && p->g_a == d_a && p->g_b == d_bhat && p->g_out == d_out && p->g_batch == batch && p->g_steps == steps) {
        if (hipGraphLaunch((hipGraphExec_t)p->g_exec, s) == hipSuccess) return SC_HIP_SUCCESS;
        (void)hipGetLastError();
    }
    for (int i = 0; i < steps; i++) if (run_ntt(p, OP_MULNTT, batch, d_a, d_bhat, d_out, stream) != SC_HIP_SUCCESS) return SC_HIP_FAILURE;
    return SC_HIP_SUCCESS;
}

// ---- timing --------------------------------------------------------------------------------

extern "C" int sc_hip_time_op(sc_hip_plan_t *p, int op, size_t batch, const int32_t *d_a, const int32_t *d_b,
                              int32_t *d_out, void *stream, int iters, float *ms)
{
    if (!p || !ms || iters <= 0) return fail("time_op: bad argument");
    if (op < 0 || op > 3) return fail("time_op: op must be 0..3");
    HIPCHK(hipSetDevice(p->ctx->device));
    hipEvent_t e0, e1;
    HIPCHK(hipEventCreate(&e0));
    HIPCHK(hipEventCreate(&e1));
    int rc = SC_HIP_SUCCESS;
    (void)hipEventRecord(e0, (hipStream_t)stream);
    for (int i = 0; i < iters && rc == SC_HIP_SUCCESS; i++) rc = run_ntt(p, op, batch, d_a, d_b, d_out, stream);
    (void)hipEventRecord(e1, (hipStream_t)stream);
    if (rc == SC_HIP_SUCCESS && hipEventSynchronize(e1) != hipSuccess) rc = fail("time_op: event synchronize failed");
    float t = 0.f;
    if (rc == SC_HIP_SUCCESS && hipEventElapsedTime(&t, e0, e1) != hipSuccess) rc = fail("time_op: elapsed time failed");
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    *ms = t / (float)iters;
    return rc;
}
