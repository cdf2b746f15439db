// ntt_kernel.h - the batched NTT kernel itself: ntt_kernel<L, MODE, LAZY, OP, STREAM> over ntt_core.h's register / LDS machinery (device side, gfx950).
// The functions behind the reference's fwd_ntt_32_* / inv_ntt_32_* and the fused products built on them (src/utils/arith/ntt_template.c.in:1541-1663).
// A header so that the kernel's own text can also be compiled as host code over tests/emu/wave_emu.h and run lane by lane under the sanitizers the GPU
// pool cannot offer (tests/test_emu_ntt.py); sc_hip.hip includes it once and holds the host side.
#pragma once
#include "ntt_core.h"
#include <type_traits>

using namespace sc;

// ------------------------------------------------------------------------------------------
// kernels
// ------------------------------------------------------------------------------------------

enum { OP_FWD = 0, OP_INV = 1, OP_MULNTT = 2, OP_MUL = 3, OP_MULNTT2 = 4 };      // OP_MULNTT2: two NTT-domain operands for the whole batch, two outputs, ONE forward transform (ring-TESLA's v1 = a1 y, v2 = a2 y)

struct NttArgs {
    const int32_t *in0;
    const int32_t *in1;
    const int32_t *in2;           // OP_MULNTT2: the second NTT-domain operand (both are broadcast) ...
    int32_t *out2;                // ... and where its product goes
    int32_t *out;
    int32_t *sink;                // 4 KB nobody reads: where the lanes of a ragged last group that lie past the batch store (plain form)
    const uint32_t *zf, *zfp;     // forward twiddles zeta[k] = g^bitrev(k) and Shoup companions
    const uint32_t *zi, *zip;     // inverse twiddles zeta[k]^-1 and companions
    unsigned long long batch;
    const uint32_t *batch_dev;    // non-null: the batch where only the device knows it (a count a previous kernel left); `batch` is then the host's upper bound
    unsigned in1_bcast;           // != 0: every polynomial is multiplied by the SAME second operand (a public key)
    ModK k;
    uint32_t c0, c0p, c1, c1p;    // inverse output scale (n^-1 [* 2^32]) and zeta[1]^-1 * scale
};

typedef int v4i __attribute__((ext_vector_type(4)));      // a native 16-byte vector: what the non-temporal builtins take

__host__ __device__ constexpr int brv4(int r) { return ((r & 1) << 3) | ((r & 2) << 1) | ((r & 4) >> 1) | ((r & 8) >> 3); }

// One wavefront handles PW polynomials at a time and loops over the batch (persistent waves:
// the per-lane twiddles are loaded once).  256 threads = 4 independent waves per workgroup;
// LDS: 4 KB per wave (one polynomial image per polynomial in flight).
//
// The loop is software-pipelined: the 16 input words of the NEXT group are requested before the
// current group's butterflies start, so every resident wave always has 4 KB of HBM reads in
// flight (the kernel is otherwise a load -> compute -> store chain per wave and runs at the
// memory latency, not the memory bandwidth).  Groups are disjoint in memory, so running ahead
// is safe when out aliases in.
template <int L, int OP>
struct Io {
    using Y = Lay<L>;
    // element offset of register r for this lane: coefficient-domain arrays are read/written in
    // round-0 layout (j = r*LP + lp); NTT-domain arrays (natural order in memory) in the last
    // round's layout with the bit reversal applied (position j = lp*16+r  <->  index bitrev_L(j))
    static __device__ __forceinline__ int off_coef(int r, int lp) { return r * Y::LP + lp; }
    static __device__ __forceinline__ int off_ntt(int r, int blp) { return brv4(r) * Y::LP + blp; }
};

// STREAM (round 4): how the kernel touches memory when the operands cannot be resident in the 256 MiB Infinity Cache anyway (the host picks it by working
// set).  Measured on n = 256 over 3 x 1 GiB (tools/exp_ntt_stream.hip, profiles/r04_ntt_stream_variants.txt): what pays is 16 bytes per lane AND the
// non-temporal policy together - the wavefront's 4 KB group moves as four 1 KB instructions (one whole polynomial each at n = 256) and is transposed into /
// out of the register layout through the wave's LDS image (padded by 16 words per polynomial at n = 256 so that the four polynomials of a wavefront fall
// on different banks) - fwd 0.67 -> 0.70, fused 0.67 -> 0.72 of 8 TB/s; either half alone loses (4-byte nt accesses 0.62, 16-byte default-policy 0.65), and on a
// working set that replays out of the Infinity Cache the non-temporal form gives the cache hits away (0.78 -> 0.70): hence a variant, not a rewrite.
// The second operand stays on 4-byte loads (non-temporal when streaming).  STREAM kernels take whole groups only; the host runs a ragged tail on the plain one.
template <int L, int MODE, bool LAZY, int OP, bool STREAM>
__global__ __launch_bounds__(256) void ntt_kernel(NttArgs a)
{
    using Y = Lay<L>;
    constexpr int N = Y::N, LP = Y::LP, PW = Y::PW;
    constexpr int NP = (STREAM && L == 8) ? N + 16 : N;            // words per polynomial in the transposition image
    constexpr int WW = STREAM ? PW * NP : 1024;                   // LDS words per wavefront
    __shared__ __attribute__((aligned(16))) uint32_t lds[4 * WW];

    const int lane = threadIdx.x & 63;
    const int wid = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));      // wave-uniform: the group walk below is scalar arithmetic
    const int pw = lane / LP, lp = lane % LP;
    uint32_t *lds_wave = lds + wid * WW;
    uint32_t *lds_poly = lds_wave + pw * N;
    const int blp = (int)(__builtin_bitreverse32((uint32_t)lp) >> (32 - (L - 4)));   // bitrev_{L-4}(lp)
    const ModK k = a.k;

    TwSet<L> twf, twi;
    if (OP != OP_INV) twf.load(a.zf, a.zfp, lp);
    if (OP != OP_FWD) twi.load(a.zi, a.zip, lp);

    if (a.batch_dev) { const unsigned long long bd = *a.batch_dev; if (bd < a.batch) a.batch = bd; }
    const unsigned long long groups = (a.batch + PW - 1) / PW;
    const unsigned long long gstride = (unsigned long long)gridDim.x * 4;
    unsigned long long g = (unsigned long long)blockIdx.x * 4 + wid;
    if (g >= groups) return;

    // in0 is read in the layout the first transform wants; image index of register r of this lane in either domain
    const int off0 = (OP == OP_INV) ? Io<L, OP>::off_ntt(0, blp) : Io<L, OP>::off_coef(0, lp);
    auto img_coef = [&](int r) { return pw * NP + r * LP + lp; };
    auto img_ntt = [&](int r) { return pw * NP + brv4(r) * LP + blp; };
    int32_t nx[16];                                      // raw words of the group about to be processed
    v4i wx[4];                                           // STREAM: the same as 4 x 16 bytes per lane, memory order
    // loads are unconditional: a group past the end of the batch reads group 0 and is never stored (a guarded load is an exec-mask save, a branch
    // and a restore per instruction: 80 scalar instructions per trip of the fused kernel)
    auto request_in0 = [&](unsigned long long gg) {
        if constexpr (STREAM) {
            const v4i *src = reinterpret_cast<const v4i *>(a.in0 + (size_t)(gg < groups ? gg : 0) * (PW * N)) + lane;
#pragma unroll
            for (int c = 0; c < 4; c++) wx[c] = __builtin_nontemporal_load(src + c * 64);
        } else {
            const unsigned long long poly = gg * PW + pw;
            const int32_t *src = a.in0 + (size_t)((gg < groups && poly < a.batch) ? poly : 0) * N + off0;
#pragma unroll
            for (int r = 0; r < 16; r++) nx[r] = src[(OP == OP_INV ? brv4(r) : r) * LP];
        }
    };
    // out: registers -> memory.  NTT = the NTT-domain order (forward output), else coefficient order.  Stores are unguarded: one guarded store is an exec-mask
    // save, a branch and a restore, and sixteen of them split the store burst into sixteen islands (the forward kernel ran 456 us against 382 us over 2 GiB
    // with them) - the lanes of a ragged last group that lie past the batch store into a 4 KB sink instead
    auto store_out = [&](uint32_t (&x)[16], size_t base, bool valid, auto ntt_order, int32_t *outp) {
        constexpr bool NTT = decltype(ntt_order)::value;
        if constexpr (STREAM) {
#pragma unroll
            for (int r = 0; r < 16; r++) lds_wave[NTT ? img_ntt(r) : img_coef(r)] = x[r];
            wave_lds_fence();
            v4i *dst = reinterpret_cast<v4i *>(outp + (base - (size_t)pw * N)) + lane;
#pragma unroll
            for (int c = 0; c < 4; c++) {
                const int cw = (c * 256) / N * NP + (c * 256) % N;                // word c * 256 of the group, in the padded image
                const v4i v = *reinterpret_cast<const v4i *>(lds_wave + cw + 4 * lane);
                __builtin_nontemporal_store(v, dst + c * 64);
            }
            wave_lds_fence();
        } else {
            int32_t *dst = (valid ? outp + base : a.sink + pw * N) + (NTT ? Io<L, OP>::off_ntt(0, blp) : Io<L, OP>::off_coef(0, lp));
#pragma unroll
            for (int r = 0; r < 16; r++) dst[(NTT ? brv4(r) : r) * LP] = (int32_t)x[r];
        }
    };
    request_in0(g);

    for (; g < groups; g += gstride) {
        const unsigned long long poly = g * PW + pw;
        const bool valid = poly < a.batch;
        const size_t base = (size_t)poly * N;
        uint32_t x[16];
        if constexpr (STREAM) {
            // the group's 4 KB as it lies in memory -> image -> this lane's 16 positions
#pragma unroll
            for (int c = 0; c < 4; c++) *reinterpret_cast<v4i *>(lds_wave + (c * 256) / N * NP + (c * 256) % N + 4 * lane) = wx[c];
            wave_lds_fence();
#pragma unroll
            for (int r = 0; r < 16; r++) nx[r] = (int32_t)lds_wave[OP == OP_INV ? img_ntt(r) : img_coef(r)];
            wave_lds_fence();
        }
#pragma unroll
        for (int r = 0; r < 16; r++) x[r] = (MODE == MODE_SMAD) ? reduce_signed(nx[r], k) : (MODE == MODE_S32) ? reduce_s32(nx[r], k) : (MODE == MODE_MAD) ? reduce_i32_small(nx[r], k) : reduce_i32(nx[r], k);

        // second operand of this group: requested now, consumed after the forward transform
        int32_t nb[16];
        constexpr bool PW_NTT = OP == OP_MULNTT || OP == OP_MULNTT2;      // the second operand is already in the NTT domain
        if (PW_NTT || OP == OP_MUL) {
            const int32_t *srcb = a.in1 + ((a.in1_bcast || !valid) ? 0 : base) + (PW_NTT ? Io<L, OP>::off_ntt(0, blp) : Io<L, OP>::off_coef(0, lp));
#pragma unroll
            for (int r = 0; r < 16; r++) {
                const int32_t *q = srcb + (PW_NTT ? brv4(r) : r) * LP;
                nb[r] = STREAM ? __builtin_nontemporal_load(q) : *q;
            }
        }
        // first operand of the next group (run-ahead)
        if (OP == OP_FWD || OP == OP_INV) request_in0(g + gstride);

        if (OP != OP_INV) fwd_core<L, MODE, LAZY>(x, twf, lds_poly, lp, k);

        if (OP == OP_FWD) {
#pragma unroll
            for (int r = 0; r < 16; r++) {
                uint32_t v = x[r];
                if (MODE == MODE_SMAD) {          // |v| < 2q(L+1) <= 22q: + 32q, unsigned Barrett step to [0, 2q), one conditional subtraction
                    v += k.q << 5;
                    v += (uint32_t)mul_i24((int32_t)__umulhi(v, k.m32), -(int32_t)k.q);
                    v = csub(v, k.q);
                } else if (MODE == MODE_S32) v = canon_s32(v, k);
                else {
                    if (LAZY) v = Mod<MODE>::mulc(v, 1u, k.one_p, k);
                    v = csub(v, k.q);
                }
                x[r] = v;
            }
            store_out(x, base, valid, std::true_type{}, a.out);
            continue;
        }

        if (OP == OP_MULNTT2) {
            // NTT(y) is kept: first product -> inverse -> out, second product (its operand requested before the first inverse) -> inverse -> out2
            uint32_t f[16];
            int32_t nb2[16];
            const int32_t *src2 = a.in2 + Io<L, OP>::off_ntt(0, blp);
#pragma unroll
            for (int r = 0; r < 16; r++) {
                f[r] = (MODE == MODE_SMAD && L > 8) ? reduce_signed((int32_t)x[r], k) : x[r];
                nb2[r] = src2[brv4(r) * LP];
            }
#pragma unroll
            for (int r = 0; r < 16; r++) {
                if (MODE == MODE_SMAD) x[r] = mulmod_ssmall(f[r], reduce_signed(nb[r], k), k);
                else if (MODE == MODE_S32) x[r] = montmul_s32(f[r], reduce_s32(nb[r], k), k);
                else if (MODE == MODE_MAD) x[r] = mulmod_small(f[r], csub(reduce_i32_small(nb[r], k), k.q), k);
                else x[r] = montmul(f[r], reduce_i32(nb[r], k), k);
            }
            inv_core<L, MODE>(x, twi, lds_poly, lp, k, a.c0, a.c0p, a.c1, a.c1p);
            store_out(x, base, valid, std::false_type{}, a.out);
#pragma unroll
            for (int r = 0; r < 16; r++) {
                if (MODE == MODE_SMAD) x[r] = mulmod_ssmall(f[r], reduce_signed(nb2[r], k), k);
                else if (MODE == MODE_S32) x[r] = montmul_s32(f[r], reduce_s32(nb2[r], k), k);
                else if (MODE == MODE_MAD) x[r] = mulmod_small(f[r], csub(reduce_i32_small(nb2[r], k), k.q), k);
                else x[r] = montmul(f[r], reduce_i32(nb2[r], k), k);
            }
            request_in0(g + gstride);
            inv_core<L, MODE>(x, twi, lds_poly, lp, k, a.c0, a.c0p, a.c1, a.c1p);
            store_out(x, base, valid, std::false_type{}, a.out2);
            continue;
        }
        if (OP == OP_MULNTT) {
#pragma unroll
            for (int r = 0; r < 16; r++) {
                if (MODE == MODE_SMAD) x[r] = mulmod_ssmall((L > 8) ? reduce_signed((int32_t)x[r], k) : x[r], reduce_signed(nb[r], k), k);   // n = 256: |x| < 18q, |y| <= 2q: below 2^31
                else if (MODE == MODE_S32) x[r] = montmul_s32(x[r], reduce_s32(nb[r], k), k);                       // 22q * 3q/2 < 2^31
                else if (MODE == MODE_MAD) x[r] = mulmod_small(x[r], csub(reduce_i32_small(nb[r], k), k.q), k);    // < 2q(L+1) * q < 2^32
                else x[r] = montmul(x[r], reduce_i32(nb[r], k), k);
            }
            request_in0(g + gstride);
        }
        if (OP == OP_MUL) {
            uint32_t y[16];
#pragma unroll
            for (int r = 0; r < 16; r++) y[r] = (MODE == MODE_SMAD) ? reduce_signed(nb[r], k) : (MODE == MODE_S32) ? reduce_s32(nb[r], k) : (MODE == MODE_MAD) ? reduce_i32_small(nb[r], k) : reduce_i32(nb[r], k);
            request_in0(g + gstride);
            fwd_core<L, MODE, (MODE == MODE_SMAD || MODE == MODE_S32)>(y, twf, lds_poly, lp, k);      // [0,2q) (signed modes: lazy, reduced below)
#pragma unroll
            for (int r = 0; r < 16; r++) {
                if (MODE == MODE_SMAD) x[r] = mulmod_ssmall((L > 8) ? reduce_signed((int32_t)x[r], k) : x[r], reduce_signed((int32_t)y[r], k), k);
                else if (MODE == MODE_S32) x[r] = montmul_s32(x[r], reduce_s32((int32_t)y[r], k), k);
                else if (MODE == MODE_MAD) x[r] = mulmod_small(x[r], csub(y[r], k.q), k);
                else x[r] = montmul(x[r], y[r], k);
            }
        }

        inv_core<L, MODE>(x, twi, lds_poly, lp, k, a.c0, a.c0p, a.c1, a.c1p);
        store_out(x, base, valid, std::false_type{}, a.out);
    }
}
