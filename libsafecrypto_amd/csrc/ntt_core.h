// ntt_core.h - register/LDS machinery of the batched negacyclic NTT (device side, gfx950).
//
// What it computes: V[k] = sum_i t[i] g^{i(2k+1)} mod q and its exact
// inverse, the functions behind the reference's fwd_ntt_32_* / inv_ntt_32_*
// (src/utils/arith/ntt_template.c.in:1541-1663).  How it computes them is different from the
// reference (twist + bit-reverse + radix-2 DIT with lazy int32 sums):
//
//   * merged-twiddle Cooley-Tukey forward / Gentleman-Sande inverse: the psi^i twist is folded
//     into the butterflies (twiddle zeta[k] = g^bitrev(k), k = 2^s + block), so there is no
//     separate pointwise pass and no bit-reversal pass;
//   * 16 coefficients per lane, n/16 lanes per polynomial (16/32/64 lanes for n=256/512/1024,
//     so 4/2/1 polynomials per 64-wide wavefront);  log2(n) stages are run as "rounds" of up
//     to four stages that are entirely in registers; between rounds the polynomial is
//     re-distributed through LDS (16 ds_write_b32 + 4 ds_read_b128 per lane, XOR-swizzled,
//     wave-private so no s_barrier);
//   * the twiddles of the first round depend only on the register index -> SGPRs; the others
//     are per-lane and stay resident in VGPRs while the wave loops over polynomials.
//
// Position/layout algebra: a position j (log2 n bits) is split into a 4-bit register index r
// taken from bits [b0+3..b0] and a lane-in-polynomial index lp made of the remaining bits.
// Round ri uses b0 = max(L-4(ri+1), 0).
#pragma once
#include "modarith.h"

namespace sc {

template <int L> struct Lay {
    static constexpr int N = 1 << L;
    static constexpr int LP = N / 16;          // lanes per polynomial
    static constexpr int PW = 64 / LP;         // polynomials per wavefront
    static constexpr int NR = (L + 3) / 4;     // rounds
    static __host__ __device__ constexpr int b0(int ri) { return (L - 4 * (ri + 1)) < 0 ? 0 : (L - 4 * (ri + 1)); }
    static __host__ __device__ constexpr int ns(int ri) { return (L - 4 * ri) > 4 ? 4 : (L - 4 * ri); }
};

__host__ __device__ constexpr int pos_of(int b0, int lp, int r)
{
    return ((lp >> b0) << (b0 + 4)) | (r << b0) | (lp & ((1 << b0) - 1));
}
__host__ __device__ constexpr int lin_of(int b0, int j)   // lp*16 + r of position j in layout b0
{
    return ((((j >> (b0 + 4)) << b0) | (j & ((1 << b0) - 1))) << 4) | ((j >> b0) & 15);
}
__host__ __device__ constexpr int swz(int a) { return a ^ (((a >> 6) & 3) << 2); }

__device__ __forceinline__ void wave_lds_fence()
{
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

// Re-distribute one polynomial from register layout A (b0 = A) to layout B through LDS.
// LDS address of position j = swz(lin_of(B, j)): the reader finds its 16 values contiguous.
template <int A, int B>
__device__ __forceinline__ void exchange(uint32_t (&x)[16], uint32_t *lds_poly, int lp)
{
    const int wa = swz(lin_of(B, pos_of(A, lp, 0)));
#pragma unroll
    for (int r = 0; r < 16; r++) {
        const int wb = swz(lin_of(B, pos_of(A, 0, r)));   // folds to a constant once unrolled
        lds_poly[wa ^ wb] = x[r];
    }
    wave_lds_fence();
    const int kx = (lp >> 2) & 3;
#pragma unroll
    for (int c = 0; c < 4; c++) {
        const uint4 v = *reinterpret_cast<const uint4 *>(lds_poly + lp * 16 + ((c ^ kx) << 2));
        x[4 * c + 0] = v.x; x[4 * c + 1] = v.y; x[4 * c + 2] = v.z; x[4 * c + 3] = v.w;
    }
    wave_lds_fence();
}

// Twiddles of round RI for this lane: stage s acting on register bit t uses
// zeta[2^s + (H << (3-t)) + c], c = r >> (t+1), H = lp >> b0.  Stored at tw[(1<<(3-t))-1+c].
template <int L, int RI>
__device__ __forceinline__ void load_tw(const uint32_t *__restrict__ z, const uint32_t *__restrict__ zp, int lp,
                                        uint32_t (&tw)[15], uint32_t (&twp)[15])
{
    constexpr int B0 = Lay<L>::b0(RI);
    const int H = (RI == 0) ? 0 : (lp >> B0);
#pragma unroll
    for (int i = 0; i < Lay<L>::ns(RI); i++) {
        const int s = 4 * RI + i, p = L - 1 - s, t = p - B0, cnt = 1 << (3 - t);
        const int kb = (1 << s) + (H << (3 - t));
#pragma unroll
        for (int c = 0; c < cnt; c++) {
            tw[cnt - 1 + c] = z[kb + c];
            twp[cnt - 1 + c] = zp[kb + c];
        }
    }
}

// One forward (Cooley-Tukey) stage on register bit T:  (a, b) -> (a + zeta b, a - zeta b)
template <int MODE, bool LAZY, int T>
__device__ __forceinline__ void ct_stage(uint32_t (&x)[16], const uint32_t (&tw)[15], const uint32_t (&twp)[15], const ModK &k)
{
#pragma unroll
    for (int r = 0; r < 16; r++) {
        if (r & (1 << T)) continue;
        const int idx = (1 << (3 - T)) - 1 + (r >> (T + 1));
        // small-modulus mode: forward values stay below 2q(L+1) < 2^18.1, so value * twiddle fits 32 bits and a Barrett step on the
        // product needs no per-twiddle companion (15-30 fewer live registers per wave: one more wave per SIMD)
        if constexpr (MODE == MODE_SMAD) {       // signed lazy residues (modarith.h): no bias, no companion
            const uint32_t t = mulmod_ssmall(x[r | (1 << T)], tw[idx], k), a = x[r];
            x[r] = a + t;
            x[r | (1 << T)] = a - t;
            continue;
        }
        const uint32_t t = (MODE == MODE_MAD) ? mulmod_small(x[r | (1 << T)], tw[idx], k) : Mod<MODE>::mulc(x[r | (1 << T)], tw[idx], twp[idx], k);
        const uint32_t a = x[r];
        if constexpr (MODE == MODE_S32) {        // signed lazy residues, 32-bit multiplier (modarith.h)
            x[r] = a + t;
            x[r | (1 << T)] = a - t;
            continue;
        }
        if (LAZY) {                      // values grow by < 2q per stage, never reduced
            x[r] = a + t;
            x[r | (1 << T)] = a + k.q2 - t;
        } else {                         // values stay in [0, 2q)
            x[r] = csub(a + t, k.q2);
            x[r | (1 << T)] = csub(a + k.q2 - t, k.q2);
        }
    }
}

// One inverse (Gentleman-Sande) stage on register bit T:  (a, b) -> (a + b, (a - b) zeta^-1).
// G < 0: inputs and outputs in [0, 2q).  G >= 0 (MODE_MAD only): inputs below 2q 2^G; the sums are left to grow (below
// 2q 2^(G+1)) and the difference is biased by 2q 2^G instead of reduced - 3 instructions saved per butterfly.  With q < 2^14 and
// at most 8 growing stages every multiplied value stays below 2^23, which is what MODE_MAD needs.
template <int MODE, int T, int G>
__device__ __forceinline__ void gs_stage(uint32_t (&x)[16], const uint32_t (&tw)[15], const uint32_t (&twp)[15], const ModK &k)
{
    static_assert(G < 0 || MODE == MODE_MAD || MODE == MODE_SMAD || MODE == MODE_S32, "growing sums need a small-modulus or signed mode");
#pragma unroll
    for (int r = 0; r < 16; r++) {
        if (r & (1 << T)) continue;
        const int idx = (1 << (3 - T)) - 1 + (r >> (T + 1));
        const uint32_t a = x[r], b = x[r | (1 << T)];
        if constexpr (MODE == MODE_SMAD || MODE == MODE_S32) {       // signed: the sums grow (below 2q 2^stage in magnitude), the difference needs no bias
            x[r] = a + b;
            x[r | (1 << T)] = Mod<MODE>::mulc(a - b, tw[idx], twp[idx], k);
        } else if constexpr (G >= 0) {
            const uint32_t d = a + (k.q2 << G) - b;
            x[r] = a + b;
            x[r | (1 << T)] = Mod<MODE>::mulc(d, tw[idx], twp[idx], k);
        } else {
            uint32_t d = a + k.q2 - b;                       // (0, 4q)
            if (MODE == MODE_U24) d = csub(d, k.q2);         // the 24-bit multiplier needs d < 2^24
            x[r] = csub(a + b, k.q2);
            x[r | (1 << T)] = Mod<MODE>::mulc(d, tw[idx], twp[idx], k);
        }
    }
}

// Last inverse stage (s = 0, register bit 3 of round 0) with the output scale folded in:
// (a, b) -> ((a + b) c0, (a - b) c1), c0 = scale, c1 = zeta^-1 scale; result canonical [0,q).
template <int MODE, int G>
__device__ __forceinline__ void gs_last_stage(uint32_t (&x)[16], uint32_t c0, uint32_t c0p, uint32_t c1, uint32_t c1p, const ModK &k)
{
    if constexpr (MODE == MODE_SMAD) {
        // signed sums of magnitude <= 2q 2^8 = 512 q: biased by 512 q (a multiple of q) they are non-negative and below 1024 q < 2^24; for
        // q >= 8192 the scale constants come centred with signed companions (modarith.h mulc_pos_centred), the result lands in [0, 2q)
        const uint32_t bias = k.q << 9;
        if (k.q < 8192u) {       // 1024 q < 2^23 (q = 7681): canonical constants, all three products on the 24-bit multiplier (measured 1.7 % faster per fused launch)
#pragma unroll
            for (int r = 0; r < 8; r++) {
                const uint32_t a = x[r], b = x[r | 8];
                x[r] = csub(Mod<MODE_MAD>::mulc(a + b + bias, c0, c0p, k), k.q);
                x[r | 8] = csub(Mod<MODE_MAD>::mulc(a - b + bias, c1, c1p, k), k.q);
            }
            return;
        }
#pragma unroll
        for (int r = 0; r < 8; r++) {
            const uint32_t a = x[r], b = x[r | 8];
            x[r] = csub(mulc_pos_centred(a + b + bias, c0, c0p, k), k.q);
            x[r | 8] = csub(mulc_pos_centred(a - b + bias, c1, c1p, k), k.q);
        }
        return;
    }
    if constexpr (MODE == MODE_S32) {
        // signed sums of magnitude <= 32q, biased by 32q (a multiple of q): non-negative and below 2^31; centred scale constants with signed
        // companions leave the product in [0, 3q/2)
        const uint32_t bias = k.q << 5;
#pragma unroll
        for (int r = 0; r < 8; r++) {
            const uint32_t a = x[r], b = x[r | 8];
            x[r] = csub(Mod<MODE_S32>::mulc(a + b + bias, c0, c0p, k), k.q);
            x[r | 8] = csub(Mod<MODE_S32>::mulc(a - b + bias, c1, c1p, k), k.q);
        }
        return;
    }
#pragma unroll
    for (int r = 0; r < 8; r++) {
        const uint32_t a = x[r], b = x[r | 8];
        uint32_t s = a + b, d = a + (G >= 0 ? (k.q2 << G) : k.q2) - b;
        if (MODE == MODE_U24) { s = csub(s, k.q2); d = csub(d, k.q2); }
        x[r] = csub(Mod<MODE>::mulc(s, c0, c0p, k), k.q);
        x[r | 8] = csub(Mod<MODE>::mulc(d, c1, c1p, k), k.q);
    }
}

// A partial round (NS < 4) always sits at b0 = 0 and acts on register bits NS-1..0; a full
// round acts on 3..0.
template <int MODE, bool LAZY, int NS>
__device__ __forceinline__ void ct_round(uint32_t (&x)[16], const uint32_t (&tw)[15], const uint32_t (&twp)[15], const ModK &k)
{
    if constexpr (NS >= 4) ct_stage<MODE, LAZY, 3>(x, tw, twp, k);
    if constexpr (NS >= 3) ct_stage<MODE, LAZY, 2>(x, tw, twp, k);
    if constexpr (NS >= 2) ct_stage<MODE, LAZY, 1>(x, tw, twp, k);
    ct_stage<MODE, LAZY, 0>(x, tw, twp, k);
}

// G0: growth exponent of the round's first stage (see gs_stage), < 0 for a round kept in [0, 2q)
template <int MODE, int NS, bool SKIP_TOP, int G0>
__device__ __forceinline__ void gs_round(uint32_t (&x)[16], const uint32_t (&tw)[15], const uint32_t (&twp)[15], const ModK &k)
{
    gs_stage<MODE, 0, G0>(x, tw, twp, k);
    if constexpr (NS >= 2) gs_stage<MODE, 1, (G0 < 0 ? -1 : G0 + 1)>(x, tw, twp, k);
    if constexpr (NS >= 3) gs_stage<MODE, 2, (G0 < 0 ? -1 : G0 + 2)>(x, tw, twp, k);
    if constexpr (NS >= 4 && !SKIP_TOP) gs_stage<MODE, 3, (G0 < 0 ? -1 : G0 + 3)>(x, tw, twp, k);
}

// Twiddle sets held by a wave for one direction.
template <int L> struct TwSet {
    uint32_t t0[15], t0p[15];     // round 0 (lane-invariant -> scalar registers)
    uint32_t t1[15], t1p[15];     // round 1
    uint32_t t2[15], t2p[15];     // round 2 (n = 512, 1024 only)
    __device__ __forceinline__ void load(const uint32_t *__restrict__ z, const uint32_t *__restrict__ zp, int lp)
    {
        load_tw<L, 0>(z, zp, lp, t0, t0p);
        load_tw<L, 1>(z, zp, lp, t1, t1p);
        if constexpr (Lay<L>::NR == 3) load_tw<L, 2>(z, zp, lp, t2, t2p);
    }
};

// Forward transform of the 16 values of this lane.  In: layout b0(0), values in [0,2q).
// Out: layout b0(NR-1) = 0 (position j = lp*16 + r holds V[bitrev_L(j)]), values in [0,2q)
// when !LAZY and in [0, 2q(L+1)) when LAZY.
template <int L, int MODE, bool LAZY>
__device__ __forceinline__ void fwd_core(uint32_t (&x)[16], const TwSet<L> &tw, uint32_t *lds_poly, int lp, const ModK &k)
{
    using Y = Lay<L>;
    ct_round<MODE, LAZY, 4>(x, tw.t0, tw.t0p, k);
    exchange<Y::b0(0), Y::b0(1)>(x, lds_poly, lp);
    ct_round<MODE, LAZY, Y::ns(1)>(x, tw.t1, tw.t1p, k);
    if constexpr (Y::NR == 3) {
        exchange<Y::b0(1), Y::b0(2)>(x, lds_poly, lp);
        ct_round<MODE, LAZY, Y::ns(2)>(x, tw.t2, tw.t2p, k);
    }
}

// Inverse transform.  In: layout 0 (position j holds V[bitrev_L(j)]), values in [0,2q).
// Out: layout b0(0) (natural coefficient order), canonical [0,q), scaled by c0 (= n^-1 or
// n^-1 2^32 when a Montgomery product preceded).
template <int L, int MODE>
__device__ __forceinline__ void inv_core(uint32_t (&x)[16], const TwSet<L> &tw, uint32_t *lds_poly, int lp, const ModK &k,
                                         uint32_t c0, uint32_t c0p, uint32_t c1, uint32_t c1p)
{
    using Y = Lay<L>;
    // small-modulus mode: the last 8 stages (rounds 1 and 0) let the sums grow; the L-8 stages of round 2 stay reduced
    constexpr bool GROW = MODE == MODE_MAD || MODE == MODE_SMAD || MODE == MODE_S32;
    static_assert(!GROW || Y::ns(1) == 4, "growth bookkeeping assumes 4 + 4 stages in rounds 1 and 0");
    if constexpr (Y::NR == 3) {
        gs_round<MODE, Y::ns(2), false, -1>(x, tw.t2, tw.t2p, k);
        if constexpr (MODE == MODE_SMAD) {       // the registers this round left as sums (up to 2q 2^ns) go back to [-q, 2q) - see modarith.h
#pragma unroll
            for (int r = 0; r < 16; r++)
                if (!(r & (1 << (Y::ns(2) - 1)))) x[r] = reduce_signed((int32_t)x[r], k);
        }
        // (the 32-bit signed mode has room for them: at most 4 B after two stages, 64 B after the round that follows)
        exchange<Y::b0(2), Y::b0(1)>(x, lds_poly, lp);
    }
    gs_round<MODE, Y::ns(1), false, (GROW ? 0 : -1)>(x, tw.t1, tw.t1p, k);
    if constexpr (MODE == MODE_S32) {            // the eight registers whose last product is more than one stage back (up to 16 B, 64 B for n = 1024)
#pragma unroll
        for (int r = 0; r < 8; r++) x[r] = reduce_s32((int32_t)x[r], k);
    }
    exchange<Y::b0(1), Y::b0(0)>(x, lds_poly, lp);
    gs_round<MODE, 4, true, (GROW ? 4 : -1)>(x, tw.t0, tw.t0p, k);
    gs_last_stage<MODE, (GROW ? 7 : -1)>(x, c0, c0p, c1, c1p, k);
}

}  // namespace sc
