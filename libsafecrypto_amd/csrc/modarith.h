// modarith.h - in-register modular arithmetic for gfx950 (device side).
//
// Two constant-multiplier code paths (Shoup: the multiplier w comes with a precomputed
// companion wp = floor(w * BETA / q)):
//   MODE_U24: BETA = 2^24, built from the full-rate 24-bit multiplies v_mul_u32_u24 /
//             v_mul_hi_u32_u24 (valid while every multiplied value is < 2^24, q < 2^23)
//   MODE_U32: BETA = 2^32, built from v_mul_hi_u32 / v_mul_lo_u32 (q < 2^30)
// Both return a*w mod q as a lazily reduced value in [0, 2q).  Data-by-data products use a
// signed 32-bit Montgomery step.  The reference reduces with a double-precision quotient
// (ntt_template.c.in:756-760,848-851); only the residue classes have to agree, and every
// kernel ends by mapping to the canonical representative the reference produces.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace sc {

enum { MODE_U24 = 1, MODE_U32 = 2, MODE_MAD = 3, MODE_SMAD = 4, MODE_S32 = 5 };

// Per-modulus scalars, passed by value as kernel arguments (they end up in SGPRs).
struct ModK {
    uint32_t q;        // modulus
    uint32_t q2;       // 2q
    uint32_t qinv;     // q^{-1} mod 2^32           (Montgomery)
    uint32_t m32;      // floor(2^32 / q)           (reduction of arbitrary int32 inputs)
    uint32_t one_p;    // floor(BETA / q)           (Shoup companion of w = 1: a generic reduce)
    uint32_t half;     // (q-1)/2                   (center)
};

// v_mul_u32_u24 / v_mul_hi_u32_u24 (full-rate 24x24-bit multiplies) through the LLVM intrinsics;
// hipcc 7.2 has no __builtin_amdgcn_mul*_u24, and the (a & 0xffffff) * (b & 0xffffff) idiom is
// only sometimes selected as the 24-bit form.
extern "C" __device__ uint32_t sc_llvm_mul_u24(uint32_t, uint32_t) __asm("llvm.amdgcn.mul.u24.i32");
extern "C" __device__ uint32_t sc_llvm_mulhi_u24(uint32_t, uint32_t) __asm("llvm.amdgcn.mulhi.u24");
extern "C" __device__ int32_t sc_llvm_mul_i24(int32_t, int32_t) __asm("llvm.amdgcn.mul.i24.i32");
__device__ __forceinline__ uint32_t mul_u24(uint32_t a, uint32_t b) { return sc_llvm_mul_u24(a, b); }
__device__ __forceinline__ int32_t mul_i24(int32_t a, int32_t b) { return sc_llvm_mul_i24(a, b); }
__device__ __forceinline__ uint32_t mulhi_u24(uint32_t a, uint32_t b) { return sc_llvm_mulhi_u24(a, b); }

__device__ __forceinline__ uint32_t csub(uint32_t x, uint32_t m) { return min(x, x - m); }  // x in [0,2m) -> [0,m)

template <int MODE> struct Mod;

template <> struct Mod<MODE_U24> {
    static constexpr uint64_t BETA = 1ull << 24;
    // a < 2^24, w < q < 2^23, wp = floor(w 2^24 / q)  ->  a*w mod q in [0,2q)
    static __device__ __forceinline__ uint32_t mulc(uint32_t a, uint32_t w, uint32_t wp, const ModK &k)
    {
        const uint32_t lo = mul_u24(a, wp);                               // bits 31:0  of a*wp
        const uint32_t hi = mulhi_u24(a, wp);                             // bits 47:32 of a*wp
        const uint32_t qh = __builtin_amdgcn_alignbit(hi, lo, 24);        // floor(a*wp / 2^24) < 2^24
        return mul_u24(a, w) - mul_u24(qh, k.q);
    }
};

template <> struct Mod<MODE_U32> {
    static constexpr uint64_t BETA = 1ull << 32;
    // a < 2^32, w < q < 2^30, wp = floor(w 2^32 / q)  ->  a*w mod q in [0,2q)
    static __device__ __forceinline__ uint32_t mulc(uint32_t a, uint32_t w, uint32_t wp, const ModK &k)
    {
        uint32_t qh = __umulhi(a, wp);
        return a * w - qh * k.q;
    }
};

// Small moduli (q < 2^14: 7681, 12289): every value a kernel multiplies stays below 2^23, so the quotient estimate
// qh = floor(a wp / 2^32) <= a fits a signed 24-bit operand and the correction a*w - qh*q is ONE multiply-add
// (v_mad_i32_i24 with the constant -q): 3 instructions per constant product instead of 4.  Same Shoup companion (BETA = 2^32)
// as MODE_U32, same result range [0, 2q).
template <> struct Mod<MODE_MAD> {
    static constexpr uint64_t BETA = 1ull << 32;
    static __device__ __forceinline__ uint32_t mulc(uint32_t a, uint32_t w, uint32_t wp, const ModK &k)
    {
        const uint32_t qh = __umulhi(a, wp);
        return mul_u24(a, w) + (uint32_t)mul_i24((int32_t)qh, -(int32_t)k.q);
    }
};

// Signed small-modulus mode (36 q^2 < 2^31: q = 7681).  Values are SIGNED lazy residues: a Barrett / Shoup quotient estimate that
// is off by one in either direction leaves r in [-q, 2q), and nothing forces r back into [0, 2q) - so a butterfly is
// (a + t, a - t) with no 2q bias, an input reduction is two instructions, and the pointwise product needs no canonical operand.
// Twiddles are centred (|w| <= q/2) with signed companions floor(w 2^32 / q).  Bounds, n = 256 (36 q^2 < 2^31: q = 7681): forward values stay
// below 18q in magnitude (inputs in [-q, 2q), eight stages of growth by at most 2q), so every product of a value with a twiddle or with another
// reduced value is below 18q * 2q < 2^31; inverse sums double per stage and stay below 2q 2^8 < 2^23 (the 24-bit multiply-add operand).
// n = 512, 1024 (L = 9, 10 stages; q^2 L < 2^31: q < 14654, so 7681 and 12289): the value the last forward stage multiplies is below 2q L and a
// twiddle below q/2, product below q^2 L < 2^31; the transform's outputs (below 2q(L+1)) are reduced to [-q, 2q) before the pointwise product
// (4 q^2 < 2^31); the L-8 stages of the first inverse round leave sums of up to 2q 2^(L-8), which are reduced again so that the eight growing
// stages that follow start from 2q as for n = 256 (512 q < 2^23 needs q < 2^14).
template <> struct Mod<MODE_SMAD> {
    static constexpr uint64_t BETA = 1ull << 32;
    // a signed with |a| < 2^23, w centred, wp = floor(w 2^32 / q) as int32  ->  a*w mod q in [-q, 2q)
    static __device__ __forceinline__ uint32_t mulc(uint32_t a, uint32_t w, uint32_t wp, const ModK &k)
    {
        const int32_t qh = __mulhi((int32_t)a, (int32_t)wp);
        return (uint32_t)mul_i24((int32_t)a, (int32_t)w) + (uint32_t)mul_i24(qh, -(int32_t)k.q);
    }
};

// Signed 32-bit mode (q < 2^25: Dilithium's 8380417, ring-TESLA's 8399873, and anything smaller).  The same idea as MODE_SMAD with the full
// 32-bit multiplier: values are signed lazy residues, a Shoup product of a signed value a (|a| < 2^31) with a centred constant and its signed
// companion lands in (-q/2, 3q/2), a forward butterfly is (a + t, a - t) (6 instructions instead of 7), an inverse one (a + b, (a - b) w) with
// the sums left to grow through the four stages of a round (6 instead of 9) and the eight registers a round leaves above 2q reduced once
// (3 instructions each), an input reduction 3 instructions instead of 4, a data-by-data Montgomery product 5 instead of 8 (its result in
// (-q, q) needs no correction).  Bounds: forward values stay below 2q(L+1) <= 22q; a round of four growing inverse stages takes inputs of
// magnitude B to sums and differences of at most 16 B (B <= 8q: 128 q < 2^31); the last inverse stage biases sums of at most 32q by 32q
// (64 q < 2^31 is where q < 2^25 comes from) and finishes with an unsigned-operand product in [0, 3q/2).
template <> struct Mod<MODE_S32> {
    static constexpr uint64_t BETA = 1ull << 32;
    static __device__ __forceinline__ uint32_t mulc(uint32_t a, uint32_t w, uint32_t wp, const ModK &k)
    {
        const int32_t qh = __mulhi((int32_t)a, (int32_t)wp);
        return a * w - (uint32_t)qh * k.q;
    }
};

// any int32 -> signed residue in (-q/2, 3q/2): three instructions
__device__ __forceinline__ uint32_t reduce_s32(int32_t x, const ModK &k)
{
    const int32_t qh = __mulhi(x, (int32_t)k.m32);
    return (uint32_t)x - (uint32_t)qh * k.q;
}

// signed a, b with |a b| < q 2^31  ->  a b 2^-32 mod q in (-q, q): five instructions
__device__ __forceinline__ uint32_t montmul_s32(uint32_t a, uint32_t b, const ModK &k)
{
    const int64_t p = (int64_t)(int32_t)a * (int32_t)b;
    const int32_t m = (int32_t)((uint32_t)p * k.qinv);
    return (uint32_t)((int32_t)(p >> 32) - __mulhi(m, (int32_t)k.q));
}

// signed lazy value with |v| < 32q -> canonical [0, q): bias, unsigned Barrett step, one conditional subtraction
__device__ __forceinline__ uint32_t canon_s32(uint32_t v, const ModK &k)
{
    v += k.q << 5;
    v -= __umulhi(v, k.m32) * k.q;
    return csub(v, k.q);
}

// a in [0, 2^24) (a biased sum), w centred, wp its signed companion -> a*w mod q in [0, 2q): the last inverse stage of the signed mode.  The
// quotient estimate |qh| <= a/2 < 2^23 fits the 24-bit multiply-add whatever q < 2^14 is (with a canonical constant it reaches a itself, which
// only fits for 1024 q < 2^23, i.e. q = 7681); the product a*w is taken modulo 2^32 by the full multiplier, which issues at the same rate.
__device__ __forceinline__ uint32_t mulc_pos_centred(uint32_t a, uint32_t w, uint32_t wp, const ModK &k)
{
    const int32_t qh = __mulhi((int32_t)a, (int32_t)wp);
    return a * w + (uint32_t)mul_i24(qh, -(int32_t)k.q);
}

// any int32 -> signed residue in [-q, 2q): two instructions (|qh| <= 2^31 / q < 2^19 fits the 24-bit operand)
__device__ __forceinline__ uint32_t reduce_signed(int32_t x, const ModK &k)
{
    const int32_t qh = __mulhi(x, (int32_t)k.m32);
    return (uint32_t)x + (uint32_t)mul_i24(qh, -(int32_t)k.q);
}

// signed x, y with |x y| < 2^31  ->  x*y mod q in [-q, 2q)
__device__ __forceinline__ uint32_t mulmod_ssmall(uint32_t x, uint32_t y, const ModK &k)
{
    const int32_t t = mul_i24((int32_t)x, (int32_t)y);                 // both factors below 2^23 in magnitude (callers' bounds): the 24-bit multiplier is exact
    const int32_t qh = __mulhi(t, (int32_t)k.m32);                     // and issues faster than v_mul_lo_u32 (measured: 1.7 % of a fused launch for 16 of them)
    return (uint32_t)t + (uint32_t)mul_i24(qh, -(int32_t)k.q);
}

// signed residue in [-q, 2q) -> canonical [0, q)
__device__ __forceinline__ uint32_t canon_signed(uint32_t r, const ModK &k) { return csub(csub(r + k.q, k.q2), k.q); }

// arbitrary int32 -> [0, 2q).  qh = floor(x m32 / 2^32) is floor(x/q) or one more, so r = x - qh q lies in [-q, 2q);
// read as unsigned, the smaller of r and r + q is the representative in [0, 2q).
__device__ __forceinline__ uint32_t reduce_i32(int32_t x, const ModK &k)
{
    const int32_t qh = __mulhi(x, (int32_t)k.m32);
    const uint32_t r = (uint32_t)x - (uint32_t)qh * k.q;
    return min(r, r + k.q);
}

// the same for q < 2^14 (|qh| < 2^19 fits a signed 24-bit operand): the correction is one multiply-add
__device__ __forceinline__ uint32_t reduce_i32_small(int32_t x, const ModK &k)
{
    const int32_t qh = __mulhi(x, (int32_t)k.m32);
    const uint32_t r = (uint32_t)x + (uint32_t)mul_i24(qh, -(int32_t)k.q);
    return min(r, r + k.q);
}

// q < 2^14: x*y for x*y < 2^32 (x a lazily grown NTT output, y canonical) -> [0, 2q): plain product, Barrett quotient, one
// multiply-add; replaces the 8-instruction Montgomery step and leaves no 2^-32 factor behind
__device__ __forceinline__ uint32_t mulmod_small(uint32_t x, uint32_t y, const ModK &k)
{
    const uint32_t t = mul_u24(x, y);                                   // x < 2q(L+1) < 2^19, y < q < 2^14
    const uint32_t qh = __umulhi(t, k.m32);
    return t + (uint32_t)mul_i24((int32_t)qh, -(int32_t)k.q);
}

// a*b*2^-32 mod q in [0,q); needs a*b < q*2^32
__device__ __forceinline__ uint32_t montmul(uint32_t a, uint32_t b, const ModK &k)
{
    uint64_t p = (uint64_t)a * b;
    uint32_t m = (uint32_t)p * k.qinv;
    int32_t t = (int32_t)(uint32_t)(p >> 32) - (int32_t)__umulhi(m, k.q);
    t += (t >> 31) & (int32_t)k.q;
    return (uint32_t)t;
}

}  // namespace sc
