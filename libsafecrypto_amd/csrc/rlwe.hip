// rlwe.hip - batched Ring-LWE encryption (RLWE-Enc) on MI355X behind the C ABI of include/sc_hip.h.
//
// Replaces, for a batch of messages under one key pair, rlwe_enc_encrypt / rlwe_enc_decrypt and, for a batch of independent
// key pairs, rlwe_enc_keygen with a_gen / r1_gen / r2_gen (src/schemes/enc/rlwe_enc/rlwe_enc.c:511-631, 633-707, 425-493,
// 205-290; parameter sets rlwe_enc_params.c:23-41; raw coding: q_bits per coefficient, MSB first).  The reference is built with
// RLWE_ENC_ENABLE_NTT_TRANSMISSION (rlwe_enc.h:24): c1 = a o NTT(e1) + NTT(e2) travels in the NTT domain, c2 = INTT(p o NTT(e1))
// + e3 + m_scale * bits in the coefficient domain.
//
// Randomness: the prng_32 word stream ("tape") of sc->prng_ctx[0], as for BLISS-B and ring-TESLA (or expanded on the device
// with sc_hip_drbg_words): 6n words per encryption (3n Gaussian samples, one prng_64 = two words, high word first, through the
// 64-bit CDF sampler, gaussian_cdf.c:639-659), n/2 + 2n + n/16 words per key pair.
//
// Mapping: ONE fused kernel per operation.  A polynomial lives in 16 registers of n/16 lanes (ntt_core.h), 4 (n = 256) or 2
// (n = 512) messages per wavefront; the samples are drawn straight into the transform's input layout, both transforms, the two
// key products, the inverse transform, the message encoding and the bit packing happen without a trip through HBM: per
// encryption the kernel reads the 24n-byte tape and the n/8-byte message and writes the 2 n q_bits / 8-byte ciphertext.
// q < 2^14 for both parameter sets, so everything runs in the small-modulus arithmetic (modarith.h MODE_MAD).  The sampler is
// a guide-table search (64-entry table + 4096-entry guide in LDS, see Sampler below).
#include "engine_internal.h"
#include <atomic>
#include "ntt_core.h"
#include "../../include/sc_hip_bench.h"

#include <string.h>

using namespace sc;

namespace {

struct RlP { int32_t set, q, q_bits, n, n_bits, m_scale, o_scale_0, o_scale_1; float sigma, tail; int32_t pad[2]; };   // rlwe_set_t (rlwe_enc_params.h:37-61)

constexpr int MAXTAB = 256;           // sampler table entries (values fit the guide's 8-bit fields)

struct RlArgs {
    const uint32_t *zf, *zfp, *zi, *zip;
    ModK k;
    uint32_t c0, c0p, c1, c1p;              // n^-1 and zeta[1]^-1 n^-1 (Shoup pairs)
    const uint32_t *szf, *szi, *szip;        // the signed small-modulus mode's tables (centred twiddles, signed companions) when the plan has them
    uint32_t sc0, sc0p, sc1, sc1p;           // and its scale constants
    const uint64_t *cdf; const uint16_t *guide; int tab_size;
    const uint32_t *key_a, *key_p, *key_r2; // NTT domain, engine order (position lp*16 + r holds index bitrev(lp*16 + r)), canonical
    unsigned long long batch;
    const uint8_t *msgs_in; uint8_t *msgs_out; size_t msg_stride;
    const uint32_t *tapes; size_t tape_stride;
    const uint8_t *ct_in; uint8_t *ct_out; size_t ct_stride;
    int16_t *kg_a, *kg_p, *kg_r2;
    int32_t m_scale, o0, o1;
};

__host__ __device__ constexpr int brv4(int r) { return ((r & 1) << 3) | ((r & 2) << 1) | ((r & 4) >> 1) | ((r & 8) >> 3); }

// gaussian_cdf_sample_64 (gaussian_cdf.c:536-553, 639-659): the largest b with cdf[b] < u (0 if none), sign from the LSB of the
// same draw.  A 4096-entry guide on the top 12 bits of u gives lo = max{b : cdf[b] < t 2^52} and cnt = max{b : cdf[b] <
// (t+1) 2^52} - lo, so the answer is lo + #{b in (lo, lo+cnt] : cdf[b] < u}.  For sigma ~ 4.5-4.9 all but the last few of the
// 4096 buckets have cnt <= 2: two unconditional probes (the table is padded with two all-ones entries, and entries above
// lo + cnt are >= (t+1) 2^52 > u, so probing them adds nothing); the rare wider bucket takes the halving search.
struct Sampler {
    const uint64_t *tab; const uint16_t *guide;
    __device__ __forceinline__ int32_t wide(uint64_t u, uint32_t lo, uint32_t hi) const
    {
        while (lo < hi) {
            const uint32_t mid = (lo + hi + 1) >> 1;
            if (tab[mid] < u) lo = mid; else hi = mid - 1;
        }
        return (int32_t)lo;
    }
};

// n Gaussian samples of one polynomial, drawn into the round-0 layout (register r of lane lp = coefficient r*LP + lp), as
// residues in [0, q] (a zero with the sign bit clear comes out as q).  w: the lane's 16 draws (two tape words each, high word first: prng_64, prng.c:963-978).
__device__ __forceinline__ void sample_regs(uint32_t (&x)[16], const uint2 (&w)[16], const Sampler &S, uint32_t q)
{
    uint32_t widest = 0;
    // four draws at a time (guide reads, then the eight probes): enough LDS reads in flight to cover their latency without
    // the 100+ live registers the fully unrolled form costs
#pragma unroll
    for (int c = 0; c < 16; c += 4) {
        uint32_t g[4];
#pragma unroll
        for (int e = 0; e < 4; e++) g[e] = S.guide[w[c + e].x >> 20];
#pragma unroll
        for (int e = 0; e < 4; e++) {
            const uint32_t lo = g[e] & 0xffu;
            // a = lo + [tab[lo+1] < u] + [tab[lo+2] < u] as two borrow chains that end in an add-with-carry (six plain add-class instructions; measured 2 % on the kernel), and
            // the draw's words stay where the load put them (a 64-bit compare wants them swapped into an aligned pair: two moves per draw)
            const uint2 t1 = *reinterpret_cast<const uint2 *>(S.tab + lo + 1), t2 = *reinterpret_cast<const uint2 *>(S.tab + lo + 2);
            uint32_t a, scratch;
            asm("v_sub_co_u32 %1, vcc, %3, %7\n\t"
                "v_subb_co_u32 %1, vcc, %4, %8, vcc\n\t"
                "v_addc_co_u32 %0, vcc, 0, %2, vcc\n\t"
                "v_sub_co_u32 %1, vcc, %5, %7\n\t"
                "v_subb_co_u32 %1, vcc, %6, %8, vcc\n\t"
                "v_addc_co_u32 %0, vcc, 0, %0, vcc"
                : "=&v"(a), "=&v"(scratch)
                : "v"(lo), "v"(t1.x), "v"(t1.y), "v"(t2.x), "v"(t2.y), "v"(w[c + e].y), "v"(w[c + e].x)
                : "vcc");
            widest = max(widest, g[e]);
            x[c + e] = (w[c + e].y & 1u) ? a : q - a;                                  // +a or -a as a residue in [0, q] (q itself stands for 0: the transforms take [0, 2q))
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    if (__builtin_expect(widest >= (3u << 8), 0)) {                                  // some bucket of this lane holds more than two candidates
#pragma unroll
        for (int r = 0; r < 16; r++) {
            const uint32_t g = S.guide[w[r].x >> 20], lo = g & 0xffu, cnt = g >> 8;
            if (cnt > 2) {
                const uint64_t u = ((uint64_t)w[r].x << 32) | w[r].y;
                const uint32_t a = (uint32_t)S.wide(u, lo, lo + cnt);
                x[r] = (w[r].y & 1u) ? a : q - a;
            }
        }
    }
}

// Loads are UNCONDITIONAL: a lane whose item is past the end of the batch is pointed at item 0 instead (its results are never
// stored).  A predicated load becomes a branch with a full s_waitcnt inside, which serialises the 16 requests.
template <int LP>
__device__ __forceinline__ void load_draws(uint2 (&w)[16], const uint32_t *words, int lp)
{
#pragma unroll
    for (int r = 0; r < 16; r++) w[r] = *reinterpret_cast<const uint2 *>(words + 2 * (r * LP + lp));
}

__device__ __forceinline__ uint32_t canon(uint32_t v, const ModK &k)               // any value below 2^23 -> [0, q)
{ return csub(Mod<MODE_MAD>::mulc(v, 1u, k.one_p, k), k.q); }

// Bit packing of the ciphertext (QB-bit big-endian fields).  The bit pattern repeats every lcm(32, QB) bits: a GROUP of CW = 32 / gcd coefficients
// fills GW = QB / gcd whole words (13 bits: 32 coefficients -> 13 words; 14 bits: 16 -> 7), so a lane that packs one group does it with
// compile-time shifts (3-4 shift-or steps and a byte swap per word) instead of a variable 64-bit window per word.  The polynomial is staged
// in LDS in natural order with the 16-byte chunks of a group XOR-rotated (pk_pos) so that 16 neighbouring lanes, each reading its own
// group with 16-byte reads, hit distinct banks.
template <int QB> struct Pk {
    static constexpr int G = (QB % 2) ? 1 : ((QB % 4) ? 2 : ((QB % 8) ? 4 : 8));   // gcd(32, QB) for QB <= 16
    static constexpr int CW = 32 / G, GW = QB / G;
    static_assert(CW == 32 || CW == 16, "group shapes of the two parameter sets");
    static __host__ __device__ constexpr int rot(int g) { return CW == 32 ? (g & 7) : ((g >> 2) & 3); }
    static __host__ __device__ constexpr int pos(int j) { return j ^ (rot(j / CW) << 2); }       // LDS word of coefficient j
};

// group g of the polynomial staged at `stage` -> GW words at dst32 + g * GW
template <int QB>
__device__ __forceinline__ void pack_group(const uint32_t *stage, uint32_t *dst32, bool valid, int g)
{
    using P = Pk<QB>;
    constexpr int CW = P::CW, GW = P::GW;
    uint32_t c[CW];
    const int rot = P::rot(g);
#pragma unroll
    for (int c4 = 0; c4 < CW / 4; c4++) {
        const uint4 v = *reinterpret_cast<const uint4 *>(stage + g * CW + ((c4 ^ rot) << 2));
        c[4 * c4] = v.x; c[4 * c4 + 1] = v.y; c[4 * c4 + 2] = v.z; c[4 * c4 + 3] = v.w;
    }
    uint32_t *out = dst32 + g * GW;
#pragma unroll
    for (int w = 0; w < GW; w++) {
        uint32_t v = 0;
#pragma unroll
        for (int i = (32 * w) / QB; i < CW && QB * i < 32 * (w + 1); i++) {
            const int sh = 32 * (w + 1) - QB * (i + 1);              // where coefficient i's lowest bit falls in word w (bit 0 = the word's last stream bit)
            v |= sh >= 0 ? c[i] << (sh & 31) : c[i] >> ((-sh) & 31);
        }
        if (valid) out[w] = __builtin_bswap32(v);
    }
}

#define RL_PROLOGUE                                                                                                  \
    using Y = Lay<L>;                                                                                                \
    constexpr int N = Y::N, LP = Y::LP, PW = Y::PW;                                                                  \
    __shared__ __attribute__((aligned(16))) uint32_t lds[4 * 1024];                                                  \
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;                                                       \
    const int pw = lane / LP, lp = lane % LP;                                                                        \
    uint32_t *lds_poly = lds + wid * 1024 + pw * N;                                                                  \
    const int blp = (int)(__builtin_bitreverse32((uint32_t)lp) >> (32 - (L - 4)));                                   \
    const ModK k = a.k;                                                                                              \
    const unsigned long long groups = (a.batch + PW - 1) / PW;                                                       \
    const unsigned long long gstride = (unsigned long long)gridDim.x * 4;

#define RL_TABLES                                                                                                    \
    __shared__ uint64_t s_tab[MAXTAB + 2];                                                                           \
    __shared__ uint16_t s_guide[4096];                                                                               \
    for (int i = threadIdx.x; i < a.tab_size + 2; i += blockDim.x) s_tab[i] = i < a.tab_size ? a.cdf[i] : ~0ull;     \
    for (int i = threadIdx.x; i < 2048; i += blockDim.x) reinterpret_cast<uint32_t *>(s_guide)[i] = reinterpret_cast<const uint32_t *>(a.guide)[i]; \
    const Sampler S{ s_tab, s_guide };

// All three kernels are persistent-wave loops that would otherwise be a chain load -> compute -> load -> ... per wave at one or
// two waves per SIMD, i.e. run at the HBM latency.  They are software-pipelined like the NTT kernels (sc_hip.hip): the words
// the NEXT step needs are requested before the current step's arithmetic starts (one polynomial's draws = 32 registers ahead).

// rlwe_enc_encrypt, rlwe_enc.c:511-631.  Register budget: the two key products a o NTT(e1), p o NTT(e1) (both below 2q < 2^15)
// are parked as one packed word per coefficient in a lane-private LDS slot while e2 is sampled and transformed, and only one
// polynomial's draws are held at a time, so two waves fit a SIMD without spills for both n.
#ifndef RL_ENC_WAVES
#define RL_ENC_WAVES(L) ((L) == 8 ? 3 : 2)   /* n = 256: 168 registers (a handful of spills) and three waves are 5-6 % faster than 181 and two; n = 512 needs 194 and gains nothing from three */
#endif
// MODE: MODE_SMAD (signed lazy residues, modarith.h: a butterfly is one instruction shorter in both directions) where the plan qualifies - both of the
// reference's parameter sets do - and MODE_MAD for any other q below 2^14.
template <int L, int QB, int MODE>
__global__ __launch_bounds__(256, RL_ENC_WAVES(L)) void rlwe_encrypt_kernel(RlArgs a)
{
    RL_PROLOGUE
    constexpr bool SM = MODE == MODE_SMAD;
    RL_TABLES
    __shared__ __attribute__((aligned(16))) uint32_t s_keys[2 * N];       // a | p, engine order
    __shared__ uint32_t s_park[4][16 * 64];                                 // per wave: slot r*64 + lane
    __shared__ uint32_t s_msg[4][N / 8];                                    // per wave: PW messages of N/32 words
    for (int i = threadIdx.x; i < 2 * N; i += blockDim.x) s_keys[i] = a.key_a[i];
    // Per-lane twiddles (rounds 1, 2) live in LDS, one row per lane-in-polynomial, and are fetched into registers for the duration of
    // one transform: held for the whole loop they are 45 (n = 256) / 69 (n = 512) registers that decide between two and three waves
    // per SIMD.  Row: forward round 1 | inverse round 1 | its companions [| the 8 + 8 + 8 of round 2]; a stride of 4 * odd words keeps the
    // 16-byte reads of 16 neighbouring rows on distinct banks.
    constexpr int TWS = (Y::NR == 3 ? 72 : 48) + 4;
    __shared__ __attribute__((aligned(16))) uint32_t s_tw[LP * TWS];
    TwSet<L> twf, twi;                                                      // round 0 (lane-invariant, scalar registers) stays here
    {
        TwSet<L> f, i;
        f.load(SM ? a.szf : a.zf, SM ? a.szf : a.zfp, lp);           // (neither mode's forward butterflies use companions)
        i.load(SM ? a.szi : a.zi, SM ? a.szip : a.zip, lp);
        if (threadIdx.x < LP) {
            uint32_t *row = s_tw + lp * TWS;
#pragma unroll
            for (int j = 0; j < 15; j++) { row[j] = f.t1[j]; row[16 + j] = i.t1[j]; row[32 + j] = i.t1p[j]; }
            if constexpr (Y::NR == 3) {
#pragma unroll
                for (int j = 0; j < 8; j++) { row[48 + j] = f.t2[7 + j]; row[56 + j] = i.t2[7 + j]; row[64 + j] = i.t2p[7 + j]; }
            }
        }
#pragma unroll
        for (int j = 0; j < 15; j++) { twf.t0[j] = f.t0[j]; twf.t0p[j] = f.t0p[j]; twi.t0[j] = i.t0[j]; twi.t0p[j] = i.t0p[j]; }
    }
    __syncthreads();
    auto fetch15 = [](uint32_t (&d)[15], const uint32_t *src) {
#pragma unroll
        for (int c = 0; c < 4; c++) {
            const uint4 v = reinterpret_cast<const uint4 *>(src)[c];
            d[4 * c] = v.x; d[4 * c + 1] = v.y; d[4 * c + 2] = v.z;
            if (c < 3) d[4 * c + 3] = v.w;
        }
    };
    auto fetch8 = [](uint32_t (&d)[15], const uint32_t *src) {
#pragma unroll
        for (int c = 0; c < 2; c++) {
            const uint4 v = reinterpret_cast<const uint4 *>(src)[c];
            d[7 + 4 * c] = v.x; d[8 + 4 * c] = v.y; d[9 + 4 * c] = v.z; d[10 + 4 * c] = v.w;
        }
    };
    // the row offset goes through an empty asm so that the reads are not hoisted out of the loop (back into 45-69 resident registers)
    auto tw_fwd = [&]() {
        int off = lp * TWS;
        asm volatile("" : "+v"(off));
        fetch15(twf.t1, s_tw + off);
        if constexpr (Y::NR == 3) fetch8(twf.t2, s_tw + off + 48);
    };
    auto tw_inv = [&]() {
        int off = lp * TWS;
        asm volatile("" : "+v"(off));
        fetch15(twi.t1, s_tw + off + 16);
        fetch15(twi.t1p, s_tw + off + 32);
        if constexpr (Y::NR == 3) { fetch8(twi.t2, s_tw + off + 56); fetch8(twi.t2p, s_tw + off + 64); }
    };
    unsigned long long g = (unsigned long long)blockIdx.x * 4 + wid;
    if (g >= groups) return;
    uint8_t *msg_lds = reinterpret_cast<uint8_t *>(&s_msg[wid][pw * (N / 32)]);
    uint32_t *park = &s_park[wid][lane];
    using PK = Pk<QB>;
    uint32_t *c1_stage = &s_park[wid][pw * N];
    uint2 wn[16];
    {
        const unsigned long long item = g * PW + pw;
        load_draws<LP>(wn, a.tapes + (item < a.batch ? (size_t)item * a.tape_stride : 0), lp);
    }
    for (; g < groups; g += gstride) {
        const unsigned long long item = g * PW + pw;
        const bool valid = item < a.batch;
        const uint32_t *tape = a.tapes + (valid ? (size_t)item * a.tape_stride : 0);
        uint32_t x[16];
        // e1 -> NTT(e1) (lazy, engine order) -> the two key products, parked.  e2's draws and the message are in flight meanwhile.
        sample_regs(x, wn, S, k.q);
        load_draws<LP>(wn, tape + 2 * N, lp);
        const uint32_t mword = reinterpret_cast<const uint32_t *>(a.msgs_in + (valid ? (size_t)item * a.msg_stride : 0))[lp & (N / 32 - 1)];
        tw_fwd();
        fwd_core<L, MODE, true>(x, twf, lds_poly, lp, k);
        {
            const uint4 *pa = reinterpret_cast<const uint4 *>(s_keys + lp * 16), *pp = reinterpret_cast<const uint4 *>(s_keys + N + lp * 16);
#pragma unroll
            for (int c = 0; c < 4; c++) {
                const uint4 va = pa[c], vp = pp[c];
                const uint32_t ka[4] = { va.x, va.y, va.z, va.w }, kp[4] = { vp.x, vp.y, vp.z, vp.w };
#pragma unroll
                for (int e = 0; e < 4; e++) {
                    if constexpr (SM) {                                             // |x| < 18q (n = 256) or reduced to [-q, 2q) first, key < q: products below 2^31;
                        const uint32_t xr = (L > 8) ? reduce_signed((int32_t)x[4 * c + e], k) : x[4 * c + e];      // both results in [-q, 2q): the first parked
                        park[(4 * c + e) * 64] = (mulmod_ssmall(xr, ka[e], k) + k.q) | (mulmod_ssmall(xr, kp[e], k) << 16);   // biased by q (below 3q < 2^16), the second as int16
                    } else                                                          // x < 2q(L+1), key < q: products below 2^32
                        park[(4 * c + e) * 64] = mulmod_small(x[4 * c + e], ka[e], k) | (mulmod_small(x[4 * c + e], kp[e], k) << 16);
                }
            }
        }
        // c1 = a o NTT(e1) + NTT(e2), NTT domain, canonical (:593-602); e3's draws in flight
        sample_regs(x, wn, S, k.q);
        load_draws<LP>(wn, tape + 4 * N, lp);
        tw_fwd();
        fwd_core<L, MODE, true>(x, twf, lds_poly, lp, k);
        uint8_t *ct = a.ct_out + (valid ? (size_t)item * a.ct_stride : 0);
        uint32_t t2[16];
        wave_lds_fence();
#pragma unroll
        for (int r = 0; r < 16; r++) {
            const uint32_t pk = park[r * 64];
            if constexpr (SM) {
                t2[r] = (uint32_t)((int32_t)pk >> 16);
                uint32_t v = (pk & 0xffffu) + x[r] + 31u * k.q;                       // (a o E1 + q) + E2, |E2| < 2q(L+1) <= 20q: + 31q makes it positive, below 2^20
                v += (uint32_t)mul_i24((int32_t)__umulhi(v, k.m32), -(int32_t)k.q);    // [0, 2q)
                x[r] = csub(v, k.q);
            } else {
                t2[r] = pk >> 16;
                x[r] = canon((pk & 0xffffu) + x[r], k);
            }
        }
        wave_lds_fence();
        // c1 is staged where the products were parked (the wave's 4 KB, one polynomial image per block) and packed together with c2 at the end
#pragma unroll
        for (int r = 0; r < 16; r++) c1_stage[PK::pos(brv4(r) * LP + blp)] = x[r];    // engine position lp*16 + r = natural index brv4(r)*LP + blp
        // c2 = INTT(p o NTT(e1)) + Encode(m) + e3 (:604-609); the next group's e1 draws in flight
        tw_inv();
        if constexpr (SM) inv_core<L, MODE>(t2, twi, lds_poly, lp, k, a.sc0, a.sc0p, a.sc1, a.sc1p);
        else inv_core<L, MODE>(t2, twi, lds_poly, lp, k, a.c0, a.c0p, a.c1, a.c1p);
        sample_regs(x, wn, S, k.q);
        {
            const unsigned long long ng = g + gstride, nitem = ng * PW + pw;
            const bool nvalid = ng < groups && nitem < a.batch;
            load_draws<LP>(wn, a.tapes + (nvalid ? (size_t)nitem * a.tape_stride : 0), lp);
        }
        if (lp < N / 32) reinterpret_cast<uint32_t *>(msg_lds)[lp] = mword;
        wave_lds_fence();
#pragma unroll
        for (int r = 0; r < 16; r++) {
            const int j = r * LP + lp;
            const uint32_t bit = ((uint32_t)msg_lds[j >> 3] >> (7 - (j & 7))) & 1u;
            lds_poly[PK::pos(j)] = csub(csub(t2[r] + x[r] + bit * (uint32_t)a.m_scale, k.q2), k.q);     // < q + q + 1 + m_scale < 4q
        }
        wave_lds_fence();
        // both polynomials' groups over the block's lanes: 2 * 8 groups on 16 lanes (n = 256, 13 bits), 2 * 32 on 32 lanes in two passes (n = 512, 14 bits)
        constexpr int GP = N / PK::CW;
        if constexpr (2 * GP == LP) {
            const bool second = lp >= GP;
            pack_group<QB>(second ? lds_poly : c1_stage, reinterpret_cast<uint32_t *>(ct + (second ? N * QB / 8 : 0)), valid, lp & (GP - 1));
        } else {
            static_assert(GP == LP, "one group per lane and polynomial");
            pack_group<QB>(c1_stage, reinterpret_cast<uint32_t *>(ct), valid, lp);
            pack_group<QB>(lds_poly, reinterpret_cast<uint32_t *>(ct + N * QB / 8), valid, lp);
        }
        wave_lds_fence();
    }
}

// rlwe_enc_decrypt, rlwe_enc.c:633-707
template <int L, int QB>
__global__ __launch_bounds__(256) void rlwe_decrypt_kernel(RlArgs a)
{
    RL_PROLOGUE
    constexpr int CTW = 2 * N * QB / 32;                        // ciphertext words
    constexpr int KW = (CTW + LP - 1) / LP;                     // per lane
    static_assert(CTW < N, "the ciphertext is staged in the polynomial's LDS image");
    TwSet<L> twi;
    twi.load(a.zi, a.zip, lp);
    uint32_t kr[16];
    {
        const uint4 *pr = reinterpret_cast<const uint4 *>(a.key_r2 + lp * 16);
#pragma unroll
        for (int c = 0; c < 4; c++) { const uint4 v = pr[c]; kr[4 * c] = v.x; kr[4 * c + 1] = v.y; kr[4 * c + 2] = v.z; kr[4 * c + 3] = v.w; }
    }
    unsigned long long g = (unsigned long long)blockIdx.x * 4 + wid;
    if (g >= groups) return;
    uint32_t cn[KW];
    auto fetch = [&](unsigned long long gg) {
        const unsigned long long item = gg * PW + pw;
        const bool ok = gg < groups && item < a.batch;
        const uint32_t *ct = reinterpret_cast<const uint32_t *>(a.ct_in + (ok ? (size_t)item * a.ct_stride : 0));
#pragma unroll
        for (int i = 0; i < KW; i++) cn[i] = ct[min(lp + LP * i, CTW - 1)];                  // unconditional (see load_draws)
    };
    fetch(g);
    for (; g < groups; g += gstride) {
        const unsigned long long item = g * PW + pw;
        const bool valid = item < a.batch;
#pragma unroll
        for (int i = 0; i < KW; i++) if (lp + LP * i < CTW) lds_poly[lp + LP * i] = __builtin_bswap32(cn[i]);
        if (lp == 0) lds_poly[CTW] = 0;                          // the word after the stream is read (and shifted out) by the last field
        wave_lds_fence();
        fetch(g + gstride);                                       // the next group's ciphertexts are in flight during the transform
        auto field = [&](int i) -> uint32_t {                     // coefficient i of the 2n-coefficient stream
            const int bo = i * QB, wi = bo >> 5, o = bo & 31;
            const uint64_t win = ((uint64_t)lds_poly[wi] << 32) | lds_poly[wi + 1];
            return (uint32_t)(win >> (64 - o - QB)) & ((1u << QB) - 1u);
        };
        uint32_t x[16], c2[16];
#pragma unroll
        for (int r = 0; r < 16; r++) {
            x[r] = mulmod_small(field(brv4(r) * LP + blp), kr[r], k);               // c1 (NTT domain, < 2^QB) o r2 -> [0, 2q)
            c2[r] = field(N + r * LP + lp);
        }
        wave_lds_fence();
        inv_core<L, MODE_MAD>(x, twi, lds_poly, lp, k, a.c0, a.c0p, a.c1, a.c1p);
        uint8_t *out = a.msgs_out + (valid ? (size_t)item * a.msg_stride : 0);
#pragma unroll
        for (int r = 0; r < 16; r++) {
            const int32_t v = (int32_t)canon(x[r] + c2[r], k);
            const unsigned long long m = __ballot(v > a.o0 && v < a.o1);             // lane = pw*LP + lp  <->  bit r*LP + lp of message pw
            const uint32_t piece = __builtin_bitreverse32((uint32_t)(m >> (pw * LP))) >> (32 - LP);   // MSB-first, LP bits
            if (valid && lp == r) {
                if (LP == 16) { out[2 * r] = (uint8_t)(piece >> 8); out[2 * r + 1] = (uint8_t)piece; }
                else { out[4 * r] = (uint8_t)(piece >> 24); out[4 * r + 1] = (uint8_t)(piece >> 16); out[4 * r + 2] = (uint8_t)(piece >> 8); out[4 * r + 3] = (uint8_t)piece; }
            }
        }
    }
}

// rlwe_enc_keygen with a_gen / r1_gen / r2_gen, rlwe_enc.c:205-290, 425-493: one key pair per polynomial slot
template <int L>
__global__ __launch_bounds__(256) void rlwe_keygen_kernel(RlArgs a)
{
    RL_PROLOGUE
    RL_TABLES
    __syncthreads();
    TwSet<L> twf;
    twf.load(a.zf, a.zfp, lp);
    unsigned long long g = (unsigned long long)blockIdx.x * 4 + wid;
    if (g >= groups) return;
    uint32_t an[16];                                              // the a-words of the group about to be processed
    auto fetch_a = [&](unsigned long long gg) {
        const unsigned long long item = gg * PW + pw;
        const bool ok = gg < groups && item < a.batch;
        const uint32_t *tape = a.tapes + (ok ? (size_t)item * a.tape_stride : 0);
#pragma unroll
        for (int r = 0; r < 16; r++) an[r] = tape[(r * LP + lp) >> 1];                        // unconditional (see load_draws)
    };
    fetch_a(g);
    for (; g < groups; g += gstride) {
        const unsigned long long item = g * PW + pw;
        const bool valid = item < a.batch;
        const uint32_t *tape = a.tapes + (valid ? (size_t)item * a.tape_stride : 0);
        uint32_t ah[16], r1h[16], x[16], rw[16];
        uint2 w[16];
        // a: 16-bit halves of n/2 words, reduced mod q (:205-225); the Gaussian draws and the r2 words in flight
#pragma unroll
        for (int r = 0; r < 16; r++) {
            const int j = r * LP + lp;
            ah[r] = reduce_i32_small((int32_t)((j & 1) ? (an[r] >> 16) : (an[r] & 0xffffu)), k);
        }
        load_draws<LP>(w, tape + N / 2, lp);
#pragma unroll
        for (int r = 0; r < 16; r++) rw[r] = tape[N / 2 + 2 * N + ((r * LP + lp) >> 4)];
        fwd_core<L, MODE_MAD, true>(ah, twf, lds_poly, lp, k);
#pragma unroll
        for (int r = 0; r < 16; r++) ah[r] = canon(ah[r], k);
        // r1: Gaussian (:227-244); the next group's a-words in flight
        sample_regs(r1h, w, S, k.q);
        fetch_a(g + gstride);
        fwd_core<L, MODE_MAD, true>(r1h, twf, lds_poly, lp, k);
        // r2: two-bit fields, 3 -> -1, 1 -> +1, else 0 (:246-285)
#pragma unroll
        for (int r = 0; r < 16; r++) {
            const int j = r * LP + lp;
            const uint32_t f = (rw[r] >> (2 * (j & 15))) & 3u;
            x[r] = (f == 3u) ? k.q - 1u : (f & 1u);
        }
        fwd_core<L, MODE_MAD, true>(x, twf, lds_poly, lp, k);
        // p = r1 - a r2 (:475-482); everything is stored as the reference stores it: SINT16, reference NTT order, canonical
        const size_t base = (size_t)item * N;
#pragma unroll
        for (int r = 0; r < 16; r++) {
            const uint32_t r2c = canon(x[r], k);
            const uint32_t t = mulmod_small(ah[r], r2c, k);                         // [0, 2q)
            const uint32_t pv = canon(r1h[r] + k.q2 - t, k);
            const int i = brv4(r) * LP + blp;
            if (valid) { a.kg_a[base + i] = (int16_t)ah[r]; a.kg_p[base + i] = (int16_t)pv; a.kg_r2[base + i] = (int16_t)r2c; }
        }
    }
}

}  // namespace

struct sc_hip_rlwe {
    sc_hip_ctx *ctx;
    sc_hip_plan *plan;
    sc_hip_cdt *cdt;
    RlP P;
    uint32_t *d_keys;         // a | p | r2, engine order
    uint16_t *d_guide;        // 4096 entries: lo | cnt << 8 (see Sampler)
    bool have_pub, have_priv;
    bool force_mad;           // sc_hip_rlwe_set_arith (test hook)
};

extern "C" void sc_hip_rlwe_destroy(sc_hip_rlwe_t *h)
{
    if (!h) return;
    if (h->d_keys) (void)sc::wipe_free(h->d_keys);
    if (h->d_guide) (void)sc::wipe_free(h->d_guide);
    if (h->cdt) sc_hip_cdt_destroy(h->cdt);
    if (h->plan) sc_hip_plan_destroy(h->plan);
    delete h;
}

extern "C" int sc_hip_rlwe_create(sc_hip_ctx_t *ctx, const sc_hip_rlwe_params_t *params, sc_hip_rlwe_t **out)
{
    if (!ctx || !params || !out) return fail("rlwe_create: null argument");
    static_assert(sizeof(RlP) == sizeof(sc_hip_rlwe_params_t), "parameter struct layout");
    RlP P; memcpy(&P, params, sizeof P);
    if (!((P.n == 256 && P.q_bits == 13) || (P.n == 512 && P.q_bits == 14)) || (1 << P.n_bits) != P.n)
        return fail("rlwe_create: (n, q_bits) must be (256, 13) or (512, 14) - the reference's parameter sets");
    if (P.q < 3 || P.q >= (1 << 14) || P.q >= (1 << P.q_bits)) return fail("rlwe_create: q must be below 2^14 and fit q_bits");
    if (P.m_scale < 0 || P.m_scale >= P.q || P.o_scale_0 < 0 || P.o_scale_1 > 65535) return fail("rlwe_create: scale parameters out of range");
    sc_hip_rlwe *h = new sc_hip_rlwe;
    memset(h, 0, sizeof *h);
    h->ctx = ctx; h->P = P;
    if (sc_hip_plan_create(ctx, P.q, P.n, &h->plan) != SC_HIP_SUCCESS) { sc_hip_rlwe_destroy(h); return SC_HIP_FAILURE; }
    if (sc_hip_cdt_create(ctx, P.tail, P.sigma, &h->cdt) != SC_HIP_SUCCESS) { sc_hip_rlwe_destroy(h); return SC_HIP_FAILURE; }
    if (h->cdt->size > MAXTAB) { sc_hip_rlwe_destroy(h); return fail("rlwe_create: tail*sigma gives a table above 256 entries"); }
    if (hipSetDevice(ctx->device) != hipSuccess || hipMalloc((void **)&h->d_keys, 3 * (size_t)P.n * 4) != hipSuccess ||
        hipMalloc((void **)&h->d_guide, 4096 * 2) != hipSuccess) { sc_hip_rlwe_destroy(h); return fail("rlwe_create: device allocation failed"); }
    {
        const std::vector<uint64_t> &cdf = h->cdt->cdf;
        const int size = h->cdt->size;
        std::vector<uint16_t> g(4096);
        int lo = 0;                                                              // max{b : cdf[b] < t 2^52}, 0 if none; monotone in t
        for (int t = 0; t < 4096; t++) {
            while (lo + 1 < size && cdf[lo + 1] < ((uint64_t)t << 52)) lo++;
            int hi = lo;                                                         // max{b : cdf[b] < (t+1) 2^52}; every entry is below 2^64
            if (t == 4095) hi = size - 1;
            else while (hi + 1 < size && cdf[hi + 1] < ((uint64_t)(t + 1) << 52)) hi++;
            g[t] = (uint16_t)(lo | ((hi - lo) << 8));
        }
        if (sc::copy_in(h->d_guide, g.data(), 4096 * 2) != hipSuccess) { sc_hip_rlwe_destroy(h); return fail("rlwe_create: cannot upload the sampler guide"); }
    }
    *out = h;
    return SC_HIP_SUCCESS;
}

// reference NTT order (SINT16, any representative) -> engine order, canonical
static void rl_to_engine(const sc_hip_rlwe *h, const int16_t *src, uint32_t *dst)
{
    const int n = h->P.n, L = h->P.n_bits, q = h->P.q;
    for (int j = 0; j < n; j++) {
        uint32_t i = 0;
        for (int b = 0; b < L; b++) i |= (((uint32_t)j >> b) & 1u) << (L - 1 - b);
        int32_t v = src[i] % q;
        dst[j] = (uint32_t)(v < 0 ? v + q : v);
    }
}

extern "C" int sc_hip_rlwe_set_pubkey(sc_hip_rlwe_t *h, const int16_t *a, const int16_t *p)
{
    if (!h || !a || !p) return fail("rlwe_set_pubkey: null argument");
    HIPCHK(hipSetDevice(h->ctx->device));
    const size_t n = (size_t)h->P.n;
    std::vector<uint32_t> t(2 * n);
    rl_to_engine(h, a, t.data()); rl_to_engine(h, p, t.data() + n);
    HIPCHK(sc::copy_in(h->d_keys, t.data(), 2 * n * 4));
    h->have_pub = true;
    return SC_HIP_SUCCESS;
}

extern "C" int sc_hip_rlwe_set_privkey(sc_hip_rlwe_t *h, const int16_t *r2)
{
    if (!h || !r2) return fail("rlwe_set_privkey: null argument");
    HIPCHK(hipSetDevice(h->ctx->device));
    const size_t n = (size_t)h->P.n;
    std::vector<uint32_t> t(n);
    rl_to_engine(h, r2, t.data());
    HIPCHK(sc::copy_in(h->d_keys + 2 * n, t.data(), n * 4));
    h->have_priv = true;
    return SC_HIP_SUCCESS;
}

extern "C" size_t sc_hip_rlwe_ct_bytes(const sc_hip_rlwe_t *h) { return h ? 2 * (size_t)h->P.n * h->P.q_bits / 8 : 0; }
extern "C" size_t sc_hip_rlwe_msg_bytes(const sc_hip_rlwe_t *h) { return h ? (size_t)h->P.n / 8 : 0; }
extern "C" size_t sc_hip_rlwe_enc_words(const sc_hip_rlwe_t *h) { return h ? 6 * (size_t)h->P.n : 0; }
extern "C" size_t sc_hip_rlwe_keygen_words(const sc_hip_rlwe_t *h) { return h ? (size_t)h->P.n / 2 + 2 * (size_t)h->P.n + (size_t)h->P.n / 16 : 0; }

static void rl_args(const sc_hip_rlwe *h, size_t batch, RlArgs &a)
{
    const sc_hip_plan *p = h->plan;
    const int n = p->n;
    memset(&a, 0, sizeof a);
    a.zf = p->d_tab[1]; a.zfp = a.zf + n; a.zi = a.zf + 2 * n; a.zip = a.zf + 3 * n;
    a.k = p->k[1];
    a.c0 = p->c_inv[1][0]; a.c0p = p->c_inv[1][1]; a.c1 = p->c_inv[1][2]; a.c1p = p->c_inv[1][3];
    if (p->smad_ok) {
        a.szf = p->d_tab_s; a.szi = p->d_tab_s + 2 * n; a.szip = p->d_tab_s + 3 * n;
        const uint32_t *c = p->q < 8192 ? p->c_inv[1] : p->c_inv_s;          // as sc_hip.hip picks them for this mode (ntt_core.h gs_last_stage)
        a.sc0 = c[0]; a.sc0p = c[1]; a.sc1 = c[2]; a.sc1p = c[3];
    }
    a.cdf = h->cdt->d_cdf; a.guide = h->d_guide; a.tab_size = h->cdt->size;
    a.key_a = h->d_keys; a.key_p = h->d_keys + n; a.key_r2 = h->d_keys + 2 * n;
    a.batch = batch;
    a.m_scale = h->P.m_scale; a.o0 = h->P.o_scale_0; a.o1 = h->P.o_scale_1;
}

// persistent waves: the grid is what the device holds at once (tables, keys and per-lane twiddles are loaded once per wave)
template <typename K>
static hipError_t rl_launch(K kernel, const sc_hip_rlwe *h, size_t batch, const RlArgs &a, void *stream)
{
    static std::atomic<int> per_cu_cache{0};                    // one static per kernel instantiation; atomic: handles of several host threads launch through here
    int per_cu = per_cu_cache.load(std::memory_order_relaxed);
    if (per_cu == 0) {
        int nb = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, kernel, 256, 0) != hipSuccess || nb < 1) nb = 2;
        per_cu = nb;
        per_cu_cache.store(nb, std::memory_order_relaxed);
    }
    const size_t pw = 1024 / (size_t)h->P.n, groups = (batch + pw - 1) / pw;
    size_t blocks = (groups + 3) / 4;
    const size_t cap = (size_t)h->ctx->cus * (size_t)per_cu;
    if (blocks > cap) blocks = cap;
    hipLaunchKernelGGL(kernel, dim3((unsigned)(blocks ? blocks : 1)), dim3(256), 0, (hipStream_t)stream, a);
    return hipGetLastError();
}

static bool aligned(const void *p, size_t a) { return (reinterpret_cast<uintptr_t>(p) & (a - 1)) == 0; }

extern "C" int sc_hip_rlwe_encrypt(sc_hip_rlwe_t *h, size_t batch, const uint8_t *d_msgs, size_t msg_stride, const uint32_t *d_tapes, size_t tape_stride,
                                   uint8_t *d_ct, size_t ct_stride, void *stream)
{
    if (!h) return fail("rlwe_encrypt: null handle");
    if (!h->have_pub) return fail("rlwe_encrypt: no public key loaded");
    if (batch == 0) return SC_HIP_SUCCESS;
    if (!d_msgs || !d_tapes || !d_ct) return fail("rlwe_encrypt: null device pointer");
    if (msg_stride < sc_hip_rlwe_msg_bytes(h) || tape_stride < sc_hip_rlwe_enc_words(h) || ct_stride < sc_hip_rlwe_ct_bytes(h)) return fail("rlwe_encrypt: stride below the record size");
    if (!aligned(d_tapes, 8) || (tape_stride & 1) || !aligned(d_ct, 4) || (ct_stride & 3) || !aligned(d_msgs, 4) || (msg_stride & 3))
        return fail("rlwe_encrypt: tapes must be 8-byte aligned (even stride), messages and ciphertexts 4-byte aligned (strides multiples of 4)");
    HIPCHK(hipSetDevice(h->ctx->device));
    RlArgs a; rl_args(h, batch, a);
    a.msgs_in = d_msgs; a.msg_stride = msg_stride; a.tapes = d_tapes; a.tape_stride = tape_stride; a.ct_out = d_ct; a.ct_stride = ct_stride;
    if (h->plan->smad_ok && !h->force_mad) HIPCHK(h->P.n == 256 ? rl_launch(rlwe_encrypt_kernel<8, 13, MODE_SMAD>, h, batch, a, stream) : rl_launch(rlwe_encrypt_kernel<9, 14, MODE_SMAD>, h, batch, a, stream));
    else HIPCHK(h->P.n == 256 ? rl_launch(rlwe_encrypt_kernel<8, 13, MODE_MAD>, h, batch, a, stream) : rl_launch(rlwe_encrypt_kernel<9, 14, MODE_MAD>, h, batch, a, stream));
    return SC_HIP_SUCCESS;
}

extern "C" int sc_hip_rlwe_set_arith(sc_hip_rlwe_t *h, int mode)
{
    if (!h) return fail("rlwe_set_arith: null handle");
    if (mode != 0 && mode != MODE_MAD) return fail("rlwe_set_arith: mode must be 0 (automatic) or 3 (unsigned multiply-add)");
    h->force_mad = mode == MODE_MAD;
    return SC_HIP_SUCCESS;
}

extern "C" int sc_hip_rlwe_decrypt(sc_hip_rlwe_t *h, size_t batch, const uint8_t *d_ct, size_t ct_stride, uint8_t *d_msgs, size_t msg_stride, void *stream)
{
    if (!h) return fail("rlwe_decrypt: null handle");
    if (!h->have_priv) return fail("rlwe_decrypt: no private key loaded");
    if (batch == 0) return SC_HIP_SUCCESS;
    if (!d_ct || !d_msgs) return fail("rlwe_decrypt: null device pointer");
    if (msg_stride < sc_hip_rlwe_msg_bytes(h) || ct_stride < sc_hip_rlwe_ct_bytes(h)) return fail("rlwe_decrypt: stride below the record size");
    if (!aligned(d_ct, 4) || (ct_stride & 3)) return fail("rlwe_decrypt: ciphertexts must be 4-byte aligned");
    HIPCHK(hipSetDevice(h->ctx->device));
    RlArgs a; rl_args(h, batch, a);
    a.ct_in = d_ct; a.ct_stride = ct_stride; a.msgs_out = d_msgs; a.msg_stride = msg_stride;
    HIPCHK(h->P.n == 256 ? rl_launch(rlwe_decrypt_kernel<8, 13>, h, batch, a, stream) : rl_launch(rlwe_decrypt_kernel<9, 14>, h, batch, a, stream));
    return SC_HIP_SUCCESS;
}

extern "C" int sc_hip_rlwe_keygen(sc_hip_rlwe_t *h, size_t batch, const uint32_t *d_tapes, size_t tape_stride, int16_t *d_a, int16_t *d_p, int16_t *d_r2, void *stream)
{
    if (!h) return fail("rlwe_keygen: null handle");
    if (batch == 0) return SC_HIP_SUCCESS;
    if (!d_tapes || !d_a || !d_p || !d_r2) return fail("rlwe_keygen: null device pointer");
    if (tape_stride < sc_hip_rlwe_keygen_words(h)) return fail("rlwe_keygen: tape stride below the words a key pair consumes");
    if (!aligned(d_tapes, 8) || (tape_stride & 1)) return fail("rlwe_keygen: tapes must be 8-byte aligned with an even stride");
    HIPCHK(hipSetDevice(h->ctx->device));
    RlArgs a; rl_args(h, batch, a);
    a.tapes = d_tapes; a.tape_stride = tape_stride; a.kg_a = d_a; a.kg_p = d_p; a.kg_r2 = d_r2;
    HIPCHK(h->P.n == 256 ? rl_launch(rlwe_keygen_kernel<8>, h, batch, a, stream) : rl_launch(rlwe_keygen_kernel<9>, h, batch, a, stream));
    return SC_HIP_SUCCESS;
}
