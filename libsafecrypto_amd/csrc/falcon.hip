// falcon.hip - batched Falcon signing (fast Fourier sampling over the LDL tree) and verification on MI355X
// behind the C ABI of include/sc_hip.h.  Binary rings only (parameter sets 0 and 2, x^n + 1, n = 512 / 1024).
//
// Replaces, for a batch of messages under one key:
//   gaussian_sample_with_tree      src/utils/arith/gpv.c:2961-3090 (ter == 0 branch)
//   gaussian_lattice_sample_fft    src/utils/arith/gpv.c:2663-2714
//   falcon_FFT / iFFT / poly_*     src/utils/arith/falcon_fft.c:203-378, 380-424, 466-483, 543-553, 694-785
//   falcon_sig_sign / _verify      src/schemes/sig/falcon/falcon.c:808-935, 955-1040 (hash to ring :534-591, norm :511-521)
//
// Arithmetic is IEEE-754 binary64 with one rounding per operation in the reference's operation order (the reference
// is built without FMA contraction, hence -ffp-contract=off / the pragma below): s1 and s2 come out bit for bit.
//
// Mapping: ONE LANE PER SIGNATURE.  The sampler is a depth-first walk over a binary tree whose two halves depend on
// each other through the Gaussian samples - there is no parallelism inside a signature below the top few levels, but a
// batch is embarrassingly parallel and every lane of a wavefront takes the same path (the tree shape is fixed, the
// sampler is branch-free), so 64 signatures run in lockstep without divergence.  Working polynomials (6n doubles per
// signature) live in HBM interleaved across the batch - element e of signature b at ws[e*B + b] - so every access of a
// wavefront is one contiguous 512-byte row; key material (basis, tree, roots) is read at wave-uniform addresses.
// The recursion is unrolled by hand into a loop over levels; the passes the reference makes between the two recursive
// calls (merge, (t1 - z1)*L + t0, split) are fused into one sweep, and the bottom 8- or 16-coefficient sub-trees run
// entirely in registers.  HBM traffic per signature is about 9n doubles per tree level above the register block.
//
// Randomness: each tree leaf builds a throw-away 64-bit CDF sampler with sigma = (float)leaf and draws two samples
// (two prng_64 = four 32-bit CSPRNG pool words, high word first, prng.c:963-976).  The engine owns no CSPRNG: each
// message travels with that word stream (4n words), and the n per-leaf CDF tables are tabulated on the host when the
// key is loaded (same long-double code path as sc_hip_cdt_build_table), in visiting order.
#pragma clang fp contract(off)
#include "engine_internal.h"
#include "keccak_dev.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

using namespace sc;

namespace {

#include "falcon_sign_dev.h"

#include "falcon_verify_dev.h"

// exp(i*pi*j/2048), j = 0..2048, in binary128 (Taylor series for the first step, then repeated multiplication); GM[k] for
// k in [m, 2m), m = 2^logm, is the root with angle pi*(2*rev_logm(k-m)+1)/(2m), rounded to double - this regenerates the
// reference's fpr_gm_tab (falcon_fft.h) bit for bit, including its (-0.0, 1.0) spelling of entry 1.
void make_gm(std::vector<double> &gm)
{
    typedef __float128 q_t;
    const q_t pi = (q_t)3.141592653589793116 + (q_t)1.2246467991473532e-16 + (q_t)(-2.9947698097183397e-33);
    const q_t x = pi / 2048, x2 = x * x;
    q_t ts = x, tc = 1, s1 = x, c1 = 1;
    for (int k = 1; k < 16; k++) {
        tc = -tc * x2 / (q_t)((2 * k - 1) * (2 * k)); c1 += tc;
        ts = -ts * x2 / (q_t)((2 * k) * (2 * k + 1)); s1 += ts;
    }
    std::vector<q_t> cr(2049), ci(2049);
    cr[0] = 1; ci[0] = 0;
    for (int j = 1; j <= 2048; j++) { cr[j] = cr[j - 1] * c1 - ci[j - 1] * s1; ci[j] = cr[j - 1] * s1 + ci[j - 1] * c1; }
    gm.assign(2048, 0.0);
    for (int logm = 0; logm < 10; logm++) {
        const unsigned m = 1u << logm;
        for (unsigned i1 = 0; i1 < m; i1++) {
            unsigned rev = 0;
            for (int bb = 0; bb < logm; bb++) rev |= ((i1 >> bb) & 1u) << (logm - 1 - bb);
            const unsigned k = m + i1, j = (2 * rev + 1) * (2048 / (2 * m));
            gm[2 * k] = (double)cr[j]; gm[2 * k + 1] = (double)ci[j];
        }
    }
    gm[2] = -0.0; gm[3] = 1.0;
}

// dlp_ibe_extract's output coding (dlp_ibe.c:991-1006): v centred (already a small signed integer), ubits bits per coefficient,
// two's complement, MSB first.  One thread per eight coefficients = ubits bytes.
// status: the sampler leaves 2 where the tape ran out; everything else is packed and marked 1.
__global__ void dlp_pack_kernel(uint32_t batch, const int32_t *__restrict__ v, uint32_t *__restrict__ status, uint8_t *__restrict__ sk, size_t sk_stride, int n, int ubits)
{
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int per = n / 8;                                       // a power of two
    const uint32_t i = (uint32_t)(t >> sc::pow2_shift(per)); const int g = (int)(t & (size_t)(per - 1));
    if (i >= batch || status[i] == 2) return;
    const int32_t *src = v + (size_t)i * n + 8 * g;
    uint8_t *o = sk + (size_t)i * sk_stride + (size_t)g * ubits;
    uint64_t acc = 0; int nb = 0, w = 0;
    const uint32_t mask = (1u << ubits) - 1u;
    for (int e = 0; e < 8; e++) {
        acc = (acc << ubits) | (uint64_t)((uint32_t)src[e] & mask); nb += ubits;
        while (nb >= 8) { o[w++] = (uint8_t)(acc >> (nb - 8)); nb -= 8; }
    }
    if (g == 0) status[i] = 1;
}

// ---- ENS / DLP signing: Klein's sampler over the expanded basis (gpv.c:2409-2492, gaussian_lattice_sample_flt, flags = 0) -------------------
//
// From the last basis row down:  d = <ci, b_gs[j]> / ||b_gs[j]||^2,  z = CDT_j(prng_64) + (SINT32)d,  ci -= z b[j].  The dot product is a
// SEQUENTIAL single-precision sum over k = 0 .. 2n-1 of (FLOAT)ci[k] * b_gs[j][k] (dot_s64_flt, gpv.c:843-855) and single-precision addition does
// not associate, so one lane owns one signature and walks k in order; what is parallel is the batch.  The update of row j and the dot product of
// row j-1 are one sweep over ci (one read and one write of the 2n-word state per row).  ci is held as int32 [2n][B] (a wavefront reads 256
// contiguous bytes per k); the reference's SINT64 is never needed for a valid key, and that is CHECKED: each lane carries a bound M >= max|ci|,
// raised by |z| max|b| per row, recomputed exactly from the state when it nears 2^31, and a lane whose state really would leave int32 stops with
// status 3 instead of wrapping.  Basis row, Gram-Schmidt row, 1/||.|| and the row's CDT are wave-uniform: scalar loads.
struct KleinArgs {
    uint32_t batch; uint32_t B;            // signatures, padded column count (a multiple of 64)
    int n, n2;
    const int32_t *__restrict__ c;         // [batch][n]: H(m)
    const int32_t *__restrict__ b;         // [2n][2n]
    const float *__restrict__ bgs;         // [2n][2n]
    const float *__restrict__ inv;         // [2n]
    const uint64_t *__restrict__ tabs;     // [2n] << ts_log
    const uint32_t *__restrict__ sizes;    // [2n]
    int ts_log;
    uint32_t maxb;
    const uint32_t *__restrict__ tapes; uint32_t tape_stride; const uint32_t *__restrict__ tape_lens;
    int32_t *__restrict__ ci;              // [2n][B]
    uint32_t *__restrict__ status;
    int want_s2;                           // the register form writes ci[0, n) as well (signatures with message recovery send s2 = centre(ci[0, n)) too)
};

constexpr int KU = 16;                     // sweep unroll: loads in flight per lane

__global__ __launch_bounds__(64) void klein_kernel(KleinArgs A)
{
    const uint32_t col = blockIdx.x * 64 + threadIdx.x;
    const bool valid = col < A.batch;
    const uint32_t item = valid ? col : 0;                      // padding lanes rerun item 0 in their own columns and report nothing
    const int n = A.n, n2 = A.n2;
    const size_t B = A.B;
    int32_t *__restrict__ ci = A.ci + col;
    const uint32_t *tape = A.tapes + (size_t)item * A.tape_stride;
    const bool short_tape = A.tape_lens && A.tape_lens[item] < 2u * (uint32_t)n2;

    float dot = 0.f;
    {
        const int32_t *cr = A.c + (size_t)item * n;
        const float *g = A.bgs + (size_t)(n2 - 1) * n2;
        for (int k = 0; k < n; k++) { const int32_t v = cr[k]; ci[(size_t)k * B] = v; dot += (float)v * g[k]; }
        for (int k = n; k < n2; k++) { ci[(size_t)k * B] = 0; dot += 0.f * g[k]; }
    }
    uint64_t M = 0x7fffffffu >> 1;                               // H(m) < 2^q_bits + q < 2^30
    bool dead = false;
    int32_t cur[KU], nxt[KU];                                    // run-ahead: chunk k + KU is requested before chunk k is worked on,
#pragma unroll                                                   // and the next row's first chunk before this row's sample is drawn
    for (int u = 0; u < KU; u++) cur[u] = ci[(size_t)u * B];
    for (int j = n2 - 1; j >= 0; j--) {
        const float inv = A.inv[j];
        const float d = dot * inv * inv;
        const uint2 w = *reinterpret_cast<const uint2 *>(tape + 2 * (size_t)(n2 - 1 - j));
        const uint64_t x = ((uint64_t)w.x << 32) | w.y;
        const uint64_t *__restrict__ cdf = A.tabs + ((size_t)j << A.ts_log);
        const int size = (int)A.sizes[j];
        uint32_t a = 0;
        for (int e = 1; e < size; e++) a += (cdf[e] < x) ? 1u : 0u;
        int32_t z = ((x & 1) ? (int32_t)a : -(int32_t)a) + (int32_t)d;
        const uint64_t az = (uint64_t)(z < 0 ? -(int64_t)z : (int64_t)z) * A.maxb;
        bool risky = !dead && M + az > 0x7fffffffull;
        if (__any(risky)) {                                      // rare: replace the running bound by the exact maximum
            uint32_t m = 0;
            for (int k = 0; k < n2; k++) { const int32_t v = ci[(size_t)k * B]; const uint32_t av = (uint32_t)(v < 0 ? -v : v); m = av > m ? av : m; }
            if (risky) { M = m; if (M + az > 0x7fffffffull) dead = true; }
        }
        if (dead) z = 0;
        M += dead ? 0 : az;
        // read through a float lvalue so that the stores to ci (int) cannot alias it and the row stays on the scalar unit
        const float *__restrict__ brow = reinterpret_cast<const float *>(A.b) + (size_t)j * n2;
        const float *__restrict__ grow = A.bgs + (size_t)(j > 0 ? j - 1 : 0) * n2;
        dot = 0.f;
        for (int k = 0; k < n2; k += 2 * KU) {                   // two chunks per trip, two register sets: no copies between them
#pragma unroll
            for (int u = 0; u < KU; u++) nxt[u] = ci[(size_t)(k + KU + u) * B];
#pragma unroll
            for (int u = 0; u < KU; u++) {
                const int32_t v = cur[u] - z * __float_as_int(brow[k + u]);
                ci[(size_t)(k + u) * B] = v;
                dot += (float)v * grow[k + u];
            }
            const int kn = (k + 2 * KU < n2) ? k + 2 * KU : 0;   // after the last chunk: the first chunk (already final for this row)
#pragma unroll
            for (int u = 0; u < KU; u++) cur[u] = ci[(size_t)(kn + u) * B];
#pragma unroll
            for (int u = 0; u < KU; u++) {
                const int32_t v = nxt[u] - z * __float_as_int(brow[k + KU + u]);
                ci[(size_t)(k + KU + u) * B] = v;
                dot += (float)v * grow[k + KU + u];
            }
        }
    }
    if (valid) A.status[col] = short_tape ? 2u : dead ? 3u : 1u;
}

// The production form: G lanes per signature, the sampler state held in REGISTERS.  The lane-per-signature kernel above needs 64 signatures per
// wavefront (16384 signatures are 256 wavefronts on 1024 SIMDs) and streams the whole state through HBM once per basis row.  Here lane r of a
// signature owns the elements k = r mod G: the update, the int -> float conversion and the product with the Gram-Schmidt row run G wide, and only
// the ADDITIONS keep the reference's order - one lane of the signature adds the G products of a group of elements one after the other (v_add_f32
// with a row_ror DPP operand, G dependent instructions per G elements; chain_g below).  A lane's share of the 2n-word state is 2n/G = NCH words of registers, so the sweep touches no memory
// of its own and only s1 is written at the end.  G = 16, 8 or 4: the lanes of a signature sit 16/G apart inside a DPP row (16/G signatures
// interleaved per row, 64/G per wavefront) and the running sum moves with row_ror:(16/G), so each ring closes on itself.  The DPP
// additions are the cost (every wavefront spends 2n of them per basis row whatever G is), so fewer lanes per signature = fewer wavefronts = less
// work, as long as enough wavefronts remain to fill the SIMDs: the host picks G from the batch size, bounded below by 128 state registers per lane.
// The basis row and the Gram-Schmidt row of a step are the same 16n bytes for every signature and known in advance: the wavefront fetches the NEXT
// step's rows with 16-byte loads at the top of a step, parks them in registers while the sweep runs out of LDS (a double buffer of 2 x 16n bytes
// shared by the workgroup's four wavefronts, each fetching a quarter), and writes them to the other buffer afterwards - no load is waited on inside the sweep, which is fully unrolled (static
// register indices).
// Round 3: the running sum no longer travels.  It stays in the signature's LAST lane, which adds the products of lanes 0, 1, ..., G-1 in that order,
// each fetched with a row_ror by the distance to its owner (the DPP operand is the product, computed well ahead, not the sum): the same additions
// in the same order in that lane - the other lanes add rotated copies nobody reads - but the dependent operand of every addition is a plain
// register.  With the sum as the DPP operand every one of the 2n additions per row waited out the two wait states a DPP read of a just-written
// register needs (an s_nop between any two of them: ~15 cycles per addition for a lone wavefront, ~5 now).
// Written as one block of assembly per product: left to itself the compiler computes all G - 1 rotated copies of every product ahead of the chain
// (v_mov_b32_dpp into registers of their own - several hundred spills) instead of folding each rotation into its addition.  The s_nop covers the two
// wait states between the instruction that produced p and the first DPP read of it (the compiler does not look inside the block).
#define SC_ROR_ADD(n) "v_add_f32_dpp %0, %1, %0 row_ror:" #n " row_mask:0xf bank_mask:0xf\n\t"
template <int G>
__device__ __forceinline__ float chain_g(float acc, float p)
{
    static_assert(G == 4 || G == 8 || G == 16, "lanes per signature");
    if constexpr (G == 4)
        asm("s_nop 1\n\t" SC_ROR_ADD(12) SC_ROR_ADD(8) SC_ROR_ADD(4) "v_add_f32 %0, %0, %1" : "+v"(acc) : "v"(p));
    else if constexpr (G == 8)
        asm("s_nop 1\n\t" SC_ROR_ADD(14) SC_ROR_ADD(12) SC_ROR_ADD(10) SC_ROR_ADD(8) SC_ROR_ADD(6) SC_ROR_ADD(4) SC_ROR_ADD(2) "v_add_f32 %0, %0, %1" : "+v"(acc) : "v"(p));
    else
        asm("s_nop 1\n\t" SC_ROR_ADD(15) SC_ROR_ADD(14) SC_ROR_ADD(13) SC_ROR_ADD(12) SC_ROR_ADD(11) SC_ROR_ADD(10) SC_ROR_ADD(9) SC_ROR_ADD(8) SC_ROR_ADD(7) SC_ROR_ADD(6)
            SC_ROR_ADD(5) SC_ROR_ADD(4) SC_ROR_ADD(3) SC_ROR_ADD(2) SC_ROR_ADD(1) "v_add_f32 %0, %0, %1" : "+v"(acc) : "v"(p));
    return acc;                                                  // in the signature's last lane: acc_in + p[0] + ... + p[G-1], left to right
}
#undef SC_ROR_ADD

constexpr int KW = 4;                      // wavefronts per workgroup: they share the LDS copy of the step's rows (loaded once, a quarter each)

template <int G, int NCH, bool PF>
__global__ __launch_bounds__(64 * KW) void kleinr_kernel(KleinArgs A)
{
    constexpr int n2 = NCH * G, n = n2 / 2, S = 16 / G;          // S signatures per DPP row
    constexpr int RV = n2 / 4, NV = (RV + 64 * KW - 1) / (64 * KW);      // float4 per row; per thread
    __shared__ float4 rows[2][2][RV];                            // [buffer][basis | Gram-Schmidt][k / 4]
    const uint32_t tid = threadIdx.x, lane = tid & 63, r = (lane & 15) / S;
    const uint32_t col = (blockIdx.x * KW + (tid >> 6)) * (64 / G) + (lane >> 4) * S + (lane & (S - 1));
    const bool valid = col < A.batch;
    const uint32_t item = valid ? col : 0;
    const uint32_t *tape = A.tapes + (size_t)item * A.tape_stride;
    const bool short_tape = A.tape_lens && A.tape_lens[item] < 2u * (uint32_t)n2;
    const float4 *__restrict__ bmat = reinterpret_cast<const float4 *>(A.b);      // int bit patterns, moved as floats
    const float4 *__restrict__ gmat = reinterpret_cast<const float4 *>(A.bgs);
    uint64_t ring = 0;                                           // the lanes of this signature
#pragma unroll
    for (int i = 0; i < G; i++) ring |= 1ull << ((lane & 48) + (lane & (S - 1)) + S * i);

    int32_t st[NCH];
    float dot = 0.f;
    {
        const int32_t *cr = A.c + (size_t)item * n + r;
        const float *g = A.bgs + (size_t)(n2 - 1) * n2 + r;
        asm volatile("s_nop 4");
#pragma unroll
        for (int c = 0; c < NCH; c++) {
            st[c] = (c < NCH / 2) ? cr[G * c] : 0;
            dot = chain_g<G>(dot, (float)st[c] * g[G * c]);
        }
#pragma unroll
        for (int i = 0; i < NV; i++) {
            const uint32_t e = tid + 64 * KW * i;
            if (e < RV) { rows[1][0][e] = bmat[(size_t)(n2 - 1) * RV + e]; rows[1][1][e] = gmat[(size_t)(n2 - 2) * RV + e]; }
        }
        __syncthreads();
    }
    uint64_t M = 0x7fffffffu >> 1;
    bool dead = false;
    // The row's Gaussian draw depends on nothing the sweep computes.  PF: tape word and table entries of step j - 1 are requested at the top of step j
    // and turned into the sample behind its sweep - fetched and searched at the top of their own step (the other form) they are two exposed memory
    // round trips (the tape from HBM, the table from L2) at the head of every one of the 2n dependent steps.  The search is shared by the
    // signature's G lanes: lane r compares entries r, r + G, ... (64 / G of them in registers; tables are padded with all-ones, and entry 0 is not
    // part of the search: gaussian_cdf_sample_64 counts b >= 1), and three or four rotate-and-add steps within the DPP row leave the count in every
    // lane.  The price is 2 * 64 / G + 2 registers held across the sweep (n = 512, 8 lanes: 167 -> 198, two wavefronts per SIMD instead of three);
    // measurements of both forms are in sc_hip_ens_dlp_sig_sign.
    constexpr int NE = 64 / G;
    const uint32_t TS = 1u << A.ts_log;
    auto ring_sum = [](uint32_t v) -> uint32_t {                  // row_ror by S, 2S, ... 8: every lane ends with the sum over its signature's G lanes
        if constexpr (S <= 1) v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x121, 0xF, 0xF, false);
        if constexpr (S <= 2) v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x122, 0xF, 0xF, false);
        if constexpr (S <= 4) v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x124, 0xF, 0xF, false);
        v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x128, 0xF, 0xF, false);
        return v;
    };
    uint2 wc = *reinterpret_cast<const uint2 *>(tape);              // step n2 - 1: first draw of the tape
    float invc = A.inv[n2 - 1];
    uint32_t ac = 0;
    if constexpr (PF) {
        const uint64_t x = ((uint64_t)wc.x << 32) | wc.y;
        const uint64_t *__restrict__ cdf = A.tabs + ((size_t)(n2 - 1) << A.ts_log);
        uint32_t cnt = 0;
        for (uint32_t e = r; e < TS; e += G) cnt += (e != 0 && cdf[e] < x) ? 1u : 0u;
        ac = ring_sum(cnt);
    }
#pragma unroll 1
    for (int j = n2 - 1; j >= 0; j--) {
        const int buf = j & 1;                                   // n2 - 1 is odd: the first step reads buffer 1
        float4 pb[NV], pg[NV];                                   // rows of step j - 1: basis row j - 1, Gram-Schmidt row j - 2
        const size_t rb = (size_t)(j > 0 ? j - 1 : 0) * RV, rg = (size_t)(j > 1 ? j - 2 : 0) * RV;
#pragma unroll
        for (int i = 0; i < NV; i++) { const uint32_t e = (tid + 64 * KW * i) % RV; pb[i] = bmat[rb + e]; pg[i] = gmat[rg + e]; }     // unconditional (wraps when RV < 256)
        const int jn = j > 0 ? j - 1 : 0;                        // the next step's draw (step 0 repeats its own: unused)
        const uint64_t *__restrict__ cdfn = A.tabs + ((size_t)(PF ? jn : j) << A.ts_log);
        uint2 wn;
        uint64_t ce[NE];
        if constexpr (PF) {
            wn = *reinterpret_cast<const uint2 *>(tape + 2 * (size_t)(n2 - 1 - jn));
#pragma unroll
            for (int i = 0; i < NE; i++) ce[i] = cdfn[(r + G * i) & (TS - 1)];
        } else wc = *reinterpret_cast<const uint2 *>(tape + 2 * (size_t)(n2 - 1 - j));
        const float inv = invc;
        invc = A.inv[jn];
        const float d = __shfl(dot, (int)(lane | (16 - S))) * inv * inv;
        const uint64_t x = ((uint64_t)wc.x << 32) | wc.y;
        if constexpr (!PF) {                                     // every lane walks the row's table, in its own step (no registers held across the sweep)
            const int size = (int)A.sizes[j];
            ac = 0;
            for (int e = 1; e < size; e++) ac += (cdfn[e] < x) ? 1u : 0u;
        }
        const uint32_t a = ac;
        int32_t z = ((x & 1) ? (int32_t)a : -(int32_t)a) + (int32_t)d;
        const uint64_t az = (uint64_t)(z < 0 ? -(int64_t)z : (int64_t)z) * A.maxb;
        bool risky = !dead && M + az > 0x7fffffffull;
        if (__any(risky)) {                                      // rare: replace the running bound by the exact maximum of this lane's elements
            uint32_t m = 0;
#pragma unroll
            for (int c = 0; c < NCH; c++) { const int32_t v = st[c]; const uint32_t av = (uint32_t)(v < 0 ? -v : v); m = av > m ? av : m; }
            if (risky) M = m;
            if (__ballot(risky && M + az > 0x7fffffffull) & ring) dead = true;
        }
        if (dead) z = 0;
        M += dead ? 0 : az;
        const float *__restrict__ brow = reinterpret_cast<const float *>(rows[buf][0]) + r;
        const float *__restrict__ grow = reinterpret_cast<const float *>(rows[buf][1]) + r;
        dot = 0.f;
        asm volatile("s_nop 4");                                 // chain_g's DPP reads are invisible to the compiler's hazard recogniser: the five wait states a DPP instruction needs behind a write of EXEC (the divergent stores above) - the sweep itself is straight-line
#pragma unroll
        for (int c = 0; c < NCH; c++) {
            st[c] -= z * __float_as_int(brow[G * c]);
            dot = chain_g<G>(dot, (float)st[c] * grow[G * c]);
        }
        if constexpr (PF) {
            const uint64_t xn = ((uint64_t)wn.x << 32) | wn.y;
            uint32_t cnt = 0;
#pragma unroll
            for (int i = 0; i < NE; i++) { const uint32_t e = r + G * i; cnt += (e != 0 && e < TS && ce[i] < xn) ? 1u : 0u; }
            for (uint32_t e = 64 + r; e < TS; e += G) cnt += (cdfn[e] < xn) ? 1u : 0u;      // tables above 64 entries: the rest, fetched on the spot
            ac = ring_sum(cnt);
            wc = wn;
        }
#pragma unroll
        for (int i = 0; i < NV; i++) {
            const uint32_t e = tid + 64 * KW * i;
            if (e < RV) { rows[buf ^ 1][0][e] = pb[i]; rows[buf ^ 1][1][e] = pg[i]; }
        }
        __syncthreads();
    }
    if (valid) {
        int32_t *__restrict__ out = A.ci + col;                  // [2n][B]; only s1 = the second half is needed (with message recovery: both)
#pragma unroll
        for (int c = 0; c < NCH; c++) if (c >= NCH / 2 || A.want_s2) out[(size_t)(G * c + r) * A.B] = st[c];
        if (r == 0) A.status[col] = short_tape ? 2u : dead ? 3u : 1u;
    }
}

// s1 = centre(ci[n, 2n)) (ens_dlp_sig.c:1273), n q_bits-bit two's complement fields, MSB first (entropy_poly_encode_32, SC_ENTROPY_NONE).
// One thread per eight coefficients = q_bits bytes.  Signatures whose status is not 1 are zero-filled.
// With message recovery (grid.y = 2n / 8) the second n fields are s2 = centre(ci[0, n)) (ens_dlp_sig.c:1434-1445: one packer stream).
__global__ void klein_pack_kernel(uint32_t batch, uint32_t B, const int32_t *__restrict__ ci, const uint32_t *__restrict__ status, uint8_t *__restrict__ sigs, size_t sig_stride,
                                  int n, int q, int q_bits)
{
    // grid: x over the signatures (consecutive threads = consecutive signatures: coalesced reads of the state), y over the n / 8 groups
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    const int g = (int)blockIdx.y;
    if (i >= batch) return;
    const bool ok = status[i] == 1;
    uint8_t *o = sigs + (size_t)i * sig_stride + (size_t)g * q_bits;
    uint64_t acc = 0; int nb = 0, w = 0;
    const uint32_t mask = (1u << q_bits) - 1u;
    const int first = 8 * g < n ? n + 8 * g : 8 * g - n;          // s1 = the state's second half, then s2 = its first
    for (int e = 0; e < 8; e++) {
        const int32_t v = ok ? centre_mod(ci[(size_t)(first + e) * B + i], q) : 0;
        acc = (acc << q_bits) | (uint64_t)((uint32_t)v & mask); nb += q_bits;
        while (nb >= 8) { o[w++] = (uint8_t)(acc >> (nb - 8)); nb -= 8; }
    }
}

}  // namespace

struct sc_hip_falcon {
    sc_hip_ctx *ctx;
    sc_hip_plan *plan;
    FalP P;
    bool have_sk, have_pk;
    double *d_sk, *d_gm;
    uint64_t *d_tabs; int ts_log;
    size_t nkeys, npk;           // keys resident: d_sk [nkeys][(n_bits+5) n], d_tabs [nkeys][n << ts_log], d_hhat [npk][n]; key 0 is "the" key of the one-key calls
    int var_spw, var_lm, var_nw; // sc_hip_falcon_set_variant (measurement hook): 0 = chosen per batch
    int force_g;                 // sc_hip_falcon_set_mapping: 0 / 64 = 8 signatures per wavefront, lower tree in LDS; 1 / 8 = lanes per signature, state in HBM
    int32_t *d_hhat;
    int32_t *d_hraw;             // key 0 as it was handed over (SINT16 widened, not reduced): sc_hip_ens_dlp_sig_verify_recovery multiplies by it in wrapping 32-bit words
    size_t cap;
    double *ws;
    int32_t *c, *s1, *s2;
    uint32_t *status, *siglens;
    sc::Workspace srt;           // sc_hip_falcon_sign_keys with the keys of the batch interleaved: the batch regrouped by key (fal_sign_grouped)
    sc::Workspace stape;         // sc_hip_falcon_sign_seeded: the batch's CSPRNG word tapes, expanded from the caller's seeds for the length of the call
    hipStream_t side = nullptr;  // ... on a stream of the handle's own, beside the hash to the ring (forked / joined with events inside the call)
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;
};

static void fal_free_ws(sc_hip_falcon *f)
{
    void *ptrs[] = { f->ws, f->c, f->s1, f->s2, f->status, f->siglens };
    for (void *p : ptrs) if (p) (void)sc::wipe_free(p);
    f->ws = nullptr; f->c = f->s1 = f->s2 = nullptr; f->status = f->siglens = nullptr; f->cap = 0;
}

// the sampler's working polynomials are products with the secret basis, s1 | s2 the lattice point before packing: zeroed behind a signing call
static void fal_wipe(sc_hip_falcon *f, size_t batch, hipStream_t s)
{
    const size_t n = f->P.n;
    if (f->ws) (void)hipMemsetAsync(f->ws, 0, ((batch + SPW_MAX - 1) / SPW_MAX * SPW_MAX) * 6 * n * sizeof(double), s);
    if (f->s1) (void)hipMemsetAsync(f->s1, 0, batch * n * 4, s);
    if (f->s2) (void)hipMemsetAsync(f->s2, 0, batch * n * 4, s);
}

static int fal_ensure_ws(sc_hip_falcon *f, size_t batch)
{
    if (batch <= f->cap) return SC_HIP_SUCCESS;
    fal_free_ws(f);
    const size_t n = f->P.n;
    HIPCHK(hipMalloc((void **)&f->ws, ((batch + SPW_MAX - 1) / SPW_MAX * SPW_MAX) * 6 * n * sizeof(double)));       // the 8-per-wavefront mapping pads to whole wavefronts
    HIPCHK(hipMalloc((void **)&f->c, batch * n * 4));
    HIPCHK(hipMalloc((void **)&f->s1, batch * n * 4));
    HIPCHK(hipMalloc((void **)&f->s2, batch * n * 4));
    HIPCHK(hipMalloc((void **)&f->status, batch * 4));
    HIPCHK(hipMalloc((void **)&f->siglens, batch * 4));
    f->cap = batch;
    return SC_HIP_SUCCESS;
}

extern "C" int sc_hip_falcon_create(sc_hip_ctx_t *ctx, const sc_hip_falcon_params_t *params, sc_hip_falcon_t **out)
{
    if (!ctx || !params || !out) return fail("falcon_create: null argument");
    static_assert(sizeof(FalP) == sizeof(sc_hip_falcon_params_t), "parameter struct layout");
    FalP P; memcpy(&P, params, sizeof P);
    if (P.set == 1) return fail("falcon_create: the ternary ring (set 1, x^n - x^(n/2) + 1) is not served");
    // n = 256 exists for the DLP signature set 0 verifier (ens_dlp_sig_params.c:45-53) and loads a public key only
    if ((P.n != 256 && P.n != 512 && P.n != 1024) || (1 << P.n_bits) != P.n) return fail("falcon_create: n must be 256 (verify only), 512 or 1024 with n_bits = log2 n");
    if (P.q_bits < 4 || P.q_bits > 30 || P.q < 3 || P.q >= (1 << P.q_bits)) return fail("falcon_create: bad q / q_bits");
    sc_hip_plan *plan = nullptr;
    if (sc_hip_plan_create(ctx, P.q, P.n, &plan) != SC_HIP_SUCCESS) return SC_HIP_FAILURE;
    HIPCHK(hipSetDevice(ctx->device));
    sc_hip_falcon *f = new sc_hip_falcon;
    memset(f, 0, sizeof *f);
    f->ctx = ctx; f->plan = plan; f->P = P;
    std::vector<double> gm;
    make_gm(gm);
    if (hipMalloc((void **)&f->d_gm, gm.size() * sizeof(double)) != hipSuccess ||
        sc::copy_in(f->d_gm, gm.data(), gm.size() * sizeof(double)) != hipSuccess) {
        sc_hip_falcon_destroy(f);
        return fail("falcon_create: device allocation failed");
    }
    *out = f;
    return SC_HIP_SUCCESS;
}

extern "C" int sc_hip_falcon_set_mapping(sc_hip_falcon_t *f, int lanes)
{
    if (!f) return fail("falcon_set_mapping: null handle");
    if (lanes != 0 && lanes != 1 && lanes != 8 && lanes != 64) return fail("falcon_set_mapping: lanes must be 0 (default), 1, 8 or 64");
    f->force_g = lanes;
    return SC_HIP_SUCCESS;
}

// measurement hook (include/sc_hip_bench.h): pin the default mapping's variant - signatures per wavefront (8 / 4) and the sub-tree level kept in LDS (7 / 6); 0, 0 = per batch
extern "C" int sc_hip_falcon_set_variant(sc_hip_falcon_t *f, int spw, int lm)
{
    if (!f) return fail("falcon_set_variant: null handle");
    { const int sp = spw & 0xFF, nwv = spw >> 8;
      if (!((sp == 0 && lm == 0) || (sp == 8 && (lm == 6 || lm == 7)) || (sp == 4 && lm == 6)) || nwv < 0 || nwv > 2) return fail("falcon_set_variant: (8, 7), (8, 6), (4, 6) or (0, 0); wavefronts per workgroup (spw >> 8) 0, 1 or 2"); }
    f->var_spw = spw & 0xFF; f->var_lm = lm; f->var_nw = (spw >> 8) & 3;      // spw | nw << 8: nw = 1 / 2 wavefronts per workgroup (0: per batch)
    return SC_HIP_SUCCESS;
}

extern "C" void sc_hip_falcon_destroy(sc_hip_falcon_t *f)
{
    if (!f) return;
    fal_free_ws(f);
    f->srt.release();
    f->stape.release();
    if (f->side) (void)hipStreamDestroy(f->side);
    if (f->ev_fork) (void)hipEventDestroy(f->ev_fork);
    if (f->ev_join) (void)hipEventDestroy(f->ev_join);
    void *ptrs[] = { f->d_sk, f->d_gm, f->d_tabs, f->d_hhat, f->d_hraw };
    for (void *p : ptrs) if (p) (void)sc::wipe_free(p);
    sc_hip_plan_destroy(f->plan);
    delete f;
}

static void leaf_sigmas(const double *tree, unsigned logn, std::vector<float> &out)
{
    if (logn == 0) { out.push_back((float)tree[0]); return; }
    const size_t n = (size_t)1 << logn;
    leaf_sigmas(tree + n + treesize(logn - 1), logn - 1, out);         // tree1 is visited first (gpv.c:2693-2700)
    leaf_sigmas(tree + n, logn - 1, out);
}

// sk: the expanded private key load_skey leaves in sc->falcon->sk (falcon.c:786-790, falcon_keygen.c:6107-6170):
// b00 | b01 | b10 | b11 in FFT form, then the normalised LDL tree; (n_bits + 5) * n doubles
extern "C" int sc_hip_falcon_set_privkey(sc_hip_falcon_t *f, const double *sk) { return sc_hip_falcon_set_privkeys(f, 1, sk); }

// a key set: nkeys expanded private keys back to back; every signature of sc_hip_falcon_sign_keys names one of them
extern "C" int sc_hip_falcon_set_privkeys(sc_hip_falcon_t *f, size_t nkeys, const double *sk)
{
    if (f && f->P.n < 512) return fail("falcon_set_privkey: the tree sampler serves n = 512 and 1024");
    if (!f || !sk || nkeys == 0) return fail("falcon_set_privkey: null argument");
    if (nkeys > 0xffffffffu) return fail("falcon_set_privkey: too many keys");
    HIPCHK(hipSetDevice(f->ctx->device));
    const size_t n = f->P.n, total = (size_t)(f->P.n_bits + 5) * n;
    std::vector<std::vector<float>> sig(nkeys);
    int ts = 2;
    for (size_t kx = 0; kx < nkeys; kx++) {
        leaf_sigmas(sk + kx * total + 4 * n, (unsigned)f->P.n_bits, sig[kx]);
        for (float sg : sig[kx]) {
            if (!(sg > 0.0f) || !(sg < 400.0f)) return fail("falcon_set_privkey: a tree leaf is not a usable standard deviation");
            const int sz = sc_hip_cdt_table_size(10.0f, sg);
            if (sz > ts) ts = sz;
        }
    }
    int ts_log = 4;                                                    // kernels are instantiated for 16, 32 and 64 entries per leaf
    while ((1 << ts_log) < ts) ts_log++;
    if (ts_log > 6) return fail("falcon_set_privkey: a tree leaf needs a CDF table of more than 64 entries (sigma > 6.4)");
    const size_t tper = n << ts_log;
    std::vector<uint64_t> tabs(nkeys * tper, ~0ULL);
    for (size_t kx = 0; kx < nkeys; kx++)
        for (size_t i = 0; i < n; i++) {
            const int sz = sc_hip_cdt_table_size(10.0f, sig[kx][i]);
            uint64_t *row = &tabs[kx * tper + (i << ts_log)];
            if (sz >= 2) { if (sc_hip_cdt_build_table(10.0f, sig[kx][i], row, sz) != SC_HIP_SUCCESS) return SC_HIP_FAILURE; }
            else row[0] = 0;                                            // a one-entry table always yields 0
        }
    if (f->d_tabs) { (void)sc::wipe_free(f->d_tabs); f->d_tabs = nullptr; }
    if (f->d_sk) { (void)sc::wipe_free(f->d_sk); f->d_sk = nullptr; }
    f->have_sk = false; f->nkeys = 0;
    HIPCHK(hipMalloc((void **)&f->d_tabs, tabs.size() * 8));
    HIPCHK(sc::copy_in(f->d_tabs, tabs.data(), tabs.size() * 8));
    HIPCHK(hipMalloc((void **)&f->d_sk, nkeys * total * sizeof(double)));
    HIPCHK(sc::copy_in(f->d_sk, sk, nkeys * total * sizeof(double)));
    f->ts_log = ts_log;
    f->nkeys = nkeys;
    f->have_sk = true;
    return SC_HIP_SUCCESS;
}

// h: the public key in the coefficient domain, SINT16[n] (the first half of sc->pubkey->key, falcon.c:772-780)
extern "C" int sc_hip_falcon_set_pubkey(sc_hip_falcon_t *f, const int16_t *h) { return sc_hip_falcon_set_pubkeys(f, 1, h); }

extern "C" int sc_hip_falcon_set_pubkeys(sc_hip_falcon_t *f, size_t nkeys, const int16_t *h)
{
    if (!f || !h || nkeys == 0) return fail("falcon_set_pubkey: null argument");
    if (nkeys > 0xffffffffu) return fail("falcon_set_pubkey: too many keys");
    HIPCHK(hipSetDevice(f->ctx->device));
    const size_t n = f->P.n;
    std::vector<int32_t> h32(nkeys * n);
    for (size_t i = 0; i < nkeys * n; i++) { int32_t v = h[i] % f->P.q; if (v < 0) v += f->P.q; h32[i] = v; }
    if (f->d_hhat) { (void)sc::wipe_free(f->d_hhat); f->d_hhat = nullptr; }
    if (f->d_hraw) { (void)sc::wipe_free(f->d_hraw); f->d_hraw = nullptr; }
    f->have_pk = false; f->npk = 0;
    HIPCHK(hipMalloc((void **)&f->d_hhat, nkeys * n * 4));
    HIPCHK(sc::copy_in(f->d_hhat, h32.data(), nkeys * n * 4));
    for (size_t i = 0; i < n; i++) h32[i] = h[i];
    HIPCHK(hipMalloc((void **)&f->d_hraw, n * 4));
    HIPCHK(sc::copy_in(f->d_hraw, h32.data(), n * 4));
    if (sc_hip_ntt_fwd(f->plan, nkeys, f->d_hhat, f->d_hhat, nullptr) != SC_HIP_SUCCESS) return SC_HIP_FAILURE;
    HIPCHK(hipStreamSynchronize(nullptr));
    f->npk = nkeys;
    f->have_pk = true;
    return SC_HIP_SUCCESS;
}

extern "C" size_t sc_hip_falcon_sig_bytes(const sc_hip_falcon_t *f)
{
    if (!f) return 0;
    return ((size_t)f->P.n * (size_t)(f->P.q_bits - 2) + 7) / 8;
}

static int fal_launch_sample(sc_hip_falcon *f, size_t batch, const int32_t *d_c, const uint32_t *d_tapes, size_t tape_stride, const uint32_t *d_tape_lens,
                             int32_t *d_s1, int32_t *d_s2, uint32_t *d_status, hipStream_t s, unsigned long long *d_prof = nullptr, const uint32_t *d_key_of = nullptr,
                             bool status_zeroed = false)                // the caller has zeroed d_status already (the signer: in front of the hash, off the path between it and the walk)
{
    const double ni = 1.0 / (double)(long)f->P.q;                    /* fpr_inverse_of, falcon_fft.h:57-60 */
    if (tape_stride < 4 * (size_t)f->P.n) {                          // no tape can hold the 4n words a signature draws
        HIPCHK(hipMemsetD32Async((hipDeviceptr_t)d_status, 2, batch, s));
        return SC_HIP_SUCCESS;
    }
    if (!status_zeroed) HIPCHK(hipMemsetAsync(d_status, 0, batch * 4, s));
    // 8 signatures per wavefront with the lower tree in LDS (fal_sample_w8_kernel) unless another mapping is forced
    const size_t sk_stride = (size_t)(f->P.n_bits + 5) * (size_t)f->P.n, tab_stride = (size_t)f->P.n << f->ts_log;
    const bool w8 = f->force_g == 0 || f->force_g == 64;
    if (w8) {
        // variant (see the kernel's head): (8, 7) with two wavefronts per workgroup at every batch size - measured sign M/s for (8, 7) one wavefront /
        // (8, 7) two wavefronts / (8, 6), n = 1024: batch 4096 3.8 / 4.6 / 3.7, 8192 6.4 / 7.6 / 5.9, 16384 7.2 / 8.8 / 8.7, 32768 7.6 / 9.0 / 9.0, 131072 8.1 / 10.2 / 10.0;
        // n = 512: 4096 7.7 / 8.8 / 7.2, 8192 12.7 / 14.8 / 11.2, 16384 14.7 / 17.5 / 17.2, 131072 17.5 / 21.7 / 21.0 (profiles/r02_falcon_variants.txt)
        int spw = f->var_spw, lm = f->var_lm;
        if (!spw) { spw = 8; lm = 7; }
        int nw = f->var_nw ? f->var_nw : 2;
        if (!(spw == 8 && lm == 7)) nw = 1;
        const dim3 grid8((unsigned)((batch + spw - 1) / spw));
#define FAL_W8B(TSL, PR, S_, L_, N_) hipLaunchKernelGGL((fal_sample_w8_kernel<TSL, PR, S_, L_, N_>), grid8, dim3(64 * N_), 0, s, (uint32_t)batch, f->ws, d_c, d_tapes, tape_stride, d_tape_lens, \
                                            f->d_sk, f->d_gm, f->d_tabs, d_s1, d_s2, d_status, f->P, ni, d_prof, d_key_of, (uint32_t)f->nkeys, sk_stride, tab_stride)
#define FAL_W8A(TSL, PR) do { if (spw == 8 && lm == 7 && nw == 2) FAL_W8B(TSL, PR, 8, 7, 2); else if (spw == 8 && lm == 7) FAL_W8B(TSL, PR, 8, 7, 1); \
                              else if (spw == 8) FAL_W8B(TSL, PR, 8, 6, 1); else FAL_W8B(TSL, PR, 4, 6, 1); } while (0)
#define FAL_W8(TSL) do { if (d_prof) FAL_W8A(TSL, true); else FAL_W8A(TSL, false); } while (0)
        switch (f->ts_log) {
        case 4: FAL_W8(4); break;
        case 5: FAL_W8(5); break;
        case 6: FAL_W8(6); break;
        default: return fail("falcon: unsupported leaf table size");
        }
#undef FAL_W8
#undef FAL_W8A
#undef FAL_W8B
        HIPCHK(hipGetLastError());
        if (!d_key_of) return SC_HIP_SUCCESS;
        // a key set: the workgroups whose 8 signatures do not share one key left theirs (status 7) to the kernel below
    }
    // the two batch-interleaved mappings: a lane per signature, or 8 lanes per signature (the state in HBM either way)
    const bool wide = f->force_g == 8 || (d_key_of && f->force_g != 1);
    const size_t threads = wide ? batch * 8 : batch;
    const int only_deferred = (w8 && d_key_of) ? 1 : 0;
    const dim3 grid((unsigned)((threads + ST - 1) / ST)), block(ST);
#define FAL_LAUNCH(TSL, G) hipLaunchKernelGGL((fal_sample_kernel<TSL, G>), grid, block, 0, s, (uint32_t)batch, batch, f->ws, d_c, d_tapes, tape_stride, d_tape_lens, \
                                              f->d_sk, f->d_gm, f->d_tabs, d_s1, d_s2, d_status, f->P, ni, d_key_of, (uint32_t)f->nkeys, sk_stride, tab_stride, only_deferred)
#define FAL_PICK(TSL) do { if (wide) FAL_LAUNCH(TSL, 8); else FAL_LAUNCH(TSL, 1); } while (0)
    switch (f->ts_log) {
    case 4: FAL_PICK(4); break;
    case 5: FAL_PICK(5); break;
    case 6: FAL_PICK(6); break;
    default: return fail("falcon: unsupported leaf table size");
    }
#undef FAL_PICK
#undef FAL_LAUNCH
    HIPCHK(hipGetLastError());
    return SC_HIP_SUCCESS;
}

// test hook (include/sc_hip_bench.h): 32-bit words of the signer's per-call secrets (the sampler's working polynomials, s1 | s2 before packing) that
// are not zero in the handle's workspace
namespace {
__global__ void fal_residue_kernel(const uint32_t *p, size_t words, unsigned long long *count)
{
    unsigned long long c = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < words; i += (size_t)gridDim.x * blockDim.x) c += p[i] != 0;
    if (c) atomicAdd(count, c);
}
}
extern "C" int sc_hip_falcon_workspace_residue(sc_hip_falcon_t *f, unsigned long long *nonzero_words)
{
    if (!f || !nonzero_words) return fail("falcon_workspace_residue: null argument");
    *nonzero_words = 0;
    if (!f->cap) return SC_HIP_SUCCESS;
    HIPCHK(hipSetDevice(f->ctx->device));
    unsigned long long *cnt = nullptr;
    HIPCHK(hipMalloc((void **)&cnt, 8));
    (void)hipMemset(cnt, 0, 8);
    const size_t n = f->P.n;
    const struct { const void *p; size_t words; } parts[3] = { { f->ws, (f->cap + SPW_MAX - 1) / SPW_MAX * SPW_MAX * 6 * n * 2 }, { f->s1, f->cap * n }, { f->s2, f->cap * n } };
    for (const auto &pt : parts) hipLaunchKernelGGL(fal_residue_kernel, dim3(1024), dim3(256), 0, nullptr, (const uint32_t *)pt.p, pt.words, cnt);
    int rc = SC_HIP_SUCCESS;
    if (hipDeviceSynchronize() != hipSuccess || hipMemcpy(nonzero_words, cnt, 8, hipMemcpyDeviceToHost) != hipSuccess) rc = fail("falcon_workspace_residue: device error");
    (void)hipFree(cnt);
    return rc;
}

static int fal_check_common(const sc_hip_falcon *f, size_t batch, const char *who)
{
    if (!f) return fail(std::string(who) + ": null handle");
    if (batch > 0x7fffffffu / 4096) return fail(std::string(who) + ": batch too large");
    return SC_HIP_SUCCESS;
}

extern "C" int sc_hip_falcon_sample(sc_hip_falcon_t *f, size_t batch, const int32_t *d_c, const uint32_t *d_tapes, size_t tape_stride, const uint32_t *d_tape_lens,
                                    int32_t *d_s1, int32_t *d_s2, uint32_t *d_status, void *stream)
{
    if (fal_check_common(f, batch, "falcon_sample")) return SC_HIP_FAILURE;
    if (!f->have_sk) return fail("falcon_sample: no private key loaded");
    if (batch == 0) return SC_HIP_SUCCESS;
    if (!d_c || !d_tapes || !d_s2) return fail("falcon_sample: null buffer");
    if (((uintptr_t)d_c & 15) != 0) return fail("falcon_sample: d_c must be 16-byte aligned");
    if (tape_stride > 0xffffffffu) return fail("falcon_sample: tape too long");
    if (((uintptr_t)d_tapes & 15) != 0 || (tape_stride & 3) != 0) return fail("falcon_sample: tapes must be 16-byte aligned rows (tape_stride a multiple of 4 words)");
    HIPCHK(hipSetDevice(f->ctx->device));
    if (fal_ensure_ws(f, batch)) return SC_HIP_FAILURE;
    const int rc = fal_launch_sample(f, batch, d_c, d_tapes, tape_stride, d_tape_lens, d_s1, d_s2, d_status ? d_status : f->status, (hipStream_t)stream);
    fal_wipe(f, batch, (hipStream_t)stream);
    return rc;
}

// measurement hook (include/sc_hip_bench.h): the default mapping's sampler with per-phase cycle counters; h_cycles[6] = shader cycles summed over the
// wavefronts: transform + targets | tree levels in HBM | the level-LM node | tree levels in LDS | register sub-trees | lattice point + inverse transforms
extern "C" int sc_hip_falcon_sample_profile(sc_hip_falcon_t *f, size_t batch, const int32_t *d_c, const uint32_t *d_tapes, size_t tape_stride,
                                            int32_t *d_s1, int32_t *d_s2, unsigned long long *h_cycles, void *stream)
{
    if (fal_check_common(f, batch, "falcon_sample_profile")) return SC_HIP_FAILURE;
    if (!f->have_sk || !d_c || !d_tapes || !d_s2 || !h_cycles || batch == 0) return fail("falcon_sample_profile: bad argument");
    if (((uintptr_t)d_c & 15) != 0 || ((uintptr_t)d_tapes & 15) != 0 || (tape_stride & 3) != 0) return fail("falcon_sample_profile: 16-byte aligned rows");
    HIPCHK(hipSetDevice(f->ctx->device));
    if (fal_ensure_ws(f, batch)) return SC_HIP_FAILURE;
    sc::DevBuf pb;
    if (pb.get(6 * sizeof(unsigned long long))) return SC_HIP_FAILURE;
    hipStream_t s = (hipStream_t)stream;
    HIPCHK(hipMemsetAsync(pb.p, 0, 6 * sizeof(unsigned long long), s));
    const int keep = f->force_g; f->force_g = 64;
    const int rc = fal_launch_sample(f, batch, d_c, d_tapes, tape_stride, nullptr, d_s1, d_s2, f->status, s, pb.as<unsigned long long>());
    f->force_g = keep;
    if (rc) return rc;
    HIPCHK(hipMemcpyAsync(h_cycles, pb.p, 6 * sizeof(unsigned long long), hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
    return SC_HIP_SUCCESS;
}

static int fal_sign_core(sc_hip_falcon_t *f, size_t batch, const uint32_t *d_key_of, const uint8_t *d_msgs, size_t msg_stride, const uint32_t *d_mlens, uint32_t fixed_len,
                         const uint32_t *d_tapes, size_t tape_stride, const uint32_t *d_tape_lens,
                         uint8_t *d_sigs, size_t sig_stride, uint32_t *d_siglens, uint32_t *d_status, void *stream, bool grouped = false, hipEvent_t tapes_ready = nullptr);

extern "C" int sc_hip_falcon_sign(sc_hip_falcon_t *f, size_t batch, const uint8_t *d_msgs, size_t msg_stride, const uint32_t *d_mlens, uint32_t fixed_len,
                                  const uint32_t *d_tapes, size_t tape_stride, const uint32_t *d_tape_lens,
                                  uint8_t *d_sigs, size_t sig_stride, uint32_t *d_siglens, uint32_t *d_status, void *stream)
{ return fal_sign_core(f, batch, nullptr, d_msgs, msg_stride, d_mlens, fixed_len, d_tapes, tape_stride, d_tape_lens, d_sigs, sig_stride, d_siglens, d_status, stream); }

// sc_hip_falcon_sign with the CSPRNG inside the call: message i signs with the generator the reference makes of seed i (prng_create(SC_ENTROPY_USER_PROVIDED ...,
// SC_PRNG_AES_CTR_DRBG) seeded through the entropy callback, as create_csprng does: src/utils/crypto/prng.c:963-1107, ctr_drbg.c:37-197) read through prng_64 /
// prng_32 (gpv.c:2671-2686 -> sampling.c).  The 4n pool words a signature draws are expanded from the seed on the device (csrc/drbg.hip) into the handle's
// workspace, consumed by the same kernels as a caller's tape, and zeroed behind the call.  Signatures are byte-identical to sc_hip_falcon_sign over
// sc_hip_drbg_words(seeds, 4n).
extern "C" int sc_hip_falcon_sign_seeded(sc_hip_falcon_t *f, size_t batch, const uint8_t *d_msgs, size_t msg_stride, const uint32_t *d_mlens, uint32_t fixed_len,
                                         const uint8_t *d_seeds, size_t seed_stride, size_t seed_len,
                                         uint8_t *d_sigs, size_t sig_stride, uint32_t *d_siglens, uint32_t *d_status, void *stream)
{
    if (fal_check_common(f, batch, "falcon_sign_seeded")) return SC_HIP_FAILURE;
    if (batch == 0) return SC_HIP_SUCCESS;
    if (!d_seeds || seed_len == 0 || seed_len > seed_stride || seed_len > 0xffffffffu) return fail("falcon_sign_seeded: bad seed buffer / length / stride");
    if (batch > 0x7fffffffu) return fail("falcon_sign_seeded: batch too large");
    HIPCHK(hipSetDevice(f->ctx->device));
    const size_t words = 4 * (size_t)f->P.n;
    if (f->stape.reserve(batch * words * 4)) return SC_HIP_FAILURE;
    uint32_t *tapes = f->stape.take<uint32_t>(batch * words);
    hipStream_t s = (hipStream_t)stream;
    // the expansion (the cipher: vector-ALU-bound, 0.25 ms per 8192 tapes) runs on a stream of the handle's own beside the hash to the ring (a chain of 25
    // dependent permutations per message: latency-bound, most issue slots idle); the sampler waits for both
    if (!f->side) {
        HIPCHK(hipStreamCreateWithFlags(&f->side, hipStreamNonBlocking));
        HIPCHK(hipEventCreateWithFlags(&f->ev_fork, hipEventDisableTiming));
        HIPCHK(hipEventCreateWithFlags(&f->ev_join, hipEventDisableTiming));
    }
    HIPCHK(hipEventRecord(f->ev_fork, s));
    HIPCHK(hipStreamWaitEvent(f->side, f->ev_fork, 0));
    if (sc::drbg_fill(f->ctx, d_seeds, seed_stride, seed_len, nullptr, (uint32_t)batch, nullptr, tapes, words * 4, 0, words / 4, 1, false, f->side)) return SC_HIP_FAILURE;
    HIPCHK(hipEventRecord(f->ev_join, f->side));
    const int rc = fal_sign_core(f, batch, nullptr, d_msgs, msg_stride, d_mlens, fixed_len, tapes, words, nullptr, d_sigs, sig_stride, d_siglens, d_status, stream, false, f->ev_join);
    if (rc != SC_HIP_SUCCESS) (void)hipStreamWaitEvent(s, f->ev_join, 0);      // (an argument error before the wait was queued: the wipe below still follows the expansion)
    f->stape.wipe(s);
    return rc;
}

// replaces: safecrypto_sign -> falcon_sig_sign for a batch of messages EACH under one of the keys loaded with sc_hip_falcon_set_privkeys: d_key_of[i] = the key's index
// (distinct keys).  d_status[i] = 6 for an index outside the key set.
extern "C" int sc_hip_falcon_sign_keys(sc_hip_falcon_t *f, size_t batch, const uint32_t *d_key_of, const uint8_t *d_msgs, size_t msg_stride, const uint32_t *d_mlens, uint32_t fixed_len,
                                       const uint32_t *d_tapes, size_t tape_stride, const uint32_t *d_tape_lens,
                                       uint8_t *d_sigs, size_t sig_stride, uint32_t *d_siglens, uint32_t *d_status, void *stream)
{
    if (!d_key_of) return fail("falcon_sign_keys: null key index buffer");
    return fal_sign_core(f, batch, d_key_of, d_msgs, msg_stride, d_mlens, fixed_len, d_tapes, tape_stride, d_tape_lens, d_sigs, sig_stride, d_siglens, d_status, stream);
}

// ---- a batch whose keys are interleaved, regrouped by key ----------------------------------------------------------------------------------------------
// The 8-signatures-per-workgroup sampler shares a key's tree and leaf tables among the 8 signatures of a workgroup; a workgroup whose signatures name
// different keys falls back to the lanes-per-signature kernel, 3x slower (bench.py: 2.6 M against 7.2 M sign/s at batch 8192 over 64 keys).  Signatures
// are independent, so the batch is regrouped: a counting sort of the message indices by key (any order within a key), the messages' rows gathered in
// that order (texts, tapes - 16 KB each, ~0.1 ms per 8192 at HBM speed), the grouped batch signed, the signatures scattered back to their rows.
__global__ void fal_mixed_groups_kernel(uint32_t batch, const uint32_t *__restrict__ key_of, uint32_t *__restrict__ mixed)
{
    const uint32_t g = blockIdx.x * blockDim.x + threadIdx.x, b0 = g * 8;
    if (b0 >= batch) return;
    const uint32_t k0 = key_of[b0];
    bool diff = false;
    for (uint32_t j = 1; j < 8 && b0 + j < batch; j++) diff |= key_of[b0 + j] != k0;
    if (diff) atomicAdd(mixed, 1u);
}
__global__ void fal_key_hist_kernel(uint32_t batch, const uint32_t *__restrict__ key_of, uint32_t nkeys, uint32_t *__restrict__ hist)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < batch) { const uint32_t k = key_of[i]; atomicAdd(&hist[k < nkeys ? k : nkeys], 1u); }       // bucket nkeys: indices outside the key set (status 6 later)
}
// one workgroup: cursor[k] = first position of key k.  Every key's run is padded to a multiple of 8 positions, so that it starts on a workgroup boundary
// (a run that straddled two workgroups would send both to the slow kernel); the padding positions stay holes and are not signed.
__global__ __launch_bounds__(1024) void fal_key_scan_kernel(const uint32_t *__restrict__ hist, uint32_t nb, uint32_t *__restrict__ cursor, uint32_t *__restrict__ total)
{
    __shared__ uint32_t part[1024];
    const uint32_t per = (nb + 1023) / 1024, lo = threadIdx.x * per, hi = lo + per < nb ? lo + per : nb;
    uint32_t sum = 0;
    for (uint32_t k = lo; k < hi; k++) sum += (hist[k] + 7u) & ~7u;
    part[threadIdx.x] = sum;
    __syncthreads();
    if (threadIdx.x == 0) { uint32_t acc = 0; for (int t = 0; t < 1024; t++) { const uint32_t v = part[t]; part[t] = acc; acc += v; } *total = acc; }
    __syncthreads();
    uint32_t acc = part[threadIdx.x];
    for (uint32_t k = lo; k < hi; k++) { cursor[k] = acc; acc += (hist[k] + 7u) & ~7u; }
}
// the rows of key k's run, holes included, carry key k (a hole under another key would make its workgroup a mixed one)
__global__ void fal_key_fill_kernel(uint32_t nb, const uint32_t *__restrict__ hist, const uint32_t *__restrict__ start, uint32_t *__restrict__ key_sorted)
{
    const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= nb) return;
    const uint32_t n = (hist[k] + 7u) & ~7u, p0 = start[k];
    for (uint32_t j = 0; j < n; j++) key_sorted[p0 + j] = k;
}
__global__ void fal_key_place_kernel(uint32_t batch, const uint32_t *__restrict__ key_of, uint32_t nkeys, uint32_t *__restrict__ cursor, uint32_t *__restrict__ perm,
                                     uint32_t *__restrict__ key_sorted)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= batch) return;
    const uint32_t k = key_of[i], pos = atomicAdd(&cursor[k < nkeys ? k : nkeys], 1u);
    perm[pos] = i; key_sorted[pos] = k;
}
// dst row p = src row perm[p] (perm[p] = 0xFFFFFFFF: a hole, left as it is); row_words 32-bit words per row, rows 4-byte aligned
__global__ void fal_gather_rows_kernel(uint32_t rows, const uint32_t *__restrict__ perm, const uint32_t *__restrict__ src, size_t src_words, uint32_t *__restrict__ dst, size_t dst_words,
                                       uint32_t row_words)
{
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t p = (uint32_t)(t / row_words), w = (uint32_t)(t % row_words);
    if (p >= rows) return;
    const uint32_t i = perm[p];
    if (i != 0xFFFFFFFFu) dst[(size_t)p * dst_words + w] = src[(size_t)i * src_words + w];
}
__global__ void fal_gather_bytes_kernel(uint32_t rows, const uint32_t *__restrict__ perm, const uint8_t *__restrict__ src, size_t src_stride, uint8_t *__restrict__ dst, size_t dst_stride,
                                        uint32_t row_bytes)
{
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t p = (uint32_t)(t / row_bytes), w = (uint32_t)(t % row_bytes);
    if (p >= rows) return;
    const uint32_t i = perm[p];
    if (i != 0xFFFFFFFFu) dst[(size_t)p * dst_stride + w] = src[(size_t)i * src_stride + w];
}
__global__ void fal_fill_lens_kernel(uint32_t rows, const uint32_t *__restrict__ perm, uint32_t *__restrict__ mlens, uint32_t mlen, uint32_t *__restrict__ tlens, uint32_t tlen)
{
    const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= rows || perm[p] == 0xFFFFFFFFu) return;
    if (mlens) mlens[p] = mlen;
    if (tlens) tlens[p] = tlen;
}
// caller's row perm[p] = grouped row p: signature bytes (up to its length), length, status
__global__ void fal_scatter_sigs_kernel(uint32_t rows, const uint32_t *__restrict__ perm, const uint8_t *__restrict__ sigs, const uint32_t *__restrict__ siglens,
                                        const uint32_t *__restrict__ status, size_t stride, uint8_t *__restrict__ out, size_t out_stride, uint32_t *__restrict__ out_lens,
                                        uint32_t *__restrict__ out_status)
{
    const uint32_t p = blockIdx.x;
    if (p >= rows) return;
    const uint32_t i = perm[p];
    if (i == 0xFFFFFFFFu) return;
    const uint32_t len = siglens[p];
    for (uint32_t w = threadIdx.x; w < len; w += blockDim.x) out[(size_t)i * out_stride + w] = sigs[(size_t)p * stride + w];
    if (threadIdx.x == 0) { out_lens[i] = len; if (out_status) out_status[i] = status[p]; }
}

static int fal_sign_core(sc_hip_falcon_t *f, size_t batch, const uint32_t *d_key_of, const uint8_t *d_msgs, size_t msg_stride, const uint32_t *d_mlens, uint32_t fixed_len,
                         const uint32_t *d_tapes, size_t tape_stride, const uint32_t *d_tape_lens,
                         uint8_t *d_sigs, size_t sig_stride, uint32_t *d_siglens, uint32_t *d_status, void *stream, bool grouped, hipEvent_t tapes_ready);

static int fal_sign_grouped(sc_hip_falcon_t *f, size_t batch, const uint32_t *d_key_of, const uint8_t *d_msgs, size_t msg_stride, const uint32_t *d_mlens, uint32_t fixed_len,
                            const uint32_t *d_tapes, size_t tape_stride, const uint32_t *d_tape_lens,
                            uint8_t *d_sigs, size_t sig_stride, uint32_t *d_siglens, uint32_t *d_status, hipStream_t s)
{
    const uint32_t B = (uint32_t)batch, nb = (uint32_t)f->nkeys + 1;
    const size_t rows_max = batch + 8 * (size_t)nb;                      // every key's run padded to a multiple of 8
    if (rows_max > 0x7fffffffu) return fail("falcon_sign_keys: batch too large");
    const size_t ms = (msg_stride + 3) & ~(size_t)3, ss = (sig_stride + 15) & ~(size_t)15;
    if (f->srt.reserve(8192 + (size_t)nb * 8 + rows_max * (4 + 4 + 4 + 4 + 4 + 4 + ms + ss + tape_stride * 4) + 16 * 256)) return SC_HIP_FAILURE;
    uint32_t *tapes_g = f->srt.take<uint32_t>(rows_max * tape_stride);   // (first: 16-byte aligned rows; the one secret slice of this workspace)
    const size_t secret_end = f->srt.mark();
    uint32_t *hist = f->srt.take<uint32_t>(nb), *cursor = f->srt.take<uint32_t>(nb), *total = f->srt.take<uint32_t>(1);
    uint32_t *perm = f->srt.take<uint32_t>(rows_max), *key_g = f->srt.take<uint32_t>(rows_max), *mlens_g = f->srt.take<uint32_t>(rows_max), *tlens_g = f->srt.take<uint32_t>(rows_max);
    uint32_t *siglens_g = f->srt.take<uint32_t>(rows_max), *status_g = f->srt.take<uint32_t>(rows_max);
    uint8_t *msgs_g = f->srt.take<uint8_t>(rows_max * ms), *sigs_g = f->srt.take<uint8_t>(rows_max * ss);
    HIPCHK(hipMemsetAsync(hist, 0, (size_t)nb * 4, s));
    hipLaunchKernelGGL(fal_key_hist_kernel, dim3((B + 255) / 256), dim3(256), 0, s, B, d_key_of, (uint32_t)f->nkeys, hist);
    hipLaunchKernelGGL(fal_key_scan_kernel, dim3(1), dim3(1024), 0, s, hist, nb, cursor, total);
    uint32_t rows = 0;
    HIPCHK(hipMemcpyAsync(&rows, total, 4, hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
    if (rows > rows_max) return fail("falcon_sign_keys: regrouping overflow");
    HIPCHK(hipMemsetAsync(perm, 0xFF, (size_t)rows * 4, s));             // holes
    hipLaunchKernelGGL(fal_key_fill_kernel, dim3((nb + 255) / 256), dim3(256), 0, s, nb, hist, cursor, key_g);
    HIPCHK(hipMemsetAsync(mlens_g, 0, (size_t)rows * 4, s));             // a hole is an empty message with an empty tape under its run's key: refused at once (status 2), not signed
    HIPCHK(hipMemsetAsync(tlens_g, 0, (size_t)rows * 4, s));
    hipLaunchKernelGGL(fal_key_place_kernel, dim3((B + 255) / 256), dim3(256), 0, s, B, d_key_of, (uint32_t)f->nkeys, cursor, perm, key_g);
    const size_t mt = (size_t)rows * msg_stride, tt = (size_t)rows * tape_stride;
    hipLaunchKernelGGL(fal_gather_bytes_kernel, dim3((unsigned)((mt + 255) / 256)), dim3(256), 0, s, rows, perm, d_msgs, msg_stride, msgs_g, ms, (uint32_t)msg_stride);
    hipLaunchKernelGGL(fal_gather_rows_kernel, dim3((unsigned)((tt + 255) / 256)), dim3(256), 0, s, rows, perm, d_tapes, tape_stride, tapes_g, tape_stride, (uint32_t)tape_stride);
    if (d_mlens) hipLaunchKernelGGL(fal_gather_rows_kernel, dim3((rows + 255) / 256), dim3(256), 0, s, rows, perm, d_mlens, (size_t)1, mlens_g, (size_t)1, 1u);
    if (d_tape_lens) hipLaunchKernelGGL(fal_gather_rows_kernel, dim3((rows + 255) / 256), dim3(256), 0, s, rows, perm, d_tape_lens, (size_t)1, tlens_g, (size_t)1, 1u);
    HIPCHK(hipGetLastError());
    // rows with fixed lengths: fill the real rows' entries (holes stay 0)
    if (!d_mlens || !d_tape_lens) hipLaunchKernelGGL(fal_fill_lens_kernel, dim3((rows + 255) / 256), dim3(256), 0, s, rows, perm, d_mlens ? (uint32_t *)nullptr : mlens_g, fixed_len,
                                                     d_tape_lens ? (uint32_t *)nullptr : tlens_g, (uint32_t)tape_stride);
    HIPCHK(hipGetLastError());
    const int rc = fal_sign_core(f, rows, key_g, msgs_g, ms, mlens_g, 0, tapes_g, tape_stride, tlens_g, sigs_g, ss, siglens_g, status_g, s, true);
    if (rc == SC_HIP_SUCCESS) {
        hipLaunchKernelGGL(fal_scatter_sigs_kernel, dim3(rows), dim3(128), 0, s, rows, perm, sigs_g, siglens_g, status_g, ss, d_sigs, sig_stride, d_siglens, d_status);
        HIPCHK(hipGetLastError());
    }
    f->srt.wipe_upto(secret_end, s);                                  // the copies of the tapes
    return rc;
}

static int fal_sign_core(sc_hip_falcon_t *f, size_t batch, const uint32_t *d_key_of, const uint8_t *d_msgs, size_t msg_stride, const uint32_t *d_mlens, uint32_t fixed_len,
                         const uint32_t *d_tapes, size_t tape_stride, const uint32_t *d_tape_lens,
                         uint8_t *d_sigs, size_t sig_stride, uint32_t *d_siglens, uint32_t *d_status, void *stream, bool grouped, hipEvent_t tapes_ready)
{
    if (fal_check_common(f, batch, "falcon_sign")) return SC_HIP_FAILURE;
    if (!f->have_sk) return fail("falcon_sign: no private key loaded");
    if (batch == 0) return SC_HIP_SUCCESS;
    if (!d_msgs || !d_tapes || !d_sigs || !d_siglens) return fail("falcon_sign: null buffer");
    if (sig_stride < sc_hip_falcon_sig_bytes(f)) return fail("falcon_sign: sig_stride smaller than sc_hip_falcon_sig_bytes()");
    if (tape_stride > 0xffffffffu) return fail("falcon_sign: tape too long");
    if (((uintptr_t)d_tapes & 15) != 0 || (tape_stride & 3) != 0) return fail("falcon_sign: tapes must be 16-byte aligned rows (tape_stride a multiple of 4 words)");
    HIPCHK(hipSetDevice(f->ctx->device));
    if (fal_ensure_ws(f, batch)) return SC_HIP_FAILURE;
    hipStream_t s = (hipStream_t)stream;
    const uint32_t B = (uint32_t)batch;
    if (d_key_of && !grouped && B >= 64 && (f->force_g == 0 || f->force_g == 64)) {
        // how many workgroups of 8 would see more than one key?  (one 4-byte read-back; a batch that is grouped already is signed where it lies)
        uint32_t mixed = 0;
        HIPCHK(hipMemsetAsync(f->siglens, 0, 4, s));
        hipLaunchKernelGGL(fal_mixed_groups_kernel, dim3(((B + 7) / 8 + 255) / 256), dim3(256), 0, s, B, d_key_of, f->siglens);
        HIPCHK(hipMemcpyAsync(&mixed, f->siglens, 4, hipMemcpyDeviceToHost, s));
        HIPCHK(hipStreamSynchronize(s));
        if ((size_t)mixed * 8 > (B + 7) / 8)
            return fal_sign_grouped(f, batch, d_key_of, d_msgs, msg_stride, d_mlens, fixed_len, d_tapes, tape_stride, d_tape_lens, d_sigs, sig_stride, d_siglens, d_status, s);
    }
    uint32_t *status = d_status ? d_status : f->status;
    const bool pre_zero = tape_stride >= 4 * (size_t)f->P.n;            // (a tape stride too short for any signature: fal_launch_sample fills the status itself)
    if (pre_zero) HIPCHK(hipMemsetAsync(status, 0, batch * 4, s));
    if ((size_t)B <= (size_t)f->ctx->cus * 4 * 64) hipLaunchKernelGGL(fal_hash2_kernel, dim3((2 * B + HT - 1) / HT), dim3(HT), 0, s, B, d_msgs, msg_stride, d_mlens, fixed_len, f->c, f->P, 0);
    else hipLaunchKernelGGL(fal_hash1_kernel, dim3((B + HT - 1) / HT), dim3(HT), 0, s, B, d_msgs, msg_stride, d_mlens, fixed_len, f->c, f->P, 0);   // two lanes per message only while SIMDs would idle (measured: +11 % verify at 8192, -8 % at 131072)
    HIPCHK(hipGetLastError());
    if (tapes_ready) HIPCHK(hipStreamWaitEvent(s, tapes_ready, 0));     // seeded call: the tapes are being expanded on the handle's side stream
    if (fal_launch_sample(f, batch, f->c, d_tapes, tape_stride, d_tape_lens, f->s1, f->s2, status, s, nullptr, d_key_of, pre_zero)) return SC_HIP_FAILURE;
    hipLaunchKernelGGL(fal_finish_kernel, dim3((B + FW - 1) / FW), dim3(64 * FW), sizeof(int32_t) * FW * (size_t)f->P.n, s, B, f->s1, f->s2, d_sigs, sig_stride, d_siglens, status, f->P,
                       f->ws, (uint32_t)((batch + SPW_MAX - 1) / SPW_MAX * SPW_MAX));
    HIPCHK(hipGetLastError());
    return SC_HIP_SUCCESS;
}

static int fal_verify_core(sc_hip_falcon_t *f, size_t batch, const uint32_t *d_key_of, const uint8_t *d_msgs, size_t msg_stride, const uint32_t *d_mlens, uint32_t fixed_len,
                           const uint8_t *d_sigs, size_t sig_stride, const uint32_t *d_siglens, uint32_t *d_verdicts, void *stream);

extern "C" int sc_hip_falcon_verify(sc_hip_falcon_t *f, size_t batch, const uint8_t *d_msgs, size_t msg_stride, const uint32_t *d_mlens, uint32_t fixed_len,
                                    const uint8_t *d_sigs, size_t sig_stride, const uint32_t *d_siglens, uint32_t *d_verdicts, void *stream)
{ return fal_verify_core(f, batch, nullptr, d_msgs, msg_stride, d_mlens, fixed_len, d_sigs, sig_stride, d_siglens, d_verdicts, stream); }

// replaces: safecrypto_verify -> falcon_sig_verify for a batch of signatures EACH under one of the public keys loaded with sc_hip_falcon_set_pubkeys
extern "C" int sc_hip_falcon_verify_keys(sc_hip_falcon_t *f, size_t batch, const uint32_t *d_key_of, const uint8_t *d_msgs, size_t msg_stride, const uint32_t *d_mlens, uint32_t fixed_len,
                                         const uint8_t *d_sigs, size_t sig_stride, const uint32_t *d_siglens, uint32_t *d_verdicts, void *stream)
{
    if (!d_key_of) return fail("falcon_verify_keys: null key index buffer");
    return fal_verify_core(f, batch, d_key_of, d_msgs, msg_stride, d_mlens, fixed_len, d_sigs, sig_stride, d_siglens, d_verdicts, stream);
}

// rows[i] = hhat[key_of[i]] (an index outside the key set: zeros, verdict 1)
__global__ void fal_gather_keys_kernel(uint32_t batch, const uint32_t *__restrict__ key_of, uint32_t nkeys, const int32_t *__restrict__ hhat, int32_t *__restrict__ rows, int n,
                                       uint32_t *__restrict__ verdicts)
{
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t i = t >> sc::pow2_shift(n / 4); const int j = (int)(t & (size_t)(n / 4 - 1));
    if (i >= batch) return;
    const uint32_t kx = key_of[i];
    int4 v = make_int4(0, 0, 0, 0);
    if (kx < nkeys) v = reinterpret_cast<const int4 *>(hhat + (size_t)kx * n)[j];
    else if (j == 0) verdicts[i] = 1;
    reinterpret_cast<int4 *>(rows + i * (size_t)n)[j] = v;
}

static int fal_verify_core(sc_hip_falcon_t *f, size_t batch, const uint32_t *d_key_of, const uint8_t *d_msgs, size_t msg_stride, const uint32_t *d_mlens, uint32_t fixed_len,
                           const uint8_t *d_sigs, size_t sig_stride, const uint32_t *d_siglens, uint32_t *d_verdicts, void *stream)
{
    if (fal_check_common(f, batch, "falcon_verify")) return SC_HIP_FAILURE;
    if (!f->have_pk) return fail("falcon_verify: no public key loaded");
    if (batch == 0) return SC_HIP_SUCCESS;
    if (!d_msgs || !d_sigs || !d_siglens || !d_verdicts) return fail("falcon_verify: null buffer");
    if (sig_stride < sc_hip_falcon_sig_bytes(f)) return fail("falcon_verify: sig_stride smaller than sc_hip_falcon_sig_bytes()");
    HIPCHK(hipSetDevice(f->ctx->device));
    if (fal_ensure_ws(f, batch)) return SC_HIP_FAILURE;
    hipStream_t s = (hipStream_t)stream;
    const uint32_t B = (uint32_t)batch;
    const size_t ne = batch * (size_t)f->P.n;
    HIPCHK(hipMemsetAsync(d_verdicts, 0, batch * 4, s));
    if ((size_t)B <= (size_t)f->ctx->cus * 4 * 64)
        hipLaunchKernelGGL(fal_hash2_unpack_kernel, dim3((2 * B + HT - 1) / HT + (unsigned)((ne / 2 + 255) / 256)), dim3(256), 0, s, B, d_msgs, msg_stride, d_mlens, fixed_len, f->c, f->P, 0,
                           d_sigs, sig_stride, d_siglens, f->s2, d_verdicts, f->P.q_bits - 2, 1);
    else {
        hipLaunchKernelGGL(fal_unpack_kernel, dim3((unsigned)((ne / 2 + 255) / 256)), dim3(256), 0, s, B, d_sigs, sig_stride, d_siglens, f->s2, d_verdicts, f->P, f->P.q_bits - 2, 1);
        hipLaunchKernelGGL(fal_hash1_kernel, dim3((B + HT - 1) / HT), dim3(HT), 0, s, B, d_msgs, msg_stride, d_mlens, fixed_len, f->c, f->P, 0);
    }   // two lanes per message only while SIMDs would idle (measured: +11 % verify at 8192, -8 % at 131072)
    HIPCHK(hipGetLastError());
    if (d_key_of) {                                                    // the public keys of the batch, gathered into the (otherwise idle) sampler workspace
        int32_t *rows = reinterpret_cast<int32_t *>(f->ws);
        hipLaunchKernelGGL(fal_gather_keys_kernel, dim3((unsigned)((ne / 4 + 255) / 256)), dim3(256), 0, s, B, d_key_of, (uint32_t)f->npk, f->d_hhat, rows, f->P.n, d_verdicts);
        HIPCHK(hipGetLastError());
        if (sc_hip_polymul_ntt(f->plan, batch, f->s2, rows, f->s1, s)) return SC_HIP_FAILURE;
    } else if (sc::polymul_ntt_bcast(f->plan, batch, f->s2, f->d_hhat, f->s1, s)) return SC_HIP_FAILURE;
    hipLaunchKernelGGL(fal_vfinish_kernel, dim3((B + FW - 1) / FW), dim3(64 * FW), sizeof(int32_t) * FW * (size_t)f->P.n, s, B, f->c, f->s1, f->s2, d_verdicts, f->P);
    HIPCHK(hipGetLastError());
    return SC_HIP_SUCCESS;
}

// replaces: safecrypto_ibe_extract -> dlp_ibe_extract (src/schemes/ibe/dlp/dlp_ibe.c:874-1043; enhanced extract, XOF id oracle, NTT-domain H(id))
// for `batch` identities under one master key.  The handle is a Falcon-shaped one created with the DLP-IBE parameters (n, q = 4206593,
// q_bits = 23, n_bits) whose private key is sc->dlp_ibe->master_tree (load_skey layout).  ubits = 8 + ceil_log2(floor(6 * 1.17 * sqrt(q / 2n)))
// (dlp_ibe.c:916-917).  d_sk: [batch][sk_stride] bytes, n * ubits / 8 each; d_status: 1 extracted, 2 tape exhausted.
extern "C" int sc_hip_dlp_ibe_extract(sc_hip_falcon_t *f, size_t batch, int ubits, const uint8_t *d_ids, size_t id_stride, const uint32_t *d_idlens, uint32_t fixed_len,
                                      const uint32_t *d_tapes, size_t tape_stride, const uint32_t *d_tape_lens, uint8_t *d_sk, size_t sk_stride, uint32_t *d_status, void *stream)
{
    if (fal_check_common(f, batch, "dlp_ibe_extract")) return SC_HIP_FAILURE;
    if (!f->have_sk) return fail("dlp_ibe_extract: no master tree loaded");
    if (batch == 0) return SC_HIP_SUCCESS;
    if (!d_ids || !d_tapes || !d_sk) return fail("dlp_ibe_extract: null buffer");
    if (ubits < 2 || ubits > 32 || sk_stride < (size_t)f->P.n * (size_t)ubits / 8) return fail("dlp_ibe_extract: bad ubits or sk_stride");
    if (tape_stride > 0xffffffffu) return fail("dlp_ibe_extract: tape too long");
    if (((uintptr_t)d_tapes & 15) != 0 || (tape_stride & 3) != 0) return fail("dlp_ibe_extract: tapes must be 16-byte aligned rows (tape_stride a multiple of 4 words)");
    HIPCHK(hipSetDevice(f->ctx->device));
    if (fal_ensure_ws(f, batch)) return SC_HIP_FAILURE;
    hipStream_t s = (hipStream_t)stream;
    const uint32_t B = (uint32_t)batch;
    uint32_t *status = d_status ? d_status : f->status;
    if ((size_t)B <= (size_t)f->ctx->cus * 4 * 64) hipLaunchKernelGGL(fal_hash2_kernel, dim3((2 * B + HT - 1) / HT), dim3(HT), 0, s, B, d_ids, id_stride, d_idlens, fixed_len, f->c, f->P, 1);
    else hipLaunchKernelGGL(fal_hash1_kernel, dim3((B + HT - 1) / HT), dim3(HT), 0, s, B, d_ids, id_stride, d_idlens, fixed_len, f->c, f->P, 1);
    HIPCHK(hipGetLastError());
    if (sc_hip_ntt_inv(f->plan, batch, f->c, f->c, s)) return SC_HIP_FAILURE;                     // H(id) is taken as an NTT-domain vector (dlp_ibe.c:935-940)
    if (fal_launch_sample(f, batch, f->c, d_tapes, tape_stride, d_tape_lens, f->s1, f->s2, status, s)) return SC_HIP_FAILURE;
    const size_t nt = batch * (size_t)(f->P.n / 8);
    hipLaunchKernelGGL(dlp_pack_kernel, dim3((unsigned)((nt + 255) / 256)), dim3(256), 0, s, B, f->s2, status, d_sk, sk_stride, f->P.n, ubits);
    HIPCHK(hipGetLastError());
    fal_wipe(f, batch, s);
    return SC_HIP_SUCCESS;
}

// replaces: safecrypto_verify -> ens_dlp_sig_verify (src/schemes/sig/ens_dlp/ens_dlp_sig.c:1564-1665, ENS and DLP signatures without message
// recovery) for `batch` messages under one public key.  The handle is a Falcon-shaped one created with the scheme's parameters {n, q, q_bits,
// n_bits, set, bd} (ens_dlp_sig_params.c:23-62; n = 512 / 1024) and loaded with h (sc_hip_falcon_set_pubkey: the first n SINT16 of the public key).
// Same arithmetic as Falcon verification - s2 = centre(H(m) - h s1), single-precision norm from the top index down - with the signature's n fields
// q_bits wide and uncentred, and the hash subtracting q when >= q (:728-742).  d_verdicts[i] = 0 valid / 1 invalid.
extern "C" int sc_hip_ens_dlp_sig_verify(sc_hip_falcon_t *f, size_t batch, const uint8_t *d_msgs, size_t msg_stride, const uint32_t *d_mlens, uint32_t fixed_len,
                                         const uint8_t *d_sigs, size_t sig_stride, const uint32_t *d_siglens, uint32_t *d_verdicts, void *stream)
{
    if (fal_check_common(f, batch, "ens_dlp_sig_verify")) return SC_HIP_FAILURE;
    if (!f->have_pk) return fail("ens_dlp_sig_verify: no public key loaded");
    if (batch == 0) return SC_HIP_SUCCESS;
    if (!d_msgs || !d_sigs || !d_siglens || !d_verdicts) return fail("ens_dlp_sig_verify: null buffer");
    if (sig_stride < ((size_t)f->P.n * (size_t)f->P.q_bits + 7) / 8) return fail("ens_dlp_sig_verify: sig_stride smaller than a signature (n q_bits / 8 bytes)");
    HIPCHK(hipSetDevice(f->ctx->device));
    if (fal_ensure_ws(f, batch)) return SC_HIP_FAILURE;
    hipStream_t s = (hipStream_t)stream;
    const uint32_t B = (uint32_t)batch;
    const size_t ne = batch * (size_t)f->P.n;
    HIPCHK(hipMemsetAsync(d_verdicts, 0, batch * 4, s));
    if ((size_t)B <= (size_t)f->ctx->cus * 4 * 64)                     // as fal_verify_core: the unpacking under the hash chain, one launch
        hipLaunchKernelGGL(fal_hash2_unpack_kernel, dim3((2 * B + HT - 1) / HT + (unsigned)((ne / 2 + 255) / 256)), dim3(256), 0, s, B, d_msgs, msg_stride, d_mlens, fixed_len, f->c, f->P, 1,
                           d_sigs, sig_stride, d_siglens, f->s2, d_verdicts, f->P.q_bits, 0);
    else {
        hipLaunchKernelGGL(fal_unpack_kernel, dim3((unsigned)((ne / 2 + 255) / 256)), dim3(256), 0, s, B, d_sigs, sig_stride, d_siglens, f->s2, d_verdicts, f->P, f->P.q_bits, 0);
        hipLaunchKernelGGL(fal_hash1_kernel, dim3((B + HT - 1) / HT), dim3(HT), 0, s, B, d_msgs, msg_stride, d_mlens, fixed_len, f->c, f->P, 1);
    }
    HIPCHK(hipGetLastError());
    if (sc::polymul_ntt_bcast(f->plan, batch, f->s2, f->d_hhat, f->s1, s)) return SC_HIP_FAILURE;          // the signature's s1 sits in the s2 buffer: the roles are swapped
    hipLaunchKernelGGL(fal_vfinish_kernel, dim3((B + FW - 1) / FW), dim3(64 * FW), sizeof(int32_t) * FW * (size_t)f->P.n, s, B, f->c, f->s1, f->s2, d_verdicts, f->P);
    HIPCHK(hipGetLastError());
    return SC_HIP_SUCCESS;
}


// ---- ENS / DLP signing ------------------------------------------------------------------------------------------------------------------
struct sc_hip_ens_dlp_signer {
    sc_hip_ctx *ctx;
    FalP P;
    bool have_key;
    int32_t *d_b; float *d_bgs, *d_inv; uint64_t *d_tabs; uint32_t *d_sizes; int ts_log; uint32_t maxb;
    sc::Workspace ws;
    int force_lanes;             // sc_hip_ens_dlp_signer_set_mapping: 0 = chosen from the batch size
    int force_pf;                // the same hook: +1 / -1 pin the form of the table search (kleinr_kernel PF), 0 = chosen from the batch size
};

extern "C" int sc_hip_ens_dlp_signer_create(sc_hip_ctx_t *ctx, const sc_hip_falcon_params_t *params, sc_hip_ens_dlp_signer_t **out)
{
    if (!ctx || !params || !out) return fail("ens_dlp_signer_create: null argument");
    FalP P; memcpy(&P, params, sizeof P);
    if ((P.n != 256 && P.n != 512 && P.n != 1024) || (1 << P.n_bits) != P.n) return fail("ens_dlp_signer_create: n must be 256, 512 or 1024 with n_bits = log2 n");
    if (P.q_bits < 4 || P.q_bits > 29 || P.q < 3 || P.q >= (1 << P.q_bits)) return fail("ens_dlp_signer_create: bad q / q_bits");
    sc_hip_ens_dlp_signer *h = new sc_hip_ens_dlp_signer();
    h->ctx = ctx; h->P = P;
    *out = h;
    return SC_HIP_SUCCESS;
}

extern "C" int sc_hip_ens_dlp_signer_set_mapping(sc_hip_ens_dlp_signer_t *h, int lanes)
{
    if (!h) return fail("ens_dlp_signer_set_mapping: null handle");
    const int form = lanes >> 8;                                 // + 0x100: table entries held across the sweep, + 0x200: fetched in their own step (0: by batch)
    lanes &= 0xff;
    if ((lanes != 0 && lanes != 1 && lanes != 4 && lanes != 8 && lanes != 16) || form < 0 || form > 2)
        return fail("ens_dlp_signer_set_mapping: lanes must be 0 (default), 1, 4, 8 or 16, optionally + 0x100 / 0x200 for the form of the table search");
    h->force_lanes = lanes;
    h->force_pf = form == 1 ? 1 : form == 2 ? -1 : 0;
    return SC_HIP_SUCCESS;
}

extern "C" void sc_hip_ens_dlp_signer_destroy(sc_hip_ens_dlp_signer_t *h)
{
    if (!h) return;
    void *ptrs[] = { h->d_b, h->d_bgs, h->d_inv, h->d_tabs, h->d_sizes };
    for (void *p : ptrs) if (p) (void)sc::wipe_free(p);
    h->ws.release();
    delete h;
}

// b, b_gs, b_gs_inv_norm: what create_gpv_matrices (ens_dlp_sig.c:364-421) leaves in sc->ens_dlp_sig (keep_matrices = 1): the expanded basis
// (SINT32 [2n][2n]), its single-precision Gram-Schmidt orthogonalisation (FLOAT [2n][2n]) and 1 / ||b_gs[j]|| (FLOAT [2n]); host pointers.
// Builds the 2n per-row CDTs the reference recreates inside every signature (create_sampler per row, gpv.c:2445-2452; sigma_j = s_f / ||b_gs[j]||
// with s_f = 2 / b_gs_inv_norm[0] rounded to FLOAT, ens_dlp_sig.c:1253; tail 10; long-double table as gaussian_cdf_create_64).
extern "C" int sc_hip_ens_dlp_signer_set_key(sc_hip_ens_dlp_signer_t *h, const int32_t *b, const float *b_gs, const float *b_gs_inv_norm)
{
    if (!h || !b || !b_gs || !b_gs_inv_norm) return fail("ens_dlp_signer_set_key: null argument");
    HIPCHK(hipSetDevice(h->ctx->device));
    const size_t n2 = 2 * (size_t)h->P.n;
    const float sig0 = (float)(2.0L / (long double)b_gs_inv_norm[0]);
    const double s_f = sig0;
    std::vector<float> sg(n2);
    std::vector<uint32_t> sizes(n2);
    int ts_log = 1;
    for (size_t j = 0; j < n2; j++) {
        sg[j] = (float)(s_f * (double)b_gs_inv_norm[j]);
        const int sz = sc_hip_cdt_table_size(10.0f, sg[j]);
        if (sz < 2 || sz > 4096) return fail("ens_dlp_signer_set_key: a row's sampler table is out of range (degenerate Gram-Schmidt norms)");
        sizes[j] = (uint32_t)sz;
        while ((1 << ts_log) < sz) ts_log++;
    }
    std::vector<uint64_t> tabs(n2 << ts_log, ~0ull);
    for (size_t j = 0; j < n2; j++)
        if (sc_hip_cdt_build_table(10.0f, sg[j], &tabs[j << ts_log], (int)sizes[j]) != SC_HIP_SUCCESS) return SC_HIP_FAILURE;
    uint32_t maxb = 1;
    for (size_t i = 0; i < n2 * n2; i++) { const int64_t v = b[i]; const uint32_t a = (uint32_t)(v < 0 ? -v : v); maxb = a > maxb ? a : maxb; }
    void *old[] = { h->d_b, h->d_bgs, h->d_inv, h->d_tabs, h->d_sizes };
    for (void *p : old) if (p) (void)sc::wipe_free(p);
    h->d_b = nullptr; h->d_bgs = h->d_inv = nullptr; h->d_tabs = nullptr; h->d_sizes = nullptr; h->have_key = false;
    HIPCHK(hipMalloc((void **)&h->d_b, n2 * n2 * 4));
    HIPCHK(hipMalloc((void **)&h->d_bgs, n2 * n2 * 4));
    HIPCHK(hipMalloc((void **)&h->d_inv, n2 * 4));
    HIPCHK(hipMalloc((void **)&h->d_tabs, tabs.size() * 8));
    HIPCHK(hipMalloc((void **)&h->d_sizes, n2 * 4));
    HIPCHK(sc::copy_in(h->d_b, b, n2 * n2 * 4));
    HIPCHK(sc::copy_in(h->d_bgs, b_gs, n2 * n2 * 4));
    HIPCHK(sc::copy_in(h->d_inv, b_gs_inv_norm, n2 * 4));
    HIPCHK(sc::copy_in(h->d_tabs, tabs.data(), tabs.size() * 8));
    HIPCHK(sc::copy_in(h->d_sizes, sizes.data(), n2 * 4));
    h->ts_log = ts_log; h->maxb = maxb; h->have_key = true;
    return SC_HIP_SUCCESS;
}

// The same from the private key alone (f | g | F | G, SINT32, 4n words = sc->privkey->key): what create_gpv_matrices (ens_dlp_sig.c:364-421) does at key
// load, on the host as there - gpv_expand_basis (gpv.c:709-760; its scratch matrix is zeroed once and only written up to each polynomial's degree, so
// a polynomial with zero top coefficients inherits the previous one's entries: order g, f, G, F), modified_gram_schmidt_fast_flt (gpv.c:1021-1117) and
// gpv_precompute_inv_flt (:977-984), every operation rounded to single precision in the reference's order (this file is built with -ffp-contract=off).
static float dot_seq(const float *a, const float *b, size_t n) { float d = 0; for (size_t k = 0; k < n; k++) d += a[k] * b[k]; return d; }

extern "C" int sc_hip_ens_dlp_signer_set_privkey(sc_hip_ens_dlp_signer_t *h, const int32_t *fgFG)
{
    if (!h || !fgFG) return fail("ens_dlp_signer_set_privkey: null argument");
    const size_t N = (size_t)h->P.n, N2 = 2 * N;
    std::vector<int32_t> b(N2 * N2), A(N * N, 0);
    const int32_t *polys[4] = { fgFG + N, fgFG, fgFG + 3 * N, fgFG + 2 * N };            // g, f, G, F
    for (int m = 0; m < 4; m++) {
        const int32_t *p = polys[m];
        size_t deg = N - 1;
        while (p[deg] == 0 && deg > 0) deg--;
        for (size_t i = 0; i < N; i++) {
            for (size_t j = i; j <= deg + i && j < N; j++) A[i * N + j] = p[j - i];
            for (size_t j = 0; j + N <= deg + i; j++) A[i * N + j] = -p[j + N - i];
        }
        const size_t r0 = (m >= 2) ? N : 0, c0 = (m & 1) ? N : 0;
        for (size_t i = 0; i < N; i++)
            for (size_t j = 0; j < N; j++) b[(r0 + i) * N2 + c0 + j] = (m & 1) ? -A[i * N + j] : A[i * N + j];
    }
    std::vector<float> gs(N2 * N2), inv(N2), v(N2), v1(N2);
    const int32_t *f = fgFG, *g = fgFG + N;
    for (size_t i = 0; i < N; i++) { gs[i] = (float)g[i]; gs[N + i] = (float)-f[i]; }
    float C_k = 0, D_k = 0;
    for (int half = 0; half < 2; half++) {
        const size_t base = half ? N : 0;
        if (half) {
            const float D_k_inv = 1 / D_k;
            for (size_t i = 0; i < N; i++) {
                gs[N * N2 + N + i] = gs[(N - 1) * N2 + N - 1 - i] * (float)h->P.q * D_k_inv;
                gs[N * N2 + i] = -gs[(N - 1) * N2 + N2 - 1 - i] * (float)h->P.q * D_k_inv;
            }
        }
        const float *r0 = &gs[base * N2];
        for (size_t i = 0; i + 1 < N; i++) { v[i] = r0[i + 1]; v[N + i] = r0[i + N + 1]; }
        v[N - 1] = -r0[0];
        v[N2 - 1] = -r0[N];
        v1 = v;
        C_k = dot_seq(r0, v1.data(), N2);
        D_k = half ? dot_seq(r0, r0, N2) : dot_seq(v.data(), v.data(), N2);
        for (size_t i = base + 1; i < base + N; i++) {
            const float aux = C_k / D_k;
            float *ri = &gs[i * N2];
            const float *rp = ri - N2;
            ri[0] = -rp[N - 1] + aux * v[N - 1];
            ri[N] = -rp[N2 - 1] + aux * v[N2 - 1];
            for (size_t j = 1; j < N; j++) {
                ri[j] = rp[j - 1] - aux * v[j - 1];
                ri[N + j] = rp[N + j - 1] - aux * v[N + j - 1];
            }
            for (size_t j = 0; j < N2; j++) v[j] -= aux * rp[j];
            const float C_ko = C_k, D_ko = D_k;
            C_k = dot_seq(ri, v1.data(), N2);
            D_k = D_ko - C_ko * C_ko / D_ko;
        }
    }
    for (size_t j = 0; j < N2; j++) {
        const double sqr = dot_seq(&gs[j * N2], &gs[j * N2], N2);
        inv[j] = (float)(1.0 / sqrt(sqr));
    }
    return sc_hip_ens_dlp_signer_set_key(h, b.data(), gs.data(), inv.data());
}

extern "C" size_t sc_hip_ens_dlp_sig_bytes(const sc_hip_ens_dlp_signer_t *h) { return h ? ((size_t)h->P.n * (size_t)h->P.q_bits + 7) / 8 : 0; }
extern "C" size_t sc_hip_ens_dlp_sig_tape_words(const sc_hip_ens_dlp_signer_t *h) { return h ? 4 * (size_t)h->P.n : 0; }

// Klein's sampler for `batch` targets c (n words each) -> the state ci [2n][B]; want_s2: the register form keeps the first half of the state too
static int klein_launch(sc_hip_ens_dlp_signer_t *h, size_t batch, size_t B, const int32_t *c, const uint32_t *d_tapes, size_t tape_stride, const uint32_t *d_tape_lens,
                        int32_t *ci, uint32_t *d_status, int want_s2, hipStream_t s)
{
    const size_t n = (size_t)h->P.n, n2 = 2 * n;
    const uint32_t nb = (uint32_t)batch;
    KleinArgs A;
    A.batch = nb; A.B = (uint32_t)B; A.n = (int)n; A.n2 = (int)n2; A.c = c; A.b = h->d_b; A.bgs = h->d_bgs; A.inv = h->d_inv; A.tabs = h->d_tabs; A.sizes = h->d_sizes;
    A.ts_log = h->ts_log; A.maxb = h->maxb; A.tapes = d_tapes; A.tape_stride = (uint32_t)tape_stride; A.tape_lens = d_tape_lens; A.ci = ci; A.status = d_status; A.want_s2 = want_s2;
    // lanes per signature.  Register form: the fewest (down to 128 state words per lane) that still give every SIMD a wavefront (two, below).
    // sc_hip_ens_dlp_signer_set_mapping: 1 = the lane-per-signature kernel (state in memory), 4 / 8 / 16 = the register form with that many lanes.
    int G = 16, in_regs = 1;
    const int gmin = (int)(n / 64);                              // 4 / 8 / 16 for n = 256 / 512 / 1024
    // (round 3, with the additions off the DPP hazard: half the lanes are worth it as soon as every SIMD still gets ONE wavefront - n = 512 at 8192: 1.60 M
    // sign/s on 8 lanes against 1.41 M on 16 - except for the step to 4 lanes at n = 256, whose 128 state words per lane against a 512-addition chain want two:
    // 7.0 M against 7.65 M on 8 lanes at 16384)
    while (G > gmin && batch * (size_t)(G / 2) / 64 >= (size_t)h->ctx->cus * 4 * ((n == 256 && G == 8) ? 2 : 1)) G /= 2;
    if (h->force_lanes == 1) { G = 1; in_regs = 0; }
    if ((h->force_lanes == 4 || h->force_lanes == 8 || h->force_lanes == 16) && h->force_lanes >= gmin) G = h->force_lanes;
    if (in_regs) {
        const dim3 grid((unsigned)((B / (64 / G) + KW - 1) / KW));
        // the next row's tape word and table entries held across the sweep (PF): measured better at every size since the additions stopped waiting on
        // DPP hazards (n = 512 at 4096 / 16384 / 65536: 1.06 / 2.34 / 2.60 M sign/s against 0.62 / 2.01 / 2.51 M for the other form on 8 lanes; n = 256 at
        // 16384: 7.7 against 6.5 M); the other form stays for the tests and as the statement of the loop without the software pipelining
        const bool pf = h->force_pf >= 0;
#define SC_KLEIN(GG, NN) do { if (pf) hipLaunchKernelGGL((kleinr_kernel<GG, NN, true>), grid, dim3(64 * KW), 0, s, A); \
                              else hipLaunchKernelGGL((kleinr_kernel<GG, NN, false>), grid, dim3(64 * KW), 0, s, A); } while (0)
        if (n == 256 && G == 4) SC_KLEIN(4, 128);
        else if (n == 256 && G == 8) SC_KLEIN(8, 64);
        else if (n == 256) SC_KLEIN(16, 32);
        else if (n == 512 && G == 8) SC_KLEIN(8, 128);
        else if (n == 512) SC_KLEIN(16, 64);
        else SC_KLEIN(16, 128);
#undef SC_KLEIN
    }
    else hipLaunchKernelGGL(klein_kernel, dim3((unsigned)(B / 64)), dim3(64), 0, s, A);
    HIPCHK(hipGetLastError());
    return SC_HIP_SUCCESS;
}

// replaces: safecrypto_sign -> ens_dlp_sig_sign (src/schemes/sig/ens_dlp/ens_dlp_sig.c:1171-1315, ENS and DLP signatures without message recovery)
// for `batch` messages under one private key: t = H(m) (SHAKE128, :728-742), Klein's sampler (gaussian_lattice_sample_flt, gpv.c:2409-2492),
// s1 = centre(ci[n, 2n)), n q_bits-bit signed fields.  d_tapes row i = the prng_32 word stream of sc->prng_ctx[0] during signature i: one prng_64
// (two words, high word first) per basis row = 4n words; rows 8-byte aligned.  d_status: 1 signed, 2 tape too short, 3 the sampler state left 32 bits.
extern "C" int sc_hip_ens_dlp_sig_sign(sc_hip_ens_dlp_signer_t *h, size_t batch, const uint8_t *d_msgs, size_t msg_stride, const uint32_t *d_mlens, uint32_t fixed_len,
                                       const uint32_t *d_tapes, size_t tape_stride, const uint32_t *d_tape_lens, uint8_t *d_sigs, size_t sig_stride, uint32_t *d_status, void *stream)
{
    if (!h) return fail("ens_dlp_sig_sign: null handle");
    if (!h->have_key) return fail("ens_dlp_sig_sign: no key loaded");
    if (batch == 0) return SC_HIP_SUCCESS;
    if (batch > 0x7fffffc0u) return fail("ens_dlp_sig_sign: batch too large");
    if (!d_msgs || !d_tapes || !d_sigs || !d_status) return fail("ens_dlp_sig_sign: null buffer");
    const size_t n = (size_t)h->P.n, n2 = 2 * n;
    if (tape_stride < 2 * n2 || tape_stride > 0xffffffffu || (tape_stride & 1) != 0 || ((uintptr_t)d_tapes & 7) != 0)
        return fail("ens_dlp_sig_sign: tapes must be 8-byte aligned rows of at least 4n words (even tape_stride)");
    if (sig_stride < sc_hip_ens_dlp_sig_bytes(h)) return fail("ens_dlp_sig_sign: sig_stride too small");
    HIPCHK(hipSetDevice(h->ctx->device));
    const size_t B = (batch + 63) & ~(size_t)63;
    if (h->ws.reserve(B * n2 * 4 + batch * n * 4 + 1024)) return SC_HIP_FAILURE;
    int32_t *ci = h->ws.take<int32_t>(B * n2);
    int32_t *c = h->ws.take<int32_t>(batch * n);
    hipStream_t s = (hipStream_t)stream;
    const uint32_t nb = (uint32_t)batch;
    if (batch <= (size_t)h->ctx->cus * 4 * 64) hipLaunchKernelGGL(fal_hash2_kernel, dim3((2 * nb + HT - 1) / HT), dim3(HT), 0, s, nb, d_msgs, msg_stride, d_mlens, fixed_len, c, h->P, 1);
    else hipLaunchKernelGGL(fal_hash1_kernel, dim3((nb + HT - 1) / HT), dim3(HT), 0, s, nb, d_msgs, msg_stride, d_mlens, fixed_len, c, h->P, 1);
    HIPCHK(hipGetLastError());
    if (klein_launch(h, batch, B, c, d_tapes, tape_stride, d_tape_lens, ci, d_status, 0, s)) return SC_HIP_FAILURE;
    hipLaunchKernelGGL(klein_pack_kernel, dim3((unsigned)((batch + 255) / 256), (unsigned)(n / 8)), dim3(256), 0, s, nb, (uint32_t)B, ci, d_status, d_sigs, sig_stride, (int)n, h->P.q, h->P.q_bits);
    HIPCHK(hipGetLastError());
    h->ws.wipe(s);                                                  // the sampler state (the lattice point before packing)
    return SC_HIP_SUCCESS;
}

// ---- ENS / DLP signatures WITH MESSAGE RECOVERY (SC_SCHEME_SIG_ENS_WITH_RECOVERY / _DLP_WITH_RECOVERY) ----------------------------------------
// ens_dlp_sig_sign_recovery / ens_dlp_sig_verify_recovery (src/schemes/sig/ens_dlp/ens_dlp_sig.c:1319-1501, 1663-1826), the XOF forms of the random
// oracles (what the default build compiles, ens_dlp_sig.h:31).  m = m1 || m2: the first (n - k)(q_bits - 1) bits ride in the target of the lattice
// point - t2 = H'(m) (k coefficients: SHAKE128 over m, zero-extended to the m1 bytes when shorter), t1 = F(t2) + m1 (F: SHAKE128 over the k words of
// t2, two bytes per coefficient, first byte high, q_bits - 1 bits kept; m1 added WITHOUT reduction) - the signature is s1 | s2 and the signer hands
// back m2.  k = k_recovery travels in the parameter block's first spare word (sc_hip_falcon_params_t::pad[0]; ens_dlp_sig_params.c:23-62).
namespace {

struct RecP { int n, k, q, q_bits; uint32_t m1_bits, m1_bytes; };

static inline RecP rec_params(const FalP &P)
{
    RecP r;
    r.n = P.n; r.k = P.pad[0]; r.q = P.q; r.q_bits = P.q_bits;
    r.m1_bits = (uint32_t)(P.n - P.pad[0]) * (uint32_t)(P.q_bits - 1); r.m1_bytes = (r.m1_bits + 7) >> 3;
    return r;
}

static inline bool rec_params_ok(const FalP &P) { return P.pad[0] >= 1 && P.pad[0] < P.n / 2 && P.q_bits > 8 && P.q_bits <= 16; }    // f_function's own range check (:754-756)

// H'(m) -> k words masked to q_bits, q subtracted when >= q (oracle_recovery_xof :686-697 / oracle_xof :660-665, h_function_xof :728-742)
template <class SP>
__device__ __forceinline__ void rec_hash(SP &sp, const uint8_t *m, uint32_t mlen, uint32_t min_len, int32_t *out, const RecP &P)
{
    sp.absorb(m, mlen);
    for (uint32_t i = mlen; i < min_len; i++) sp.absorb_byte(0);
    sp.finalize(0x1F);
    const uint32_t mask = (1u << P.q_bits) - 1u;
    for (int i = 0; i < P.k; i++) {
        uint32_t y = 0;
        for (int j = 0; j < 4; j++) y |= (uint32_t)sp.squeeze_byte() << (8 * j);
        y &= mask;
        out[i] = (int32_t)(y - (y >= (uint32_t)P.q ? (uint32_t)P.q : 0u));
    }
}

// the sponge of F absorbed and finalised (f_function :746-792): the k words of t2 as they lie in memory
template <class SP>
__device__ __forceinline__ void rec_f_begin(SP &sp, const int32_t *t2, const RecP &P)
{
    for (int i = 0; i < P.k; i++) {
        const uint32_t v = (uint32_t)t2[i];
        for (int j = 0; j < 4; j++) sp.absorb_byte((uint8_t)(v >> (8 * j)));
    }
    sp.finalize(0x1F);
}

// One lane per message: the target c (n words) and the remainder m2.
__global__ __launch_bounds__(HT) void rec_target_kernel(uint32_t batch, const uint8_t *__restrict__ msgs, size_t msg_stride, const uint32_t *__restrict__ mlens, uint32_t fixed_len,
                                                       int32_t *__restrict__ c, uint8_t *__restrict__ m2, size_t m2_stride, uint32_t *__restrict__ m2lens, RecP P)
{
    __shared__ uint64_t st[25 * HT];
    const uint32_t b = blockIdx.x * HT + threadIdx.x;
    Sponge<HT> sp; sp.init(st, threadIdx.x, 168);
    if (b >= batch) return;
    const uint32_t mlen = mlens ? mlens[b] : fixed_len;
    const uint8_t *m = msgs + (size_t)b * msg_stride;
    int32_t *cb = c + (size_t)b * P.n;
    rec_hash(sp, m, mlen, P.m1_bytes, cb + P.n - P.k, P);
    sp.init(st, threadIdx.x, 168);
    rec_f_begin(sp, cb + P.n - P.k, P);
    // add_recovery_msg (:797-832): q_bits - 1 bits per coefficient from the first min(m1 bytes, mlen) bytes, MSB first, zero behind them
    const uint32_t avail = mlen < P.m1_bytes ? mlen : P.m1_bytes, fb = (uint32_t)P.q_bits - 1u, fmask = (1u << fb) - 1u;
    uint64_t acc = 0; uint32_t nb = 0, rp = 0;
    for (int i = 0; i < P.n - P.k; i++) {
        const uint32_t hi = sp.squeeze_byte(), lo = sp.squeeze_byte();
        while (nb < fb) { acc = (acc << 8) | (uint64_t)(rp < avail ? m[rp] : 0); rp++; nb += 8; }
        const uint32_t v = (uint32_t)(acc >> (nb - fb)) & fmask;
        nb -= fb;
        cb[i] = (int32_t)((((hi << 8) | lo) & fmask) + v);
    }
    // m2 (:1455-1479): what lies behind the m1 bits, moved up to a byte boundary; the reference's loop gives the last byte no bits of a successor, and keeps
    // nothing at all of a message that ends inside the last m1 byte
    uint32_t l2 = 0;
    if (mlen > P.m1_bytes) {
        const uint32_t sh = P.m1_bits & 7u, base = P.m1_bits >> 3, masku = 0xFFu >> sh, maskl = masku ^ 0xFFu;
        uint8_t *o = m2 + (size_t)b * m2_stride;
        l2 = mlen - base;
        if (l2 > m2_stride) l2 = (uint32_t)m2_stride;                  // (the host checked the strides; a length beyond its row is the caller's error)
        for (uint32_t i = 0; i < l2; i++) {
            const uint32_t cur = m[base + i], nxt = (base + i + 1 < mlen) ? m[base + i + 1] : 0u;
            o[i] = (uint8_t)(((cur & masku) << sh) | ((nxt & maskl) >> (8 - sh)));
        }
    }
    m2lens[b] = l2;
}

// The verifier in four kernels: the two sponges (F over t2, H' over the rebuilt message: seven and six permutations at n = 512) are lane-per-signature
// kernels; run by one lane of a wavefront that also does the wide work they were 98 % of its instructions on 1 / 64 of its lanes (2.8 ms per 16384 ENS
// signatures; the wide work - unpacking, norm, the wrapping schoolbook product - is a tenth of that).  verdicts[] is zeroed by the caller; a stage that refuses a
// signature sets 1 and the later stages pass it by.
//
// (1) a wavefront per signature: unpack s1 | s2, the norm, t = s1 h + s2 (schoolbook, 32-bit words that wrap: small_mul_mod_ring :892-912), normalised -> tbuf
__global__ __launch_bounds__(64) void rec_v1_kernel(uint32_t batch, const uint8_t *__restrict__ sigs, size_t sig_stride, const uint32_t *__restrict__ siglens,
                                                   size_t m2_stride, const uint32_t *__restrict__ m2lens, const int32_t *__restrict__ hraw, size_t out_stride,
                                                   int32_t *__restrict__ tbuf, uint32_t *__restrict__ verdicts, RecP P, float bd)
{
    extern __shared__ __attribute__((aligned(16))) int32_t rv_lds[];          // s1 | s2 | t | h (n words each)
    const int lane = threadIdx.x, n = P.n;
    const uint32_t b = blockIdx.x;
    int32_t *s1 = rv_lds, *s2 = s1 + n, *t = s2 + n, *hh = t + n;
    const uint32_t l2 = m2lens ? m2lens[b] : 0u, siglen = siglens[b];
    const uint32_t base = P.m1_bits >> 3;
    const uint32_t olen = l2 ? base + l2 : P.m1_bytes;                         // (m1_bits + 8 l2) >> 3 (:1757)
    if ((size_t)siglen * 8 < (size_t)2 * n * P.q_bits || siglen > sig_stride || (size_t)l2 > m2_stride || (size_t)olen > out_stride) { if (lane == 0) verdicts[b] = 1; return; }
    const uint8_t *sg = sigs + (size_t)b * sig_stride;
    for (int i = lane; i < 2 * n; i += 64) {                                   // a field (q_bits <= 16) and its skew fit the four bytes from its first one
        const uint32_t bp = (uint32_t)i * P.q_bits, b0 = bp >> 3;
        uint32_t w = 0;
        if ((size_t)b0 + 4 <= sig_stride) { __builtin_memcpy(&w, sg + b0, 4); w = __builtin_bswap32(w); }
        else for (uint32_t k = 0; k < 4; k++) w |= (b0 + k < siglen ? (uint32_t)sg[b0 + k] : 0u) << (24 - 8 * k);      // the row's last bytes
        const uint32_t val = (w >> (32 - (bp & 7) - P.q_bits)) & ((1u << P.q_bits) - 1u);
        s1[i] = (int32_t)(val << (32 - P.q_bits)) >> (32 - P.q_bits);       // (s2 follows s1 in LDS)
    }
    for (int i = lane; i < n; i += 64) hh[i] = hraw[i];
    __syncthreads();
    for (int i = lane; i < n; i += 64) t[i] = s1[i] * s1[i] + s2[i] * s2[i];
    __syncthreads();
    int bad = 0;
    if (lane == 0) bad = norm_fails_lds(t, n, bd) ? 1 : 0;                     // check_norm_bd on the fields as decoded (:1726-1729)
    bad = __shfl(bad, 0);
    if (bad) { if (lane == 0) verdicts[b] = 1; return; }
    __syncthreads();
    // t[j] = sum_k s1[k] H[n + j - k] with the shift table H[m] = -h[m] (m < n), h[m - n] (m >= n) in t | hh (the sign of the wrapped part is in the table);
    // a lane owns the coefficients j = lane + 64 c: a step is one broadcast read of s1[k], n / 64 reads at constant offsets from one address and as many
    // 32-bit multiply-adds that wrap - the reference's words
    for (int i = lane; i < n; i += 64) t[i] = -hh[i];
    __syncthreads();
    constexpr int MAXC = 16;                                                   // n <= 1024
    const int C = n / 64;
    uint32_t acc[MAXC];
#pragma unroll
    for (int c = 0; c < MAXC; c++) acc[c] = 0;
    const int32_t *row0 = t + n + lane;                                        // H = t | hh: contiguous in LDS
    // (both factors fit 24 signed bits - a q_bits <= 16 field, a SINT16 key coefficient - so a term is one v_mad_i32_i24: exact product, sum that wraps)
#pragma unroll 4
    for (int k = 0; k < n; k++) {
        const int32_t sk = s1[k];
        const int32_t *row = row0 - k;
#pragma unroll
        for (int c = 0; c < MAXC; c++) if (c < C) acc[c] += (uint32_t)__mul24(sk, row[64 * c]);
    }
    int32_t *to = tbuf + (size_t)b * n;
#pragma unroll
    for (int c = 0; c < MAXC; c++) {
        if (c < C) {
            const int j = lane + 64 * c;
            int32_t v = (int32_t)(acc[c] + (uint32_t)s2[j]) % P.q;             // normalize_32: any correct reduction of a 32-bit word gives this canonical value
            if (v < 0) v += P.q;
            to[j] = v;
        }
    }
}

// (2) a lane per signature: F(t2) -> fbuf (n - k words), two bytes per coefficient (f_function :746-792)
__global__ __launch_bounds__(HT) void rec_v2_kernel(uint32_t batch, const int32_t *__restrict__ tbuf, int32_t *__restrict__ fbuf, const uint32_t *__restrict__ verdicts, RecP P)
{
    __shared__ uint64_t st[25 * HT];
    const uint32_t b = blockIdx.x * HT + threadIdx.x;
    Sponge<HT> sp; sp.init(st, threadIdx.x, 168);
    if (b >= batch || verdicts[b]) return;
    rec_f_begin(sp, tbuf + (size_t)b * P.n + P.n - P.k, P);
    const uint32_t fmask = (1u << (P.q_bits - 1)) - 1u;
    int32_t *fo = fbuf + (size_t)b * P.n;
    for (int i = 0; i < P.n - P.k; i++) { const uint32_t hi = sp.squeeze_byte(), lo = sp.squeeze_byte(); fo[i] = (int32_t)(((hi << 8) | lo) & fmask); }
}

// (3) a wavefront per signature: m1 = t1 - F(t2) with its range check (recover_msg :852-856), the message rebuilt in the output row
__global__ __launch_bounds__(64) void rec_v3_kernel(uint32_t batch, const int32_t *__restrict__ tbuf, const int32_t *__restrict__ fbuf, const uint8_t *__restrict__ m2, size_t m2_stride,
                                                   const uint32_t *__restrict__ m2lens, uint8_t *__restrict__ out, size_t out_stride, uint32_t *__restrict__ outlens,
                                                   uint32_t *__restrict__ verdicts, RecP P)
{
    extern __shared__ __attribute__((aligned(16))) int32_t rv_lds[];          // m1's coefficients
    const int lane = threadIdx.x, n = P.n;
    const uint32_t b = blockIdx.x;
    if (verdicts[b]) { if (lane == 0) outlens[b] = 0; return; }
    int32_t *s1 = rv_lds;
    const int32_t *t = tbuf + (size_t)b * n, *fo = fbuf + (size_t)b * n;
    const uint32_t l2 = m2lens ? m2lens[b] : 0u;
    const uint32_t sh = P.m1_bits & 7u, base = P.m1_bits >> 3;
    const uint32_t olen = l2 ? base + l2 : P.m1_bytes;
    const uint32_t bound = (1u << (P.q_bits - 1)) - 1u;
    int over = 0;
    for (int i = lane; i < n - P.k; i += 64) {
        int32_t v = t[i] - fo[i];                                              // in (-2^(q_bits-1), q): one conditional addition normalises it (:1782-1788)
        if (v < 0) v += P.q;
        s1[i] = v;
        over |= ((uint32_t)v > bound) ? 1 : 0;
    }
    if (__any(over)) { if (lane == 0) { verdicts[b] = 1; outlens[b] = 0; } return; }
    __syncthreads();
    // the row: m1's bits from the front; m2 behind them, moved back down by the bits it was moved up; the byte they share takes m1's bits on top
    uint8_t *ob = out + (size_t)b * out_stride;
    const uint32_t fb = (uint32_t)P.q_bits - 1u, masku = 0xFFu >> sh, maskl = masku ^ 0xFFu;
    for (uint32_t j = lane; j < olen; j += 64) {
        uint32_t byte = 0;
        if (j < P.m1_bytes) {
            for (int k = 0; k < 8; k++) {
                const uint32_t bp = 8 * j + k;
                if (bp < P.m1_bits) { const uint32_t ci = bp / fb, bi = bp - ci * fb; byte |= (((uint32_t)s1[ci] >> (fb - 1 - bi)) & 1u) << (7 - k); }
            }
        }
        if (l2 && j >= base) {
            const uint32_t i = j - base;
            const uint8_t *mb = m2 + (size_t)b * m2_stride;
            const uint32_t lo = ((uint32_t)mb[i] >> sh) & masku, hi = i ? (((uint32_t)mb[i - 1] << (8 - sh)) & maskl) : 0u;
            byte = (j == base && sh) ? ((byte & maskl) | lo) : (hi | lo);
        }
        ob[j] = (uint8_t)byte;
    }
    if (lane == 0) outlens[b] = olen;
}

// (4) a lane per signature: H'(m) over the rebuilt message against t2 (:1797-1812)
__global__ __launch_bounds__(HT) void rec_v4_kernel(uint32_t batch, const int32_t *__restrict__ tbuf, const uint8_t *__restrict__ out, size_t out_stride, uint32_t *__restrict__ outlens,
                                                   uint32_t *__restrict__ verdicts, int32_t *__restrict__ hbuf, RecP P)
{
    __shared__ uint64_t st[25 * HT];
    const uint32_t b = blockIdx.x * HT + threadIdx.x;
    Sponge<HT> sp; sp.init(st, threadIdx.x, 168);
    if (b >= batch || verdicts[b]) return;
    int32_t *hv = hbuf + (size_t)b * P.n;                                       // (the F row is done with: its first k words take the hash)
    rec_hash(sp, out + (size_t)b * out_stride, outlens[b], 0, hv, P);
    const int32_t *t2 = tbuf + (size_t)b * P.n + P.n - P.k;
    int neq = 0;
    for (int i = 0; i < P.k; i++) neq |= (hv[i] != t2[i]) ? 1 : 0;
    if (neq) { verdicts[b] = 1; outlens[b] = 0; }
}

}  // namespace

extern "C" size_t sc_hip_ens_dlp_sig_rec_bytes(const sc_hip_ens_dlp_signer_t *h) { return h ? (2 * (size_t)h->P.n * (size_t)h->P.q_bits + 7) / 8 : 0; }
extern "C" size_t sc_hip_ens_dlp_sig_rec_m1_bytes(const sc_hip_ens_dlp_signer_t *h) { return (h && rec_params_ok(h->P)) ? rec_params(h->P).m1_bytes : 0; }

// replaces: safecrypto_sign_with_recovery -> ens_dlp_sig_sign_recovery (ens_dlp_sig.c:1319-1501) for `batch` messages under one private key.
// Out of place where the reference works in the caller's buffer: d_m2 row i / d_m2lens[i] = what the reference leaves in *m / *m_len.
// d_status as sc_hip_ens_dlp_sig_sign.  Signatures are 2 n q_bits / 8 bytes (s1 | s2).
extern "C" int sc_hip_ens_dlp_sig_sign_recovery(sc_hip_ens_dlp_signer_t *h, size_t batch, const uint8_t *d_msgs, size_t msg_stride, const uint32_t *d_mlens, uint32_t fixed_len,
                                                const uint32_t *d_tapes, size_t tape_stride, const uint32_t *d_tape_lens, uint8_t *d_sigs, size_t sig_stride,
                                                uint8_t *d_m2, size_t m2_stride, uint32_t *d_m2lens, uint32_t *d_status, void *stream)
{
    if (!h) return fail("ens_dlp_sig_sign_recovery: null handle");
    if (!h->have_key) return fail("ens_dlp_sig_sign_recovery: no key loaded");
    if (!rec_params_ok(h->P)) return fail("ens_dlp_sig_sign_recovery: the handle's parameters carry no k_recovery (params.pad[0]) or q_bits is outside 9..16");
    if (batch == 0) return SC_HIP_SUCCESS;
    if (batch > 0x7fffffc0u) return fail("ens_dlp_sig_sign_recovery: batch too large");
    if (!d_msgs || !d_tapes || !d_sigs || !d_status || !d_m2 || !d_m2lens) return fail("ens_dlp_sig_sign_recovery: null buffer");
    const size_t n = (size_t)h->P.n, n2 = 2 * n;
    const RecP R = rec_params(h->P);
    if (tape_stride < 2 * n2 || tape_stride > 0xffffffffu || (tape_stride & 1) != 0 || ((uintptr_t)d_tapes & 7) != 0)
        return fail("ens_dlp_sig_sign_recovery: tapes must be 8-byte aligned rows of at least 4n words (even tape_stride)");
    if (sig_stride < sc_hip_ens_dlp_sig_rec_bytes(h)) return fail("ens_dlp_sig_sign_recovery: sig_stride too small (2 n q_bits / 8 bytes)");
    const size_t longest = d_mlens ? msg_stride : (size_t)fixed_len;
    if (longest > (R.m1_bits >> 3) && m2_stride < longest - (R.m1_bits >> 3)) return fail("ens_dlp_sig_sign_recovery: m2_stride too small for the longest message's remainder");
    HIPCHK(hipSetDevice(h->ctx->device));
    const size_t B = (batch + 63) & ~(size_t)63;
    if (h->ws.reserve(B * n2 * 4 + batch * n * 4 + 1024)) return SC_HIP_FAILURE;
    int32_t *ci = h->ws.take<int32_t>(B * n2);
    int32_t *c = h->ws.take<int32_t>(batch * n);
    hipStream_t s = (hipStream_t)stream;
    const uint32_t nb = (uint32_t)batch;
    hipLaunchKernelGGL(rec_target_kernel, dim3((nb + HT - 1) / HT), dim3(HT), 0, s, nb, d_msgs, msg_stride, d_mlens, fixed_len, c, d_m2, m2_stride, d_m2lens, R);
    HIPCHK(hipGetLastError());
    if (klein_launch(h, batch, B, c, d_tapes, tape_stride, d_tape_lens, ci, d_status, 1, s)) return SC_HIP_FAILURE;
    hipLaunchKernelGGL(klein_pack_kernel, dim3((unsigned)((batch + 255) / 256), (unsigned)(n2 / 8)), dim3(256), 0, s, nb, (uint32_t)B, ci, d_status, d_sigs, sig_stride, (int)n, h->P.q, h->P.q_bits);
    HIPCHK(hipGetLastError());
    h->ws.wipe(s);
    return SC_HIP_SUCCESS;
}

// replaces: safecrypto_verify_with_recovery -> ens_dlp_sig_verify_recovery (ens_dlp_sig.c:1663-1826) for `batch` signatures under one public key (the
// Falcon-shaped handle of sc_hip_ens_dlp_sig_verify, created with k_recovery in params.pad[0]).  In: the signatures and the remainders m2 (d_m2lens null:
// none).  Out: d_verdicts[i] = 0 valid / 1 invalid, and for a valid signature the rebuilt message in row i of d_msgs_out (d_mlens_out[i] bytes: the m1
// bytes, then what m2 carried) - what the reference leaves in *m when it is handed m2 in a ZEROED buffer (it takes the low bits of the byte m1 and m2
// share from the caller's buffer when m2 is empty).  A refused signature leaves length 0.
extern "C" int sc_hip_ens_dlp_sig_verify_recovery(sc_hip_falcon_t *f, size_t batch, const uint8_t *d_sigs, size_t sig_stride, const uint32_t *d_siglens,
                                                  const uint8_t *d_m2, size_t m2_stride, const uint32_t *d_m2lens,
                                                  uint8_t *d_msgs_out, size_t out_stride, uint32_t *d_mlens_out, uint32_t *d_verdicts, void *stream)
{
    if (fal_check_common(f, batch, "ens_dlp_sig_verify_recovery")) return SC_HIP_FAILURE;
    if (!f->have_pk) return fail("ens_dlp_sig_verify_recovery: no public key loaded");
    if (!rec_params_ok(f->P)) return fail("ens_dlp_sig_verify_recovery: the handle's parameters carry no k_recovery (params.pad[0]) or q_bits is outside 9..16");
    if (batch == 0) return SC_HIP_SUCCESS;
    if (!d_sigs || !d_siglens || !d_msgs_out || !d_mlens_out || !d_verdicts || (d_m2lens && !d_m2)) return fail("ens_dlp_sig_verify_recovery: null buffer");
    const RecP R = rec_params(f->P);
    if (out_stride < R.m1_bytes) return fail("ens_dlp_sig_verify_recovery: out_stride smaller than the bytes a signature carries");
    HIPCHK(hipSetDevice(f->ctx->device));
    hipStream_t s = (hipStream_t)stream;
    if (fal_ensure_ws(f, batch)) return SC_HIP_FAILURE;
    const uint32_t B = (uint32_t)batch;
    const size_t n = (size_t)f->P.n;
    HIPCHK(hipMemsetAsync(d_verdicts, 0, batch * 4, s));
    HIPCHK(hipMemsetAsync(d_mlens_out, 0, batch * 4, s));
    hipLaunchKernelGGL(rec_v1_kernel, dim3(B), dim3(64), 4 * n * sizeof(int32_t), s, B, d_sigs, sig_stride, d_siglens, m2_stride, d_m2lens, f->d_hraw, out_stride, f->s1, d_verdicts, R, f->P.bd);
    hipLaunchKernelGGL(rec_v2_kernel, dim3((B + HT - 1) / HT), dim3(HT), 0, s, B, f->s1, f->s2, d_verdicts, R);
    hipLaunchKernelGGL(rec_v3_kernel, dim3(B), dim3(64), n * sizeof(int32_t), s, B, f->s1, f->s2, d_m2, m2_stride, d_m2lens, d_msgs_out, out_stride, d_mlens_out, d_verdicts, R);
    hipLaunchKernelGGL(rec_v4_kernel, dim3((B + HT - 1) / HT), dim3(HT), 0, s, B, f->s1, d_msgs_out, out_stride, d_mlens_out, d_verdicts, f->s2, R);
    HIPCHK(hipGetLastError());
    return SC_HIP_SUCCESS;
}

// ---- host-buffer convenience (sc::sign_host / sc::verify_host in engine_internal.h); Falcon reports no attempt count (it never retries) ----

extern "C" int sc_hip_falcon_sign_host(sc_hip_falcon_t *f, size_t batch, const uint8_t *h_msgs, size_t msg_stride, const uint32_t *h_mlens, uint32_t fixed_len,
                                       const uint32_t *h_tapes, size_t tape_stride, const uint32_t *h_tape_lens,
                                       uint8_t *h_sigs, size_t sig_stride, uint32_t *h_siglens, uint32_t *h_status)
{
    if (!f) return fail("falcon_sign_host: null handle");
    return sc::sign_host(f->ctx->device, batch, h_msgs, msg_stride, h_mlens, h_tapes, tape_stride, h_tape_lens, h_sigs, sig_stride, h_siglens, nullptr, h_status,
        [&](const uint8_t *m, const uint32_t *ml, const uint32_t *tp, const uint32_t *tl, uint8_t *sg, uint32_t *sl, uint32_t *, uint32_t *st) {
            return sc_hip_falcon_sign(f, batch, m, msg_stride, ml, fixed_len, tp, tape_stride, tl, sg, sig_stride, sl, st, nullptr);
        });
}

extern "C" int sc_hip_falcon_verify_host(sc_hip_falcon_t *f, size_t batch, const uint8_t *h_msgs, size_t msg_stride, const uint32_t *h_mlens, uint32_t fixed_len,
                                         const uint8_t *h_sigs, size_t sig_stride, const uint32_t *h_siglens, uint32_t *h_verdicts)
{
    if (!f) return fail("falcon_verify_host: null handle");
    return sc::verify_host(f->ctx->device, batch, h_msgs, msg_stride, h_mlens, h_sigs, sig_stride, h_siglens, h_verdicts,
        [&](const uint8_t *m, const uint32_t *ml, const uint8_t *sg, const uint32_t *sl, uint32_t *v) {
            return sc_hip_falcon_verify(f, batch, m, msg_stride, ml, fixed_len, sg, sig_stride, sl, v, nullptr);
        });
}
