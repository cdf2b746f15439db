// tesla.hip - batched ring-TESLA sign / verify on MI355X behind the C ABI of include/sc_hip.h.
//
// Replaces, for a batch of messages under one key, ring_tesla_sign / ring_tesla_verify
// (src/schemes/sig/ring_tesla/ring_tesla.c:752-910, 922-1040; raw signature coding) with test_w / test_rejection (:542-595) and
// the random oracle of src/schemes/sig/ring_tesla/random_oracle.c: poly_rounding + SHA3-512 (the reference's permutation,
// keccak_dev.h) over the big-endian words, the Salsa20-keyed stream (random_stream) and populate_c.
//
// Two behaviours of the reference are reproduced, not repaired: salsa20_core stores its result with SC_BIG_ENDIAN_32_COPY
// (src/utils/crypto/salsa/salsa20.c:93), so every 32-bit word of the stream is byte-swapped relative to Salsa20; and
// poly_rounding computes (p - modn(p)) >> d with modn over the ring modulus q, which is 0 for every centred or normalised
// coefficient.  The rounding is still evaluated here as written; when it is 0 for a whole message - always, with the reference's
// arithmetic - the hash resumes from the sponge state after the 56 whole all-zero blocks, kept per handle.
//
// Randomness: n prng_32 words of sc->prng_ctx[0] per attempt (ring_tesla.c:818-822); each message travels with its word tape
// as for BLISS (or the tape is expanded on the device with sc_hip_drbg_words).  Mapping: y / products on the batched NTT kernels
// (the public a1, a2 are NTT-domain constants of the reference, passed in as data), one lane per message for the hash and the
// Salsa20 stream, one wavefront per message for the sparse products, the three tests and the packing.
#include "engine_internal.h"
#include "keccak_dev.h"
#include "drbg_dev.h"
namespace sc { const DrbgConst &drbg_const(); }

#include <string.h>

using namespace sc;

namespace {

constexpr int HT = 64;

struct TsP { int32_t n, n_bits, q, q_bits, b, b_bits, omega, d, bound, u, set, pad; };    // ring_tesla_set_t fields (ring_tesla_params.h:43-67)

struct St { uint32_t *pos, *status; };

// Attempts are evaluated S at a time per message ("slots": slot = aidx * S + k tries the k-th next attempt of active message
// aidx, tape words [pos + k n, pos + (k+1) n)) and the LOWEST accepted k wins - exactly the attempt the reference's loop would have
// stopped at.  While most of the batch is active S is 1-2; as the batch thins out the same slot budget goes deeper, so the long
// tail of the rejection loop (P(accept) ~ 0.32) costs a few rounds instead of twenty.
// The round loop runs on the device: rs = { messages still unsigned, slots per message, slots } of the round, derived by its first kernel from the count its
// predecessor's last kernel left behind; every kernel of a round takes its sizes from there (the by-value arguments are the host's upper bounds, which
// size the grids) - so the host queues the rounds its schedule predicts without looking at a count in between (a look idles the device for 25-50 us).
enum { TRS_NACT = 0, TRS_S = 1, TRS_NSLOTS = 2 };
__global__ void ts_rs_kernel(uint32_t *rs, uint32_t *cnt, uint32_t first_nact, uint32_t slot_cap)
{
    const uint32_t nact = first_nact ? first_nact : *cnt;
    uint32_t S = nact ? slot_cap / nact : 0;
    if (S > 32) S = 32;
    if (nact && S < 1) S = 1;
    rs[TRS_NACT] = nact; rs[TRS_S] = S; rs[TRS_NSLOTS] = nact * S;
    *cnt = 0;
}

// A slot's state: does it draw (1) or does its message's tape end before it (2), and where on the tape.  The mask
// kernels work it out themselves (wave-uniform: three scalar loads per kilobyte of mask) and leave need[] / sbase[] behind for the round's later kernels - a
// launch of its own for it was 5 us of a 290 us round
__device__ __forceinline__ uint32_t ts_slot_state(const uint32_t *active, uint32_t slot, uint32_t S, const St &s, const uint32_t *tape_lens, uint32_t tape_stride, int n, uint32_t &item, uint32_t &pos)
{
    const uint32_t aidx = slot / S, k = slot % S;
    item = active ? active[aidx] : aidx;
    const uint32_t len = tape_lens ? tape_lens[item] : tape_stride;
    pos = s.pos[item] + k * (uint32_t)n;
    return pos + (uint32_t)n > len ? 2u : 1u;
}

// y = (word & (2^(b_bits+1) - 1)) - b   (ring_tesla.c:818-822)
// A wavefront takes 256 consecutive coefficients of one slot per step (a lane four of them: 16 bytes of tape in, 16 bytes out), so the slot's state -
// does it draw, which message, where on its tape - is wave-uniform and read once per kilobyte rather than per coefficient.
__global__ __launch_bounds__(256) void ts_y_kernel(const uint32_t *active, uint32_t nslots, uint32_t S, uint32_t *need, uint32_t *sbase, const uint32_t *tapes, size_t tape_stride, int32_t *y, TsP P,
                                                   const uint32_t *rs, St st, const uint32_t *tape_lens)
{
    if (rs) { nslots = rs[TRS_NSLOTS]; S = rs[TRS_S]; }
    const int lps = 31 - __clz(P.n / 256);                                // log2 of the steps per slot (n a power of two, >= 256: tesla_create)
    const size_t steps = (size_t)nslots << lps, nwaves = (size_t)gridDim.x * 4;
    const int lane = threadIdx.x & 63;
    const uint32_t msk = (1u << (P.b_bits + 1)) - 1u;
    for (size_t ws = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6); ws < steps; ws += nwaves) {
        const uint32_t slot = (uint32_t)__builtin_amdgcn_readfirstlane((int)(ws >> lps));
        const int i = (((int)ws & ((1 << lps) - 1)) << 8) + 4 * lane;
        int4 *dst = reinterpret_cast<int4 *>(y + (size_t)slot * P.n + i);
        uint32_t item, pos;
        const uint32_t nd = ts_slot_state(active, slot, S, st, tape_lens, (uint32_t)tape_stride, P.n, item, pos);
        if (i == 0 && lane == 0) { need[slot] = nd; sbase[slot] = pos; }     // (the slot's first step files its state)
        if (nd != 1) { *dst = make_int4(0, 0, 0, 0); continue; }
        const uint32_t *tp = tapes + (size_t)item * tape_stride + pos + i;
        uint32_t w[4];
        if ((reinterpret_cast<uintptr_t>(tp) & 15) == 0) { const uint4 q = *reinterpret_cast<const uint4 *>(tp); w[0] = q.x; w[1] = q.y; w[2] = q.z; w[3] = q.w; }
        else { w[0] = tp[0]; w[1] = tp[1]; w[2] = tp[2]; w[3] = tp[3]; }
        *dst = make_int4((int32_t)(w[0] & msk) - P.b, (int32_t)(w[1] & msk) - P.b, (int32_t)(w[2] & msk) - P.b, (int32_t)(w[3] & msk) - P.b);
    }
}

// The same masks for a SEEDED call (sc_hip_tesla_sign_seeded): the n words of an attempt are blocks pos/4 .. pos/4 + n/4 - 1 of the message's own AES-256
// CTR-DRBG stream (csrc/drbg_dev.h; prng_32 order: the words of a block come out as w1, w0, w3, w2), produced where they are consumed - no tape exists.
// 32 lanes per slot: every lane runs the key schedule of its message's seed (a quarter of the cost of its n/128 blocks), then block j, j + 32, ...
__global__ __launch_bounds__(256) void ts_y_seeded_kernel(const uint32_t *active, uint32_t nslots, uint32_t S, uint32_t *need, uint32_t *sbase,
                                                          const uint8_t *__restrict__ seeds, size_t seed_stride, uint32_t seed_len, int32_t *y, TsP P, const uint32_t *rs, sc::DrbgConst C, St st)
{
    __shared__ sc::AesLdsWide L;
    L.fill(C);
    __syncthreads();
    if (rs) { nslots = rs[TRS_NSLOTS]; S = rs[TRS_S]; }
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x, slot = t >> 5, j = t & 31;
    if (slot >= nslots) return;
    int4 *dst = reinterpret_cast<int4 *>(y + (size_t)slot * P.n);
    const uint32_t nblk = (uint32_t)P.n / 4;
    uint32_t item, pos;
    const uint32_t nd = ts_slot_state(active, slot, S, st, nullptr, 0xfffffff0u, P.n, item, pos);      // (a generator does not run out: ts_sign_core's tape_stride for a seeded call)
    if (j == 0) { need[slot] = nd; sbase[slot] = pos; }
    if (nd != 1) { for (uint32_t i = j; i < nblk; i += 32) dst[i] = make_int4(0, 0, 0, 0); return; }
    const uint32_t msk = (1u << (P.b_bits + 1)) - 1u;
    sc::Drbg d;
    d.init(seeds + (size_t)item * seed_stride, seed_len, C, L);
    d.counter += pos / 4 + j;
    for (uint32_t i = j; i < nblk; i += 32) {
        uint32_t w[4];
        d.block(w, L);
        d.counter += 31;
        dst[i] = make_int4((int32_t)(w[1] & msk) - P.b, (int32_t)(w[0] & msk) - P.b, (int32_t)(w[3] & msk) - P.b, (int32_t)(w[2] & msk) - P.b);
    }
}

struct ZeroSix { void *p[6]; size_t n16; };
__global__ void ts_zero6_kernel(ZeroSix z)
{
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (int a = 0; a < 6; a++) {
        uint4 *q = reinterpret_cast<uint4 *>(z.p[a]);
        if (!q) continue;
        for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < z.n16; i += stride) q[i] = make_uint4(0, 0, 0, 0);
    }
}

__device__ __forceinline__ int32_t centre_q(int32_t x, int32_t q)   // ntt center_32 on a canonical value
{ return x > ((q - 1) >> 1) ? x - q : x; }

// poly_rounding (random_oracle.c:28-40) on both polynomials; cen != 0: the inputs are canonical and are centred first (signing
// hashes centre(v), ring_tesla.c:843-849; verification hashes normalised w).  nz[aidx] = 1 if any rounded value is non-zero.
__global__ __launch_bounds__(256) void ts_round_kernel(uint32_t nact, const int32_t *v1, const int32_t *v2, int32_t *r1, int32_t *r2, uint32_t *nz, int cen, TsP P,
                                                       const uint32_t *rs = nullptr)
{
    if (rs) nact = rs[TRS_NSLOTS];
    const int lane = threadIdx.x & 63;
    const uint32_t aidx = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (aidx >= nact) return;
    // r1, r2 are written only where something is non-zero (ts_oracle_kernel reads them only then, and with the reference's rounding - the remainder is
    // taken modulo q, not 2^d - that is the rare case): the kernel is bandwidth-bound and the 8n bytes of zeros per slot were half of what it moved
    uint32_t any = 0;
    int32_t ra[8], rb[8];                                            // n = 512 (sc_hip_tesla_create)
#pragma unroll
    for (int t = 0; t < 8; t++) {
        const int i = lane + 64 * t;
        ra[t] = rb[t] = 0;
        if (i < P.n) {
            const size_t k = (size_t)aidx * P.n + i;
            const int32_t a = cen ? centre_q(v1[k], P.q) : v1[k], b = cen ? centre_q(v2[k], P.q) : v2[k];
            // x % q is x itself for |x| < q - what both callers hand in (centred or canonical values): the division only for anything else
            const int32_t ma = (a > -P.q && a < P.q) ? a : a % P.q, mb = (b > -P.q && b < P.q) ? b : b % P.q;
            ra[t] = (a - ma) >> P.d; rb[t] = (b - mb) >> P.d;
            any |= (uint32_t)(ra[t] | rb[t]);
        }
    }
    for (int o = 32; o > 0; o >>= 1) any |= (uint32_t)__shfl_xor((int)any, o);
    if (any) {
#pragma unroll
        for (int t = 0; t < 8; t++) {
            const int i = lane + 64 * t;
            if (i < P.n) { r1[(size_t)aidx * P.n + i] = ra[t]; r2[(size_t)aidx * P.n + i] = rb[t]; }
        }
    }
    if (lane == 0) nz[aidx] = any != 0;
}

// once per handle: the SHA3-512 sponge after `blocks` all-zero 72-byte blocks
__global__ void ts_zero_prefix_kernel(uint32_t blocks, uint64_t *state)
{
    __shared__ uint64_t st[25 * HT];
    Sponge<HT> sp; sp.init(st, threadIdx.x, 72);
    if (threadIdx.x != 0) return;
    for (uint32_t b = 0; b < blocks; b++) sp.permute();               // absorbing zeros leaves the state unchanged before each permutation
    sp.save(state);
}

__device__ __forceinline__ uint32_t rotl(uint32_t x, int k) { return (x << k) | (x >> (32 - k)); }

// oracle (random_oracle.c:42-66) then f_function (:162-176): md = SHA3-512(BE32(r1) | BE32(r2) | m); Salsa20 stream keyed with
// md[0..32), nonce 1..8, 64-bit block counter, 800 bytes, each word stored byte-swapped (salsa20.c:93); omega distinct positions
// from big-endian byte pairs.  mode 0 (sign): writes md and c.  mode 1 (verify): c comes from the signature's digest (sigmd) and
// the recomputed md is compared with it.
__global__ __launch_bounds__(HT) void ts_oracle_kernel(const uint32_t *active, uint32_t nact, uint32_t S, const uint32_t *need, const uint8_t *msgs, size_t msg_stride, const uint32_t *mlens,
                                                       uint32_t fixed_len, const int32_t *r1, const int32_t *r2, const uint32_t *nz, const uint64_t *zstate, uint32_t zblocks,
                                                       uint8_t *md_out, int mode, uint32_t *verdict, const uint8_t *sigs, size_t sig_stride, uint32_t md_off, TsP P,
                                                       const uint32_t *rs = nullptr)
{
    __shared__ uint64_t st[25 * HT];
    if (rs) { nact = rs[TRS_NSLOTS]; S = rs[TRS_S]; }
    const uint32_t aidx = blockIdx.x * HT + threadIdx.x;
    Sponge<HT> sp; sp.init(st, threadIdx.x, 72);
    if (aidx >= nact) return;
    if (need && need[aidx] == 2) return;
    const uint32_t item = active ? active[aidx / S] : aidx / S;        // aidx counts slots here
    const uint32_t mlen = mlens ? mlens[item] : fixed_len;
    const int n = P.n;
    if (nz[aidx] == 0) {
        sp.load(st, threadIdx.x, 72, zstate);
        for (uint32_t k = zblocks * 9; k < (uint32_t)n; k++) sp.absorb_u64(0);        // 8n zero bytes in all, zblocks*72 already in the state
    } else {
        for (int k = 0; k < 2; k++) {
            const int32_t *r = (k ? r2 : r1) + (size_t)aidx * n;
            for (int i = 0; i < n; i += 2) {
                const uint32_t a = (uint32_t)r[i], b = (uint32_t)r[i + 1];
                sp.absorb_u64((uint64_t)__builtin_bswap32(a) | ((uint64_t)__builtin_bswap32(b) << 32));
            }
        }
    }
    sp.absorb(msgs + (size_t)item * msg_stride, mlen);
    sp.finalize(0x06);
    uint8_t md[64];
    for (int b = 0; b < 64; b++) md[b] = sp.squeeze_byte();
    if (mode == 0) {
        uint4 *o = reinterpret_cast<uint4 *>(md_out + (size_t)aidx * 64);
        const uint4 *mi = reinterpret_cast<const uint4 *>(md);
        for (int k = 0; k < 4; k++) o[k] = mi[k];
    } else {
        const uint8_t *sm = sigs + (size_t)item * sig_stride + md_off;
        uint32_t neq = 0;
        for (int b = 0; b < 64; b++) neq |= (uint32_t)(md[b] ^ sm[b]);
        if (neq) verdict[item] = 1;
    }
}

// f_function alone: c from a 64-byte digest (md rows for signing, the signature's digest for verification)
__device__ __forceinline__ void ts_f_body(uint32_t block, const uint32_t *active, uint32_t nact, const uint32_t *need, const uint8_t *md_rows, size_t md_stride, uint32_t md_off, int by_item,
                                          int32_t *cidx, const TsP &P, const uint32_t *rs)
{
    __shared__ uint32_t used[16 * HT];                                 // n <= 512 flags per lane, bit-packed
    if (rs) nact = rs[TRS_NSLOTS];
    const uint32_t aidx = block * HT + threadIdx.x;
    if (aidx >= nact) return;
    if (need && need[aidx] == 2) return;
    const uint32_t item = active ? active[aidx] : aidx;
    const uint8_t *md = md_rows + (size_t)(by_item ? item : aidx) * md_stride + md_off;
    uint32_t key[8];
    for (int w = 0; w < 8; w++) key[w] = (uint32_t)md[4 * w] | ((uint32_t)md[4 * w + 1] << 8) | ((uint32_t)md[4 * w + 2] << 16) | ((uint32_t)md[4 * w + 3] << 24);
    uint32_t *fl = used + threadIdx.x;
    for (int w = 0; w < 16; w++) fl[w * HT] = 0;
    int cnt = 0;
    for (uint32_t blk = 0; blk < 13 && cnt < P.omega; blk++) {           // 800 bytes = 12.5 blocks
        const uint32_t in[16] = { 0x61707865u, key[0], key[1], key[2], key[3], 0x3320646eu, 0x04030201u, 0x08070605u, blk, 0u, 0x79622d32u, key[4], key[5], key[6], key[7], 0x6b206574u };
        uint32_t x[16];
#pragma unroll
        for (int i = 0; i < 16; i++) x[i] = in[i];
#pragma unroll 1
        for (int rnd = 0; rnd < 10; rnd++) {
            x[ 4] ^= rotl(x[ 0] + x[12],  7); x[ 8] ^= rotl(x[ 4] + x[ 0],  9); x[12] ^= rotl(x[ 8] + x[ 4], 13); x[ 0] ^= rotl(x[12] + x[ 8], 18);
            x[ 9] ^= rotl(x[ 5] + x[ 1],  7); x[13] ^= rotl(x[ 9] + x[ 5],  9); x[ 1] ^= rotl(x[13] + x[ 9], 13); x[ 5] ^= rotl(x[ 1] + x[13], 18);
            x[14] ^= rotl(x[10] + x[ 6],  7); x[ 2] ^= rotl(x[14] + x[10],  9); x[ 6] ^= rotl(x[ 2] + x[14], 13); x[10] ^= rotl(x[ 6] + x[ 2], 18);
            x[ 3] ^= rotl(x[15] + x[11],  7); x[ 7] ^= rotl(x[ 3] + x[15],  9); x[11] ^= rotl(x[ 7] + x[ 3], 13); x[15] ^= rotl(x[11] + x[ 7], 18);
            x[ 1] ^= rotl(x[ 0] + x[ 3],  7); x[ 2] ^= rotl(x[ 1] + x[ 0],  9); x[ 3] ^= rotl(x[ 2] + x[ 1], 13); x[ 0] ^= rotl(x[ 3] + x[ 2], 18);
            x[ 6] ^= rotl(x[ 5] + x[ 4],  7); x[ 7] ^= rotl(x[ 6] + x[ 5],  9); x[ 4] ^= rotl(x[ 7] + x[ 6], 13); x[ 5] ^= rotl(x[ 4] + x[ 7], 18);
            x[11] ^= rotl(x[10] + x[ 9],  7); x[ 8] ^= rotl(x[11] + x[10],  9); x[ 9] ^= rotl(x[ 8] + x[11], 13); x[10] ^= rotl(x[ 9] + x[ 8], 18);
            x[12] ^= rotl(x[15] + x[14],  7); x[13] ^= rotl(x[12] + x[15],  9); x[14] ^= rotl(x[13] + x[12], 13); x[15] ^= rotl(x[14] + x[13], 18);
        }
        const int words = (blk == 12) ? 8 : 16;                           // the last block contributes 32 bytes
#pragma unroll
        for (int i = 0; i < 16; i++) {
            if (i < words && cnt < P.omega) {
                const uint32_t v = x[i] + in[i];                          // stream bytes of this word, in order: v>>24, v>>16, v>>8, v
                for (int h = 0; h < 2 && cnt < P.omega; h++) {
                    const int pos = (int)((h ? v : (v >> 16)) & 0xFFFFu) & (P.n - 1);
                    const uint32_t bit = 1u << (pos & 31), wv = fl[(pos >> 5) * HT];
                    if (!(wv & bit)) { fl[(pos >> 5) * HT] = wv | bit; cidx[(size_t)aidx * P.omega + cnt] = pos; cnt++; }
                }
            }
        }
    }
}
__global__ __launch_bounds__(HT) void ts_f_kernel(const uint32_t *active, uint32_t nact, const uint32_t *need, const uint8_t *md_rows, size_t md_stride, uint32_t md_off, int by_item,
                                                  int32_t *cidx, TsP P, const uint32_t *rs = nullptr)
{ ts_f_body(blockIdx.x, active, nact, need, md_rows, md_stride, md_off, by_item, cidx, P, rs); }

__device__ __forceinline__ int32_t wave_or(int32_t v) { for (int o = 32; o > 0; o >>= 1) v |= __shfl_xor(v, o); return v; }

constexpr int MAXW = 32;                   // omega of a parameter set (sc_hip_tesla_create checks it)

__device__ __forceinline__ int32_t norm_q(int32_t x, int32_t q) { int32_t r = x % q; return r < 0 ? r + q : r; }     // normalize_32
// normalize_32 of a value known to lie in (-q, 2q): no division (x % q by a run-time q is some forty instructions)
__device__ __forceinline__ int32_t norm_near(int32_t x, int32_t q) { x += (x >> 31) & q; return x >= q ? x - q : x; }

// test_w (ring_tesla.c:542-578) on a normalised value
__device__ __forceinline__ bool w_bad(int32_t val, const TsP &P)
{
    const int32_t d_sub = 1 << (P.d - 1);
    int32_t left = (val & ((1 << P.d) - 1)) - d_sub + 1;
    if (left < 0) left = -left;
    return left > d_sub - P.bound;
}

// The sparse products of the deciding kernel through a SHIFT TABLE (round 4): for a polynomial t, T[m] = t[m] (m < n), T[m] = -t[m - n] (n <= m < 2n), so that
// the term of mul_32_sparse (ntt.c:381-422) for position p at coefficient j - +t[j + n - p] when j < p, -t[j - p] otherwise - is T[n + j - p]: one LDS read
// at an address that is the same for the lane's n / 64 coefficients up to a constant, and one addition - no comparison, no select, no wrap-around mask
// (the plain form - a comparison, a select and a wrap-around mask per term, as round 3's kernels had it - spends seven vector instructions per term).  acc[c] = the sum over the omega positions for j = lane + 64 c.
// my_pos: lane k holds position k (one vector load per slot, omega <= 64); the loop reads them with v_readlane - no memory access inside the chain of
// additions.  (Measured, per 32768 slots: this form 92 us, positions fetched by scalar loads inside the loop 93 us, positions in a register array behind an
// unrolled guarded loop over the 32 a parameter set may have - the plain form's way - 120 us.)
constexpr int TS_MAXC = 8;                                             // n <= 512
template <typename T>
__device__ __forceinline__ void sparse_shift(int32_t (&acc)[TS_MAXC], const T *tab, int my_pos, int omega, int n, int lane)
{
    const int C = n / 64;
#pragma unroll
    for (int c = 0; c < TS_MAXC; c++) acc[c] = 0;
    for (int k = 0; k < omega; k++) {                                  // wave-uniform
        const T *row = tab + (n + lane - __builtin_amdgcn_readlane(my_pos, k));
#pragma unroll
        for (int c = 0; c < TS_MAXC; c++) if (c < C) acc[c] += (int32_t)row[64 * c];
    }
}

// one wavefront per slot (ring_tesla.c:852-880): w1 = v1 - e1*c, w2 = v2 - e2*c normalised and tested, z = y + s*c centred and
// tested.  v1, v2 are canonical products; the key polynomials sit in LDS as shift tables (built when the key is loaded: `key` = three tables of 2n
// entries).  acc[slot] = 1 and z[slot] are left for the selection.
// Round 4: 125 -> 69 us per 32768 slots (ring-TESLA-I signing 13.7 -> 16 M/s at batch 16384) - the shift tables (-> 92 us), then the tables copied from
// memory in 16-byte pieces instead of rebuilt per workgroup and the slot's rows of v1, v2 and y requested together before the first test (three
// dependent round trips to memory became one).  Measured and dropped: fewer, persistent workgroups walking the slots (147 us: a slot is a chain of
// dependent steps, it is the slots beside it that hide them).
__global__ __launch_bounds__(256) void ts_decide_kernel(uint32_t nslots, const uint32_t *need, const int16_t *key, const int32_t *v1, const int32_t *v2,
                                                        const int32_t *y, const int32_t *cidx, int32_t *zout, uint32_t *acc, TsP P, const uint32_t *rs)
{
    extern __shared__ __attribute__((aligned(16))) int16_t ks[];       // shift tables of s | e1 | e2: 2n entries each
    if (rs) nslots = rs[TRS_NSLOTS];
    if (blockIdx.x * 4 >= nslots) return;                              // (whole workgroup)
    const int n = P.n, C = n / 64;
    for (int i = threadIdx.x; i < 6 * n / 8; i += blockDim.x) reinterpret_cast<uint4 *>(ks)[i] = reinterpret_cast<const uint4 *>(key)[i];
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const uint32_t slot = __builtin_amdgcn_readfirstlane(blockIdx.x * 4 + (threadIdx.x >> 6));
    if (slot >= nslots) return;
    if (need[slot] == 2) { if (lane == 0) acc[slot] = 0; return; }
    const int pos = lane < P.omega ? cidx[(size_t)slot * P.omega + lane] : 0;
    int32_t r1[TS_MAXC], r2[TS_MAXC], ry[TS_MAXC];
#pragma unroll
    for (int c = 0; c < TS_MAXC; c++) {
        const size_t k = (size_t)slot * n + lane + 64 * (c < C ? c : 0);
        r1[c] = v1[k]; r2[c] = v2[k]; ry[c] = y[k];
    }
    // Every value normalised below is a centred residue (at most (q-1)/2 in magnitude) or a mask coefficient (at most b) plus omega key coefficients of
    // 16 bits: inside (-q, q) because omega * 2^15 < q / 2 and b < q / 2 (checked by sc_hip_tesla_create) - normalize_32 without the division.
    // the reference tests w1, then w2, then z and leaves at the first failure (ring_tesla.c:852-880): so does the wavefront
    int32_t sp[TS_MAXC];
    int32_t bad = 0;
    sparse_shift(sp, ks + 2 * n, pos, P.omega, n, lane);
#pragma unroll
    for (int c = 0; c < TS_MAXC; c++) {
        if (c < C) {
            const int32_t a = norm_near(centre_q(r1[c], P.q) - sp[c], P.q);
            bad |= w_bad(a, P) ? 1 : 0;
        }
    }
    bad = wave_or(bad);
    if (!bad) {
        sparse_shift(sp, ks + 4 * n, pos, P.omega, n, lane);
#pragma unroll
        for (int c = 0; c < TS_MAXC; c++) {
            if (c < C) {
                const int32_t b = norm_near(centre_q(r2[c], P.q) - sp[c], P.q);
                bad |= w_bad(b, P) ? 1 : 0;
            }
        }
        bad = wave_or(bad);
    }
    const int32_t th = P.b - P.u;
    if (!bad) {
        sparse_shift(sp, ks, pos, P.omega, n, lane);
#pragma unroll
        for (int c = 0; c < TS_MAXC; c++) {
            if (c < C) {
                const size_t k = (size_t)slot * n + lane + 64 * c;
                const int32_t z = centre_q(norm_near(sp[c] + ry[c], P.q), P.q);
                zout[k] = z;
                bad |= (z < -th || z > th) ? 1 : 0;
            }
        }
        bad = wave_or(bad);
    }
    if (lane == 0) acc[slot] = bad ? 0u : 1u;
}

// one wavefront per active message: the lowest accepted slot becomes the signature (z: q_bits-bit two's complement MSB first -
// three bytes per coefficient for q_bits = 24 - then the digest, ring_tesla.c:882-900); a slot beyond the end of the tape before
// any acceptance ends the message with status 2; otherwise the message goes to the next round S attempts further on.
constexpr int SEL_PER = 4;                  // messages per wavefront of ts_select_kernel
__global__ __launch_bounds__(256) void ts_select_kernel(const uint32_t *active, uint32_t nact, uint32_t S, const uint32_t *need, const uint32_t *acc, St s, const int32_t *zs,
                                                        const uint8_t *md, uint8_t *sigs, size_t sig_stride, uint32_t *siglens, uint32_t *trials,
                                                        uint32_t *next_active, uint32_t *next_count, TsP P, const uint32_t *rs)
{
    // A wavefront takes SEL_PER messages in turn and the workgroup files the ones that go on to the next round with ONE addition to the global count:
    // with a wavefront per message and an atomic each, the 6 K messages a first round leaves behind queued up on one address - 90 us of a 400 us round
    // (round 2: 24 us for 2.5 K, round 3: 9 us for 1 K) against ~10 us for the packing itself.
    __shared__ uint32_t go_cnt, go_base, go_items[4 * SEL_PER];
    if (rs) { nact = rs[TRS_NACT]; S = rs[TRS_S]; }
    const int lane = threadIdx.x & 63, n = P.n;
    if (threadIdx.x == 0) go_cnt = 0;
    __syncthreads();
    const int per = nact <= gridDim.x * 4u ? 1 : SEL_PER;               // a later round's few messages: one each (the grid was sized for the first round's)
    for (int turn = 0; turn < per; turn++) {
    const uint32_t aidx = (blockIdx.x * 4 + (threadIdx.x >> 6)) * (uint32_t)per + turn;
    if (aidx >= nact) break;
    const uint32_t item = active ? active[aidx] : aidx;
    uint32_t win = S, dead = S;                                      // first accepted slot / first slot past the tape
    for (uint32_t k = 0; k < S; k++) {
        const uint32_t slot = aidx * S + k;
        if (need[slot] == 2) { dead = k; break; }
        if (acc[slot]) { win = k; break; }
    }
    if (win == S) {
        if (lane == 0) {
            if (dead < S) { trials[item] += dead; s.pos[item] += dead * (uint32_t)n; s.status[item] = 2; siglens[item] = 0; }
            else { trials[item] += S; s.pos[item] += S * (uint32_t)n; go_items[atomicAdd(&go_cnt, 1u)] = item; }
        }
        continue;
    }
    const uint32_t slot = aidx * S + win;
    const int32_t *z = zs + (size_t)slot * n;
    uint8_t *out = sigs + (size_t)item * sig_stride;
    if (P.q_bits == 24 && (reinterpret_cast<uintptr_t>(out) & 3) == 0) {
        // four 24-bit fields are three big-endian words (a row that starts on a word boundary)
        for (int gq = lane; gq < n / 4; gq += 64) {
            const int4 v = *reinterpret_cast<const int4 *>(z + 4 * gq);
            const uint32_t a = (uint32_t)v.x & 0xFFFFFFu, b = (uint32_t)v.y & 0xFFFFFFu, c = (uint32_t)v.z & 0xFFFFFFu, d = (uint32_t)v.w & 0xFFFFFFu;
            uint32_t *ow = reinterpret_cast<uint32_t *>(out) + 3 * gq;
            ow[0] = __builtin_bswap32((a << 8) | (b >> 16)); ow[1] = __builtin_bswap32((b << 16) | (c >> 8)); ow[2] = __builtin_bswap32((c << 24) | d);
        }
    } else if (P.q_bits == 24) {
        for (int j = lane; j < n; j += 64) {
            const uint32_t v = (uint32_t)z[j];
            out[3 * j] = (uint8_t)(v >> 16); out[3 * j + 1] = (uint8_t)(v >> 8); out[3 * j + 2] = (uint8_t)v;
        }
    } else if (lane == 0) {
        uint64_t a = 0; int nb = 0; uint32_t w = 0;
        for (int j = 0; j < n; j++) {
            a = (a << P.q_bits) | ((uint32_t)z[j] & ((1u << P.q_bits) - 1u)); nb += P.q_bits;
            while (nb >= 8) { out[w++] = (uint8_t)(a >> (nb - 8)); nb -= 8; }
        }
        if (nb) out[w++] = (uint8_t)(a << (8 - nb));
    }
    const uint32_t zb = ((uint32_t)n * P.q_bits + 7) / 8;
    out[zb + lane] = md[(size_t)slot * 64 + lane];
    if (lane == 0) { siglens[item] = zb + 64; s.status[item] = 1; trials[item] += win + 1; s.pos[item] += (win + 1) * (uint32_t)n; }
    }
    __syncthreads();
    if (threadIdx.x == 0 && go_cnt) go_base = atomicAdd(next_count, go_cnt);
    __syncthreads();
    if (threadIdx.x < go_cnt) next_active[go_base + threadIdx.x] = go_items[threadIdx.x];
}

// ---- verify ---------------------------------------------------------------------------------------------------------------

// z from the signature (ring_tesla.c:984-996) + test_rejection; one wavefront per signature
__device__ __forceinline__ void ts_unpack_body(uint32_t block, uint32_t batch, const uint8_t *sigs, size_t sig_stride, const uint32_t *siglens, int32_t *z, uint32_t *verdict, const TsP &P)
{
    const int lane = threadIdx.x & 63, n = P.n;
    const uint32_t item = block * 4 + (threadIdx.x >> 6);
    if (item >= batch) return;
    const uint32_t zb = ((uint32_t)n * P.q_bits + 7) / 8;
    if (siglens[item] < zb + 64) { for (int j = lane; j < n; j += 64) z[(size_t)item * n + j] = 0; if (lane == 0) verdict[item] = 1; return; }
    const uint8_t *s = sigs + (size_t)item * sig_stride;
    const int32_t th = P.b - P.u;
    int32_t bad = 0;
    for (int j = lane; j < n; j += 64) {
        uint32_t val = 0;
        if (P.q_bits == 24) val = ((uint32_t)s[3 * j] << 16) | ((uint32_t)s[3 * j + 1] << 8) | s[3 * j + 2];
        else for (int b = 0; b < P.q_bits; b++) { const uint32_t bp = (uint32_t)j * P.q_bits + b; val = (val << 1) | ((s[bp >> 3] >> (7 - (bp & 7))) & 1u); }
        const int32_t zz = (int32_t)(val << (32 - P.q_bits)) >> (32 - P.q_bits);
        z[(size_t)item * n + j] = zz;
        bad |= (zz < -th || zz > th) ? 1 : 0;
    }
    bad = wave_or(bad);
    if (bad && lane == 0) verdict[item] = 1;
}
// The verifier's two readers of the signature as ONE launch: the first workgroups (their first HT lanes, a lane per signature) turn the digest into the
// challenge's positions (F: twelve Salsa20 blocks in a row), the rest unpack z under them.
__global__ __launch_bounds__(256) void ts_unpack_f_kernel(uint32_t batch, const uint8_t *sigs, size_t sig_stride, const uint32_t *siglens, int32_t *z, uint32_t *verdict, TsP P,
                                                         uint32_t md_off, int32_t *cidx)
{
    const uint32_t nbf = (batch + HT - 1) / HT;
    if (blockIdx.x >= nbf) ts_unpack_body(blockIdx.x - nbf, batch, sigs, sig_stride, siglens, z, verdict, P);
    else if (threadIdx.x < HT) ts_f_body(blockIdx.x, nullptr, batch, nullptr, sigs, sig_stride, md_off, 1, cidx, P, nullptr);
}

// w = normalize(prod - t*c)  (ring_tesla.c:1014-1026); one wavefront per signature, in place on the products.  The two public polynomials sit in LDS as
// SHIFT TABLES (sparse_shift above; built when the key is loaded, copied in 16-byte pieces): a term of a sparse product is one LDS read at an address the
// lane's eight coefficients share up to a constant, and one addition.  Round 3 had the plain polynomials in LDS (they were fetched from global memory
// before that: 173 us of a 330 us call) under persistent workgroups, seven vector instructions per term: 105 us per 16384 signatures, 42 % of a
// verification call; round 4: a wavefront per signature, its two rows requested up front: 105 -> 33 us (ring-TESLA-I verification 66 -> 88-91 M/s at batch 16384).
__global__ __launch_bounds__(256) void ts_vw_kernel(uint32_t batch, int32_t *w1, int32_t *w2, const int32_t *ttab, const int32_t *cidx, TsP P, sc::ModK mk)
{
    extern __shared__ __attribute__((aligned(16))) int32_t tl[];      // tables of t1 | t2: 2n entries each
    const int lane = threadIdx.x & 63, n = P.n, C = n / 64;
    for (int i = threadIdx.x; i < 4 * n / 4; i += blockDim.x) reinterpret_cast<uint4 *>(tl)[i] = reinterpret_cast<const uint4 *>(ttab)[i];
    __syncthreads();
    const uint32_t item = __builtin_amdgcn_readfirstlane(blockIdx.x * 4 + (threadIdx.x >> 6));
    if (item >= batch) return;
    const int pos = lane < P.omega ? cidx[(size_t)item * P.omega + lane] : 0;
    int32_t r1[TS_MAXC], r2[TS_MAXC], sp[TS_MAXC];
#pragma unroll
    for (int c = 0; c < TS_MAXC; c++) {
        const size_t k = (size_t)item * n + lane + 64 * (c < C ? c : 0);
        r1[c] = w1[k]; r2[c] = w2[k];
    }
    // any int32 -> [0, q) by a Barrett step (modarith.h) instead of the division normalize_32 compiles to
    sparse_shift(sp, tl, pos, P.omega, n, lane);
#pragma unroll
    for (int c = 0; c < TS_MAXC; c++) if (c < C) w1[(size_t)item * n + lane + 64 * c] = (int32_t)sc::csub(sc::reduce_i32(r1[c] - sp[c], mk), mk.q);
    sparse_shift(sp, tl + 2 * n, pos, P.omega, n, lane);
#pragma unroll
    for (int c = 0; c < TS_MAXC; c++) if (c < C) w2[(size_t)item * n + lane + 64 * c] = (int32_t)sc::csub(sc::reduce_i32(r2[c] - sp[c], mk), mk.q);
}

}  // namespace

struct sc_hip_tesla {
    sc_hip_ctx *ctx;
    sc_hip_plan *plan;
    TsP P;
    bool have_priv, have_pub;
    int32_t *d_a1, *d_a2, *d_t1, *d_t2;
    int32_t *d_ttab = nullptr;           // t1 | t2 as the verifier's shift tables (t | -t, 2n entries each)
    int16_t *d_key;                      // s | e1 | e2, each as a shift table of 2n entries (t | -t)
    uint64_t *d_zstate; uint32_t zblocks;
    size_t cap, slot_cap;                // messages / slots (attempts evaluated side by side) the workspace holds
    St st;
    uint32_t *act[2], *cnt, *rs, *need, *sbase, *nz, *trials, *acc;
    int32_t *y, *v1, *v2, *r1, *r2, *cidx, *zs;
    uint8_t *md;
    uint32_t *h_pin = nullptr;           // pinned host word the call's look at the device's count lands in (a copy to pageable memory blocks the host before the closing work is queued)
};

static void ts_free_ws(sc_hip_tesla *t)
{
    void *ptrs[] = { t->st.pos, t->st.status, t->act[0], t->act[1], t->cnt, t->rs, t->need, t->sbase, t->nz, t->trials, t->acc, t->y, t->v1, t->v2, t->r1, t->r2, t->cidx, t->zs, t->md };
    for (void *p : ptrs) if (p) (void)sc::wipe_free(p);
    t->st.pos = t->st.status = nullptr; t->act[0] = t->act[1] = t->cnt = t->rs = t->need = t->sbase = t->nz = t->trials = t->acc = nullptr;
    t->y = t->v1 = t->v2 = t->r1 = t->r2 = t->cidx = t->zs = nullptr; t->md = nullptr; t->cap = t->slot_cap = 0;
}

// per-message state for `batch` messages, per-attempt arrays for `slots` attempts (slots >= batch)
static int ts_ensure_ws(sc_hip_tesla *t, size_t batch, size_t slots)
{
    if (batch <= t->cap && slots <= t->slot_cap) return SC_HIP_SUCCESS;
    if (batch < t->cap) batch = t->cap;
    if (slots < t->slot_cap) slots = t->slot_cap;
    ts_free_ws(t);
    const size_t n = t->P.n;
#define WS(ptr, bytes) HIPCHK(hipMalloc((void **)&(ptr), (bytes)))
    WS(t->st.pos, batch * 4); WS(t->st.status, batch * 4); WS(t->act[0], batch * 4); WS(t->act[1], batch * 4); WS(t->cnt, 64); WS(t->rs, 64); WS(t->trials, batch * 4);
    WS(t->need, slots * 4); WS(t->sbase, slots * 4); WS(t->nz, slots * 4); WS(t->acc, slots * 4);
    WS(t->y, slots * n * 4); WS(t->v1, slots * n * 4); WS(t->v2, slots * n * 4); WS(t->r1, slots * n * 4); WS(t->r2, slots * n * 4); WS(t->zs, slots * n * 4);
    WS(t->cidx, slots * (size_t)t->P.omega * 4); WS(t->md, slots * 64);
#undef WS
    t->cap = batch; t->slot_cap = slots;
    return SC_HIP_SUCCESS;
}

extern "C" int sc_hip_tesla_create(sc_hip_ctx_t *ctx, const sc_hip_tesla_params_t *params, sc_hip_tesla_t **out)
{
    if (!ctx || !params || !out) return fail("tesla_create: null argument");
    static_assert(sizeof(TsP) == sizeof(sc_hip_tesla_params_t), "parameter struct layout");
    TsP P; memcpy(&P, params, sizeof P);
    if (P.n != 512 || (1 << P.n_bits) != P.n) return fail("tesla_create: n must be 512");
    if (P.omega > MAXW || (long long)P.omega * 32768 >= P.q / 2 || P.b >= P.q / 2) return fail("tesla_create: omega above 32, or omega * 2^15 or b not below q / 2 (the decide kernel's division-free normalisation)");
    if (P.omega < 1 || P.omega > 64 || P.d < 2 || P.d > 30 || P.q_bits < 8 || P.q_bits > 30 || P.b_bits < 1 || P.b_bits > 30 || P.q < 3) return fail("tesla_create: parameters out of range");
    sc_hip_plan *plan = nullptr;
    if (sc_hip_plan_create(ctx, P.q, P.n, &plan) != SC_HIP_SUCCESS) return SC_HIP_FAILURE;
    HIPCHK(hipSetDevice(ctx->device));
    sc_hip_tesla *t = new sc_hip_tesla;
    memset(t, 0, sizeof *t);
    t->ctx = ctx; t->plan = plan; t->P = P;
    t->zblocks = (uint32_t)(8 * P.n / 72);
    if (hipMalloc((void **)&t->d_zstate, 25 * 8) != hipSuccess) { sc_hip_tesla_destroy(t); return fail("tesla_create: device allocation failed"); }
    hipLaunchKernelGGL(ts_zero_prefix_kernel, dim3(1), dim3(HT), 0, nullptr, t->zblocks, t->d_zstate);
    if (hipDeviceSynchronize() != hipSuccess) { sc_hip_tesla_destroy(t); return fail("tesla_create: kernel failed"); }
    *out = t;
    return SC_HIP_SUCCESS;
}

extern "C" void sc_hip_tesla_destroy(sc_hip_tesla_t *t)
{
    if (!t) return;
    ts_free_ws(t);
    if (t->h_pin) (void)hipHostFree(t->h_pin);
    void *ptrs[] = { t->d_a1, t->d_a2, t->d_t1, t->d_t2, t->d_key, t->d_zstate, t->d_ttab };
    for (void *p : ptrs) if (p) (void)sc::wipe_free(p);
    sc_hip_plan_destroy(t->plan);
    delete t;
}

static int ts_upload32(int32_t **dst, const int32_t *src, size_t n)
{
    if (!*dst) HIPCHK(hipMalloc((void **)dst, n * 4));
    HIPCHK(sc::copy_in(*dst, src, n * 4));
    return SC_HIP_SUCCESS;
}

// a1, a2: the reference's public polynomials, NTT domain (a1_0 / a2_0, ring_tesla_params.h:85-88); s, e1, e2: the private key as
// stored (SINT16 [3][n], ring_tesla.c:703-730; NULL for a verify-only handle); t1, t2: the public key (SINT32 [2][n]; NULL for sign-only)
extern "C" int sc_hip_tesla_set_key(sc_hip_tesla_t *t, const int32_t *a1, const int32_t *a2, const int16_t *s, const int16_t *e1, const int16_t *e2,
                                    const int32_t *t1, const int32_t *t2)
{
    if (!t || !a1 || !a2) return fail("tesla_set_key: null argument");
    if ((s == nullptr) != (e1 == nullptr) || (s == nullptr) != (e2 == nullptr) || (t1 == nullptr) != (t2 == nullptr)) return fail("tesla_set_key: s, e1, e2 go together; t1, t2 go together");
    HIPCHK(hipSetDevice(t->ctx->device));
    const size_t n = t->P.n;
    if (ts_upload32(&t->d_a1, a1, n) || ts_upload32(&t->d_a2, a2, n)) return SC_HIP_FAILURE;
    if (s) {
        // the three small polynomials as the deciding kernel's shift tables (sparse_shift): T[m] = t[m], T[n + m] = -t[m]
        if (!t->d_key) HIPCHK(hipMalloc((void **)&t->d_key, 6 * n * 2));
        std::vector<int16_t> tab(6 * n);
        const int16_t *src[3] = { s, e1, e2 };
        for (int p = 0; p < 3; p++)
            for (size_t m = 0; m < n; m++) { tab[2 * n * p + m] = src[p][m]; tab[2 * n * p + n + m] = (int16_t)-src[p][m]; }
        HIPCHK(sc::copy_in(t->d_key, tab.data(), 6 * n * 2));
        for (auto &v : tab) v = 0;
        t->have_priv = true;
    }
    if (t1) {
        if (ts_upload32(&t->d_t1, t1, n) || ts_upload32(&t->d_t2, t2, n)) return SC_HIP_FAILURE;
        std::vector<int32_t> tt(4 * n);
        for (size_t m = 0; m < n; m++) { tt[m] = t1[m]; tt[n + m] = -t1[m]; tt[2 * n + m] = t2[m]; tt[3 * n + m] = -t2[m]; }
        if (ts_upload32(&t->d_ttab, tt.data(), 4 * n)) return SC_HIP_FAILURE;
        t->have_pub = true;
    }
    return SC_HIP_SUCCESS;
}

extern "C" size_t sc_hip_tesla_sig_bytes(const sc_hip_tesla_t *t)
{ return t ? ((size_t)t->P.n * t->P.q_bits + 7) / 8 + 64 : 0; }

static int ts_sign_core(sc_hip_tesla_t *t, size_t batch, const uint8_t *d_msgs, size_t msg_stride, const uint32_t *d_mlens, uint32_t fixed_len,
                        const uint32_t *d_tapes, size_t tape_stride, const uint32_t *d_tape_lens,
                        uint8_t *d_sigs, size_t sig_stride, uint32_t *d_siglens, uint32_t *d_trials, uint32_t *d_status, void *stream,
                        const uint8_t *d_seeds, size_t seed_stride, size_t seed_len)
{
    if (!t) return fail("tesla_sign: null handle");
    if (!t->have_priv) return fail("tesla_sign: no private key loaded");
    if (batch == 0) return SC_HIP_SUCCESS;
    if (batch > 0x7fffffffu / 2048) return fail("tesla_sign: batch too large");
    const bool seeded = d_seeds != nullptr;
    if (seeded) {
        if (seed_len == 0 || seed_len > seed_stride || seed_len > 0xffffffffu) return fail("tesla_sign_seeded: bad seed length / stride");
        d_tape_lens = nullptr; tape_stride = 0xfffffff0u;                 // a generator does not run out (the 32-bit word position does, after 2^23 attempts)
    }
    if (!d_msgs || (!d_tapes && !seeded) || !d_sigs || !d_siglens) return fail("tesla_sign: null buffer");
    if (sig_stride < sc_hip_tesla_sig_bytes(t)) return fail("tesla_sign: sig_stride smaller than sc_hip_tesla_sig_bytes()");
    if (tape_stride > 0xffffffffu) return fail("tesla_sign: tape too long");
    HIPCHK(hipSetDevice(t->ctx->device));
    // Attempts evaluated side by side per round: two per message while that is at most 16 K slots, one per message beyond.  Measured (tools/bench_seeded.py,
    // ms per call tape-resident / seeded, one against two attempts per message): batch 4096 0.59 / 0.67 against 0.41 / 0.50; 8192 0.62 / 0.76 against 0.59 / 0.72;
    // 16384 0.89 / 1.20 against 0.89 / 1.27; 65536 2.47 / 3.13 against 3.00 / 4.24 - a round's kernels are throughput-bound from ~16 K slots up, and every
    // speculative attempt costs its 2 KB of mask (in a seeded call: of cipher output) whether it is needed or not.
    size_t slot_target = batch * 2 < 16384 ? batch * 2 : (batch > 16384 ? batch : 16384);
    if (slot_target < 4096) slot_target = 4096;
    if (ts_ensure_ws(t, batch, slot_target)) return SC_HIP_FAILURE;
    hipStream_t s = (hipStream_t)stream;
    const TsP &P = t->P;
    const int n = P.n;
    St st = t->st;
    HIPCHK(hipMemsetAsync(st.pos, 0, batch * 4, s)); HIPCHK(hipMemsetAsync(st.status, 0, batch * 4, s));
    uint32_t *trials = d_trials ? d_trials : t->trials;
    HIPCHK(hipMemsetAsync(trials, 0, batch * 4, s));
    // Rounds are queued as the schedule predicts them (the slot rule run forward with the scheme's rejection probability, 0.68 per attempt, until fewer than four
    // messages are expected to be left: three rounds at 16384 messages), then the count is looked at ONCE and single rounds follow while messages are left.
    const uint32_t cap = (uint32_t)t->slot_cap, B = (uint32_t)batch;
    uint32_t todo = 0;
    { double left = (double)batch; while (left >= 4.0 && todo < 64) {     /* (a queued round that finds nobody left still costs two transforms over the slot budget) */ double S = cap / (left < 1.0 ? 1.0 : left); if (S > 32) S = 32; if (S < 1) S = 1; left *= pow(0.68, floor(S)); todo++; } }
    if (!todo) todo = 1;
    uint32_t round = 0, left = B;
    bool planned = true;                                                  // the queued rounds take their sizes from the device; a round behind the look has the host's
    while (left > 0) {
        for (uint32_t i = 0; i < todo; i++, round++) {
            if (round >= 9999) { left = 0; break; }                       // the reference's own cap (ring_tesla.c:814); leftovers keep status 0
            const uint32_t *active = round ? t->act[(round - 1) & 1] : nullptr;
            uint32_t *next = t->act[round & 1];
            const uint32_t *rs = planned ? t->rs : nullptr;
            uint32_t nact = B, S = 1, nslots = cap;                       // planned: upper bounds (they size the grids)
            if (planned) hipLaunchKernelGGL(ts_rs_kernel, dim3(1), dim3(1), 0, s, t->rs, t->cnt, round ? 0u : B, cap);
            else {
                nact = left; S = cap / nact; if (S > 32) S = 32; if (S < 1) S = 1; nslots = nact * S;
                HIPCHK(hipMemsetAsync(t->cnt, 0, 4, s));
            }
            const size_t ne = (size_t)nslots * n;
            size_t yb = (ne / 4 + 255) / 256;                             // a lane per four coefficients; persistent beyond 16 workgroups per CU
            if (yb > (size_t)t->ctx->cus * 16) yb = (size_t)t->ctx->cus * 16;
            if (seeded) hipLaunchKernelGGL(ts_y_seeded_kernel, dim3((unsigned)(((size_t)nslots * 32 + 255) / 256)), dim3(256), 0, s, active, nslots, S, t->need, t->sbase,
                                           d_seeds, seed_stride, (uint32_t)seed_len, t->y, P, rs, sc::drbg_const(), st);
            else hipLaunchKernelGGL(ts_y_kernel, dim3((unsigned)yb), dim3(256), 0, s, active, nslots, S, t->need, t->sbase, d_tapes, tape_stride, t->y, P, rs, st, d_tape_lens);
            HIPCHK(hipGetLastError());
            // (a planned round's products run over the whole slot budget: rows past the round's slots hold what an earlier round left there and are read by nobody)
            if (sc::polymul_ntt_bcast2(t->plan, nslots, t->y, t->d_a1, t->d_a2, t->v1, t->v2, s)) return SC_HIP_FAILURE;      // v1 = a1 y, v2 = a2 y: one forward transform of y for both
            hipLaunchKernelGGL(ts_round_kernel, dim3((nslots + 3) / 4), dim3(256), 0, s, nslots, t->v1, t->v2, t->r1, t->r2, t->nz, 1, P, rs);
            hipLaunchKernelGGL(ts_oracle_kernel, dim3((nslots + HT - 1) / HT), dim3(HT), 0, s, active, nslots, S, t->need, d_msgs, msg_stride, d_mlens, fixed_len, t->r1, t->r2, t->nz,
                               t->d_zstate, t->zblocks, t->md, 0, (uint32_t *)nullptr, (const uint8_t *)nullptr, (size_t)0, 0u, P, rs);
            hipLaunchKernelGGL(ts_f_kernel, dim3((nslots + HT - 1) / HT), dim3(HT), 0, s, (const uint32_t *)nullptr, nslots, t->need, t->md, (size_t)64, 0u, 0, t->cidx, P, rs);
            hipLaunchKernelGGL(ts_decide_kernel, dim3((nslots + 3) / 4), dim3(256), sizeof(int16_t) * 6 * (size_t)n, s, nslots, t->need, t->d_key, t->v1, t->v2, t->y, t->cidx, t->zs, t->acc, P, rs);
            hipLaunchKernelGGL(ts_select_kernel, dim3((nact + 4 * SEL_PER - 1) / (4 * SEL_PER)), dim3(256), 0, s, active, nact, S, t->need, t->acc, st, t->zs, t->md, d_sigs, sig_stride, d_siglens, trials, next, t->cnt, P, rs);
            HIPCHK(hipGetLastError());
        }
        if (!left) {                                                          // the reference's round cap: no look follows, the closing work does
            if (d_status) HIPCHK(hipMemcpyAsync(d_status, st.status, batch * 4, hipMemcpyDeviceToDevice, s));
            const size_t used = t->slot_cap * (size_t)n * 4;
            ZeroSix z6{ { t->y, t->v1, t->v2, t->r1, t->r2, t->zs }, used / 16 };
            size_t zb = (used / 16 + 255) / 256;
            if (zb > (size_t)t->ctx->cus * 8) zb = (size_t)t->ctx->cus * 8;
            hipLaunchKernelGGL(ts_zero6_kernel, dim3((unsigned)(zb ? zb : 1)), dim3(256), 0, s, z6);
            HIPCHK(hipGetLastError());
            break;
        }
        // The look at the count - and in front of it, not behind it, the call's closing work: the call usually ends here, so the zeroing of the masks y, their
        // products and the rejected candidates (one launch over the six arrays; every round's slots are cut from the handle's slot budget, so that is how far they
        // were written) and the copy of the statuses run while the host waits for the count instead of after it has woken up.  A round that follows rewrites
        // what it reads and is zeroed again.
        if (!t->h_pin) HIPCHK(hipHostMalloc((void **)&t->h_pin, 64, hipHostMallocDefault));
        HIPCHK(hipMemcpyAsync(t->h_pin, t->cnt, 4, hipMemcpyDeviceToHost, s));
        if (d_status) HIPCHK(hipMemcpyAsync(d_status, st.status, batch * 4, hipMemcpyDeviceToDevice, s));
        {
            const size_t used = t->slot_cap * (size_t)n * 4;
            ZeroSix z6{ { t->y, t->v1, t->v2, t->r1, t->r2, t->zs }, used / 16 };
            size_t zb = (used / 16 + 255) / 256;
            if (zb > (size_t)t->ctx->cus * 8) zb = (size_t)t->ctx->cus * 8;
            hipLaunchKernelGGL(ts_zero6_kernel, dim3((unsigned)(zb ? zb : 1)), dim3(256), 0, s, z6);
            HIPCHK(hipGetLastError());
        }
        HIPCHK(hipStreamSynchronize(s));
        left = *(volatile uint32_t *)t->h_pin;
        todo = 1; planned = false;
    }
    return SC_HIP_SUCCESS;
}

extern "C" int sc_hip_tesla_sign(sc_hip_tesla_t *t, size_t batch, const uint8_t *d_msgs, size_t msg_stride, const uint32_t *d_mlens, uint32_t fixed_len,
                                 const uint32_t *d_tapes, size_t tape_stride, const uint32_t *d_tape_lens,
                                 uint8_t *d_sigs, size_t sig_stride, uint32_t *d_siglens, uint32_t *d_trials, uint32_t *d_status, void *stream)
{
    if (!d_tapes && batch) return fail("tesla_sign: null buffer");
    return ts_sign_core(t, batch, d_msgs, msg_stride, d_mlens, fixed_len, d_tapes, tape_stride, d_tape_lens, d_sigs, sig_stride, d_siglens, d_trials, d_status, stream, nullptr, 0, 0);
}

// sc_hip_tesla_sign with the CSPRNG inside the call: message i draws its masks from the generator the reference makes of seed i (AES-256 CTR-DRBG,
// src/utils/crypto/prng.c:963-1107, ctr_drbg.c:37-197; ring_tesla.c:818-822 reads it through prng_32).  The words of an attempt are produced by the kernel
// that turns them into the mask (ts_y_seeded_kernel): nothing is expanded ahead and no tape exists.  Byte-identical to sc_hip_tesla_sign over
// sc_hip_drbg_words(seeds, ...) rows that are long enough.
extern "C" int sc_hip_tesla_sign_seeded(sc_hip_tesla_t *t, size_t batch, const uint8_t *d_msgs, size_t msg_stride, const uint32_t *d_mlens, uint32_t fixed_len,
                                        const uint8_t *d_seeds, size_t seed_stride, size_t seed_len,
                                        uint8_t *d_sigs, size_t sig_stride, uint32_t *d_siglens, uint32_t *d_trials, uint32_t *d_status, void *stream)
{
    if (!d_seeds && batch) return fail("tesla_sign_seeded: null seed buffer");
    return ts_sign_core(t, batch, d_msgs, msg_stride, d_mlens, fixed_len, nullptr, 0, nullptr, d_sigs, sig_stride, d_siglens, d_trials, d_status, stream, d_seeds, seed_stride, seed_len);
}

extern "C" int sc_hip_tesla_verify(sc_hip_tesla_t *t, size_t batch, const uint8_t *d_msgs, size_t msg_stride, const uint32_t *d_mlens, uint32_t fixed_len,
                                   const uint8_t *d_sigs, size_t sig_stride, const uint32_t *d_siglens, uint32_t *d_verdicts, void *stream)
{
    if (!t) return fail("tesla_verify: null handle");
    if (!t->have_pub) return fail("tesla_verify: no public key loaded");
    if (batch == 0) return SC_HIP_SUCCESS;
    if (batch > 0x7fffffffu / 2048) return fail("tesla_verify: batch too large");
    if (!d_msgs || !d_sigs || !d_siglens || !d_verdicts) return fail("tesla_verify: null buffer");
    if (sig_stride < sc_hip_tesla_sig_bytes(t)) return fail("tesla_verify: sig_stride smaller than sc_hip_tesla_sig_bytes()");
    HIPCHK(hipSetDevice(t->ctx->device));
    if (ts_ensure_ws(t, batch, batch)) return SC_HIP_FAILURE;
    hipStream_t s = (hipStream_t)stream;
    const TsP &P = t->P;
    const uint32_t B = (uint32_t)batch;
    const uint32_t md_off = ((uint32_t)P.n * P.q_bits + 7) / 8;
    HIPCHK(hipMemsetAsync(d_verdicts, 0, batch * 4, s));
    hipLaunchKernelGGL(ts_unpack_f_kernel, dim3((B + HT - 1) / HT + (B + 3) / 4), dim3(256), 0, s, B, d_sigs, sig_stride, d_siglens, t->y, d_verdicts, P, md_off, t->cidx);
    HIPCHK(hipGetLastError());
    if (sc::polymul_ntt_bcast2(t->plan, batch, t->y, t->d_a1, t->d_a2, t->v1, t->v2, s)) return SC_HIP_FAILURE;
    {
        hipLaunchKernelGGL(ts_vw_kernel, dim3((B + 3) / 4), dim3(256), sizeof(int32_t) * 4 * (size_t)P.n, s, B, t->v1, t->v2, t->d_ttab, t->cidx, P, t->plan->k[1]);
    }
    hipLaunchKernelGGL(ts_round_kernel, dim3((B + 3) / 4), dim3(256), 0, s, B, t->v1, t->v2, t->r1, t->r2, t->nz, 0, P);
    hipLaunchKernelGGL(ts_oracle_kernel, dim3((B + HT - 1) / HT), dim3(HT), 0, s, (const uint32_t *)nullptr, B, 1u, (const uint32_t *)nullptr, d_msgs, msg_stride, d_mlens, fixed_len, t->r1, t->r2, t->nz,
                       t->d_zstate, t->zblocks, (uint8_t *)nullptr, 1, d_verdicts, d_sigs, sig_stride, md_off, P);
    HIPCHK(hipGetLastError());
    return SC_HIP_SUCCESS;
}

// ---- host-buffer convenience (sc::sign_host / sc::verify_host in engine_internal.h) ----

extern "C" int sc_hip_tesla_sign_host(sc_hip_tesla_t *t, size_t batch, const uint8_t *h_msgs, size_t msg_stride, const uint32_t *h_mlens, uint32_t fixed_len,
                                      const uint32_t *h_tapes, size_t tape_stride, const uint32_t *h_tape_lens,
                                      uint8_t *h_sigs, size_t sig_stride, uint32_t *h_siglens, uint32_t *h_trials, uint32_t *h_status)
{
    if (!t) return fail("tesla_sign_host: null handle");
    return sc::sign_host(t->ctx->device, batch, h_msgs, msg_stride, h_mlens, h_tapes, tape_stride, h_tape_lens, h_sigs, sig_stride, h_siglens, h_trials, h_status,
        [&](const uint8_t *m, const uint32_t *ml, const uint32_t *tp, const uint32_t *tl, uint8_t *sg, uint32_t *sl, uint32_t *tr, uint32_t *st) {
            return sc_hip_tesla_sign(t, batch, m, msg_stride, ml, fixed_len, tp, tape_stride, tl, sg, sig_stride, sl, tr, st, nullptr);
        });
}

extern "C" int sc_hip_tesla_verify_host(sc_hip_tesla_t *t, size_t batch, const uint8_t *h_msgs, size_t msg_stride, const uint32_t *h_mlens, uint32_t fixed_len,
                                        const uint8_t *h_sigs, size_t sig_stride, const uint32_t *h_siglens, uint32_t *h_verdicts)
{
    if (!t) return fail("tesla_verify_host: null handle");
    return sc::verify_host(t->ctx->device, batch, h_msgs, msg_stride, h_mlens, h_sigs, sig_stride, h_siglens, h_verdicts,
        [&](const uint8_t *m, const uint32_t *ml, const uint8_t *sg, const uint32_t *sl, uint32_t *v) {
            return sc_hip_tesla_verify(t, batch, m, msg_stride, ml, fixed_len, sg, sig_stride, sl, v, nullptr);
        });
}
