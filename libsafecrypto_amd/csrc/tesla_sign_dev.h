// tesla_sign_dev.h - the device side of the batched ring-TESLA signer: every kernel of a signing round (csrc/tesla.hip ts_sign_core queues them).
// Replaces ring_tesla_sign (src/schemes/sig/ring_tesla/ring_tesla.c:752-910) an attempt per slot: the masks, poly_rounding and the oracle of
// random_oracle.c, f_function, the sparse products through shift tables with test_w / test_rejection, the selection and the packing.  Included by tesla.hip
// INSIDE its anonymous namespace, behind keccak_dev.h and drbg_dev.h (hence no includes here); a header so that the kernels' own text can also be compiled
// as host code over tests/emu/wave_emu.h and a signing call be run lane by lane under the sanitizers the GPU pool cannot offer (tests/test_emu_tesla_sign.py).

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
