// bliss.hip - batched BLISS-B sign / verify on MI355X behind the C ABI of include/sc_hip.h.
//
// Replaces, for a batch of messages under one key, bliss_b_sign / bliss_b_verify
// (src/schemes/sig/bliss_b/bliss_b.c:1266-1535, 1583-1740; raw signature coding, 64-bit CDT sampler, no blinding):
// get_vector_32 x2 (sampling.c:556-580 -> gaussian_cdf.c:536-659), fwd_ntt/mul_32_pointwise_16/inv_ntt
// (bliss_b.c:1378-1384), round_and_drop (:681-703), oracle (:552-620, SHA3-512 - the reference's permutation,
// keccak_dev.h), greedy_sc (:622-679), the +-(x,y) choice, the exp/cosh rejection (:1407-1428), signature_gen
// (:1045-1075), check_norms (:1010-1043) and the MSB-first bit packing (:1447-1463).
//
// Randomness: the reference draws everything from ONE CSPRNG context as a stream of 32-bit pool words
// (src/utils/crypto/prng.c:963-1048: prng_64 = two words, high first; prng_var(n) = a 32-bit reservoir refilled
// by prng_32; prng_bit = prng_var(1); prng_double = prng_var(27), prng_var(26)).  The engine owns no CSPRNG: each
// message comes with its "tape" of such words (the host's reference-compatible generator fills it) and the
// consumption rules are reproduced here, so a signature is the same pure function of (key, message, tape) as in
// the reference - byte for byte (tests/golden/bliss_ref.txt).  A tape that runs out is reported per message.
#include "engine_internal.h"
#include "keccak_dev.h"
#include "huffman_dev.h"

#include <algorithm>
#include <initializer_list>
#include <utility>
#include <math.h>
#include <string.h>
#include <vector>

using namespace sc;

namespace {

#include "bliss_sign_dev.h"

#include "bliss_verify_dev.h"

// ---- SC_ENTROPY_HUFFMAN_STATIC coding of the signature ---------------------------------------------------------------------------
// sc->coding_signature.type = SC_ENTROPY_HUFFMAN_STATIC (bliss_b.c:197-200): t and z go through entropy_poly_encode_32 / _16 with SIGNED_COEFF
// (src/utils/entropy/entropy.c:70-112, 147-190): per coefficient the beta = max(bits - 7, 0) low bits of |v| raw, |v| >> beta as a codeword of a static
// prefix code (entropy_huffman.c:22-57, 155-190; huffman.c:386-403), a sign bit unless v = 0; the kappa indices stay raw n_bits fields; one MSB-first bit
// stream flushed to whole bytes (packer.c:299-351).  The code tables are the reference's data (src/utils/entropy/huffman.h) and come in through
// sc_hip_bliss_set_entropy.  Coding is a pure format conversion of the raw signature, done as a pass of its own behind the signer / in front of the verifier.
// one wavefront per signed message: the raw row into LDS, every lane makes the tokens of its coefficients, an exclusive scan of their lengths places
// them, LDS atomics assemble the stream, the coded row replaces the raw one.  status 6: a coefficient outside the code's alphabet (encode_huffman's
// SC_OUT_OF_BOUNDS) or a coded signature longer than the row.
__global__ __launch_bounds__(256) void bl_huff_pack_kernel(uint8_t *sigs, size_t sig_stride, uint32_t *siglens, uint32_t *status, uint32_t count, BlissP P,
                                                           HuffT t1, HuffT t2, uint32_t capw)
{
    extern __shared__ uint32_t hlds[];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, n = P.n;
    const uint32_t item = blockIdx.x * 4 + wave;
    if (item >= count || status[item] != 1) return;
    const uint32_t rawbytes = (uint32_t)(n * (P.z1_bits + P.z2_bits) + P.kappa * P.n_bits + 7) >> 3, raww = (rawbytes + 3) / 4 + 1;
    uint32_t *raw = hlds + (size_t)wave * (raww + capw), *out = raw + raww;
    uint8_t *row = sigs + (size_t)item * sig_stride;
    for (uint32_t w = lane; w < raww; w += 64) {
        uint32_t v = 0;
        for (int b = 0; b < 4; b++) { const uint32_t i = 4 * w + b; v = (v << 8) | (i < rawbytes ? row[i] : 0u); }
        raw[w] = v;
    }
    for (uint32_t w = lane; w < capw; w += 64) out[w] = 0;
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier();
    uint32_t base = 0; bool bad = false;
    for (int part = 0; part < 2; part++) {
        const int bits = part ? P.z2_bits : P.z1_bits, beta = bits > 7 ? bits - 7 : 0;
        const HuffT &T = part ? t2 : t1;
        const uint32_t first = part ? (uint32_t)(n * P.z1_bits) : 0u;
        for (int c = 0; c < n / 64; c++) {
            const int j = lane + 64 * c;
            const uint32_t f = lds_get_bits(raw, first + (uint32_t)(j * bits), bits);
            const int32_t v = (int32_t)(f << (32 - bits)) >> (32 - bits);
            const uint32_t a = (uint32_t)(v < 0 ? -v : v), sym = a >> beta, sb = v != 0;
            uint64_t tok = 0; uint32_t len = 0;
            if (sym < T.depth) {
                const uint32_t cl = T.len[sym];
                tok = ((uint64_t)(a & ((1u << beta) - 1u)) << (cl + sb)) | ((uint64_t)T.code[sym] << sb) | (uint64_t)(v < 0);
                len = (uint32_t)beta + cl + sb;
            } else bad = true;
            const uint32_t incl = wave_incl_scan(len, lane);
            lds_put_bits(out, capw, base + incl - len, tok, len);
            base += __shfl(incl, 63);
        }
    }
    {
        const uint32_t first = (uint32_t)(n * (P.z1_bits + P.z2_bits));
        const uint32_t len = lane < P.kappa ? (uint32_t)P.n_bits : 0u;
        const uint64_t tok = lane < P.kappa ? lds_get_bits(raw, first + (uint32_t)(lane * P.n_bits), P.n_bits) : 0u;
        const uint32_t incl = wave_incl_scan(len, lane);
        lds_put_bits(out, capw, base + incl - len, tok, len);
        base += __shfl(incl, 63);
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier();
    const uint32_t nbytes = (base + 7) >> 3;
    const bool fail = __any(bad) || nbytes > sig_stride || (nbytes + 3) / 4 > capw;
    const uint32_t keep = fail ? 0u : nbytes, clear = rawbytes > keep ? rawbytes : keep;
    for (uint32_t i = lane; i < clear; i += 64) row[i] = i < keep ? (uint8_t)(out[i >> 2] >> (24 - 8 * (i & 3))) : (uint8_t)0;
    if (lane == 0) { siglens[item] = keep; if (fail) status[item] = 6; }
}

// replaces bl_unpack_kernel when the signatures are coded: a variable-length stream decodes serially, so one lane per signature (entropy_poly_decode_32 /
// _16 + the raw indices, bliss_b.c:1641-1657); decoding tables in LDS (a 10-bit lookup resolves all but the rare long codewords), t and z leave the lane
// in 16-byte pieces.  Only the first lpw lanes of a wavefront take a signature: the host spreads a batch over enough wavefronts to give every SIMD two
// (a wavefront costs the same per instruction with 8 lanes active as with 64, and its lanes diverge less).  A stream that ends early is invalid (the
// reference's pack_decode of the indices fails).
__global__ __launch_bounds__(64) void bl_huff_unpack_kernel(const uint8_t *sigs, size_t sig_stride, const uint32_t *siglens, uint32_t count, uint32_t lpw, int32_t *t, int16_t *z,
                                                            int32_t *cidx, uint32_t *verdict, BlissP P, HuffT t1, HuffT t2)
{
    __shared__ uint32_t tl[4][256];
    __shared__ uint32_t lut[2][1 << HUFF_LUT_BITS];
    for (uint32_t i = threadIdx.x; i < 256; i += 64) {
        tl[0][i] = i < t1.depth ? t1.dleft[i] : 0; tl[1][i] = i < t1.depth ? t1.dls[i] : 0;
        tl[2][i] = i < t2.depth ? t2.dleft[i] : 0; tl[3][i] = i < t2.depth ? t2.dls[i] : 0;
    }
    for (uint32_t i = threadIdx.x; i < (1u << t1.lutbits); i += 64) lut[0][i] = t1.lut[i];
    for (uint32_t i = threadIdx.x; i < (1u << t2.lutbits); i += 64) lut[1][i] = t2.lut[i];
    __syncthreads();
    const uint32_t item = blockIdx.x * lpw + threadIdx.x;
    if (threadIdx.x >= lpw || item >= count) return;
    const int n = P.n;
    const uint32_t sl = siglens[item];
    BitR r{ sigs + (size_t)item * sig_stride, (uint32_t)((sl < sig_stride ? sl : sig_stride) * 8), (uint32_t)sig_stride, 0, 0, 0, 0, false };
    int4 *tp = reinterpret_cast<int4 *>(t + (size_t)item * n);
    for (int j = 0; j < n; j += 4) {
        int4 q;
        q.x = huff_get(r, P.z1_bits, lut[0], t1.lutbits, tl[0], tl[1], t1.depth); q.y = huff_get(r, P.z1_bits, lut[0], t1.lutbits, tl[0], tl[1], t1.depth);
        q.z = huff_get(r, P.z1_bits, lut[0], t1.lutbits, tl[0], tl[1], t1.depth); q.w = huff_get(r, P.z1_bits, lut[0], t1.lutbits, tl[0], tl[1], t1.depth);
        tp[j >> 2] = q;
    }
    int4 *zp = reinterpret_cast<int4 *>(z + (size_t)item * n);
    for (int j = 0; j < n; j += 8) {
        uint32_t h[4];
        for (int e = 0; e < 4; e++) {
            const uint32_t lo = (uint32_t)huff_get(r, P.z2_bits, lut[1], t2.lutbits, tl[2], tl[3], t2.depth) & 0xFFFFu;
            const uint32_t hi = (uint32_t)huff_get(r, P.z2_bits, lut[1], t2.lutbits, tl[2], tl[3], t2.depth) & 0xFFFFu;
            h[e] = lo | (hi << 16);
        }
        zp[j >> 3] = make_int4((int)h[0], (int)h[1], (int)h[2], (int)h[3]);
    }
    for (int k = 0; k < P.kappa; k++) cidx[(size_t)item * P.kappa + k] = (int32_t)r.get(P.n_bits);
    if (r.over) verdict[item] = 1;
}

// ---- the arithmetic coder (SC_FLAG_0_ENTROPY_BAC; src/utils/entropy/bac.c) -------------------------------------------------------------------------
// t and z of a signature each become a byte-aligned block - a 2-byte big-endian length, then that many bytes of a 64-bit binary range coder's output
// (a coefficient is `bits` binary decisions, the probability of a 0 at a node of that tree is dist[prefix | 1 << k]; bac.c:162-258) - and the kappa
// indices follow raw (bliss_b.c:1453-1463).  A range coder is serial from the first bit of a signature to the last: one lane per signature, the first
// lpw lanes of a wavefront only (as for the Huffman decoder), the two distributions in LDS where they fit.
constexpr uint32_t BAC_LDS_ENTRIES = 4096 + 16;                   // z1_bits <= 12 and z2_bits <= 4: every reference parameter set
struct BacDist { const uint64_t *t, *z; };
__device__ __forceinline__ const uint64_t *bac_stage(const BacDist &d, int bits_t, int bits_z, uint64_t *lds, const uint64_t **dz)
{
    const uint32_t nt = 1u << bits_t, nz = 1u << bits_z;
    if (nt + nz > BAC_LDS_ENTRIES) { *dz = d.z; return d.t; }       // wider fields than the reference has: the tables stay where they are
    for (uint32_t i = threadIdx.x; i < nt; i += blockDim.x) lds[i] = d.t[i];
    for (uint32_t i = threadIdx.x; i < nz; i += blockDim.x) lds[nt + i] = d.z[i];
    __syncthreads();
    *dz = lds + nt;
    return lds;
}
struct BacOut {                  // the byte stream of one block, with the reference's carry propagation into bytes already written (bac.c:151-160)
    uint8_t *buf; uint32_t optr, cap; bool ovf;
    __device__ __forceinline__ void byte(uint32_t ob)
    {
        if (optr < cap) buf[optr] = (uint8_t)ob; else ovf = true;
        for (int i = (int)optr - 1; ob >= 0x100u && i >= 0; i--) { ob >>= 8; if ((uint32_t)i < cap) { ob += buf[i]; buf[i] = (uint8_t)ob; } }
        optr++;
    }
};
// one block: n fields of `bits` bits read from r; returns the bytes written at out (header included)
__device__ __forceinline__ uint32_t bac_encode_block(BitR &r, int n, int bits, const uint64_t *dist, uint8_t *out, uint32_t cap, bool *ovf)
{
    BacOut o{ out + 2, 0, cap > 2 ? cap - 2 : 0, false };
    uint64_t b = 0, l = ~0ull;
    uint32_t data = 0; int ocnt = 0;
    for (int i = 0; i < n; i++) {
        const uint32_t f = r.get(bits);
        const uint32_t iwrd = (f + (1u << (bits - 1))) & ((1u << bits) - 1u);          // offset + the signed field (bac.c:191; entropy.c:95-97)
        for (int icnt = bits; icnt--;) {
            uint64_t c = __umul64hi(l, dist[(iwrd & (0xFFFFFFFEu << icnt)) | (1u << icnt)]);
            if (0 == ((iwrd >> icnt) & 1)) l = c;
            else { b += c; l -= c; if (b < c) data++; }
            if (l) while (l < 0x8000000000000000ull) {
                data = (data << 1) | ((uint32_t)(b >> 63) & 1u); ocnt++;
                if (ocnt >= 8) { o.byte(data); ocnt = 0; data = 0; }
                b <<= 1; l <<= 1;
            }
        }
    }
    while (ocnt < 8) { data = (data << 1) ^ (uint32_t)(b >> 63); b <<= 1; ocnt++; }
    o.byte(data);
    while (b != 0) { o.byte((uint32_t)(b >> 56)); b <<= 8; }
    if (cap >= 2) { out[0] = (uint8_t)(o.optr >> 8); out[1] = (uint8_t)o.optr; }
    *ovf |= o.ovf || cap < 2 || o.optr > 0xFFFFu;
    return o.optr + 2;
}
// raw signature rows -> coded rows in tmp (row pitch tmp_stride); status 6 for a row that would not fit
__global__ __launch_bounds__(64) void bl_bac_pack_kernel(const uint8_t *sigs, size_t sig_stride, uint32_t *siglens, uint32_t *status, uint32_t count, uint32_t lpw, BlissP P, BacDist d,
                                                         uint8_t *tmp, size_t tmp_stride, uint32_t cap)
{
    __shared__ uint64_t lds[BAC_LDS_ENTRIES];
    const uint64_t *dz; const uint64_t *dt = bac_stage(d, P.z1_bits, P.z2_bits, lds, &dz);
    const uint32_t item = blockIdx.x * lpw + threadIdx.x;
    if (threadIdx.x >= lpw || item >= count || status[item] != 1) return;
    const uint32_t rawbytes = (uint32_t)(((size_t)P.n * (P.z1_bits + P.z2_bits) + (size_t)P.kappa * P.n_bits + 7) / 8);
    BitR r{ sigs + (size_t)item * sig_stride, rawbytes * 8, (uint32_t)sig_stride, 0, 0, 0, 0, false };
    uint8_t *out = tmp + (size_t)item * tmp_stride;
    bool ovf = false;
    uint32_t at = bac_encode_block(r, P.n, P.z1_bits, dt, out, cap, &ovf);
    at += bac_encode_block(r, P.n, P.z2_bits, dz, out + (at < cap ? at : cap), at < cap ? cap - at : 0, &ovf);
    uint64_t acc = 0; int nb = 0;                                      // the indices, MSB first, from a byte boundary
    for (int k = 0; k < P.kappa; k++) {
        acc = (acc << P.n_bits) | r.get(P.n_bits); nb += P.n_bits;
        while (nb >= 8) { if (at < cap) out[at] = (uint8_t)(acc >> (nb - 8)); else ovf = true; at++; nb -= 8; }
    }
    if (nb) { if (at < cap) out[at] = (uint8_t)(acc << (8 - nb)); else ovf = true; at++; }
    if (ovf) { status[item] = 6; siglens[item] = 0; } else siglens[item] = at;
}
// coded rows out of tmp into the signature rows (a wavefront per row), the rest of the raw row and the tmp row zeroed
__global__ __launch_bounds__(256) void bl_bac_move_kernel(uint8_t *sigs, size_t sig_stride, const uint32_t *siglens, const uint32_t *status, uint32_t count, uint8_t *tmp, size_t tmp_stride,
                                                          uint32_t rawbytes, uint32_t cap)
{
    const uint32_t item = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (item >= count) return;
    const uint32_t st = status[item];
    if (st != 1 && st != 6) return;                                    // (rows the coder did not touch)
    const uint32_t keep = st == 1 ? siglens[item] : 0u, clear = rawbytes > keep ? rawbytes : keep;
    uint8_t *row = sigs + (size_t)item * sig_stride, *src = tmp + (size_t)item * tmp_stride;
    for (uint32_t i = lane; i < clear; i += 64) row[i] = i < keep ? src[i] : (uint8_t)0;
    for (uint32_t i = lane; i < cap; i += 64) src[i] = 0;
}
// coded rows -> t, z, the indices (replaces bl_unpack_kernel; bac_decode_64_32 / _16, bac.c:260-334, bliss_b.c:1650-1657).  The decoder's bit fetch
// is restated with its state (ibyt, icnt, iptr against the block's stated length): past that length it shifts by a negative count - undefined in C, a
// shift by the count mod 32 on the x86 the reference runs on.  Anything read past the signature's own end makes it invalid.
__device__ __forceinline__ void bac_decode_block(BitR &r, int n, int bits, const uint64_t *dist, int32_t *o32, int16_t *o16)
{
    uint32_t length = r.get(8) << 8; length |= r.get(8);
    uint64_t v = (uint64_t)r.get(32) << 32; v |= r.get(32);
    uint64_t b = 0, l = ~0ull;
    uint32_t ibyt = 0; int icnt = 0; int32_t iptr = 8;
    const uint32_t offset = 1u << (bits - 1);
    for (int i = 0; i < n; i++) {
        uint32_t owrd = 0;
        for (int ocnt = bits; ocnt--;) {
            const uint64_t c = __umul64hi(l, dist[(owrd & (0xFFFFFFFEu << ocnt)) | (1u << ocnt)]);
            if (v - b < c) l = c;
            else { b += c; l -= c; owrd |= 1u << ocnt; }
            while (l < 0x8000000000000000ull) {
                icnt--;
                if (icnt < 0 && iptr < (int32_t)length) { ibyt = r.get(8); iptr++; icnt = 7; }
                v = (v << 1) + ((ibyt >> (icnt & 31)) & 1u);
                b <<= 1; l <<= 1;
            }
        }
        const int32_t val = (int32_t)(owrd - offset);
        if (o32) o32[i] = val; else o16[i] = (int16_t)val;
    }
}
__global__ __launch_bounds__(64) void bl_bac_unpack_kernel(const uint8_t *sigs, size_t sig_stride, const uint32_t *siglens, uint32_t count, uint32_t lpw, int32_t *t, int16_t *z,
                                                           int32_t *cidx, uint32_t *verdict, BlissP P, BacDist d)
{
    __shared__ uint64_t lds[BAC_LDS_ENTRIES];
    const uint64_t *dz; const uint64_t *dt = bac_stage(d, P.z1_bits, P.z2_bits, lds, &dz);
    const uint32_t item = blockIdx.x * lpw + threadIdx.x;
    if (threadIdx.x >= lpw || item >= count) return;
    const uint32_t sl = siglens[item];
    BitR r{ sigs + (size_t)item * sig_stride, (uint32_t)((sl < sig_stride ? sl : sig_stride) * 8), (uint32_t)sig_stride, 0, 0, 0, 0, false };
    bac_decode_block(r, P.n, P.z1_bits, dt, t + (size_t)item * P.n, nullptr);
    bac_decode_block(r, P.n, P.z2_bits, dz, nullptr, z + (size_t)item * P.n);
    for (int k = 0; k < P.kappa; k++) cidx[(size_t)item * P.kappa + k] = (int32_t)r.get(P.n_bits);
    if (r.over) verdict[item] = 1;
}

// ---- key generation (bliss_b_keygen, bliss_b.c:1124-1264) ------------------------------------------------------------------
// poly_32_uniform_rand (src/utils/arith/poly_32.c:132-156) is a serial placement loop over one CSPRNG word per attempt (bit 0:
// sign, the bits above: position; an occupied position costs the word and is drawn again): one lane per key pair, the polynomial
// as a private byte column in LDS.  with_g != 0 (first round): g is drawn first, stored as the key stores it (2g + 1 at
// coefficient 0, :1175-1176) and staged for its transform; then one candidate f.  need: 1 = candidate staged, 2 = tape exhausted.
__global__ __launch_bounds__(HT) void bl_kg_sample_kernel(const uint32_t *active, uint32_t nact, int with_g, const uint32_t *tapes, size_t tape_stride,
                                                          const uint32_t *tape_lens, uint32_t *pos, uint32_t *status, uint32_t *need,
                                                          int16_t *g16, int32_t *tg, int32_t *tf, int n, int nz0, int nz1)
{
    extern __shared__ int8_t col[];                                // n * HT
    const uint32_t aidx = blockIdx.x * HT + threadIdx.x;
    if (aidx >= nact) return;
    const uint32_t item = active ? active[aidx] : aidx;
    const uint32_t *w = tapes + (size_t)item * tape_stride;
    const uint32_t len = tape_lens ? tape_lens[item] : (uint32_t)tape_stride;
    uint32_t p = pos[item];
    int8_t *mine = col + threadIdx.x;
    const int c[2] = { nz0, nz1 };
    auto draw = [&]() -> bool {
        for (int i = 0; i < n; i++) mine[i * HT] = 0;
        for (int j = 0; j < 2; j++) {
            int i = 0;
            while (i < c[j]) {
                if (p >= len) return false;
                const uint32_t r = w[p++];
                const uint32_t idx = (r >> 1) & (uint32_t)(n - 1);
                if (mine[idx * HT] == 0) { mine[idx * HT] = (int8_t)((r & 1) ? j - 2 : 2 - j); i++; }
            }
        }
        return true;
    };
    bool ok = true;
    if (with_g) {
        ok = draw();
        if (ok) for (int i = 0; i < n; i++) { const int32_t v = 2 * (int32_t)mine[i * HT] + (i == 0); g16[(size_t)item * n + i] = (int16_t)v; tg[(size_t)aidx * n + i] = v; }
    }
    if (ok) ok = draw();
    if (ok) for (int i = 0; i < n; i++) tf[(size_t)aidx * n + i] = mine[i * HT];
    else for (int i = 0; i < n; i++) { tf[(size_t)aidx * n + i] = 1; if (with_g) tg[(size_t)aidx * n + i] = 0; }      // keeps the dense transforms defined
    pos[item] = p;
    need[aidx] = ok ? 1u : 2u;
    if (!ok) status[item] = 2;
}

// dense copy of NTT(2g+1) of the active key pairs (the numerator of the division)
__global__ void bl_kg_gather_kernel(const uint32_t *active, uint32_t nact, const int32_t *th, int32_t *num, int n)
{
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int ln = 31 - __clz(n);                                         // n is a power of two (bliss_create): no 64-bit division per element
    const uint32_t aidx = (uint32_t)(t >> ln); const int i = (int)(t & (size_t)(n - 1));
    if (aidx >= nact) return;
    const uint32_t item = active ? active[aidx] : aidx;
    num[t] = th[(size_t)item * n + i];
}

// rc == 0: NTT(f) was invertible, num = NTT(2g+1) / NTT(f): a = normalise(-num) (:1197-1209), f and a stored, status 1;
// otherwise the key pair goes to the next round (a fresh f).  One wavefront per active key pair.
__global__ __launch_bounds__(256) void bl_kg_finish_kernel(const uint32_t *active, uint32_t nact, const uint32_t *need, const int32_t *rc, const int32_t *num,
                                                           const int32_t *tf, int16_t *f16, int16_t *a16, uint32_t *status, uint32_t *attempts,
                                                           uint32_t *next_active, uint32_t *next_count, int n, int32_t q)
{
    const int lane = threadIdx.x & 63;
    const uint32_t aidx = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (aidx >= nact) return;
    const uint32_t item = active ? active[aidx] : aidx;
    if (need[aidx] == 2) return;
    if (lane == 0) attempts[item] += 1;
    if (rc[aidx] != 0) { if (lane == 0) next_active[atomicAdd(next_count, 1u)] = item; return; }
    for (int i = lane; i < n; i += 64) {
        int32_t x = num[(size_t)aidx * n + i] % q;
        if (x < 0) x += q;
        a16[(size_t)item * n + i] = (int16_t)(x ? q - x : 0);
        f16[(size_t)item * n + i] = (int16_t)tf[(size_t)aidx * n + i];
    }
    if (lane == 0) status[item] = 1;
}

}  // namespace

// ------------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------------

struct sc_hip_bliss {
    sc_hip_ctx *ctx;
    sc_hip_plan *plan;
    sc_hip_cdt *cdt;
    BlissP P;
    bool have_key;
    int16_t *d_f, *d_g;
    int32_t *d_a;
    int32_t *d_acorr;            // A[d] = <f, x^d f> + <g, x^d g>, d = 0 .. n-1: what greedy_sc's sign decisions need of the key (bl_decide_kernel)
    size_t cap;
    St st;
    uint32_t *act[2], *cnt, *need, *sbase, *ok, *trials;
    int32_t *tu, *tin, *v, *cidx, *ain;
    int16_t *z;
    int speculate;               // 1: later pairs of attempts side by side (bl_sign_core); 0: the reference's loop round by round; 2: self-test of the redo path
    uint32_t spw;                // tape words a Gaussian sample takes: 2 (one prng_64: the 64-bit sampler), 4 / 6 behind sc_hip_bliss_set_cdt_table_hp
    St st2; int32_t *tu2; uint8_t *sigs2; uint32_t *siglens2, *trials2, *msg_of2; size_t cap2, stride2;     // the pairs' state, vectors and output rows
    int entropy;                 // 0: raw signature fields, 1: SC_ENTROPY_HUFFMAN_STATIC (sc_hip_bliss_set_entropy), 2: SC_ENTROPY_BAC (sc_hip_bliss_set_entropy_bac)
    uint64_t *d_bac;             // SC_ENTROPY_BAC: the coder's two distributions, 2^z1_bits + 2^z2_bits entries (sc_hip_bliss_set_entropy_bac)
    uint32_t *d_huff;            // the two code tables and their decoding orders, one allocation
    HuffT ht[2];
    double tie_rel;              // accept / reject draws closer to the threshold than this (relative) are reported, not decided: status 5
    // sc_hip_bliss_sign_seeded: the batch's CSPRNG word tapes, expanded from the caller's seeds as far as the attempts get (rows of seed_words words)
    uint32_t *seed_tape, *seed_list; size_t seed_rows, seed_words;
    uint32_t *h_pin;             // 64 bytes of pinned host memory: where the call's one look at the device's counts lands (a copy to pageable memory makes the runtime wait
                                 // for it BEFORE the closing wipes behind it are queued: 35 us of idle device per call)
};

static void bl_free_ws(sc_hip_bliss *b)
{
    void *ptrs[] = { b->st.pos, b->st.var_buf, b->st.iter, b->st.cur, b->st.status, b->st.var_bits, b->act[0], b->act[1], b->cnt, b->need, b->sbase, b->ok,
                     b->trials, b->tu, b->tin, b->v, b->cidx, b->z, b->ain };
    for (void *p : ptrs) if (p) (void)sc::wipe_free(p);
    memset(&b->st, 0, sizeof b->st);
    b->act[0] = b->act[1] = b->cnt = b->need = b->sbase = b->ok = b->trials = nullptr;
    b->tu = b->tin = b->v = b->cidx = b->ain = nullptr; b->z = nullptr; b->cap = 0;
}

static int bl_ensure_ws(sc_hip_bliss *b, size_t batch)
{
    if (batch <= b->cap) return SC_HIP_SUCCESS;
    bl_free_ws(b);
    const size_t n = b->P.n;
#define WS(ptr, bytes) HIPCHK(hipMalloc((void **)&(ptr), (bytes)))
    WS(b->st.pos, batch * 4); WS(b->st.var_buf, batch * 4); WS(b->st.iter, batch * 4); WS(b->st.cur, batch * 4); WS(b->st.status, batch * 4); WS(b->st.var_bits, batch * 4);
    WS(b->act[0], batch * 4); WS(b->act[1], batch * 4); WS(b->cnt, 64); WS(b->need, batch * 4); WS(b->sbase, batch * 4); WS(b->ok, batch * 4); WS(b->trials, batch * 4);
    WS(b->tu, batch * 2 * n * 4); WS(b->tin, batch * n * 4); WS(b->v, batch * n * 4); WS(b->cidx, batch * (size_t)b->P.kappa * 4); WS(b->z, batch * n * 2); WS(b->ain, batch * n * 4);
#undef WS
    b->cap = batch;
    return SC_HIP_SUCCESS;
}

extern "C" int sc_hip_bliss_create(sc_hip_ctx_t *ctx, const sc_hip_bliss_params_t *params, sc_hip_bliss_t **out)
{
    if (!ctx || !params || !out) return fail("bliss_create: null argument");
    static_assert(sizeof(BlissP) == sizeof(sc_hip_bliss_params_t), "parameter struct layout");
    BlissP P; memcpy(&P, params, sizeof P);
    if (P.n != 256 && P.n != 512) return fail("bliss_create: n must be 256 or 512");
    if (P.kappa < 1 || P.kappa > 64 || P.n_bits < 1 || (1 << P.n_bits) != P.n) return fail("bliss_create: bad kappa / n_bits");
    if (P.z1_bits < 2 || P.z1_bits > 24 || P.z2_bits < 2 || P.z2_bits > 16 || P.d < 1 || P.d > 20 || P.p < 2) return fail("bliss_create: bad packing / rounding parameters");
    sc_hip_plan *plan = nullptr; sc_hip_cdt *cdt = nullptr;
    if (sc_hip_plan_create(ctx, P.q, P.n, &plan) != SC_HIP_SUCCESS) return SC_HIP_FAILURE;
    if (sc_hip_cdt_create(ctx, P.tail, P.sigma, &cdt) != SC_HIP_SUCCESS) { sc_hip_plan_destroy(plan); return SC_HIP_FAILURE; }
    sc_hip_bliss *b = new sc_hip_bliss;
    memset(b, 0, sizeof *b);
    b->ctx = ctx; b->plan = plan; b->cdt = cdt; b->P = P;
    b->tie_rel = 0x1p-40;
    b->speculate = 1;
    b->spw = 2;
    *out = b;
    return SC_HIP_SUCCESS;
}

// The sampler table of the handle.  sc_hip_bliss_create tabulates it on THIS host with the reference's long-double recurrence (gaussian_cdf_create_64);
// glibc's x87 expl is not bit-identical across CPU vendors (3 of 2048 entries differed by one unit between two hosts), so a caller
// that holds the reference's own table - the libsafecrypto-side binding does: sc->sc_gauss - hands it over and the samples are the reference's on any host.
extern "C" int sc_hip_bliss_set_cdt_table(sc_hip_bliss_t *b, const uint64_t *h_cdf, int size)
{
    if (!b || !h_cdf) return fail("bliss_set_cdt_table: null argument");
    sc_hip_cdt *cdt = nullptr;
    if (sc_hip_cdt_create_from_table(b->ctx, h_cdf, size, &cdt) != SC_HIP_SUCCESS) return SC_HIP_FAILURE;
    if (hipSetDevice(b->ctx->device) != hipSuccess || hipDeviceSynchronize() != hipSuccess) {      // nothing in flight may still read the old table
        sc_hip_cdt_destroy(cdt);                                       // (the handle keeps the table it had)
        return fail("bliss_set_cdt_table: device synchronisation failed");
    }
    sc_hip_cdt_destroy(b->cdt);
    b->cdt = cdt;
    b->spw = 2;
    return SC_HIP_SUCCESS;
}
// the sampler of a context created with SC_FLAG_0_SAMPLE_128BIT / _192BIT (what the reference's BLISS-B harness asks for, test/functional/func_alg_bliss_b.c:118): the context's
// own table (gauss_cdf_high_t::cdf_128 / cdf_192 - MPFR arithmetic, the library's), a sample = precision / 64 prng_64 draws of the tape
extern "C" int sc_hip_bliss_set_cdt_table_hp(sc_hip_bliss_t *b, int precision_bits, const uint64_t *h_cdf, int size)
{
    if (!b || !h_cdf) return fail("bliss_set_cdt_table_hp: null argument");
    sc_hip_cdt *cdt = nullptr;
    if (sc_hip_cdt_hp_create_from_table(b->ctx, precision_bits, h_cdf, size, &cdt) != SC_HIP_SUCCESS) return SC_HIP_FAILURE;
    if (hipSetDevice(b->ctx->device) != hipSuccess || hipDeviceSynchronize() != hipSuccess) {
        sc_hip_cdt_destroy(cdt);
        return fail("bliss_set_cdt_table_hp: device synchronisation failed");
    }
    sc_hip_cdt_destroy(b->cdt);
    b->cdt = cdt;
    b->spw = (uint32_t)(precision_bits / 32);
    return SC_HIP_SUCCESS;
}

// the margin inside which the accept / reject draw is reported (status 5) instead of decided; 2^-40 by default.  A margin of 1 reports every
// attempt that reaches the test (self-test of the reporting path).
extern "C" int sc_hip_bliss_set_tie_margin(sc_hip_bliss_t *b, double relative)
{
    if (!b) return fail("bliss_set_tie_margin: null handle");
    if (!(relative >= 0.0) || relative > 1.0) return fail("bliss_set_tie_margin: margin must be in [0, 1]");
    b->tie_rel = relative;
    return SC_HIP_SUCCESS;
}

// sc->coding_signature.type for the handle (bliss_b.c:197-200): 0 = SC_ENTROPY_NONE, 1 = SC_ENTROPY_HUFFMAN_STATIC with the two tables the reference's coder
// would pick for z1_bits and z2_bits (entropy_huffman.c:22-34: huff_table_gaussian_<min(bits, 7) - 1>, _2 below 4 bits) as (code, bits) per symbol
extern "C" int sc_hip_bliss_set_entropy(sc_hip_bliss_t *b, int type, const uint32_t *codes_t, const uint16_t *bits_t, uint32_t depth_t,
                                        const uint32_t *codes_z, const uint16_t *bits_z, uint32_t depth_z)
{
    if (!b) return fail("bliss_set_entropy: null handle");
    if (type != 0 && type != 1) return fail("bliss_set_entropy: type must be 0 (none) or 1 (static Huffman); the arithmetic coder has sc_hip_bliss_set_entropy_bac");
    if (type == 0) { b->entropy = 0; return SC_HIP_SUCCESS; }
    if (!codes_t || !bits_t || !codes_z || !bits_z) return fail("bliss_set_entropy: null table");
    const uint32_t depth[2] = { depth_t, depth_z };
    const uint32_t *codes[2] = { codes_t, codes_z };
    const uint16_t *bits[2] = { bits_t, bits_z };
    const int fbits[2] = { b->P.z1_bits, b->P.z2_bits };
    std::vector<uint32_t> h;
    size_t at[2];
    uint32_t lutbits[2];
    for (int t = 0; t < 2; t++) {
        at[t] = h.size();
        if (sc::huff_build_table(codes[t], bits[t], depth[t], fbits[t], &h, &lutbits[t], "bliss_set_entropy")) return SC_HIP_FAILURE;
    }
    if (b->P.n % 64) return fail("bliss_set_entropy: n must be a multiple of 64");
    HIPCHK(hipSetDevice(b->ctx->device));
    if (b->d_huff) { (void)hipFree(b->d_huff); b->d_huff = nullptr; }
    HIPCHK(hipMalloc((void **)&b->d_huff, h.size() * 4));
    HIPCHK(sc::copy_in(b->d_huff, h.data(), h.size() * 4));
    for (int t = 0; t < 2; t++) {
        const uint32_t *base = b->d_huff + at[t];
        b->ht[t] = HuffT{ base, base + depth[t], base + 2 * depth[t], base + 3 * depth[t], base + 4 * depth[t], depth[t], lutbits[t] };
    }
    b->entropy = 1;
    return SC_HIP_SUCCESS;
}

// replaces: the signature coding of SC_FLAG_0_ENTROPY_BAC (bliss_b.c:198, 397-398, 1453-1463, 1650-1657; src/utils/entropy/bac.c).  dist_t / dist_z: the
// coder's distributions for t (2^z1_bits entries) and z (2^z2_bits) as entropy_dist_create / gauss_freq_bac_64 leave them in sc->dist[2], sc->dist[3] -
// long double arithmetic of the caller's host, handed over as data like the sampler's table (sc_hip_bliss_set_cdt_table).
extern "C" int sc_hip_bliss_set_entropy_bac(sc_hip_bliss_t *b, const uint64_t *dist_t, const uint64_t *dist_z)
{
    if (!b) return fail("bliss_set_entropy_bac: null handle");
    if (!dist_t || !dist_z) return fail("bliss_set_entropy_bac: null table");
    if (b->P.z1_bits < 1 || b->P.z1_bits > 16 || b->P.z2_bits < 1 || b->P.z2_bits > 16) return fail("bliss_set_entropy_bac: field widths of 1..16 bits");
    const size_t nt = (size_t)1 << b->P.z1_bits, nz = (size_t)1 << b->P.z2_bits;
    HIPCHK(hipSetDevice(b->ctx->device));
    if (b->d_bac) { (void)hipFree(b->d_bac); b->d_bac = nullptr; }
    HIPCHK(hipMalloc((void **)&b->d_bac, (nt + nz) * 8));
    HIPCHK(sc::copy_in(b->d_bac, dist_t, nt * 8));
    HIPCHK(sc::copy_in(b->d_bac + nt, dist_z, nz * 8));
    b->entropy = 2;
    return SC_HIP_SUCCESS;
}

static void bl_free_ws2(sc_hip_bliss *b);
// measurement / test hook (include/sc_hip_bench.h): 0 = the reference's loop round by round, 1 = the later pairs of attempts side by side (default)
extern "C" int sc_hip_bliss_set_speculation(sc_hip_bliss_t *b, int on)
{
    if (!b) return fail("bliss_set_speculation: null handle");
    if (on < 0 || on > 3) return fail("bliss_set_speculation: 0, 1, 2 or 3");
    b->speculate = on;
    return SC_HIP_SUCCESS;
}

extern "C" void sc_hip_bliss_destroy(sc_hip_bliss_t *b)
{
    if (!b) return;
    bl_free_ws(b);
    bl_free_ws2(b);
    if (b->seed_tape) (void)sc::wipe_free(b->seed_tape);
    if (b->seed_list) (void)hipFree(b->seed_list);
    if (b->h_pin) (void)hipHostFree(b->h_pin);
    if (b->d_f) (void)sc::wipe_free(b->d_f);
    if (b->d_g) (void)sc::wipe_free(b->d_g);
    if (b->d_acorr) (void)sc::wipe_free(b->d_acorr);
    if (b->d_a) (void)sc::wipe_free(b->d_a);
    if (b->d_huff) (void)hipFree(b->d_huff);
    if (b->d_bac) (void)hipFree(b->d_bac);
    sc_hip_cdt_destroy(b->cdt);
    sc_hip_plan_destroy(b->plan);
    delete b;
}

// f, g: private key (SINT16[n] each, bliss_b.c:1328-1330; may be NULL for a verify-only handle); a: public key in the NTT domain
extern "C" int sc_hip_bliss_set_key(sc_hip_bliss_t *b, const int16_t *f, const int16_t *g, const int16_t *a)
{
    if (!b || !a) return fail("bliss_set_key: null argument");
    if ((f == nullptr) != (g == nullptr)) return fail("bliss_set_key: f and g go together");
    HIPCHK(hipSetDevice(b->ctx->device));
    const size_t n = b->P.n;
    std::vector<int32_t> a32(n);
    for (size_t i = 0; i < n; i++) a32[i] = a[i];
    if (!b->d_a) HIPCHK(hipMalloc((void **)&b->d_a, n * 4));
    HIPCHK(sc::copy_in(b->d_a, a32.data(), n * 4));
    if (f) {
        if (!b->d_f) HIPCHK(hipMalloc((void **)&b->d_f, n * 2));
        if (!b->d_g) HIPCHK(hipMalloc((void **)&b->d_g, n * 2));
        HIPCHK(sc::copy_in(b->d_f, f, n * 2));
        HIPCHK(sc::copy_in(b->d_g, g, n * 2));
        if (!b->d_acorr) HIPCHK(hipMalloc((void **)&b->d_acorr, n * 4));
        hipLaunchKernelGGL(bl_acorr_kernel, dim3((unsigned)((n + 63) / 64)), dim3(64), 0, nullptr, b->d_f, b->d_g, b->d_acorr, (int)n);
        HIPCHK(hipGetLastError());
        HIPCHK(hipStreamSynchronize(nullptr));
    }
    b->have_key = true;
    return SC_HIP_SUCCESS;
}

extern "C" size_t sc_hip_bliss_sig_bytes(const sc_hip_bliss_t *b)
{
    if (!b) return 0;
    return ((size_t)b->P.n * (b->P.z1_bits + b->P.z2_bits) + (size_t)b->P.kappa * b->P.n_bits + 7) / 8;
}

// one key for the batch (kf / kg / ka16 null: the handle's), or a key per message: kf, kg, ka16 = SINT16 [batch][n] rows as sc_hip_bliss_keygen writes them
static int bl_sign_core(sc_hip_bliss_t *b, const int16_t *kf, const int16_t *kg, const int16_t *ka16, size_t batch, const uint8_t *d_msgs, size_t msg_stride,
                        const uint32_t *d_mlens, uint32_t fixed_len, const uint32_t *d_tapes, size_t tape_stride, const uint32_t *d_tape_lens,
                        uint8_t *d_sigs, size_t sig_stride, uint32_t *d_siglens, uint32_t *d_trials, uint32_t *d_status, void *stream,
                        const uint8_t *d_seeds = nullptr, size_t seed_stride = 0, size_t seed_len = 0);

extern "C" int sc_hip_bliss_sign(sc_hip_bliss_t *b, size_t batch, const uint8_t *d_msgs, size_t msg_stride, const uint32_t *d_mlens, uint32_t fixed_len,
                                 const uint32_t *d_tapes, size_t tape_stride, const uint32_t *d_tape_lens,
                                 uint8_t *d_sigs, size_t sig_stride, uint32_t *d_siglens, uint32_t *d_trials, uint32_t *d_status, void *stream)
{
    if (!b) return fail("bliss_sign: null handle");
    if (!b->have_key || !b->d_f) return fail("bliss_sign: no private key loaded");
    return bl_sign_core(b, nullptr, nullptr, nullptr, batch, d_msgs, msg_stride, d_mlens, fixed_len, d_tapes, tape_stride, d_tape_lens, d_sigs, sig_stride, d_siglens, d_trials, d_status, stream);
}

// sc_hip_bliss_sign with the CSPRNG inside the call: message i signs with the generator the reference makes of seed i (AES-256 CTR-DRBG as create_csprng /
// prng_create instantiate it: src/utils/crypto/prng.c:963-1107, ctr_drbg.c:37-197), read through prng_32 / prng_64 / prng_var exactly as bliss_b_sign reads
// sc->prng_ctx[0] (bliss_b.c:1372-1373, sampling.c:211-228).  The words are expanded on the device as far as a message's attempts get: 4n + a few for the
// first pair of attempts of every message, the later pairs' only for the messages that fail both.  Byte-identical to sc_hip_bliss_sign over
// sc_hip_drbg_words(seeds, ...) rows; a message that would need more than 26 attempts (p ~ 1e-12) ends with status 2.
extern "C" int sc_hip_bliss_sign_seeded(sc_hip_bliss_t *b, size_t batch, const uint8_t *d_msgs, size_t msg_stride, const uint32_t *d_mlens, uint32_t fixed_len,
                                        const uint8_t *d_seeds, size_t seed_stride, size_t seed_len,
                                        uint8_t *d_sigs, size_t sig_stride, uint32_t *d_siglens, uint32_t *d_trials, uint32_t *d_status, void *stream)
{
    if (!b) return fail("bliss_sign_seeded: null handle");
    if (!b->have_key || !b->d_f) return fail("bliss_sign_seeded: no private key loaded");
    if (!d_seeds) return fail("bliss_sign_seeded: null seed buffer");
    return bl_sign_core(b, nullptr, nullptr, nullptr, batch, d_msgs, msg_stride, d_mlens, fixed_len, nullptr, 0, nullptr, d_sigs, sig_stride, d_siglens, d_trials, d_status, stream,
                        d_seeds, seed_stride, seed_len);
}

// replaces: safecrypto_sign -> bliss_b_sign for a batch of messages EACH under its own key pair (distinct keys)
extern "C" int sc_hip_bliss_sign_keys(sc_hip_bliss_t *b, size_t batch, const int16_t *d_f, const int16_t *d_g, const int16_t *d_a,
                                      const uint8_t *d_msgs, size_t msg_stride, const uint32_t *d_mlens, uint32_t fixed_len,
                                      const uint32_t *d_tapes, size_t tape_stride, const uint32_t *d_tape_lens,
                                      uint8_t *d_sigs, size_t sig_stride, uint32_t *d_siglens, uint32_t *d_trials, uint32_t *d_status, void *stream)
{
    if (!b) return fail("bliss_sign_keys: null handle");
    if (!d_f || !d_g || !d_a) return fail("bliss_sign_keys: null key buffer");
    return bl_sign_core(b, d_f, d_g, d_a, batch, d_msgs, msg_stride, d_mlens, fixed_len, d_tapes, tape_stride, d_tape_lens, d_sigs, sig_stride, d_siglens, d_trials, d_status, stream);
}

// ---- the rejection loop, and the later pairs of attempts side by side ------------------------------------------------------------------------
// What one run of the round loop works on: a set of items (messages, or pairs of attempts of messages), their state, vectors and output rows.
struct BlRun {
    St st; int32_t *tu; uint8_t *sigs; size_t sig_stride; uint32_t *siglens, *trials;
    uint32_t items; const uint32_t *active0; uint32_t max_iter;
    const uint32_t *items_dev = nullptr;      // the item count where only the device knows it (`items` is then the host's upper bound): a pass queued without a look
};
struct BlIn {
    const int16_t *kf, *kg, *ka16; const uint8_t *msgs; size_t msg_stride; const uint32_t *mlens; uint32_t fixed_len;
    const uint32_t *tapes; size_t tape_stride; const uint32_t *tape_lens;
    // seeded call (sc_hip_bliss_sign_seeded): `tapes` is the handle's own buffer, filled from these seeds as the passes need it
    const uint8_t *seeds = nullptr; size_t seed_stride = 0, seed_len = 0;
};

// several arrays zeroed by one launch (a hipMemsetAsync each costs a launch gap of its own: seven in front of a signing call, four behind each pass)
struct ZeroList { uint8_t *p[8]; size_t bytes[8]; int count; };
__global__ void bl_zero_kernel(ZeroList L)
{
    const size_t stride = (size_t)gridDim.x * blockDim.x, t0 = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (int a = 0; a < L.count; a++) {
        uint4 *q = reinterpret_cast<uint4 *>(L.p[a]);                 // (hipMalloc'd bases: 16-byte aligned)
        const size_t n16 = L.bytes[a] / 16;
        for (size_t i = t0; i < n16; i += stride) q[i] = make_uint4(0, 0, 0, 0);
        if (t0 < (L.bytes[a] & 15)) L.p[a][n16 * 16 + t0] = 0;
    }
}
static int bl_zero(sc_hip_bliss_t *b, std::initializer_list<std::pair<void *, size_t>> arrays, hipStream_t s)
{
    ZeroList L{};
    size_t most = 0;
    for (const auto &a : arrays) {
        if (!a.first || !a.second) continue;
        if (L.count == 8) return fail("bliss: zero list too long");
        L.p[L.count] = static_cast<uint8_t *>(a.first); L.bytes[L.count] = a.second; L.count++;
        if (a.second > most) most = a.second;
    }
    if (!L.count) return SC_HIP_SUCCESS;
    size_t blocks = (most / 16 + 255) / 256;
    if (blocks < 1) blocks = 1;
    if (blocks > (size_t)b->ctx->cus * 8) blocks = (size_t)b->ctx->cus * 8;
    hipLaunchKernelGGL(bl_zero_kernel, dim3((unsigned)blocks), dim3(256), 0, s, L);
    HIPCHK(hipGetLastError());
    return SC_HIP_SUCCESS;
}

// one round over the list `active` (null: identity).  nact is the number of items, or - with nd, the device-side count - the host's upper bound on it:
// the grids are sized for the bound and the kernels read the count.
static int bl_round_launch(sc_hip_bliss_t *b, const BlRun &r, const BlIn &in, hipStream_t s, const uint32_t *active, uint32_t nact, const uint32_t *nd,
                           uint32_t *next, uint32_t *next_cnt)
{
    const BlissP &P = b->P;
    const int n = P.n;
    const St &st = r.st;
    const int hpw = b->cdt->hp_words;                                  // 0: the 64-bit sampler; 2 / 3: 128- / 192-bit precision (sc_hip_bliss_set_cdt_table_hp)
    if (hpw) hipLaunchKernelGGL(bl_begin_hp_kernel, dim3((nact + 255) / 256), dim3(256), 0, s, active, nact, st, in.tape_lens, (uint32_t)in.tape_stride, n, (uint32_t)(4 * n * hpw),
                                b->need, b->sbase, r.siglens, in.mlens, nd);
    else hipLaunchKernelGGL(bl_begin_kernel, dim3((nact + 255) / 256), dim3(256), 0, s, active, nact, st, in.tape_lens, (uint32_t)in.tape_stride, n, b->need, b->sbase, r.siglens, in.mlens, nd);
    const size_t nsamp = (size_t)nact * 2 * n;
    size_t sblocks = (nsamp + 256 * 32 - 1) / (256 * 32);
    const size_t slds = sc::cdt_lds_bytes(b->cdt), sper = (size_t)(160 * 1024) / slds;
    if (sblocks > (size_t)b->ctx->cus * (sper > 4 ? 4 : sper)) sblocks = (size_t)b->ctx->cus * (sper > 4 ? 4 : sper);
    const CdtGuide &cg = b->cdt->g;
#define BL_SAMPLE(SS) hipLaunchKernelGGL(bl_sample_kernel<SS>, dim3((unsigned)sblocks), dim3(256), slds, s, active, nact, b->need, b->sbase, st.cur, \
                       in.tapes, in.tape_stride, b->cdt->d_cdf, b->cdt->d_guide, cg, b->cdt->size, r.tu, n, st.msg_of, nd)
#define BL_SAMPLE_HP(SS) hipLaunchKernelGGL(bl_sample_hp_kernel<SS>, dim3((unsigned)sblocks), dim3(256), slds, s, active, nact, b->need, b->sbase, st.cur, \
                       in.tapes, in.tape_stride, b->cdt->d_cdf, b->cdt->d_guide, cg, b->cdt->size, r.tu, n, hpw, st.msg_of, nd)
    if (hpw) { switch (cg.S) { case 0: BL_SAMPLE_HP(0); break; case 1: BL_SAMPLE_HP(1); break; case 2: BL_SAMPLE_HP(2); break; case 3: BL_SAMPLE_HP(3); break; default: BL_SAMPLE_HP(4); break; } }
    else switch (cg.S) { case 0: BL_SAMPLE(0); break; case 1: BL_SAMPLE(1); break; case 2: BL_SAMPLE(2); break; case 3: BL_SAMPLE(3); break; default: BL_SAMPLE(4); break; }
#undef BL_SAMPLE_HP
#undef BL_SAMPLE
    const size_t ne = (size_t)nact * n;
    hipLaunchKernelGGL(bl_stage_kernel, dim3((unsigned)((ne + 255) / 256)), dim3(256), 0, s, active, nact, b->need, st.cur, r.tu, b->tin, n, in.ka16, b->ain, st.msg_of, nd);
    HIPCHK(hipGetLastError());
    // (with a device-side count the product runs over the bound: rows past the count hold what an earlier round left there and are read by nobody)
    if (in.ka16 ? sc_hip_polymul_ntt(b->plan, nact, b->tin, b->ain, b->v, s) : sc::polymul_ntt_bcast(b->plan, nact, b->tin, b->d_a, b->v, s, nd)) return SC_HIP_FAILURE;
    hipLaunchKernelGGL(bl_round_kernel, dim3((unsigned)((ne + 255) / 256)), dim3(256), 0, s, active, nact, st.cur, r.tu, b->v, b->z, P, nd);
    // two lanes per message while the one-lane kernel would not give every SIMD a wavefront (measured: +10 % sign at 16384, +12 % at 65536)
    if ((size_t)nact <= (size_t)b->ctx->cus * 4 * 64)
        hipLaunchKernelGGL(bl_oracle2_kernel, dim3((2 * nact + HT - 1) / HT), dim3(HT), 0, s, active, nact, b->need, in.msgs, in.msg_stride, in.mlens, in.fixed_len, b->z, b->cidx, b->ok,
                           0, (uint32_t *)nullptr, P, st.msg_of, nd);
    else
        hipLaunchKernelGGL(bl_oracle1_kernel, dim3((nact + HT - 1) / HT), dim3(HT), 0, s, active, nact, b->need, in.msgs, in.msg_stride, in.mlens, in.fixed_len, b->z, b->cidx, b->ok,
                           0, (uint32_t *)nullptr, P, st.msg_of, nd);
#define BL_DECIDE(KM) hipLaunchKernelGGL(bl_decide_kernel<KM>, dim3((nact + 3) / 4), dim3(256), sizeof(int32_t) * 8 * (size_t)n + (in.kf ? 0 : sizeof(uint32_t) * 8 * (size_t)n), s, active, nact, b->need, b->ok, st, \
                       in.tapes, in.tape_stride, in.tape_lens, \
                       in.kf ? in.kf : b->d_f, in.kg ? in.kg : b->d_g, r.tu, b->v, b->z, b->cidx, r.sigs, r.sig_stride, r.siglens, r.trials, next, next_cnt, P, b->tie_rel, \
                       in.kf ? (size_t)n : (size_t)0, r.max_iter, nd, in.kf ? (const int32_t *)nullptr : b->d_acorr)
    if (P.kappa <= 32) BL_DECIDE(32); else BL_DECIDE(64);
#undef BL_DECIDE
    HIPCHK(hipGetLastError());
    return SC_HIP_SUCCESS;
}

static int bl_rounds(sc_hip_bliss_t *b, const BlRun &r, const BlIn &in, hipStream_t s)
{
    if (r.items == 0) return SC_HIP_SUCCESS;
    if (r.max_iter == 2) {
        // An item is one pair of attempts: two rounds, and an item that fails both ends with status 7 - there is no third.  The second round's list and
        // its count stay on the device (its kernels read them; grids sized for the first round's count), so nothing is read back here: a read-back
        // idles the device for 40-50 us, a fifth of such a round.
        HIPCHK(hipMemsetAsync(b->cnt, 0, 8, s));
        if (bl_round_launch(b, r, in, s, r.active0, r.items, r.items_dev, b->act[0], b->cnt)) return SC_HIP_FAILURE;
        return bl_round_launch(b, r, in, s, b->act[0], r.items, b->cnt, b->act[1], b->cnt + 1);
    }
    uint32_t nact = r.items;
    const uint32_t *active = r.active0;
    for (uint32_t round = 0; nact > 0; round++) {
        if (round > 100000) return fail("bliss_sign: rejection loop did not terminate");
        uint32_t *next = b->act[round & 1];
        HIPCHK(hipMemsetAsync(b->cnt, 0, 4, s));
        if (bl_round_launch(b, r, in, s, active, nact, nullptr, next, b->cnt)) return SC_HIP_FAILURE;
        uint32_t cnt = 0;
        HIPCHK(hipMemcpyAsync(&cnt, b->cnt, 4, hipMemcpyDeviceToHost, s));
        HIPCHK(hipStreamSynchronize(s));
        nact = cnt;
        active = next;
    }
    return SC_HIP_SUCCESS;
}

// The reference's loop is sequential in the attempts of one message, but only PAIRS of attempts depend on each other: attempt 2p draws fresh vectors, attempt
// 2p+1 re-uses them modified (bliss_b.c:1366-1373); and an attempt's position on the CSPRNG word tape does not depend on any VALUE: a fresh attempt takes
// 4n words, every attempt 1 + 27 + 26 reservoir bits (prng_var: a 32-bit word whenever the reservoir runs short).  So pair p of a message can start from a
// state the host computes by running the reservoir's bit count forward - and the pairs of a message can run side by side, the lowest-numbered successful
// attempt being the signature (what the sequential loop stops at).  (An attempt whose oracle fails skips its reservoir draws, :1395-1397 - it never does in
// practice; such a message is redone in order, status 8 below.)
struct ResSim {
    uint32_t pos = 0, lrp = 0; int vb = 0;
    void attempt(bool fresh, uint32_t n, uint32_t spw = 2)            // spw: tape words a sample (2: one prng_64; 4 / 6: a 128- / 192-bit sampler's two / three)
    {
        if (fresh) pos += 2 * n * spw;
        const int req[3] = { 1, 27, 26 };
        for (int nreq : req) { if (vb < nreq) { const int bits = nreq - vb; lrp = pos; pos++; vb = 32 - bits; } else vb -= nreq; }
    }
};
// the CSPRNG words bliss_b_sign has drawn when its `attempts`-th attempt ends (what a tape must hold for a signature that takes that many): host arithmetic
// only - the same bookkeeping that places the later pairs
static size_t bl_tape_words(int n, unsigned attempts, uint32_t spw)
{
    if (n <= 0) return 0;
    ResSim sim;
    for (unsigned k = 0; k < attempts; k++) sim.attempt((k & 1u) == 0, (uint32_t)n, spw);
    return sim.pos;
}
extern "C" size_t sc_hip_bliss_tape_words(int n, unsigned attempts) { return bl_tape_words(n, attempts, 2); }
extern "C" size_t sc_hip_bliss_tape_words_hp(int n, unsigned attempts, int precision_bits)
{
    if (precision_bits != 64 && precision_bits != 128 && precision_bits != 192) return 0;
    return bl_tape_words(n, attempts, (uint32_t)(precision_bits / 32));
}

constexpr int BL_PAIRS = 4;                    // later pairs of a message evaluated side by side
struct PairInit { uint32_t pos[BL_PAIRS], lrp[BL_PAIRS]; int32_t vb[BL_PAIRS]; };

// the messages that go on to their later pairs (status 7 -> list7, cnt[0]) and the ones to be redone in order (status 8 -> list8, cnt[1]): one pass, one read-back
__global__ void bl_collect_kernel(const uint32_t *status, uint32_t count, uint32_t *list7, uint32_t *list8, uint32_t *cnt)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    const uint32_t st = status[i];
    if (st == 7) list7[atomicAdd(&cnt[0], 1u)] = i;
    else if (st == 8) list8[atomicAdd(&cnt[1], 1u)] = i;
}
// a later pass queued WITHOUT a look at how many messages go on: the device checks the count against what the pass was sized for and derives the item count
// (out[0] = messages, out[1] = items, out[2] = messages again, kept for the closing wipe of a seeded call, out[3] = the pass was refused)
// More messages than the pass was sized for (a quarter of the batch where an eighth is expected: tens of standard deviations away, but possible): the pass
// serves NOBODY - serving some would leave the others to skip pairs - and out[3] tells the look behind it to start over with the count in hand.
__global__ void bl_blind_kernel(const uint32_t *cnt7, uint32_t bound, uint32_t pp, uint32_t *out)
{
    const bool over = cnt7[0] > bound;
    const uint32_t nu = over ? 0u : cnt7[0];
    out[0] = nu; out[1] = nu * pp; out[2] = nu; out[3] = over ? 1u : 0u;
}
// item j * pp + k = pair p0 + k of message list[j]
__global__ void bl_pair_init_kernel(const uint32_t *list, uint32_t nu, uint32_t pp, PairInit pi, St v, uint32_t *msg_of, const uint32_t *tapes, size_t tape_stride,
                                    const uint32_t *tape_lens, uint32_t *siglens, uint32_t *trials, const uint32_t *nud)
{
    if (nud) nu = min(nu, *nud);
    const uint32_t vid = blockIdx.x * blockDim.x + threadIdx.x;
    if (vid >= nu * pp) return;
    const uint32_t m = list[vid / pp], k = vid % pp;
    const uint32_t len = tape_lens ? tape_lens[m] : (uint32_t)tape_stride;
    msg_of[vid] = m;
    v.pos[vid] = pi.pos[k]; v.var_bits[vid] = pi.vb[k]; v.iter[vid] = 0; v.cur[vid] = 0; siglens[vid] = 0; trials[vid] = 0;
    uint32_t buf = 0, status = 0;
    if (pi.vb[k] > 0) { if (pi.lrp[k] < len) buf = tapes[(size_t)m * tape_stride + pi.lrp[k]] >> (32 - pi.vb[k]); else status = 2; }   // (the tape ended in an earlier pair)
    v.var_buf[vid] = buf; v.status[vid] = status;
}
// one wavefront per message of the list: the first of its pairs that did not fail twice decides it
__global__ __launch_bounds__(64) void bl_resolve_kernel(const uint32_t *msg_of, uint32_t pp, uint32_t p0, const uint32_t *vstatus, const uint8_t *vsigs, const uint32_t *vsiglens,
                                                        const uint32_t *vtrials, size_t sig_stride, uint32_t *status, uint8_t *sigs, uint32_t *siglens, uint32_t *trials,
                                                        const uint32_t *nud)
{
    if (nud && blockIdx.x >= *nud) return;
    const uint32_t j = blockIdx.x, m = msg_of[j * pp];
    uint32_t k = 0;
    while (k < pp && vstatus[j * pp + k] == 7) k++;
    if (k == pp) return;                                              // every pair failed twice: the message stays at status 7 and gets the next pairs
    const uint32_t vid = j * pp + k, st = vstatus[vid];
    if (st == 1) {
        const uint32_t len = vsiglens[vid];
        for (uint32_t i = threadIdx.x; i < len; i += 64) sigs[(size_t)m * sig_stride + i] = vsigs[(size_t)vid * sig_stride + i];
        if (threadIdx.x == 0) { siglens[m] = len; trials[m] = 2 * (p0 + k) + vtrials[vid]; }
    } else if (threadIdx.x == 0) siglens[m] = 0;
    if (threadIdx.x == 0) status[m] = st;
}
// self-test (sc_hip_bliss_set_speculation(h, 2)): every third message that goes on to its later pairs is declared off its precomputed tape positions
__global__ void bl_force_redo_kernel(uint32_t *status, uint32_t count)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < count && status[i] == 7 && i % 3 == 0) status[i] = 8;
}
__global__ void bl_reset_kernel(const uint32_t *list, uint32_t count, St s)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    const uint32_t m = list[i];
    s.pos[m] = 0; s.var_buf[m] = 0; s.var_bits[m] = 0; s.iter[m] = 0; s.cur[m] = 0; s.status[m] = 0;
}

static void bl_free_ws2(sc_hip_bliss *b)
{
    void *ptrs[] = { b->st2.pos, b->st2.var_buf, b->st2.iter, b->st2.cur, b->st2.status, b->st2.var_bits, b->tu2, b->sigs2, b->siglens2, b->trials2, b->msg_of2 };
    for (void *p : ptrs) if (p) (void)sc::wipe_free(p);
    memset(&b->st2, 0, sizeof b->st2);
    b->tu2 = nullptr; b->sigs2 = nullptr; b->siglens2 = b->trials2 = b->msg_of2 = nullptr; b->cap2 = 0; b->stride2 = 0;
}
static int bl_ensure_ws2(sc_hip_bliss *b, size_t items, size_t sig_stride)
{
    if (items <= b->cap2 && sig_stride <= b->stride2) return SC_HIP_SUCCESS;
    if (items < b->cap2) items = b->cap2;
    if (sig_stride < b->stride2) sig_stride = b->stride2;
    bl_free_ws2(b);
    const size_t n = b->P.n;
#define WS(ptr, bytes) HIPCHK(hipMalloc((void **)&(ptr), (bytes)))
    WS(b->st2.pos, items * 4); WS(b->st2.var_buf, items * 4); WS(b->st2.iter, items * 4); WS(b->st2.cur, items * 4); WS(b->st2.status, items * 4); WS(b->st2.var_bits, items * 4);
    WS(b->tu2, items * 2 * n * 4); WS(b->sigs2, items * sig_stride); WS(b->siglens2, items * 4); WS(b->trials2, items * 4); WS(b->msg_of2, items * 4);
#undef WS
    b->cap2 = items; b->stride2 = sig_stride;
    return SC_HIP_SUCCESS;
}

// the Gaussian vectors t, u (with the signature they give s), their products and the unpacked candidates of a call
static int bl_wipe_vectors(sc_hip_bliss_t *b, size_t batch, hipStream_t s)
{
    const size_t n = (size_t)b->P.n;
    return bl_zero(b, { { b->tu, batch * 2 * n * 4 }, { b->tin, batch * n * 4 }, { b->v, batch * n * 4 }, { b->z, batch * n * 2 } }, s);
}

// the later pairs of a seeded call reach at most this many attempts on the handle's own tape rows (a message needs more with probability ~0.12^13; it ends with status 2)
constexpr unsigned BL_SEED_ATTEMPTS = 2 + 2 * 4 * 3;

static int bl_sign_core(sc_hip_bliss_t *b, const int16_t *kf, const int16_t *kg, const int16_t *ka16, size_t batch, const uint8_t *d_msgs, size_t msg_stride,
                        const uint32_t *d_mlens, uint32_t fixed_len, const uint32_t *d_tapes, size_t tape_stride, const uint32_t *d_tape_lens,
                        uint8_t *d_sigs, size_t sig_stride, uint32_t *d_siglens, uint32_t *d_trials, uint32_t *d_status, void *stream,
                        const uint8_t *d_seeds, size_t seed_stride, size_t seed_len)
{
    if (batch == 0) return SC_HIP_SUCCESS;
    if (batch > 0x7fffffffu / 2048) return fail("bliss_sign: batch too large");
    const bool seeded = d_seeds != nullptr;
    if (seeded) {
        // Seeded call: the tape rows are the handle's own, filled from the seeds as far as the passes below consume them - the first pair of attempts of every
        // message, then the later pairs of the messages that get that far (a tape position does not depend on any value, see ResSim) - and zeroed behind the call.
        if (seed_len == 0 || seed_len > seed_stride || seed_len > 0xffffffffu) return fail("bliss_sign_seeded: bad seed length / stride");
        HIPCHK(hipSetDevice(b->ctx->device));
        const size_t words = (bl_tape_words(b->P.n, BL_SEED_ATTEMPTS, b->spw) + 3) & ~(size_t)3;
        if (batch > b->seed_rows || words != b->seed_words) {
            if (b->seed_tape) (void)sc::wipe_free(b->seed_tape);
            if (b->seed_list) (void)hipFree(b->seed_list);
            b->seed_tape = b->seed_list = nullptr; b->seed_rows = 0;
            HIPCHK(hipMalloc((void **)&b->seed_tape, batch * words * 4));
            HIPCHK(hipMalloc((void **)&b->seed_list, batch * 4));
            b->seed_rows = batch; b->seed_words = words;
        }
        d_tapes = b->seed_tape; tape_stride = words; d_tape_lens = nullptr;
    }
    if (!d_msgs || !d_tapes || !d_sigs || !d_siglens) return fail("bliss_sign: null buffer");
    if (!d_mlens && fixed_len == 0) return fail("bliss_sign: empty messages are rejected by the reference's oracle (bliss_b.c:571-574)");
    if (sig_stride < sc_hip_bliss_sig_bytes(b)) return fail("bliss_sign: sig_stride smaller than sc_hip_bliss_sig_bytes()");
    if (tape_stride > 0xffffffffu) return fail("bliss_sign: tape too long");
    HIPCHK(hipSetDevice(b->ctx->device));
    if (bl_ensure_ws(b, batch)) return SC_HIP_FAILURE;
    hipStream_t s = (hipStream_t)stream;
    const BlissP &P = b->P;
    const uint32_t B = (uint32_t)batch;
    St st = b->st;
    st.msg_of = nullptr;
    uint32_t *trials = d_trials ? d_trials : b->trials;
    if (bl_zero(b, { { st.pos, batch * 4 }, { st.var_buf, batch * 4 }, { st.var_bits, batch * 4 }, { st.iter, batch * 4 }, { st.cur, batch * 4 }, { st.status, batch * 4 } }, s))
        return SC_HIP_FAILURE;
    HIPCHK(hipMemsetAsync(trials, 0, batch * 4, s));               // (the caller's array: any alignment)
    BlIn in{ kf, kg, ka16, d_msgs, msg_stride, d_mlens, fixed_len, d_tapes, tape_stride, d_tape_lens };
    in.seeds = d_seeds; in.seed_stride = seed_stride; in.seed_len = seed_len;
    const bool spec = b->speculate != 0;
    // seeded: blocks (16 bytes = 4 words) of every row filled so far, of the rows of the later passes (seed_list[0..seed_nu)), and whether some rows were filled whole
    size_t blk_all = 0, blk_later = 0; uint32_t seed_nu = 0; bool seed_whole = false;
    const uint32_t *seed_nud = nullptr;
    auto seed_fill = [&](const uint32_t *list, uint32_t count, const uint32_t *count_dev, size_t from, size_t to) -> int {
        return sc::drbg_fill(b->ctx, d_seeds, seed_stride, seed_len, list, count, count_dev, b->seed_tape, tape_stride * 4, from, to, 1, false, s);
    };
    if (seeded) {
        blk_all = spec ? (bl_tape_words(P.n, 2, b->spw) + 3) / 4 : tape_stride / 4;      // the plain loop's consumption is not known in advance: whole rows
        if (seed_fill(nullptr, B, nullptr, 0, blk_all)) return SC_HIP_FAILURE;
        blk_later = blk_all;
    }
    bool dirty = true;                                                // the handle's vector arrays hold this call's secrets
    bool status_out = false;                                          // the caller's status array holds the final statuses
    // every message: its first pair of attempts (all of them when the handle runs the plain loop)
    if (bl_rounds(b, BlRun{ st, b->tu, d_sigs, sig_stride, d_siglens, trials, B, nullptr, spec ? 2u : 0u }, in, s)) return SC_HIP_FAILURE;
    if (spec) {
        if (b->speculate == 2) hipLaunchKernelGGL(bl_force_redo_kernel, dim3((B + 255) / 256), dim3(256), 0, s, st.status, B);
        ResSim sim;
        sim.attempt(true, (uint32_t)P.n, b->spw); sim.attempt(false, (uint32_t)P.n, b->spw);   // pair 0 is behind us
        uint32_t *ulist = b->act[0];                                  // (the round loop is idle here: its two lists are free)
        uint32_t n8 = 0;
        ResSim sim_saved; size_t blk_saved = 0; bool was_blind = false, refused = false;
        for (uint32_t p0 = 1;; ) {
            if (p0 > 50000) return fail("bliss_sign: rejection loop did not terminate");
            if (!b->h_pin) HIPCHK(hipHostMalloc((void **)&b->h_pin, 64, hipHostMallocDefault));
            volatile uint32_t *both = b->h_pin;
            uint32_t *cnt7 = b->cnt + 8;                              // (b->cnt[0..1] belong to the round loop)
            HIPCHK(hipMemsetAsync(cnt7, 0, 8, s));
            hipLaunchKernelGGL(bl_collect_kernel, dim3((B + 255) / 256), dim3(256), 0, s, st.status, B, ulist, b->act[1], cnt7);
            // The first later pass is queued WITHOUT a look (a look idles the device for ~45 us and leaves the pass's first launches exposed one by one): one
            // message in eight goes on, so the pass is sized for a sixth of the batch (+ 64), the device checks the count against that (bl_blind_kernel) and every kernel of the
            // pass takes its count from the device; whoever is not served - beyond the bound, or failing all four pairs - is found by the look behind it.
            const bool blind = p0 == 1 && B >= 1024 && !refused;
            uint32_t nu, pp = BL_PAIRS;
            const uint32_t *nud = nullptr, *nvd = nullptr;
            if (blind) {
                nu = b->speculate == 3 ? 1u : B / 6 + 64;            // an eighth is expected (+- 0.3 % of the batch at 16384): ten standard deviations of room at 1024, more above; (3: self-test of the refusal path - a pass sized for one message)
                hipLaunchKernelGGL(bl_blind_kernel, dim3(1), dim3(1), 0, s, cnt7, nu, pp, b->cnt + 10);
                nud = b->cnt + 10; nvd = b->cnt + 11;
                sim_saved = sim; blk_saved = blk_later; was_blind = true;
            } else {
                if (p0 > 1 && dirty) {
                    // behind a pass of later pairs this look almost always finds nobody left: the call's closing wipes are queued in front of the wait, so
                    // the device zeroes while the host waits for the two counts (should a message be left, the arrays are simply used and zeroed again).
                    // They are queued in front of the read-back, too: by now the device has caught up with the host, every call here is exposed, and the
                    // device-to-host copy is the slowest of them to issue (~20 us each: the kernel trace showed two such gaps in a row - the counts and the
                    // refusal flag are one copy now)
                    if (bl_wipe_vectors(b, batch, s)) return SC_HIP_FAILURE;
                    dirty = false;
                    if (d_status && !b->entropy) { HIPCHK(hipMemcpyAsync(d_status, st.status, batch * 4, hipMemcpyDeviceToDevice, s)); status_out = true; }
                }
                b->h_pin[5] = 0;
                HIPCHK(hipMemcpyAsync(b->h_pin, cnt7, was_blind ? 24 : 8, hipMemcpyDeviceToHost, s));      // cnt[8..9]: the two counts; cnt[13]: the unlooked pass was refused
                HIPCHK(hipStreamSynchronize(s));
                if (was_blind && both[5]) {                              // the unlooked pass was refused (see bl_blind_kernel): nothing was served, start the later pairs over
                    sim = sim_saved; blk_later = blk_saved; p0 = 1; refused = true; seed_nu = 0; seed_nud = nullptr;
                }
                was_blind = false;
                nu = both[0];
                n8 = both[1];                                         // (status 8 stays: the last pass, the one that finds no 7, leaves the whole list in act[1])
                if (nu == 0) break;
                while (pp > 1 && (size_t)nu * pp > batch) pp >>= 1;   // the per-round arrays are sized for the batch
            }
            PairInit pi{};
            for (uint32_t k = 0; k < pp; k++) {
                pi.pos[k] = sim.pos; pi.lrp[k] = sim.lrp; pi.vb[k] = sim.vb;
                sim.attempt(true, (uint32_t)P.n, b->spw); sim.attempt(false, (uint32_t)P.n, b->spw);
            }
            const uint32_t nv = nu * pp;
            if (seeded) {
                // the words these pairs draw, for the messages that go on (every later pass works on a subset of the first one's list)
                if (p0 == 1) { HIPCHK(hipMemcpyAsync(b->seed_list, ulist, (size_t)nu * 4, hipMemcpyDeviceToDevice, s)); seed_nu = nu; seed_nud = blind ? b->cnt + 12 : nullptr; }
                size_t to = ((size_t)sim.pos + 3) / 4;
                if (to > tape_stride / 4) to = tape_stride / 4;
                if (to > blk_later) { if (seed_fill(ulist, nu, nud, blk_later, to)) return SC_HIP_FAILURE; blk_later = to; }
            }
            if (bl_ensure_ws2(b, nv, sig_stride)) return SC_HIP_FAILURE;
            St v = b->st2;
            hipLaunchKernelGGL(bl_pair_init_kernel, dim3((nv + 255) / 256), dim3(256), 0, s, ulist, nu, pp, pi, v, b->msg_of2, d_tapes, tape_stride, d_tape_lens, b->siglens2, b->trials2, nud);
            HIPCHK(hipGetLastError());
            v.msg_of = b->msg_of2;
            BlRun pass{ v, b->tu2, b->sigs2, sig_stride, b->siglens2, b->trials2, nv, nullptr, 2u };
            pass.items_dev = nvd;
            if (bl_rounds(b, pass, in, s)) return SC_HIP_FAILURE;
            dirty = true; status_out = false;
            hipLaunchKernelGGL(bl_resolve_kernel, dim3(nu), dim3(64), 0, s, b->msg_of2, pp, p0, v.status, b->sigs2, b->siglens2, b->trials2, sig_stride, st.status, d_sigs, d_siglens, trials, nud);
            HIPCHK(hipGetLastError());
            // the later pairs' candidate signatures - valid signatures under fresh randomness, all but one per message discarded - their vectors and their
            // bookkeeping do not outlive the call either: zeroed as far as this pass used them (a pass queued without a look: as far as it was sized for)
            if (bl_zero(b, { { b->sigs2, (size_t)nv * sig_stride }, { b->siglens2, (size_t)nv * 4 }, { b->trials2, (size_t)nv * 4 }, { b->tu2, (size_t)nv * 2 * P.n * 4 } }, s))
                return SC_HIP_FAILURE;
            p0 += pp;
        }
        // messages an oracle failure threw off the precomputed tape positions: from the start, in order
        if (n8) {
            uint32_t *redo = b->act[1];                                // (the round loop's first round reads this list and writes act[0]; later rounds alternate)
            hipLaunchKernelGGL(bl_reset_kernel, dim3((n8 + 255) / 256), dim3(256), 0, s, redo, n8, st);
            HIPCHK(hipGetLastError());
            if (seeded) { if (seed_fill(redo, n8, nullptr, 0, tape_stride / 4)) return SC_HIP_FAILURE; seed_whole = true; }
            if (bl_rounds(b, BlRun{ st, b->tu, d_sigs, sig_stride, d_siglens, trials, n8, redo, 0u }, in, s)) return SC_HIP_FAILURE;
            dirty = true; status_out = false;
        }
    }
    if (b->entropy == 2) {
        // the arithmetic coder: coded rows are built in the rows of b->v (n words each - zeroed with the other vectors below) and moved over the raw ones
        const uint32_t rawbytes = (uint32_t)sc_hip_bliss_sig_bytes(b);
        const size_t tmp_stride = (size_t)P.n * 4;
        const uint32_t cap = (uint32_t)(sig_stride < tmp_stride ? sig_stride : tmp_stride);
        uint32_t lpw = 64;
        while (lpw > 1 && (size_t)(B + lpw - 1) / lpw < (size_t)b->ctx->cus * 4) lpw >>= 1;
        const BacDist bd{ b->d_bac, b->d_bac + ((size_t)1 << P.z1_bits) };
        hipLaunchKernelGGL(bl_bac_pack_kernel, dim3((B + lpw - 1) / lpw), dim3(64), 0, s, d_sigs, sig_stride, d_siglens, st.status, B, lpw, P, bd, reinterpret_cast<uint8_t *>(b->v), tmp_stride, cap);
        hipLaunchKernelGGL(bl_bac_move_kernel, dim3((B + 3) / 4), dim3(256), 0, s, d_sigs, sig_stride, d_siglens, st.status, B, reinterpret_cast<uint8_t *>(b->v), tmp_stride, rawbytes, cap);
        HIPCHK(hipGetLastError());
        dirty = true;
    } else if (b->entropy) {
        const uint32_t rawbytes = (uint32_t)sc_hip_bliss_sig_bytes(b), raww = (rawbytes + 3) / 4 + 1;
        uint32_t capw = (uint32_t)((sig_stride + 3) / 4);
        if (capw > 2 * raww) capw = 2 * raww;                       // a coded signature twice the raw size is refused (status 6) rather than given more LDS
        hipLaunchKernelGGL(bl_huff_pack_kernel, dim3((B + 3) / 4), dim3(256), sizeof(uint32_t) * 4 * (size_t)(raww + capw), s, d_sigs, sig_stride, d_siglens, st.status, B, P,
                           b->ht[0], b->ht[1], capw);
        HIPCHK(hipGetLastError());
    }
    if (d_status && !status_out) HIPCHK(hipMemcpyAsync(d_status, st.status, batch * 4, hipMemcpyDeviceToDevice, s));
    // the Gaussian vectors t, u (with the signature they give s), their products and the unpacked candidates do not outlive the call
    // (zeroed as far as this batch used them: the arrays are sized for the largest batch the handle has seen)
    if (dirty && bl_wipe_vectors(b, batch, s)) return SC_HIP_FAILURE;
    if (seeded) {
        // the expanded CSPRNG words are per-call secrets like the vectors: zeroed as far as they were written
        if (seed_whole) HIPCHK(hipMemsetAsync(b->seed_tape, 0, batch * tape_stride * 4, s));
        else {
            if (sc::drbg_fill(b->ctx, d_seeds, seed_stride, seed_len, nullptr, B, nullptr, b->seed_tape, tape_stride * 4, 0, blk_all, 1, true, s)) return SC_HIP_FAILURE;
            if (seed_nu && blk_later > blk_all &&
                sc::drbg_fill(b->ctx, d_seeds, seed_stride, seed_len, b->seed_list, seed_nu, seed_nud, b->seed_tape, tape_stride * 4, blk_all, blk_later, 1, true, s)) return SC_HIP_FAILURE;
        }
    }
    return SC_HIP_SUCCESS;
}

static int bl_verify_core(sc_hip_bliss_t *b, const int16_t *ka16, size_t batch, const uint8_t *d_msgs, size_t msg_stride, const uint32_t *d_mlens, uint32_t fixed_len,
                          const uint8_t *d_sigs, size_t sig_stride, const uint32_t *d_siglens, uint32_t *d_verdicts, void *stream);

extern "C" int sc_hip_bliss_verify(sc_hip_bliss_t *b, size_t batch, const uint8_t *d_msgs, size_t msg_stride, const uint32_t *d_mlens, uint32_t fixed_len,
                                   const uint8_t *d_sigs, size_t sig_stride, const uint32_t *d_siglens, uint32_t *d_verdicts, void *stream)
{
    if (!b) return fail("bliss_verify: null handle");
    if (!b->have_key) return fail("bliss_verify: no public key loaded");
    return bl_verify_core(b, nullptr, batch, d_msgs, msg_stride, d_mlens, fixed_len, d_sigs, sig_stride, d_siglens, d_verdicts, stream);
}

// replaces: safecrypto_verify -> bliss_b_verify for a batch of signatures EACH under its own public key (d_a: SINT16 [batch][n], NTT domain)
extern "C" int sc_hip_bliss_verify_keys(sc_hip_bliss_t *b, size_t batch, const int16_t *d_a, const uint8_t *d_msgs, size_t msg_stride, const uint32_t *d_mlens, uint32_t fixed_len,
                                        const uint8_t *d_sigs, size_t sig_stride, const uint32_t *d_siglens, uint32_t *d_verdicts, void *stream)
{
    if (!b) return fail("bliss_verify_keys: null handle");
    if (!d_a) return fail("bliss_verify_keys: null key buffer");
    return bl_verify_core(b, d_a, batch, d_msgs, msg_stride, d_mlens, fixed_len, d_sigs, sig_stride, d_siglens, d_verdicts, stream);
}

__global__ void bl_widen_kernel(const int16_t *in, int32_t *out, size_t len)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < len) out[i] = (int32_t)in[i];
}

static int bl_verify_core(sc_hip_bliss_t *b, const int16_t *ka16, size_t batch, const uint8_t *d_msgs, size_t msg_stride, const uint32_t *d_mlens, uint32_t fixed_len,
                          const uint8_t *d_sigs, size_t sig_stride, const uint32_t *d_siglens, uint32_t *d_verdicts, void *stream)
{
    if (batch == 0) return SC_HIP_SUCCESS;
    if (batch > 0x7fffffffu / 2048) return fail("bliss_verify: batch too large");
    if (!d_msgs || !d_sigs || !d_siglens || !d_verdicts) return fail("bliss_verify: null buffer");
    if (!d_mlens && fixed_len == 0) return fail("bliss_verify: empty messages are rejected by the reference's oracle");
    if (sig_stride < sc_hip_bliss_sig_bytes(b)) return fail("bliss_verify: sig_stride smaller than sc_hip_bliss_sig_bytes()");
    HIPCHK(hipSetDevice(b->ctx->device));
    if (bl_ensure_ws(b, batch)) return SC_HIP_FAILURE;
    hipStream_t s = (hipStream_t)stream;
    const BlissP &P = b->P;
    const uint32_t B = (uint32_t)batch;
    const int n = P.n;
    HIPCHK(hipMemsetAsync(d_verdicts, 0, batch * 4, s));
    const size_t ne = batch * (size_t)n;
    if (b->entropy == 2) {
        uint32_t lpw = 64;
        while (lpw > 1 && (size_t)(B + lpw - 1) / lpw < (size_t)b->ctx->cus * 4) lpw >>= 1;
        const BacDist bd{ b->d_bac, b->d_bac + ((size_t)1 << P.z1_bits) };
        hipLaunchKernelGGL(bl_bac_unpack_kernel, dim3((B + lpw - 1) / lpw), dim3(64), 0, s, d_sigs, sig_stride, d_siglens, B, lpw, b->tin, b->z, b->cidx, d_verdicts, P, bd);
    } else if (b->entropy) {
        uint32_t lpw = 64;                                            // signatures per wavefront: every SIMD two wavefronts where the batch allows
        while (lpw > 1 && (size_t)(B + lpw - 1) / lpw < (size_t)b->ctx->cus * 8) lpw >>= 1;
        hipLaunchKernelGGL(bl_huff_unpack_kernel, dim3((B + lpw - 1) / lpw), dim3(64), 0, s, d_sigs, sig_stride, d_siglens, B, lpw, b->tin, b->z, b->cidx, d_verdicts, P, b->ht[0], b->ht[1]);
    }
    else hipLaunchKernelGGL(bl_unpack_kernel, dim3(B), dim3(256), (size_t)((P.n * (P.z1_bits + P.z2_bits) + P.kappa * P.n_bits + 7) / 8 + 16 + 3) & ~(size_t)3, s, d_sigs, sig_stride, d_siglens, B, b->tin, b->z, b->cidx, d_verdicts, P);
    hipLaunchKernelGGL(bl_vnorm_kernel, dim3((B + 3) / 4), dim3(256), 0, s, b->tin, b->z, B, d_verdicts, P);
    HIPCHK(hipGetLastError());
    if (ka16) {
        hipLaunchKernelGGL(bl_widen_kernel, dim3((unsigned)((ne + 255) / 256)), dim3(256), 0, s, ka16, b->ain, ne);
        HIPCHK(hipGetLastError());
        if (sc_hip_polymul_ntt(b->plan, batch, b->tin, b->ain, b->v, s)) return SC_HIP_FAILURE;
    } else if (sc::polymul_ntt_bcast(b->plan, batch, b->tin, b->d_a, b->v, s)) return SC_HIP_FAILURE;
    if (n == 512) hipLaunchKernelGGL(bl_vfix_kernel<8>, dim3((B + 3) / 4), dim3(256), 0, s, b->v, b->cidx, b->z, B, P);
    else if (n == 256) hipLaunchKernelGGL(bl_vfix_kernel<4>, dim3((B + 3) / 4), dim3(256), 0, s, b->v, b->cidx, b->z, B, P);
    else if (n == 1024) hipLaunchKernelGGL(bl_vfix_kernel<16>, dim3((B + 3) / 4), dim3(256), 0, s, b->v, b->cidx, b->z, B, P);
    else {
        hipLaunchKernelGGL(bl_vparity_kernel, dim3((unsigned)((ne + 255) / 256)), dim3(256), 0, s, b->v, ne, P.q);
        hipLaunchKernelGGL(bl_vcidx_kernel, dim3((B + 255) / 256), dim3(256), 0, s, b->v, b->cidx, B, P);
        hipLaunchKernelGGL(bl_vz_kernel, dim3((unsigned)((ne + 255) / 256)), dim3(256), 0, s, b->v, b->z, ne, P);
    }
    if ((size_t)B <= (size_t)b->ctx->cus * 4 * 64)
        hipLaunchKernelGGL(bl_oracle2_kernel, dim3((2 * B + HT - 1) / HT), dim3(HT), 0, s, (const uint32_t *)nullptr, B, (const uint32_t *)nullptr, d_msgs, msg_stride, d_mlens, fixed_len,
                           b->z, b->cidx, (uint32_t *)nullptr, 1, d_verdicts, P);
    else
        hipLaunchKernelGGL(bl_oracle1_kernel, dim3((B + HT - 1) / HT), dim3(HT), 0, s, (const uint32_t *)nullptr, B, (const uint32_t *)nullptr, d_msgs, msg_stride, d_mlens, fixed_len,
                           b->z, b->cidx, (uint32_t *)nullptr, 1, d_verdicts, P);
    HIPCHK(hipGetLastError());
    return SC_HIP_SUCCESS;
}

// replaces: safecrypto_keygen -> bliss_b_keygen (bliss_b.c:1124-1264) for `batch` independent key pairs; key pair i draws from tape i
// (the prng_32 word stream of sc->prng_ctx[0]; sc_hip_drbg_words makes it from a seed).  nz0, nz1 = bliss_set_t::nz.  Outputs
// [batch][n] SINT16 as the reference stores them: f, g (= 2g + 1), a (NTT domain).  d_status[i]: 1 generated, 2 tape exhausted.
extern "C" int sc_hip_bliss_keygen(sc_hip_bliss_t *b, size_t batch, int nz0, int nz1, const uint32_t *d_tapes, size_t tape_stride, const uint32_t *d_tape_lens,
                                   int16_t *d_f, int16_t *d_g, int16_t *d_a, uint32_t *d_attempts, uint32_t *d_status, void *stream)
{
    if (!b) return fail("bliss_keygen: null handle");
    if (batch == 0) return SC_HIP_SUCCESS;
    if (batch > 0x7fffffffu / 2048) return fail("bliss_keygen: batch too large");
    if (!d_tapes || !d_f || !d_g || !d_a || !d_status) return fail("bliss_keygen: null buffer");
    const int n = b->P.n;
    if (nz0 < 0 || nz1 < 0 || nz0 + nz1 > n / 2) return fail("bliss_keygen: nz out of range");
    HIPCHK(hipSetDevice(b->ctx->device));
    hipStream_t s = (hipStream_t)stream;
    const uint32_t B = (uint32_t)batch;
    const size_t pn = batch * (size_t)n * 4;
    DevBuf th, tg, tf, uh, num, pos, need, rc, act0, act1, cnt, att;
    if (th.get(pn) || tg.get(pn) || tf.get(pn) || uh.get(pn) || num.get(pn) || pos.get(batch * 4) || need.get(batch * 4) || rc.get(batch * 4) ||
        act0.get(batch * 4) || act1.get(batch * 4) || cnt.get(64) || att.get(batch * 4)) return SC_HIP_FAILURE;
    uint32_t *attempts = d_attempts ? d_attempts : att.as<uint32_t>();
    HIPCHK(hipMemsetAsync(pos.p, 0, batch * 4, s));
    HIPCHK(hipMemsetAsync(attempts, 0, batch * 4, s));
    HIPCHK(hipMemsetAsync(d_status, 0, batch * 4, s));
    uint32_t nact = B;
    const uint32_t *active = nullptr;
    for (uint32_t round = 0; nact > 0; round++) {
        if (round > 1000) return fail("bliss_keygen: the search for an invertible f did not terminate");
        uint32_t *next = (round & 1) ? act1.as<uint32_t>() : act0.as<uint32_t>();
        HIPCHK(hipMemsetAsync(cnt.p, 0, 4, s));
        hipLaunchKernelGGL(bl_kg_sample_kernel, dim3((nact + HT - 1) / HT), dim3(HT), (size_t)n * HT, s, active, nact, round == 0 ? 1 : 0, d_tapes, tape_stride, d_tape_lens,
                           pos.as<uint32_t>(), d_status, need.as<uint32_t>(), d_g, tg.as<int32_t>(), tf.as<int32_t>(), n, nz0, nz1);
        HIPCHK(hipGetLastError());
        if (round == 0 && sc_hip_ntt_fwd(b->plan, nact, tg.as<int32_t>(), th.as<int32_t>(), s)) return SC_HIP_FAILURE;       // round 0: dense index == key index
        if (sc_hip_ntt_fwd(b->plan, nact, tf.as<int32_t>(), uh.as<int32_t>(), s)) return SC_HIP_FAILURE;
        const size_t ne = (size_t)nact * n;
        hipLaunchKernelGGL(bl_kg_gather_kernel, dim3((unsigned)((ne + 255) / 256)), dim3(256), 0, s, active, nact, th.as<int32_t>(), num.as<int32_t>(), n);
        HIPCHK(hipGetLastError());
        if (sc_hip_ntt_div(b->plan, nact, (size_t)n, num.as<int32_t>(), uh.as<int32_t>(), rc.as<int32_t>(), s)) return SC_HIP_FAILURE;
        hipLaunchKernelGGL(bl_kg_finish_kernel, dim3((nact + 3) / 4), dim3(256), 0, s, active, nact, need.as<uint32_t>(), rc.as<int32_t>(), num.as<int32_t>(), tf.as<int32_t>(),
                           d_f, d_a, d_status, attempts, next, cnt.as<uint32_t>(), n, b->P.q);
        HIPCHK(hipGetLastError());
        uint32_t c = 0;
        HIPCHK(hipMemcpyAsync(&c, cnt.p, 4, hipMemcpyDeviceToHost, s));
        HIPCHK(hipStreamSynchronize(s));
        nact = c;
        active = next;
    }
    return SC_HIP_SUCCESS;
}

// ---- host-buffer convenience (the shape safecrypto_sign / safecrypto_verify have, for a batch; sc::sign_host in engine_internal.h) ----

extern "C" int sc_hip_bliss_sign_host(sc_hip_bliss_t *b, size_t batch, const uint8_t *h_msgs, size_t msg_stride, const uint32_t *h_mlens, uint32_t fixed_len,
                                      const uint32_t *h_tapes, size_t tape_stride, const uint32_t *h_tape_lens,
                                      uint8_t *h_sigs, size_t sig_stride, uint32_t *h_siglens, uint32_t *h_trials, uint32_t *h_status)
{
    if (!b) return fail("bliss_sign_host: null handle");
    return sc::sign_host(b->ctx->device, batch, h_msgs, msg_stride, h_mlens, h_tapes, tape_stride, h_tape_lens, h_sigs, sig_stride, h_siglens, h_trials, h_status,
        [&](const uint8_t *m, const uint32_t *ml, const uint32_t *t, const uint32_t *tl, uint8_t *sg, uint32_t *sl, uint32_t *tr, uint32_t *st) {
            return sc_hip_bliss_sign(b, batch, m, msg_stride, ml, fixed_len, t, tape_stride, tl, sg, sig_stride, sl, tr, st, nullptr);
        });
}

extern "C" int sc_hip_bliss_verify_host(sc_hip_bliss_t *b, size_t batch, const uint8_t *h_msgs, size_t msg_stride, const uint32_t *h_mlens, uint32_t fixed_len,
                                        const uint8_t *h_sigs, size_t sig_stride, const uint32_t *h_siglens, uint32_t *h_verdicts)
{
    if (!b) return fail("bliss_verify_host: null handle");
    return sc::verify_host(b->ctx->device, batch, h_msgs, msg_stride, h_mlens, h_sigs, sig_stride, h_siglens, h_verdicts,
        [&](const uint8_t *m, const uint32_t *ml, const uint8_t *sg, const uint32_t *sl, uint32_t *v) {
            return sc_hip_bliss_verify(b, batch, m, msg_stride, ml, fixed_len, sg, sig_stride, sl, v, nullptr);
        });
}
