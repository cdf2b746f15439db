// huffman_dev.h - SC_ENTROPY_HUFFMAN_STATIC on the device: the pieces the signature coders share (bliss.hip: t and z of a BLISS-B signature; entropy.hip:
// rows of one polynomial + raw tail for Falcon, the ENS / DLP signatures and ring-TESLA).
// Reference: entropy_poly_encode_32 / _16 with SIGNED_COEFF (src/utils/entropy/entropy.c:70-112, 147-190) -> encode_huffman_signed_32
// (entropy_huffman.c:22-57): per coefficient the beta = max(bits - 7, 0) low bits of |v| raw, |v| >> beta as a codeword of a static prefix code
// (huffman.c:386-403), a sign bit unless v = 0; one MSB-first bit stream flushed to whole bytes (packer.c:299-351).  The code tables are the reference's
// data (src/utils/entropy/huffman.h) and come in through the C ABI.
#pragma once
#include "engine_internal.h"
#include <algorithm>
#include <cmath>
#include <vector>

namespace sc {

struct HuffT {
    const uint32_t *code, *len;      // per symbol: codeword (right-aligned) and its length
    const uint32_t *dleft, *dls;     // decoding order (shortest first): codeword left-aligned in 32 bits; length | symbol << 8
    const uint32_t *lut;             // indexed by the next lutbits bits of the stream: length | symbol << 8 of the codeword that starts there, 0 = longer than lutbits
    uint32_t depth, lutbits;
};
constexpr int HUFF_LUT_BITS = 10;

__device__ __forceinline__ uint32_t wave_incl_scan(uint32_t v, int lane)
{
    for (int o = 1; o < 64; o <<= 1) { const uint32_t u = __shfl_up(v, o); if (lane >= o) v += u; }
    return v;
}
// `len` (< 64) bits of tok, MSB first, at bit offset off of a big-endian word stream; words at or past capw are dropped (the length check catches them)
__device__ __forceinline__ void lds_put_bits(uint32_t *out, uint32_t capw, uint32_t off, uint64_t tok, uint32_t len)
{
    if (len == 0) return;
    const uint64_t T = tok << (64 - len);
    const uint32_t w = off >> 5, sh = off & 31;
    const uint64_t U = T >> sh;
    const uint32_t a = (uint32_t)(U >> 32), b = (uint32_t)U, c = sh ? ((uint32_t)T << (32 - sh)) : 0u;
    if (a && w < capw) atomicOr(&out[w], a);
    if (b && w + 1 < capw) atomicOr(&out[w + 1], b);
    if (c && w + 2 < capw) atomicOr(&out[w + 2], c);
}
__device__ __forceinline__ uint32_t lds_get_bits(const uint32_t *raw, uint32_t off, int bits)      // bits <= 32; raw has one spare word behind the data
{
    const uint32_t w = off >> 5, sh = off & 31;
    const uint64_t two = ((uint64_t)raw[w] << 32) | raw[w + 1];
    return (uint32_t)(two >> (64 - sh - bits)) & (bits == 32 ? 0xFFFFFFFFu : ((1u << bits) - 1u));
}


// MSB-first reader over one signature row, one per lane
struct BitR {
    const uint8_t *p; uint32_t nbits, row_bytes, pos, fetched; uint64_t acc; int have; bool over;
    __device__ __forceinline__ uint32_t peek32()
    {
        if (have <= 32) {
            uint32_t w = 0;
            if (fetched + 4 <= row_bytes && ((uintptr_t)(p + fetched) & 3) == 0) w = __builtin_bswap32(*reinterpret_cast<const uint32_t *>(p + fetched));   // (bytes past the signature's end, if read, are never consumed unnoticed: take() flags it)
            else for (int b = 0; b < 4; b++) { const uint32_t i = fetched + b; w = (w << 8) | (i < row_bytes ? p[i] : 0u); }
            fetched += 4;
            acc |= (uint64_t)w << (32 - have);
            have += 32;
        }
        return (uint32_t)(acc >> 32);
    }
    __device__ __forceinline__ void take(int k) { acc <<= k; have -= k; pos += (uint32_t)k; if (pos > nbits) over = true; }
    __device__ __forceinline__ uint32_t get(int k) { if (k == 0) return 0; const uint32_t v = peek32() >> (32 - k); take(k); return v; }
};
__device__ __forceinline__ int32_t huff_get(BitR &r, int bits, const uint32_t *lut, uint32_t lutbits, const uint32_t *dleft, const uint32_t *dls, uint32_t depth)
{
    const int beta = bits > 7 ? bits - 7 : 0;                               // (beta + the lookup's index + the sign bit fit one 32-bit look-ahead: beta <= 17)
    const uint32_t pk = r.peek32();
    uint32_t a = beta ? pk >> (32 - beta) : 0u;
    const uint32_t rest = pk << beta;
    uint32_t e = lut[rest >> (32 - lutbits)];
    if (e) {                                                                 // raw bits, codeword and sign bit out of one look-ahead
        const uint32_t len = e & 0xFFu;
        a |= (e >> 8) << beta;
        const uint32_t neg = (rest << len) >> 31;
        r.take(beta + (int)len + (a != 0));
        return (a != 0 && neg) ? -(int32_t)a : (int32_t)a;
    }
    r.take(beta);                                                            // a codeword longer than the table's index: walk the candidates, shortest first
    const uint32_t pk2 = r.peek32();
    uint32_t i = 0;
    for (; i + 1 < depth; i++) if (((pk2 ^ dleft[i]) >> (32 - (dls[i] & 0xFFu))) == 0) break;     // a complete prefix code: the last candidate needs no test
    e = dls[i];
    r.take((int)(e & 0xFFu));
    a |= (e >> 8) << beta;
    return (a != 0 && r.get(1)) ? -(int32_t)a : (int32_t)a;
}


// Host side: checks that (codes, bits) is a complete prefix code and appends the device image of the table to h:
// code[depth] | len[depth] | dleft[depth] | dls[depth] | lut[1 << lutbits]   (HuffT's five arrays, in this order).  field_bits: width of the raw field.
inline int huff_build_table(const uint32_t *codes, const uint16_t *bits, uint32_t depth, int field_bits, std::vector<uint32_t> *h, uint32_t *lutbits_out, const char *who)
{
    const std::string w(who);
    if (depth < 2 || depth > 256) return fail(w + ": table depth must be 2..256");
    const int beta = field_bits > 7 ? field_bits - 7 : 0;
    // a prefix code, complete (every bit string decodes - the decoder's last candidate is taken untested), codewords of at most 32 bits, tokens below 64
    double kraft = 0.0;
    for (uint32_t i = 0; i < depth; i++) {
        const uint32_t L = bits[i];
        if (L < 1 || L > 32 || beta + (int)L + 1 >= 64 || (L < 32 && (codes[i] >> L))) return fail(w + ": bad codeword");
        kraft += ldexp(1.0, -(int)L);
        for (uint32_t j = 0; j < i; j++) {
            const uint32_t Lj = bits[j], m = L < Lj ? L : Lj;
            if ((codes[i] >> (L - m)) == (codes[j] >> (Lj - m))) return fail(w + ": not a prefix code");
        }
    }
    if (kraft != 1.0) return fail(w + ": the code is not complete");
    std::vector<uint32_t> order(depth);
    for (uint32_t i = 0; i < depth; i++) order[i] = i;
    std::stable_sort(order.begin(), order.end(), [&](uint32_t x, uint32_t y) { return bits[x] < bits[y]; });
    for (uint32_t i = 0; i < depth; i++) h->push_back(codes[i]);
    for (uint32_t i = 0; i < depth; i++) h->push_back(bits[i]);
    for (uint32_t i = 0; i < depth; i++) h->push_back(codes[order[i]] << (32 - bits[order[i]]));
    for (uint32_t i = 0; i < depth; i++) h->push_back((uint32_t)bits[order[i]] | (order[i] << 8));
    uint32_t maxlen = 0;
    for (uint32_t i = 0; i < depth; i++) maxlen = bits[i] > maxlen ? bits[i] : maxlen;
    const uint32_t lutbits = maxlen < (uint32_t)HUFF_LUT_BITS ? maxlen : (uint32_t)HUFF_LUT_BITS;
    for (uint32_t idx = 0; idx < (1u << lutbits); idx++) {
        uint32_t e = 0;
        for (uint32_t i = 0; i < depth && !e; i++)
            if (bits[i] <= lutbits && (idx >> (lutbits - bits[i])) == codes[i]) e = (uint32_t)bits[i] | (i << 8);
        h->push_back(e);
    }
    *lutbits_out = lutbits;
    return SC_HIP_SUCCESS;
}

}  // namespace sc
