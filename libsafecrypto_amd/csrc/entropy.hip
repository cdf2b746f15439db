// entropy.hip - SC_ENTROPY_HUFFMAN_STATIC for the signature schemes whose signature is ONE polynomial of fixed-width signed fields followed by raw 8-bit
// fields, as a format conversion of rows in device memory (include/sc_hip.h: sc_hip_huffman_*):
//     Falcon            s2: n fields of q_bits - 2 bits                         (src/schemes/sig/falcon/falcon.c:892-903, decode :984-993)
//     ENS / DLP sign.   s1: n fields of q_bits bits                             (src/schemes/sig/ens_dlp/ens_dlp_sig.c, entropy = sc->coding_signature.type :131-148)
//     ring-TESLA        z: n fields of q_bits bits, then the 64 digest bytes    (src/schemes/sig/ring_tesla/ring_tesla.c:887-890)
// The reference codes such a signature with entropy_poly_encode_32(packer, n, p, bits, SIGNED_COEFF, SC_ENTROPY_HUFFMAN_STATIC, ...) followed by raw
// entropy_poly_encode_8 fields in the SAME MSB-first bit stream (src/utils/entropy/entropy.c:70-112; encode_huffman_signed_32, entropy_huffman.c:22-57):
// per coefficient the beta = max(bits - 7, 0) low bits of |v|, the codeword of |v| >> beta, a sign bit unless v = 0.  The signers of csrc/ write raw
// rows and the verifiers read raw rows; these two passes sit behind / in front of them, as bl_huff_pack_kernel / bl_huff_unpack_kernel do for BLISS-B.
// (BLISS-B's signature is two polynomials and kappa raw indices and keeps its own pair in bliss.hip; the arithmetic coder SC_ENTROPY_BAC is not served.)
#include "engine_internal.h"
#include "huffman_dev.h"

#include <string.h>

using namespace sc;

struct sc_hip_huffman {
    sc_hip_ctx *ctx;
    uint32_t *d_tab;
    HuffT t;
    int field_bits;
};

namespace {

// one wavefront per row: the raw row into LDS, every lane makes the tokens of its coefficients, a scan of their lengths places them, LDS atomics assemble
// the stream, the coded row replaces the raw one.  status 6: a coefficient outside the code's alphabet (encode_huffman's SC_OUT_OF_BOUNDS) or a coded
// row longer than the row's pitch.
__global__ __launch_bounds__(256) void huff_rows_pack_kernel(uint8_t *rows, size_t stride, uint32_t *lens, uint32_t *status, uint32_t count, int n, int bits, int tail,
                                                            HuffT T, uint32_t raww, uint32_t capw)
{
    extern __shared__ uint32_t hlds[];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const uint32_t item = blockIdx.x * 4 + wave;
    if (item >= count || (status && status[item] != 1)) return;
    const uint32_t rawbits = (uint32_t)n * (uint32_t)bits + 8u * (uint32_t)tail, rawbytes = (rawbits + 7) >> 3;
    uint32_t *raw = hlds + (size_t)wave * (raww + capw), *out = raw + raww;
    uint8_t *row = rows + (size_t)item * stride;
    for (uint32_t w = lane; w < raww; w += 64) {
        uint32_t v = 0;
        for (int b = 0; b < 4; b++) { const uint32_t i = 4 * w + b; v = (v << 8) | (i < rawbytes ? row[i] : 0u); }
        raw[w] = v;
    }
    for (uint32_t w = lane; w < capw; w += 64) out[w] = 0;
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier();
    const int beta = bits > 7 ? bits - 7 : 0;
    uint32_t base = 0; bool bad = false;
    for (int c = 0; c < (n + 63) / 64; c++) {
        const int j = lane + 64 * c;
        uint64_t tok = 0; uint32_t len = 0;
        if (j < n) {
            const uint32_t f = lds_get_bits(raw, (uint32_t)(j * bits), bits);
            const int32_t v = (int32_t)(f << (32 - bits)) >> (32 - bits);
            const uint32_t a = (uint32_t)(v < 0 ? -v : v), sym = a >> beta, sb = v != 0;
            if (sym < T.depth) {
                const uint32_t cl = T.len[sym];
                tok = ((uint64_t)(a & ((1u << beta) - 1u)) << (cl + sb)) | ((uint64_t)T.code[sym] << sb) | (uint64_t)(v < 0);
                len = (uint32_t)beta + cl + sb;
            } else bad = true;
        }
        const uint32_t incl = wave_incl_scan(len, lane);
        lds_put_bits(out, capw, base + incl - len, tok, len);
        base += __shfl(incl, 63);
    }
    for (int c = 0; c < (tail + 63) / 64; c++) {                          // the raw 8-bit fields behind the polynomial
        const int j = lane + 64 * c;
        const uint32_t len = j < tail ? 8u : 0u;
        const uint64_t tok = j < tail ? lds_get_bits(raw, (uint32_t)(n * bits + 8 * j), 8) : 0u;
        const uint32_t incl = wave_incl_scan(len, lane);
        lds_put_bits(out, capw, base + incl - len, tok, len);
        base += __shfl(incl, 63);
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier();
    const uint32_t nbytes = (base + 7) >> 3;
    const bool fail_ = __any(bad) || nbytes > stride || (nbytes + 3) / 4 > capw;
    const uint32_t keep = fail_ ? 0u : nbytes, clear = rawbytes > keep ? rawbytes : keep;
    for (uint32_t i = lane; i < clear && i < stride; i += 64) row[i] = i < keep ? (uint8_t)(out[i >> 2] >> (24 - 8 * (i & 3))) : (uint8_t)0;
    if (lane == 0) { lens[item] = keep; if (fail_ && status) status[item] = 6; }
}

// A variable-length stream decodes serially: one lane per row (the first lpw lanes of a wavefront take one, as in bl_huff_unpack_kernel); a 10-bit lookup in
// LDS resolves all but the rare long codewords; the raw fields leave through a register bit-writer as whole bytes.
__global__ __launch_bounds__(64) void huff_rows_unpack_kernel(const uint8_t *rows, size_t stride, const uint32_t *lens, uint32_t count, uint32_t lpw, uint8_t *out, size_t out_stride,
                                                             uint32_t *outlens, uint32_t *bad, int n, int bits, int tail, HuffT T)
{
    __shared__ uint32_t tl[2][256];
    __shared__ uint32_t lut[1 << HUFF_LUT_BITS];
    for (uint32_t i = threadIdx.x; i < 256; i += 64) { tl[0][i] = i < T.depth ? T.dleft[i] : 0; tl[1][i] = i < T.depth ? T.dls[i] : 0; }
    for (uint32_t i = threadIdx.x; i < (1u << T.lutbits); i += 64) lut[i] = T.lut[i];
    __syncthreads();
    const uint32_t item = blockIdx.x * lpw + threadIdx.x;
    if (threadIdx.x >= lpw || item >= count) return;
    const uint32_t sl = lens[item];
    BitR r{ rows + (size_t)item * stride, (uint32_t)((sl < stride ? sl : stride) * 8), (uint32_t)stride, 0, 0, 0, 0, false };
    uint8_t *dst = out + (size_t)item * out_stride;
    uint64_t acc = 0; int nb = 0; uint32_t nbytes = 0;
    auto put = [&](uint32_t v, int w) {
        acc = (acc << w) | (uint64_t)(v & (w == 32 ? 0xFFFFFFFFu : ((1u << w) - 1u))); nb += w;
        while (nb >= 8) { dst[nbytes++] = (uint8_t)(acc >> (nb - 8)); nb -= 8; }
    };
    for (int j = 0; j < n; j++) put((uint32_t)huff_get(r, bits, lut, T.lutbits, tl[0], tl[1], T.depth), bits);
    for (int j = 0; j < tail; j++) put(r.get(8), 8);
    if (nb) { dst[nbytes++] = (uint8_t)(acc << (8 - nb)); nb = 0; }
    outlens[item] = nbytes;
    if (bad) bad[item] = r.over ? 1u : 0u;
}

}  // namespace

// codes[i] / bits[i]: codeword (right-aligned) and length of symbol i of the table the reference's coder would walk for fields of `field_bits` bits
// (entropy_huffman.c:27-31: huff_table_gaussian_<min(field_bits, 7) - 1>, _2 below 4 bits), depth = its number of symbols
extern "C" int sc_hip_huffman_create(sc_hip_ctx_t *ctx, const uint32_t *codes, const uint16_t *bits, uint32_t depth, int field_bits, sc_hip_huffman_t **out)
{
    if (!ctx || !codes || !bits || !out) return fail("huffman_create: null argument");
    if (field_bits < 2 || field_bits > 24) return fail("huffman_create: field width must be 2..24 bits");
    std::vector<uint32_t> h;
    uint32_t lutbits = 0;
    if (huff_build_table(codes, bits, depth, field_bits, &h, &lutbits, "huffman_create")) return SC_HIP_FAILURE;
    HIPCHK(hipSetDevice(ctx->device));
    sc_hip_huffman *t = new sc_hip_huffman;
    memset(t, 0, sizeof *t);
    t->ctx = ctx; t->field_bits = field_bits;
    if (hipMalloc((void **)&t->d_tab, h.size() * 4) != hipSuccess) { delete t; return fail("huffman_create: device allocation failed"); }
    if (sc::copy_in(t->d_tab, h.data(), h.size() * 4) != hipSuccess) { (void)hipFree(t->d_tab); delete t; return fail("huffman_create: table upload failed"); }
    const uint32_t *base = t->d_tab;
    t->t = HuffT{ base, base + depth, base + 2 * depth, base + 3 * depth, base + 4 * depth, depth, lutbits };
    *out = t;
    return SC_HIP_SUCCESS;
}

extern "C" void sc_hip_huffman_destroy(sc_hip_huffman_t *t)
{
    if (!t) return;
    if (t->d_tab) (void)hipFree(t->d_tab);
    delete t;
}

extern "C" int sc_hip_huffman_encode_rows(sc_hip_huffman_t *t, size_t batch, uint8_t *d_rows, size_t row_stride, uint32_t *d_lens, uint32_t *d_status,
                                          int n, int tail_bytes, void *stream)
{
    if (!t || !d_rows || !d_lens) return fail("huffman_encode_rows: null argument");
    if (batch == 0) return SC_HIP_SUCCESS;
    if (n < 1 || n > 4096 || tail_bytes < 0 || tail_bytes > 4096) return fail("huffman_encode_rows: n / tail out of range");
    const int bits = t->field_bits;
    const size_t rawbytes = ((size_t)n * bits + 8 * (size_t)tail_bytes + 7) / 8;
    if (row_stride < rawbytes) return fail("huffman_encode_rows: rows shorter than the raw signature");
    HIPCHK(hipSetDevice(t->ctx->device));
    const uint32_t raww = (uint32_t)((rawbytes + 3) / 4 + 1), capw = (uint32_t)((row_stride + 3) / 4);
    const size_t lds = sizeof(uint32_t) * 4 * (size_t)(raww + capw);
    if (lds > 64 * 1024) return fail("huffman_encode_rows: row too long for the LDS image");
    hipLaunchKernelGGL(huff_rows_pack_kernel, dim3((unsigned)((batch + 3) / 4)), dim3(256), lds, (hipStream_t)stream, d_rows, row_stride, d_lens, d_status, (uint32_t)batch,
                       n, bits, tail_bytes, t->t, raww, capw);
    HIPCHK(hipGetLastError());
    return SC_HIP_SUCCESS;
}

extern "C" int sc_hip_huffman_decode_rows(sc_hip_huffman_t *t, size_t batch, const uint8_t *d_rows, size_t row_stride, const uint32_t *d_lens,
                                          uint8_t *d_out, size_t out_stride, uint32_t *d_outlens, uint32_t *d_bad, int n, int tail_bytes, void *stream)
{
    if (!t || !d_rows || !d_lens || !d_out || !d_outlens) return fail("huffman_decode_rows: null argument");
    if (batch == 0) return SC_HIP_SUCCESS;
    if (n < 1 || n > 4096 || tail_bytes < 0 || tail_bytes > 4096) return fail("huffman_decode_rows: n / tail out of range");
    const int bits = t->field_bits;
    const size_t rawbytes = ((size_t)n * bits + 8 * (size_t)tail_bytes + 7) / 8;
    if (out_stride < rawbytes) return fail("huffman_decode_rows: output rows shorter than the raw signature");
    if (d_out == d_rows) return fail("huffman_decode_rows: the raw rows need a buffer of their own");
    HIPCHK(hipSetDevice(t->ctx->device));
    // a lane per row; only the first lpw lanes of a wavefront take one, so that small batches still put a wavefront on every SIMD (bl_huff_unpack_kernel)
    uint32_t lpw = 64;
    while (lpw > 8 && (batch + lpw - 1) / lpw < (size_t)t->ctx->cus * 8) lpw >>= 1;
    hipLaunchKernelGGL(huff_rows_unpack_kernel, dim3((unsigned)((batch + lpw - 1) / lpw)), dim3(64), 0, (hipStream_t)stream, d_rows, row_stride, d_lens, (uint32_t)batch, lpw,
                       d_out, out_stride, d_outlens, d_bad, n, bits, tail_bytes, t->t);
    HIPCHK(hipGetLastError());
    return SC_HIP_SUCCESS;
}
