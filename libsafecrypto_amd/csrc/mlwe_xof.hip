// mlwe_xof.hip - the XOF-driven forms of the module-LWE helpers on MI355X behind the C ABI of include/sc_hip.h.
//
// Replaces, for a batch of seeds, /root/reference/src/utils/arith/module_lwe.c:550-560 (uniform_random_ring_q_xof, called once per matrix entry by
// create_rand_product_32_xof / _16_xof, :751-912), :330-344 (uniform_rand_sample_xof), :347-367 (uniform_rand_sample_small_xof) and :449-485
// (binomial_rand_sample_xof, the branch the library build compiles) on an xof context prepared as the scheme code does: xof_init, xof_absorb(seed),
// xof_final (src/schemes/sig/dilithium/dilithium.c:866-870) - SHAKE128 or SHAKE256 with the reference's permutation (keccak_dev.h).
// The product itself is sc_hip_matvec on the expanded matrix (dilithium.hip), exactly as for the CSPRNG form.
//
// Mapping: a sponge is a sequential chain, so the unit of parallelism is the seed - one lane per seed, the state in LDS (Sponge<HT>), 21 (SHAKE128) or 17
// (SHAKE256) 64-bit words read per permutation.  A matrix of k l polynomials is k l n / 42 permutations per key (183 for Dilithium-III): at 16384
// keys that is one wavefront on each of 256 SIMDs for 183 x ~10 us; what it costs at a handful of keys is one wavefront's chain and nothing helps
// that but a shorter chain (once per key, not per operation).  The kernels are instruction-bound (about 4600 VALU instructions per 168
// bytes of stream); the scattered 8-byte stores of a lane's row are noise beside that.
//
// What the samplers write is EXACTLY what the reference's functions write, including what they leave alone: uniform_rand_sample_small_xof fills whole
// blocks of 1024 values only, binomial_rand_sample_xof writes every block of 1024 at the start of the output (include/sc_hip.h).
#include "engine_internal.h"
#include "keccak_dev.h"

using namespace sc;

namespace {

constexpr int HT = 64;

// the 32-bit word `pos` of the current rate block
template <int NT>
__device__ __forceinline__ uint32_t block_word(const Sponge<NT> &sp, uint32_t pos) { return (uint32_t)(sp.st[(pos >> 1) * NT] >> (32 * (pos & 1))); }

// uniform_random_ring_q_xof x npoly: total = npoly * n words a seed, each masked to q_bits, minus q when >= q.  skip_words: 32-bit words of the stream to pass over first.
template <int RATE>
__global__ __launch_bounds__(HT) void mlwe_xof_ring_kernel(uint32_t nkeys, const uint8_t *__restrict__ seeds, size_t seed_stride, uint32_t seed_len, uint64_t skip_words,
                                                           uint32_t total, int32_t *__restrict__ out, uint32_t q, uint32_t mask)
{
    __shared__ uint64_t st[25 * HT];
    const uint32_t i = blockIdx.x * HT + threadIdx.x;
    Sponge<HT> sp; sp.init(st, threadIdx.x, RATE);
    if (i >= nkeys) return;
    sp.absorb(seeds + (size_t)i * seed_stride, seed_len);
    sp.finalize(0x1F);
    constexpr uint32_t RW = RATE / 4;
    uint64_t skip = skip_words;
    while (skip >= RW) { sp.permute(); skip -= RW; }
    uint32_t pos = (uint32_t)skip;
    int32_t *o = out + (size_t)i * total;
    for (uint32_t k = 0; k < total;) {
        for (; pos < RW && k < total; pos++, k++) {
            uint32_t v = block_word(sp, pos) & mask;
            v -= (v >= q) ? q : 0u;
            o[k] = (int32_t)v;
        }
        if (k < total) { sp.permute(); pos = 0; }
    }
}

// a byte reader over the squeezed stream starting `skip` bytes in
template <int NT>
__device__ __forceinline__ void skip_bytes(Sponge<NT> &sp, uint64_t skip)
{
    while (skip >= (uint64_t)sp.rate) { sp.permute(); skip -= sp.rate; }
    sp.pos = (int)skip;
}

// kind 0: uniform_rand_sample_xof - a 32-bit little-endian word per value, masked to bits + 1 bits, halved while above 2 eta (twice at most), eta - value
// kind 1: uniform_rand_sample_small_xof - whole blocks of 512 bytes -> 1024 values, low nibble then high nibble, one conditional halving
// kind 2: binomial_rand_sample_xof - blocks of eta * 256 bytes -> 1024 values, every block written at the start of the row
template <int RATE>
__global__ __launch_bounds__(HT) void mlwe_xof_sample_kernel(uint32_t nseeds, const uint8_t *__restrict__ seeds, size_t seed_stride, uint32_t seed_len, uint64_t skip,
                                                             int kind, int32_t eta, uint32_t mask, uint32_t count, int32_t *__restrict__ out, size_t out_stride)
{
    __shared__ uint64_t st[25 * HT];
    const uint32_t i = blockIdx.x * HT + threadIdx.x;
    Sponge<HT> sp; sp.init(st, threadIdx.x, RATE);
    if (i >= nseeds) return;
    sp.absorb(seeds + (size_t)i * seed_stride, seed_len);
    sp.finalize(0x1F);
    skip_bytes(sp, skip);
    int32_t *o = out + (size_t)i * out_stride;
    if (kind == 0) {
        for (uint32_t k = 0; k < count; k++) {
            uint32_t w = 0;
#pragma unroll
            for (int b = 0; b < 4; b++) w |= (uint32_t)sp.squeeze_byte() << (8 * b);
            int32_t t = (int32_t)(w & mask);
            t >>= (t > 2 * eta) ? 1 : 0;
            t >>= (t > 2 * eta) ? 1 : 0;
            o[k] = eta - t;
        }
    } else if (kind == 1) {
        const uint32_t blocks = count >> 10;
        for (uint32_t k = 0; k < blocks * 1024; k += 2) {
            const uint32_t r = sp.squeeze_byte();
            int32_t t = (int32_t)(r & mask);
            if (t > 2 * eta) t >>= 1;
            const int32_t a = eta - t;
            t = (int32_t)((r >> 4) & mask);
            if (t > 2 * eta) t >>= 1;
            *reinterpret_cast<int2 *>(o + k) = make_int2(a, eta - t);
        }
    } else {
        const uint32_t blocks = count >> 10;
        for (uint32_t blk = 0; blk < blocks; blk++)
            for (uint32_t j = 0; j < 256; j++) {
                uint32_t a[8] = { 0, 0, 0, 0, 0, 0, 0, 0 };
                for (int32_t e = 0; e < eta; e++) {
                    const uint32_t p = sp.squeeze_byte();
#pragma unroll
                    for (int b = 0; b < 8; b++) a[b] += p & (1u << b);                     // the bit stays at its position (module_lwe.c:464-482)
                }
                *reinterpret_cast<int4 *>(o + 4 * j) = make_int4((int32_t)a[0] - (int32_t)(a[1] >> 1), (int32_t)(a[2] >> 2) - (int32_t)(a[3] >> 3),
                                                                  (int32_t)(a[4] >> 4) - (int32_t)(a[5] >> 5), (int32_t)(a[6] >> 6) - (int32_t)(a[7] >> 7));
            }
    }
}

int check_seeds(const char *who, sc_hip_ctx *ctx, size_t nseeds, const void *d_seeds, size_t seed_stride, size_t seed_len, int shake_bits)
{
    if (!ctx) return fail(std::string(who) + ": null context");
    if (!d_seeds && nseeds) return fail(std::string(who) + ": null seed buffer");
    if (seed_len == 0 || seed_len > seed_stride || seed_len > 0xffffffffu) return fail(std::string(who) + ": bad seed length / stride");
    if (shake_bits != 128 && shake_bits != 256) return fail(std::string(who) + ": shake_bits must be 128 or 256");
    if (nseeds > 0x7fffffffu) return fail(std::string(who) + ": too many seeds");
    return SC_HIP_SUCCESS;
}

}  // namespace

extern "C" int sc_hip_mlwe_expand_matrix_xof(sc_hip_ctx_t *ctx, size_t nkeys, const uint8_t *d_seeds, size_t seed_stride, size_t seed_len, int shake_bits, size_t skip_bytes,
                                             int npoly, int n, int32_t q, int q_bits, int32_t *d_out, void *stream)
{
    if (check_seeds("mlwe_expand_matrix_xof", ctx, nkeys, d_seeds, seed_stride, seed_len, shake_bits)) return SC_HIP_FAILURE;
    if (nkeys == 0 || npoly == 0) return SC_HIP_SUCCESS;
    if (!d_out) return fail("mlwe_expand_matrix_xof: null output");
    if (npoly < 0 || n < 1 || n > 4096 || q < 2 || q_bits < 1 || q_bits > 31) return fail("mlwe_expand_matrix_xof: bad argument");
    if ((skip_bytes & 3) != 0) return fail("mlwe_expand_matrix_xof: the stream position must be a multiple of 4 bytes (a matrix entry is a 32-bit word of the stream)");
    if ((uint64_t)npoly * (uint64_t)n > 0xffffffffull) return fail("mlwe_expand_matrix_xof: too large");
    HIPCHK(hipSetDevice(ctx->device));
    const uint32_t total = (uint32_t)((uint64_t)npoly * n), mask = (q_bits == 31) ? 0x7fffffffu : (1u << q_bits) - 1u;
    const dim3 grid((unsigned)((nkeys + HT - 1) / HT)), block(HT);
    if (shake_bits == 128)
        hipLaunchKernelGGL(mlwe_xof_ring_kernel<168>, grid, block, 0, (hipStream_t)stream, (uint32_t)nkeys, d_seeds, seed_stride, (uint32_t)seed_len, (uint64_t)(skip_bytes / 4), total, d_out, (uint32_t)q, mask);
    else
        hipLaunchKernelGGL(mlwe_xof_ring_kernel<136>, grid, block, 0, (hipStream_t)stream, (uint32_t)nkeys, d_seeds, seed_stride, (uint32_t)seed_len, (uint64_t)(skip_bytes / 4), total, d_out, (uint32_t)q, mask);
    HIPCHK(hipGetLastError());
    return SC_HIP_SUCCESS;
}

extern "C" int sc_hip_mlwe_sample_xof(sc_hip_ctx_t *ctx, int kind, size_t nseeds, const uint8_t *d_seeds, size_t seed_stride, size_t seed_len, int shake_bits, size_t skip_bytes,
                                      int32_t eta, int bits, size_t count, int32_t *d_out, size_t out_stride, size_t *pos_after, void *stream)
{
    if (check_seeds("mlwe_sample_xof", ctx, nseeds, d_seeds, seed_stride, seed_len, shake_bits)) return SC_HIP_FAILURE;
    if (kind < SC_HIP_XOF_UNIFORM || kind > SC_HIP_XOF_BINOMIAL) return fail("mlwe_sample_xof: kind must be SC_HIP_XOF_UNIFORM, _UNIFORM_SMALL or _BINOMIAL");
    if (eta < 1 || (kind == SC_HIP_XOF_BINOMIAL ? eta > 5 : eta > (1 << 20))) return fail("mlwe_sample_xof: eta out of range (binomial: 1..5, the reference's 5 * 256-byte block)");
    if (kind != SC_HIP_XOF_BINOMIAL && (bits < 0 || bits > 30)) return fail("mlwe_sample_xof: bits out of range");
    if (kind == SC_HIP_XOF_UNIFORM_SMALL && bits > 3) return fail("mlwe_sample_xof: the nibble form holds bits + 1 <= 4 bits a value");
    if (count > 0xffffffffu || out_stride < count) return fail("mlwe_sample_xof: count / row pitch");
    // bytes of stream the reference's function consumes (the restatement's header has the two places where that is not count times a constant)
    size_t used = 0;
    if (kind == SC_HIP_XOF_UNIFORM) used = count * 4;
    else if (kind == SC_HIP_XOF_UNIFORM_SMALL) used = (count >> 10) * 512;
    else used = (count >> 10) * (size_t)eta * 256;
    if (pos_after) *pos_after = skip_bytes + used;
    if (nseeds == 0 || count == 0) return SC_HIP_SUCCESS;
    if (!d_out) return fail("mlwe_sample_xof: null output");
    if (kind != SC_HIP_XOF_UNIFORM && (((uintptr_t)d_out & 15) != 0 || (out_stride & 3) != 0)) return fail("mlwe_sample_xof: rows must be 16-byte aligned (pitch a multiple of 4 values)");
    HIPCHK(hipSetDevice(ctx->device));
    const uint32_t mask = (1u << (bits + 1)) - 1u;
    const dim3 grid((unsigned)((nseeds + HT - 1) / HT)), block(HT);
    if (shake_bits == 128)
        hipLaunchKernelGGL(mlwe_xof_sample_kernel<168>, grid, block, 0, (hipStream_t)stream, (uint32_t)nseeds, d_seeds, seed_stride, (uint32_t)seed_len, (uint64_t)skip_bytes, kind, eta, mask, (uint32_t)count, d_out, out_stride);
    else
        hipLaunchKernelGGL(mlwe_xof_sample_kernel<136>, grid, block, 0, (hipStream_t)stream, (uint32_t)nseeds, d_seeds, seed_stride, (uint32_t)seed_len, (uint64_t)skip_bytes, kind, eta, mask, (uint32_t)count, d_out, out_stride);
    HIPCHK(hipGetLastError());
    return SC_HIP_SUCCESS;
}
