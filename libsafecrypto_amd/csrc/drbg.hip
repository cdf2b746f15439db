// drbg.hip - batched expansion of seeds with the reference's default CSPRNG (AES-256 CTR-DRBG, see drbg_dev.h) behind the C ABI
// of include/sc_hip.h: one generator per seed, one lane per generator.  This is what create_csprng(sc, seed, len) followed by
// prng_mem / prng_32 calls yields on the host (src/utils/arith/module_lwe.c:914-940, src/utils/crypto/prng.c:963-1107), so
// per-message randomness (Kyber's noise from r, signing tapes) and per-key expansions (the matrix from rho) can be produced
// where they are consumed instead of being uploaded.
#include "engine_internal.h"
#include "drbg_dev.h"

#include <string.h>

using namespace sc;

namespace sc {

// FIPS-197 on the host, only to build the constants of DrbgConst (S-box; the three instantiation blocks under the zero key)
namespace {
uint8_t h_xtime(uint8_t x) { return (uint8_t)((x << 1) ^ ((x >> 7) * 0x1b)); }
uint8_t h_gmul(uint8_t a, uint8_t b) { uint8_t r = 0; while (b) { if (b & 1) r ^= a; a = h_xtime(a); b >>= 1; } return r; }

void h_aes256_zero_key_encrypt(const uint8_t sbox[256], const uint8_t in[16], uint8_t out[16])
{
    uint8_t w[60][4];
    memset(w, 0, 32);
    uint8_t rc = 1;
    for (int i = 8; i < 60; i++) {
        uint8_t t[4]; memcpy(t, w[i - 1], 4);
        if (i % 8 == 0) { const uint8_t t0 = t[0]; t[0] = sbox[t[1]] ^ rc; t[1] = sbox[t[2]]; t[2] = sbox[t[3]]; t[3] = sbox[t0]; rc = h_xtime(rc); }
        else if (i % 8 == 4) for (int j = 0; j < 4; j++) t[j] = sbox[t[j]];
        for (int j = 0; j < 4; j++) w[i][j] = w[i - 8][j] ^ t[j];
    }
    const uint8_t *rk = &w[0][0];
    uint8_t s[16], t[16];
    for (int i = 0; i < 16; i++) s[i] = in[i] ^ rk[i];
    for (int r = 1; r <= 14; r++) {
        for (int c = 0; c < 4; c++) for (int row = 0; row < 4; row++) t[4 * c + row] = sbox[s[4 * ((c + row) & 3) + row]];
        if (r < 14) {
            for (int c = 0; c < 4; c++) {
                const uint8_t a0 = t[4 * c], a1 = t[4 * c + 1], a2 = t[4 * c + 2], a3 = t[4 * c + 3];
                s[4 * c] = (uint8_t)(h_xtime(a0) ^ h_xtime(a1) ^ a1 ^ a2 ^ a3);
                s[4 * c + 1] = (uint8_t)(a0 ^ h_xtime(a1) ^ h_xtime(a2) ^ a2 ^ a3);
                s[4 * c + 2] = (uint8_t)(a0 ^ a1 ^ h_xtime(a2) ^ h_xtime(a3) ^ a3);
                s[4 * c + 3] = (uint8_t)(h_xtime(a0) ^ a0 ^ a1 ^ a2 ^ h_xtime(a3));
            }
        } else memcpy(s, t, 16);
        for (int i = 0; i < 16; i++) s[i] ^= rk[16 * r + i];
    }
    memcpy(out, s, 16);
}
}  // namespace

static DrbgConst make_drbg_const()
{
    DrbgConst c;
    for (int x = 0; x < 256; x++) {
        uint8_t inv = 0;
        if (x) for (int y = 1; y < 256; y++) if (h_gmul((uint8_t)x, (uint8_t)y) == 1) { inv = (uint8_t)y; break; }
        uint8_t s = inv, r = inv;
        for (int i = 0; i < 4; i++) { r = (uint8_t)((r << 1) | (r >> 7)); s ^= r; }
        c.sbox[x] = s ^ 0x63;
    }
    for (uint32_t ctr = 1; ctr <= 3; ctr++) {                      // ctr_drbg.c:104-128: block index 2, 1, 0
        uint8_t blk[16];
        for (int w = 0; w < 4; w++) { blk[4 * w] = (uint8_t)ctr; blk[4 * w + 1] = blk[4 * w + 2] = blk[4 * w + 3] = 0; }
        h_aes256_zero_key_encrypt(c.sbox, blk, c.init_bytes + 16 * (3 - ctr));
    }
    return c;
}

const DrbgConst &drbg_const()
{
    static const DrbgConst c = make_drbg_const();                  // (a function-local static's initialisation is thread-safe: calls come from any host thread)
    return c;
}

}  // namespace sc

namespace {

// Blocks [b0, b1) of the streams of `count` generators (rows `list[0..count)`, or rows 0..count-1; count_dev: the count lives on the device and `count`
// bounds it), 2^lg lanes per generator.  CTR mode: block b of a stream is AES_key(counter0 + b) whatever came before it, so the lanes of a generator take
// blocks b0 + j, b0 + j + G, ... after each has run the (same) key schedule - a lane-per-generator walk of a 16 KB Falcon tape is 1024 sequential AES blocks
// on 128 wavefronts; with G = 32 it is 32 blocks per lane on 4096 wavefronts, and the G lanes of a generator write 16 G contiguous bytes per step.
// words != 0: prng_32 order (w1, w0, w3, w2 per block), else prng_mem's byte order.  zero != 0: the same walk storing zeros (the wipe behind a seeded call).
__global__ __launch_bounds__(512) void drbg_fill_kernel(uint32_t count, const uint32_t *__restrict__ count_dev, const uint32_t *__restrict__ list,
                                                        const uint8_t *__restrict__ seeds, size_t seed_stride, uint32_t seed_len,
                                                        uint8_t *__restrict__ out, size_t out_stride, uint32_t b0, uint32_t b1, uint32_t lg, int words, int zero, DrbgConst C)
{
    __shared__ AesLds64 T;                                             // a table copy per lane: one v_perm_b32 forms a look-up's address (drbg_dev.h)
    if (!zero) { T.fill(C); __syncthreads(); }
    const AesLane64 L(T);
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x, G = 1u << lg;
    const uint32_t gen = t >> lg, j = t & (G - 1);
    uint32_t cnt = count;
    if (count_dev) { const uint32_t c = *count_dev; cnt = c < count ? c : count; }
    if (gen >= cnt) return;
    const uint32_t row = list ? list[gen] : gen;
    uint4 *o = reinterpret_cast<uint4 *>(out + (size_t)row * out_stride);
    if (zero) { for (uint32_t b = b0 + j; b < b1; b += G) o[b] = make_uint4(0, 0, 0, 0); return; }
    Drbg d;
    d.init(seeds + (size_t)row * seed_stride, seed_len, C, L);
    d.counter += b0 + j;
    uint32_t b = b0 + j;
    for (; b + G < b1; b += 2 * G) {                                  // two blocks of the lane side by side (aes256_encrypt2)
        const uint32_t c0 = d.counter, c1 = d.counter + G;
        const uint32_t in0[4] = { c0, c0, c0, c0 }, in1[4] = { c1, c1, c1, c1 };
        uint32_t w[4], v[4];
        aes256_encrypt2(d.aes, in0, in1, w, v, L);
        d.counter += 2 * G;
        o[b] = words ? make_uint4(w[1], w[0], w[3], w[2]) : make_uint4(w[0], w[1], w[2], w[3]);
        o[b + G] = words ? make_uint4(v[1], v[0], v[3], v[2]) : make_uint4(v[0], v[1], v[2], v[3]);
    }
    if (b < b1) {
        uint32_t w[4];
        d.block(w, L);
        o[b] = words ? make_uint4(w[1], w[0], w[3], w[2]) : make_uint4(w[0], w[1], w[2], w[3]);
    }
}

// uniform_random_ring_q_csprng (src/utils/arith/module_lwe.c:519-547, n = 256) for polynomial p of key i: the p-th prng_mem(512)
// of the generator seeded with rho_i -> 256 little-endian 16-bit words, masked to q_bits, minus q when >= q.  One lane per
// (key, polynomial): CTR mode lets every lane start at its own block after its own key schedule.
__global__ __launch_bounds__(256) void drbg_matrix_kernel(uint32_t nkeys, uint32_t npoly, const uint8_t *__restrict__ rho, size_t rho_stride, uint32_t rho_len,
                                                          int32_t *__restrict__ out, int32_t q, uint32_t mask, DrbgConst C)
{
    __shared__ AesLdsWide L;
    L.fill(C);
    __syncthreads();
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= nkeys * npoly) return;
    const uint32_t key = t / npoly, p = t % npoly;
    Drbg d;
    d.init(rho + (size_t)key * rho_stride, rho_len, C, L);
    d.counter += p * 32;                                               // 512 bytes = 32 blocks per polynomial
    int4 *o = reinterpret_cast<int4 *>(out + (size_t)t * 256);
    for (int b = 0; b < 32; b++) {
        uint32_t w[4]; int32_t a[8];
        d.block(w, L);
#pragma unroll
        for (int j = 0; j < 8; j++) { int32_t v = (int32_t)(((w[j >> 1] >> (16 * (j & 1))) & 0xFFFFu) & mask); v -= (v >= q) ? q : 0; a[j] = v; }
        o[2 * b] = make_int4(a[0], a[1], a[2], a[3]); o[2 * b + 1] = make_int4(a[4], a[5], a[6], a[7]);
    }
}

int run_expand(sc_hip_ctx *ctx, size_t batch, const uint8_t *d_seeds, size_t seed_stride, size_t seed_len, void *d_out, size_t out_stride, size_t nbytes, int words,
               void *stream, const char *who)
{
    if (!ctx) return fail(std::string(who) + ": null context");
    if (batch == 0 || nbytes == 0) return SC_HIP_SUCCESS;
    if (!d_seeds || !d_out) return fail(std::string(who) + ": null buffer");
    if (seed_len == 0 || seed_len > seed_stride || seed_len > 0xffffffffu) return fail(std::string(who) + ": bad seed length / stride");
    if ((nbytes & 15) != 0 || out_stride < nbytes || (out_stride & 15) != 0 || ((uintptr_t)d_out & 15) != 0)
        return fail(std::string(who) + ": output rows must be 16-byte aligned multiples of 16 bytes");
    if (batch > 0x7fffffffu || nbytes / 16 > 0xffffffffu) return fail(std::string(who) + ": too large");
    return sc::drbg_fill(ctx, d_seeds, seed_stride, seed_len, nullptr, (uint32_t)batch, nullptr, d_out, out_stride, 0, nbytes / 16, words, false, (hipStream_t)stream);
}

}  // namespace

extern "C" int sc_hip_drbg_mem(sc_hip_ctx_t *ctx, size_t batch, const uint8_t *d_seeds, size_t seed_stride, size_t seed_len,
                               uint8_t *d_out, size_t out_stride, size_t nbytes, void *stream)
{ return run_expand(ctx, batch, d_seeds, seed_stride, seed_len, d_out, out_stride, nbytes, 0, stream, "drbg_mem"); }

extern "C" int sc_hip_drbg_words(sc_hip_ctx_t *ctx, size_t batch, const uint8_t *d_seeds, size_t seed_stride, size_t seed_len,
                                 uint32_t *d_out, size_t out_stride_words, size_t nwords, void *stream)
{ return run_expand(ctx, batch, d_seeds, seed_stride, seed_len, d_out, out_stride_words * 4, nwords * 4, 1, stream, "drbg_words"); }

// replaces: the npoly calls of uniform_random_ring_q_csprng(create_csprng(sc, rho, rho_len), ...) with which the schemes expand
// their public matrices (module_lwe.c:519-547; kyber_kem.c / dilithium.c key generation and create_rand_product_*_csprng), n = 256
extern "C" int sc_hip_mlwe_expand_matrix(sc_hip_ctx_t *ctx, size_t nkeys, const uint8_t *d_rho, size_t rho_stride, size_t rho_len, int npoly, int32_t q, int q_bits,
                                         int32_t *d_out, void *stream)
{
    if (!ctx) return fail("mlwe_expand_matrix: null context");
    if (nkeys == 0 || npoly == 0) return SC_HIP_SUCCESS;
    if (!d_rho || !d_out) return fail("mlwe_expand_matrix: null buffer");
    if (rho_len == 0 || rho_len > rho_stride || npoly < 0 || q < 2 || q_bits < 1 || q_bits > 30) return fail("mlwe_expand_matrix: bad argument");
    if (nkeys * (size_t)npoly > 0x7fffffffu || ((uintptr_t)d_out & 15) != 0) return fail("mlwe_expand_matrix: too large / misaligned output");
    HIPCHK(hipSetDevice(ctx->device));
    const size_t threads = nkeys * (size_t)npoly;
    hipLaunchKernelGGL(drbg_matrix_kernel, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (uint32_t)nkeys, (uint32_t)npoly, d_rho, rho_stride,
                       (uint32_t)rho_len, d_out, q, (1u << q_bits) - 1u, sc::drbg_const());
    HIPCHK(hipGetLastError());
    return SC_HIP_SUCCESS;
}

namespace sc {
// see engine_internal.h
int drbg_fill(sc_hip_ctx *ctx, const uint8_t *d_seeds, size_t seed_stride, size_t seed_len, const uint32_t *d_list, uint32_t count, const uint32_t *d_count,
              void *d_out, size_t out_stride_bytes, size_t blk0, size_t blk1, int words, bool zero, hipStream_t s)
{
    if (count == 0 || blk1 <= blk0) return SC_HIP_SUCCESS;
    if (blk1 > 0xffffffffu) return fail("drbg_fill: stream position too large");
    // lanes per generator: enough to give every SIMD four wavefronts (the cipher is a chain of dependent table look-ups), not more than the blocks to make
    const size_t nblk = blk1 - blk0, want = (size_t)ctx->cus * 4 * 4 * 64;
    uint32_t lg = 0;
    while (lg < 6 && ((size_t)count << lg) < want && ((size_t)2 << lg) <= nblk) lg++;
    const size_t threads = (size_t)count << lg;
    hipLaunchKernelGGL(drbg_fill_kernel, dim3((unsigned)((threads + 511) / 512)), dim3(512), 0, s, count, d_count, d_list, d_seeds, seed_stride, (uint32_t)seed_len,
                       (uint8_t *)d_out, out_stride_bytes, (uint32_t)blk0, (uint32_t)blk1, lg, words, zero ? 1 : 0, drbg_const());
    HIPCHK(hipGetLastError());
    return SC_HIP_SUCCESS;
}
}  // namespace sc
