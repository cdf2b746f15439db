// kyber.hip - batched Kyber (round 1: q = 7681, n = 256) CPA encryption / decryption behind the C ABI of sc_hip.h.
//
// Replaces kyber_cpa_enc / kyber_cpa_dec (src/utils/arith/module_lwe.c:1268-1401, 1415-1481) as
// kyber_kem_encapsulation / _decapsulation call them (src/schemes/kem/kyber/kyber_kem.c:866,992-1001; keys hold NTT(t)
// and NTT(s): kyber_kem_params.h:43,46), with binomial_rand_sample_csprng (:380-412), create_rand_product_16_csprng in
// its TRANSPOSED form (:588-748) and mlwe_compress / mlwe_decompress (:43-69, reproduced literally, wrap-around included).
// At the CPA level what the reference derives with its AES-CTR-DRBG is an input: the k*k matrix polynomials in generation
// order (once per key) and, per message, the (2k+1)*n*eta/4 noise bytes its prng_mem calls return for the seed r.
//
// KEM level (kyber_kem_encapsulation / kyber_kem_decapsulation, kyber_kem.c:855-1040, with the reference's build options:
// SHA3-512 with its own permutation as the hash, the AES CTR-DRBG for the oracle bytes and the noise): G = SHA3-512(rho | t as
// 16-bit LE | m) -> a DRBG seeded with md[0..32) yields K_bar | r | d (kem_g_function :757-800, kyber_oracle_bytes :745-755);
// a DRBG seeded with r yields the noise (module_lwe.c:1290-1340); K = SHA3-512(K_bar | u | v as 16-bit LE | d)[0..32)
// (kem_h_function :690-743); ciphertext = u (d_u bits) | v (d_v bits) | d, MSB-first.  The part of G's input that is the same
// for every message (rho | t, 21 whole SHA3-512 blocks of it) is absorbed once per key.
#include "engine_internal.h"
#include "mlwe_kernels.h"
#include "keccak_dev.h"
#include "drbg_dev.h"

using namespace sc;
using namespace sc_mlwe;

namespace {

struct KyP { int32_t n, q, q_bits; uint32_t q_inv; int32_t q_norm, k, eta, d_u, d_v, d_t, set, pad[5]; };

// mlwe_compress on one value, module_lwe.c:43-56
__device__ __forceinline__ int32_t ky_compress(int32_t x, int d, uint32_t q, uint32_t q_inv, int q_norm)
{
    int32_t t = (int32_t)((uint32_t)x << d);
    t = (int32_t)((uint32_t)t + (q >> 1));
    t = (int32_t)(((uint64_t)(int64_t)t * (uint64_t)q_inv) >> (32 + q_norm));
    return (int32_t)((uint32_t)t & ((1u << d) - 1u));
}

#define KY_ITEMS(item, count) for (uint32_t item = blockIdx.y; item < (count); item += gridDim.y)
static inline dim3 ky_grid(size_t per, size_t batch) { return dim3((unsigned)((per + 255) / 256), (unsigned)(batch < 65535 ? batch : 65535)); }

// binomial_rand_sample_csprng: coefficient group g (4 values) of a sampler reads its eta bytes [g*eta, (g+1)*eta)
__global__ void ky_binomial_kernel(const uint8_t *noise, size_t noise_stride, uint32_t count, int32_t *r, int32_t *e1, int32_t *e2, KyP P)
{
    // grid: x over the groups of an item, y over the items (KY_ITEMS): no division of a flat thread index by the runtime group count - as a 64-bit
    // division per thread that was most of what these one-load-one-store kernels executed
    const int kn4 = P.k * P.n / 4, n4 = P.n / 4, per = 2 * kn4 + n4;
    const int g = (int)(blockIdx.x * blockDim.x + threadIdx.x);
    if (g >= per) return;
    KY_ITEMS(item, count) {
        const uint8_t *b = noise + (size_t)item * noise_stride + (size_t)g * P.eta;
        uint32_t a[8] = { 0, 0, 0, 0, 0, 0, 0, 0 };
        for (int e = 0; e < P.eta; e++) { const uint32_t v = b[e]; for (int i = 0; i < 8; i++) a[i] += v & (1u << i); }
        const int32_t c0 = (int32_t)a[0] - (int32_t)(a[1] >> 1), c1 = (int32_t)(a[2] >> 2) - (int32_t)(a[3] >> 3), c2 = (int32_t)(a[4] >> 4) - (int32_t)(a[5] >> 5),
                      c3 = (int32_t)(a[6] >> 6) - (int32_t)(a[7] >> 7);
        if (g < kn4) *reinterpret_cast<int4 *>(r + ((size_t)item * kn4 + g) * 4) = make_int4(c0, c1, c2, c3);        // r feeds the transform: 32-bit words
        else {
            // e1, e2 (|value| <= eta) only meet ky_enc_finish_kernel: a signed byte each - these kernels are bandwidth-bound, and as 32-bit words the two
            // noise vectors were 8 KB written and read back per operation
            const uint32_t packed = ((uint32_t)c0 & 0xFFu) | (((uint32_t)c1 & 0xFFu) << 8) | (((uint32_t)c2 & 0xFFu) << 16) | ((uint32_t)c3 << 24);
            if (g < 2 * kn4) reinterpret_cast<uint32_t *>(e1)[(size_t)item * kn4 + (g - kn4)] = packed;
            else reinterpret_cast<uint32_t *>(e2)[(size_t)item * n4 + (g - 2 * kn4)] = packed;
        }
    }
}

// u = compress(u0 + e1, d_u) ; v = compress(v0 + [m bit] floor(q/2) + e2, d_v)   (module_lwe.c:1355-1395)
// us, vs: words between the u0 / v0 rows of consecutive items (the fused product leaves them side by side: [k + 1][n] per item)
__global__ void ky_enc_finish_kernel(const int32_t *u0, size_t us, const int32_t *v0, size_t vs, const int32_t *e1, const int32_t *e2, const uint8_t *m, uint32_t count,
                                     int32_t *u, int32_t *v, KyP P)
{
    // four coefficients per thread (16-byte loads and stores), grid as ky_binomial_kernel
    const int e = 4 * (int)(blockIdx.x * blockDim.x + threadIdx.x);
    if (e >= (P.k + 1) * P.n) return;
    KY_ITEMS(item, count) {
        if (e < P.k * P.n) {
            const size_t i = (size_t)item * P.k * P.n + e;
            const int4 a = *reinterpret_cast<const int4 *>(u0 + (size_t)item * us + e);
            const uint32_t pb = reinterpret_cast<const uint32_t *>(e1)[i >> 2];        // four signed bytes (ky_binomial_kernel)
            const int4 b = make_int4((int32_t)(int8_t)pb, (int32_t)(int8_t)(pb >> 8), (int32_t)(int8_t)(pb >> 16), (int32_t)(int8_t)(pb >> 24));
            *reinterpret_cast<int4 *>(u + i) = make_int4(ky_compress(a.x + b.x, P.d_u, (uint32_t)P.q, P.q_inv, P.q_norm), ky_compress(a.y + b.y, P.d_u, (uint32_t)P.q, P.q_inv, P.q_norm),
                                                         ky_compress(a.z + b.z, P.d_u, (uint32_t)P.q, P.q_inv, P.q_norm), ky_compress(a.w + b.w, P.d_u, (uint32_t)P.q, P.q_inv, P.q_norm));
        } else {
            const int j = e - P.k * P.n; const size_t i = (size_t)item * P.n + j;
            const uint32_t mb = m[(size_t)item * 32 + (j >> 3)];               // j is a multiple of 4: its four bits sit in one byte, most significant first
            const int32_t h = P.q >> 1;
            const int4 a = *reinterpret_cast<const int4 *>(v0 + (size_t)item * vs + j);
            const uint32_t pb = reinterpret_cast<const uint32_t *>(e2)[i >> 2];
            const int4 b = make_int4((int32_t)(int8_t)pb, (int32_t)(int8_t)(pb >> 8), (int32_t)(int8_t)(pb >> 16), (int32_t)(int8_t)(pb >> 24));
            const int32_t b0 = (mb & (0x80u >> (j & 7))) ? h : 0, b1 = (mb & (0x40u >> (j & 7))) ? h : 0, b2 = (mb & (0x20u >> (j & 7))) ? h : 0, b3 = (mb & (0x10u >> (j & 7))) ? h : 0;
            *reinterpret_cast<int4 *>(v + i) = make_int4(ky_compress(a.x + b0 + b.x, P.d_v, (uint32_t)P.q, P.q_inv, P.q_norm), ky_compress(a.y + b1 + b.y, P.d_v, (uint32_t)P.q, P.q_inv, P.q_norm),
                                                         ky_compress(a.z + b2 + b.z, P.d_v, (uint32_t)P.q, P.q_inv, P.q_norm), ky_compress(a.w + b3 + b.w, P.d_v, (uint32_t)P.q, P.q_inv, P.q_norm));
        }
    }
}

// mlwe_decompress(u, d_u): t = u*q >> d  (module_lwe.c:58-69)
__global__ void ky_decompress_kernel(const int32_t *in, int32_t *out, size_t len, int d, int32_t q)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < len) out[i] = (int32_t)(in[i] * q) >> d;
}

// m byte: 8 x compress(decompress(v) - x, 1), most significant bit first (module_lwe.c:1462-1478)
__global__ void ky_dec_finish_kernel(const int32_t *v, const int32_t *x, uint8_t *m, uint32_t count, KyP P)
{
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t item = (uint32_t)(t / 32); const int byte = (int)(t % 32);
    if (item >= count) return;
    uint32_t b = 0;
    for (int j = 0; j < 8; j++) {
        const size_t i = (size_t)item * P.n + 8 * byte + j;
        const int32_t dv = (int32_t)(v[i] * P.q) >> P.d_v;
        b |= (uint32_t)(ky_compress(dv - x[i], 1, (uint32_t)P.q, P.q_inv, P.q_norm) & 1) << (7 - j);
    }
    m[(size_t)item * 32 + byte] = (uint8_t)b;
}


// key generation: s and e (k polynomials each) from the 2k noise blocks of a key's stream (module_lwe.c:1215-1216)
__global__ void ky_kg_binomial_kernel(const uint8_t *stream, size_t stride, size_t noise_off, uint32_t count, int32_t *s, int32_t *e, KyP P)
{
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int kn4 = P.k * P.n / 4, per = 2 * kn4;
    const uint32_t item = (uint32_t)(t / per); const int g = (int)(t % per);
    if (item >= count) return;
    const uint8_t *b = stream + (size_t)item * stride + noise_off + (size_t)g * P.eta;
    uint32_t a[8] = { 0, 0, 0, 0, 0, 0, 0, 0 };
    for (int x = 0; x < P.eta; x++) { const uint32_t v = b[x]; for (int i = 0; i < 8; i++) a[i] += v & (1u << i); }
    int32_t *dst = (g < kn4) ? s + ((size_t)item * kn4 + g) * 4 : e + ((size_t)item * kn4 + (g - kn4)) * 4;
    dst[0] = (int32_t)a[0] - (int32_t)(a[1] >> 1);
    dst[1] = (int32_t)(a[2] >> 2) - (int32_t)(a[3] >> 3);
    dst[2] = (int32_t)(a[4] >> 4) - (int32_t)(a[5] >> 5);
    dst[3] = (int32_t)(a[6] >> 6) - (int32_t)(a[7] >> 7);
}

// t = normalise(decompress(compress(centre(A s + e), d_t)))   (module_lwe.c:1251-1255); as canonical, e small signed
__global__ void ky_kg_t_kernel(const int32_t *as, const int32_t *e, int32_t *t, size_t len, KyP P)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= len) return;
    int32_t v = (as[i] + e[i]) % P.q;                                  // centre_32: the representative in [-(q-1)/2, (q-1)/2]
    const int32_t q2 = (P.q - 1) >> 1;
    if (v < -q2) v += P.q;
    if (v > q2) v -= P.q;
    v = ky_compress(v, P.d_t, (uint32_t)P.q, P.q_inv, P.q_norm);
    v = (int32_t)(v * P.q) >> P.d_t;                                   // mlwe_decompress (module_lwe.c:58-69)
    v %= P.q; if (v < 0) v += P.q;
    t[i] = v;
}

// rho and z of every key out of its stream: bytes [0, 32) and [z_off, z_off + 32)
__global__ void ky_kg_slices_kernel(const uint8_t *stream, size_t stride, size_t z_off, uint32_t count, uint8_t *rho, uint8_t *z)
{
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t item = (uint32_t)(t / 64); const int b = (int)(t % 64);
    if (item >= count) return;
    const uint8_t *s = stream + (size_t)item * stride;
    if (b < 32) rho[(size_t)item * 32 + b] = s[b]; else z[(size_t)item * 32 + (b - 32)] = s[z_off + (b - 32)];
}

constexpr int KHT = 64;      // KEM hash kernels: lanes per workgroup (LDS sponge states)

// once per key: the sponge state after the whole 72-byte blocks of rho | t16
__global__ void ky_gprefix_kernel(const uint8_t *__restrict__ prefix, uint32_t full_bytes, uint64_t *__restrict__ state)
{
    __shared__ uint64_t st[25 * KHT];
    Sponge<KHT> sp; sp.init(st, threadIdx.x, 72);
    if (threadIdx.x != 0) return;
    sp.absorb(prefix, full_bytes);
    sp.save(state);
}

// G and the two DRBG expansions, one lane per message: krd [batch][96] = K_bar | r | d ; noise [batch][noise_stride]
// PK: a public key per message - the prefix rho | t (16 bits per coefficient) is absorbed from the message's own key (rho [batch][32],
// t [batch][k*n] as sc_hip_kyber_keygen writes them) instead of resuming from the one key's saved state
template <class SP>
__device__ __forceinline__ void ky_absorb_key(SP &sp, const uint8_t *__restrict__ rho, const int32_t *__restrict__ t, int kn)
{
    const uint64_t *r = reinterpret_cast<const uint64_t *>(rho);
    for (int w = 0; w < 4; w++) sp.absorb_u64(r[w]);
    const int4 *pt = reinterpret_cast<const int4 *>(t);
    for (int g = 0; g < kn / 4; g++) { const int4 c = pt[g]; sp.absorb_u64((uint64_t)(c.x & 0xFFFF) | ((uint64_t)(c.y & 0xFFFF) << 16) | ((uint64_t)(c.z & 0xFFFF) << 32) | ((uint64_t)(c.w & 0xFFFF) << 48)); }
}

template <bool PK>
__global__ __launch_bounds__(KHT) void ky_g_kernel(uint32_t batch, const uint8_t *__restrict__ m, const uint64_t *__restrict__ gstate, const uint8_t *__restrict__ tail, uint32_t tail_len,
                                                   uint8_t *__restrict__ krd, uint8_t *__restrict__ noise, size_t noise_stride, uint32_t noise_blocks, DrbgConst C,
                                                   const uint8_t *__restrict__ rho, const int32_t *__restrict__ tk, int kn)
{
    __shared__ uint64_t st[25 * KHT];
    __shared__ AesLds L;
    L.fill(C);
    __syncthreads();
    const uint32_t i = blockIdx.x * KHT + threadIdx.x;
    Sponge<KHT> sp;
    if (PK) sp.init(st, threadIdx.x, 72); else sp.load(st, threadIdx.x, 72, gstate);
    if (i >= batch) return;
    if (PK) ky_absorb_key(sp, rho + (size_t)i * 32, tk + (size_t)i * kn, kn); else sp.absorb(tail, tail_len);
    sp.absorb(m + (size_t)i * 32, 32);
    sp.finalize(0x06);
    uint8_t md[32];
    for (int b = 0; b < 32; b++) md[b] = sp.squeeze_byte();
    Drbg d;
    d.init(md, 32, C, L);
    uint4 *o = reinterpret_cast<uint4 *>(krd + (size_t)i * 96);
    uint32_t w[4], r32[8];
    for (int b = 0; b < 6; b++) {
        d.block(w, L);
        o[b] = make_uint4(w[0], w[1], w[2], w[3]);
        if (b == 2) { r32[0] = w[0]; r32[1] = w[1]; r32[2] = w[2]; r32[3] = w[3]; }
        if (b == 3) { r32[4] = w[0]; r32[5] = w[1]; r32[6] = w[2]; r32[7] = w[3]; }
    }
    uint8_t r[32];
    for (int b = 0; b < 32; b++) r[b] = (uint8_t)(r32[b >> 2] >> (8 * (b & 3)));
    d.init(r, 32, C, L);
    uint4 *no = reinterpret_cast<uint4 *>(noise + (size_t)i * noise_stride);
    for (uint32_t b = 0; b < noise_blocks; b++) { d.block(w, L); no[b] = make_uint4(w[0], w[1], w[2], w[3]); }
}

// H: key = SHA3-512(K_bar | u16 | v16 | d)[0..32); d from krd (encapsulation) or from the ciphertext (decapsulation)
__global__ __launch_bounds__(KHT) void ky_h_kernel(uint32_t batch, const uint8_t *__restrict__ krd, const int32_t *__restrict__ u, const int32_t *__restrict__ v,
                                                   const uint8_t *__restrict__ dsrc, size_t d_stride, uint8_t *__restrict__ keys, KyP P)
{
    __shared__ uint64_t st[25 * KHT];
    const uint32_t i = blockIdx.x * KHT + threadIdx.x;
    Sponge<KHT> sp; sp.init(st, threadIdx.x, 72);
    if (i >= batch) return;
    const uint64_t *kb = reinterpret_cast<const uint64_t *>(krd + (size_t)i * 96);
    for (int w = 0; w < 4; w++) sp.absorb_u64(kb[w]);
    const int kn = P.k * P.n;
    const int4 *pu = reinterpret_cast<const int4 *>(u + (size_t)i * kn), *pv = reinterpret_cast<const int4 *>(v + (size_t)i * P.n);
    for (int g = 0; g < kn / 4; g++) { const int4 c = pu[g]; sp.absorb_u64((uint64_t)(c.x & 0xFFFF) | ((uint64_t)(c.y & 0xFFFF) << 16) | ((uint64_t)(c.z & 0xFFFF) << 32) | ((uint64_t)(c.w & 0xFFFF) << 48)); }
    for (int g = 0; g < P.n / 4; g++) { const int4 c = pv[g]; sp.absorb_u64((uint64_t)(c.x & 0xFFFF) | ((uint64_t)(c.y & 0xFFFF) << 16) | ((uint64_t)(c.z & 0xFFFF) << 32) | ((uint64_t)(c.w & 0xFFFF) << 48)); }
    sp.absorb(dsrc + (size_t)i * d_stride, 32);
    sp.finalize(0x06);
    uint8_t *o = keys + (size_t)i * 32;
    for (int b = 0; b < 32; b++) o[b] = sp.squeeze_byte();
}

// The same two kernels with two lanes per message (Sponge2, keccak_dev.h; the CTR-mode generator's blocks alternate between the
// lanes).  Used while the one-lane kernels have no more wavefronts than the chip has SIMDs: both are serial chains (H: 30
// permutations; G: 2 permutations + 118 AES-256 blocks through LDS tables) and run at the latency of one wavefront.
template <bool PK>
__global__ __launch_bounds__(KHT) void ky_g2_kernel(uint32_t batch, const uint8_t *__restrict__ m, const uint64_t *__restrict__ gstate, const uint8_t *__restrict__ tail, uint32_t tail_len,
                                                    uint8_t *__restrict__ krd, uint8_t *__restrict__ noise, size_t noise_stride, uint32_t noise_blocks, DrbgConst C,
                                                    const uint8_t *__restrict__ rho, const int32_t *__restrict__ tk, int kn)
{
    __shared__ uint32_t st[25 * KHT];
    __shared__ AesLds L;
    L.fill(C);
    __syncthreads();
    const uint32_t i = (blockIdx.x * KHT + threadIdx.x) >> 1;
    const uint32_t hh = threadIdx.x & 1;
    Sponge2<KHT> sp;
    if (PK) sp.init(st, threadIdx.x, 72); else sp.load(st, threadIdx.x, 72, gstate);
    if (i >= batch) return;
    if (PK) ky_absorb_key(sp, rho + (size_t)i * 32, tk + (size_t)i * kn, kn); else sp.absorb(tail, tail_len);
    sp.absorb(m + (size_t)i * 32, 32);
    sp.finalize(0x06);
    uint8_t md[32];
    for (int b = 0; b < 32; b++) md[b] = sp.squeeze_byte();
    Drbg d;
    d.init(md, 32, C, L);
    uint4 *o = reinterpret_cast<uint4 *>(krd + (size_t)i * 96);
    uint32_t w[4], mine[4] = { 0, 0, 0, 0 };
    d.counter += hh;                                                  // lane hh produces blocks hh, hh + 2, ...
    for (int b = (int)hh; b < 6; b += 2) {
        d.block(w, L); d.counter++;
        o[b] = make_uint4(w[0], w[1], w[2], w[3]);
        if (b == 2 || b == 3) { mine[0] = w[0]; mine[1] = w[1]; mine[2] = w[2]; mine[3] = w[3]; }
    }
    // r = blocks 2 (lane 0) and 3 (lane 1) of K_bar | r | d
    uint32_t r32[8];
#pragma unroll
    for (int e = 0; e < 4; e++) { const uint32_t other = partner_half(mine[e]); r32[e] = hh ? other : mine[e]; r32[4 + e] = hh ? mine[e] : other; }
    uint8_t r[32];
    for (int b = 0; b < 32; b++) r[b] = (uint8_t)(r32[b >> 2] >> (8 * (b & 3)));
    d.init(r, 32, C, L);
    d.counter += hh;
    uint4 *no = reinterpret_cast<uint4 *>(noise + (size_t)i * noise_stride);
    for (uint32_t b = hh; b < noise_blocks; b += 2) { d.block(w, L); d.counter++; no[b] = make_uint4(w[0], w[1], w[2], w[3]); }
}

// The input is K_bar (4 words) | u (k n / 4 words of four 16-bit coefficients) | v (n / 4 words) | d (4 words): whole 64-bit words, 29.3 blocks of nine for
// Kyber's k = 3.  The sponge's state stays in registers from block to block and the nine words of the NEXT block are requested before the current one is
// permuted.  (Measured: the same 517 us per 65536 values as the word-by-word form through the LDS image - at two wavefronts per SIMD the 31 permutations
// are issue-bound, not waiting for their input; and one lane per value at that batch, one wavefront per SIMD, is 4 % slower.)
__global__ __launch_bounds__(KHT) void ky_h2_kernel(uint32_t batch, const uint8_t *__restrict__ krd, const int32_t *__restrict__ u, const int32_t *__restrict__ v,
                                                    const uint8_t *__restrict__ dsrc, size_t d_stride, uint8_t *__restrict__ keys, KyP P)
{
    const uint32_t i0 = (blockIdx.x * KHT + threadIdx.x) >> 1;
    const int h = (int)(threadIdx.x & 1);
    if (blockIdx.x * (KHT / 2) >= batch) return;                      // (whole workgroup)
    const bool valid = i0 < batch;
    const uint32_t i = valid ? i0 : 0;                                // unconditional loads: a lane pair past the batch reads item 0 and stores nothing
    const int kn = P.k * P.n;
    const int nu = kn / 4, nv = P.n / 4, total = 4 + nu + nv + 4;
    const uint32_t *kb = reinterpret_cast<const uint32_t *>(krd + (size_t)i * 96) + h;
    const int2 *pu = reinterpret_cast<const int2 *>(u + (size_t)i * kn) + h, *pv = reinterpret_cast<const int2 *>(v + (size_t)i * P.n) + h;
    const uint8_t *dp = dsrc + (size_t)i * d_stride + 4 * h;
    auto word = [&](int idx) -> uint32_t {                            // half h of stream word idx (idx is the same for every lane: uniform branches)
        if (idx >= total) return 0u;
        if (idx < 4) return kb[2 * idx];
        if (idx < 4 + nu) { const int2 c = pu[2 * (idx - 4)]; return ((uint32_t)c.x & 0xFFFFu) | (((uint32_t)c.y & 0xFFFFu) << 16); }
        if (idx < 4 + nu + nv) { const int2 c = pv[2 * (idx - 4 - nu)]; return ((uint32_t)c.x & 0xFFFFu) | (((uint32_t)c.y & 0xFFFFu) << 16); }
        const uint8_t *q = dp + 8 * (idx - 4 - nu - nv);              // d: the ciphertext's tail in decapsulation - any alignment
        return (uint32_t)q[0] | ((uint32_t)q[1] << 8) | ((uint32_t)q[2] << 16) | ((uint32_t)q[3] << 24);
    };
    uint32_t x[25], cur[9], nx[9];
#pragma unroll
    for (int w = 0; w < 25; w++) x[w] = 0;
#pragma unroll
    for (int w = 0; w < 9; w++) cur[w] = word(w);
    int base = 0;
    for (; base + 9 <= total; base += 9) {
#pragma unroll
        for (int w = 0; w < 9; w++) nx[w] = word(base + 9 + w);
#pragma unroll
        for (int w = 0; w < 9; w++) x[w] ^= cur[w];
        keccak_permute_halves<false>(x, h);
#pragma unroll
        for (int w = 0; w < 9; w++) cur[w] = nx[w];
    }
    const int rem = total - base;                                     // 0..8 words of the last block, then the SHA-3 pad
#pragma unroll
    for (int w = 0; w < 9; w++) { x[w] ^= cur[w]; if (w == rem && h == 0) x[w] ^= 0x06u; }
    if (h) x[8] ^= 0x80u << 24;
    keccak_permute_halves<false>(x, h);
    if (valid) {
        uint8_t *o = keys + (size_t)i * 32 + 4 * h;                    // word w of the digest: bytes 8 w + 4 h .. + 3 from this lane
#pragma unroll
        for (int w = 0; w < 4; w++) { o[8 * w] = (uint8_t)x[w]; o[8 * w + 1] = (uint8_t)(x[w] >> 8); o[8 * w + 2] = (uint8_t)(x[w] >> 16); o[8 * w + 3] = (uint8_t)(x[w] >> 24); }
    }
}

// ciphertext = u (d_u bits per coefficient) | v (d_v bits) MSB first, then d (kyber_kem.c:888-896).  Eight coefficients of
// width d are exactly d bytes: one lane packs one such group (u groups first, then v groups, then the 32 bytes of d by the
// last 32 lanes of the message's range).
__global__ void ky_pack_kernel(uint32_t batch, const int32_t *__restrict__ u, const int32_t *__restrict__ v, const uint8_t *__restrict__ krd, uint8_t *__restrict__ ct, size_t ct_stride, KyP P)
{
    const uint32_t ug = (uint32_t)(P.k * P.n / 8), vg = (uint32_t)(P.n / 8), per = ug + vg + 32;
    const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= per) return;
    const uint32_t ubytes = ug * (uint32_t)P.d_u, vbytes = vg * (uint32_t)P.d_v;
    const bool isu = j < ug;
    const int d = isu ? P.d_u : P.d_v;
    KY_ITEMS(i, batch) {
    uint8_t *out = ct + (size_t)i * ct_stride;
    if (j >= ug + vg) { const uint32_t b = j - ug - vg; out[ubytes + vbytes + b] = krd[(size_t)i * 96 + 64 + b]; continue; }
    const int32_t *src = isu ? u + (size_t)i * P.k * P.n + (size_t)j * 8 : v + (size_t)i * P.n + (size_t)(j - ug) * 8;
    uint8_t *o = out + (isu ? (size_t)j * d : (size_t)ubytes + (size_t)(j - ug) * d);
    const int4 lo = *reinterpret_cast<const int4 *>(src), hi = *reinterpret_cast<const int4 *>(src + 4);
    const uint32_t c[8] = { (uint32_t)lo.x, (uint32_t)lo.y, (uint32_t)lo.z, (uint32_t)lo.w, (uint32_t)hi.x, (uint32_t)hi.y, (uint32_t)hi.z, (uint32_t)hi.w };
    uint32_t acc = 0; int nb = 0, w = 0;
#pragma unroll
    for (int e = 0; e < 8; e++) {
        acc = (acc << d) | (c[e] & ((1u << d) - 1u)); nb += d;
        while (nb >= 8) { o[w++] = (uint8_t)(acc >> (nb - 8)); nb -= 8; }
    }
    }
}

// kyber_kem.c:976-983: one lane unpacks the d bytes of a group of eight coefficients
__global__ void ky_unpack_kernel(uint32_t batch, const uint8_t *__restrict__ ct, size_t ct_stride, int32_t *__restrict__ u, int32_t *__restrict__ v, KyP P)
{
    const uint32_t ug = (uint32_t)(P.k * P.n / 8), vg = (uint32_t)(P.n / 8), per = ug + vg;
    const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= per) return;
    const bool isu = j < ug;
    const int d = isu ? P.d_u : P.d_v;
    KY_ITEMS(i, batch) {
    const uint8_t *src = ct + (size_t)i * ct_stride + (isu ? (size_t)j * d : (size_t)ug * P.d_u + (size_t)(j - ug) * d);
    int32_t *dst = isu ? u + (size_t)i * P.k * P.n + (size_t)j * 8 : v + (size_t)i * P.n + (size_t)(j - ug) * 8;
    uint32_t acc = 0, c[8]; int nb = 0, r = 0;
#pragma unroll
    for (int e = 0; e < 8; e++) {
        while (nb < d) { acc = (acc << 8) | src[r++]; nb += 8; }
        c[e] = (acc >> (nb - d)) & ((1u << d) - 1u); nb -= d;
    }
    *reinterpret_cast<int4 *>(dst) = make_int4((int32_t)c[0], (int32_t)c[1], (int32_t)c[2], (int32_t)c[3]);
    *reinterpret_cast<int4 *>(dst + 4) = make_int4((int32_t)c[4], (int32_t)c[5], (int32_t)c[6], (int32_t)c[7]);
    }
}

}  // namespace

namespace sc { const DrbgConst &drbg_const(); }

struct sc_hip_kyber {
    sc_hip_ctx *ctx;
    sc_hip_plan *plan;
    KyP P;
    bool have_pub, have_priv;
    uint32_t *d_A, *d_t, *d_s;            // internal order, Montgomery form: A^T-ready [k][k], t [1][k], s [1][k]
    uint32_t *d_As;                       // the same A^T rows with t as row k, plain values | Shoup companions (dil_yaw_kernel: u0 | v0 in one launch)
    uint32_t *d_Ss;                       // s as a 1 x k matrix in that form (decryption's s . u the same way)
    size_t cap;
    int32_t *r, *e1, *e2, *u0, *v0;
    uint32_t *yh;
    // KEM level
    bool have_kem;
    uint64_t *d_gstate; uint8_t *d_gtail; uint32_t gtail_len;      // G's per-key prefix: sponge state after its whole blocks + the rest
    size_t kem_cap;
    uint8_t *krd, *noise, *mbuf; int32_t *ku, *kv;
    // a key per message: the matrices / t / s of the batch in the kernels' internal Montgomery form (staged per call)
    size_t keys_cap; bool keys_priv;
    int32_t *pk_gen; uint32_t *pk_A, *pk_t, *pk_s;
    hipStream_t side; hipEvent_t ev_fork, ev_join;     // a key per item: the matrices are expanded on a stream of the handle's own beside the hashing of the messages
};

// the key operands of one CPA call: the handle's one key (strides 0) or a key per message
struct KyKeys {
    const uint32_t *A, *t, *s; size_t sA, st;      // internal order, Montgomery form; strides between items (0: one key for the batch)
    hipEvent_t ready;                               // non-null: A is being written on another stream; the first product that reads it waits for this
    bool nat;                                       // a key per item: A as the expansion wrote it ([col][row][n], natural order, plain), t and s the key pair's own
                                                    // NTT-domain arrays - read in place by the products (DilNtt::key_nat), the transform of the other operand carries the Montgomery factor
};

static int ky_ensure(sc_hip_kyber *h, size_t batch)
{
    if (batch <= h->cap) return SC_HIP_SUCCESS;
    void *old[] = { h->r, h->e1, h->e2, h->u0, h->v0, h->yh };
    for (void *p : old) if (p) (void)sc::wipe_free(p);
    h->r = h->e1 = h->e2 = h->u0 = h->v0 = nullptr; h->yh = nullptr; h->cap = 0;
    const size_t kn = (size_t)h->P.k * N;
    HIPCHK(hipMalloc((void **)&h->r, batch * kn * 4)); HIPCHK(hipMalloc((void **)&h->e1, batch * kn * 4)); HIPCHK(hipMalloc((void **)&h->e2, batch * N * 4));
    // u0: room for the fused product's [k + 1][n] rows per item
    HIPCHK(hipMalloc((void **)&h->u0, batch * (kn + N) * 4)); HIPCHK(hipMalloc((void **)&h->v0, batch * N * 4)); HIPCHK(hipMalloc((void **)&h->yh, batch * kn * 4));
    h->cap = batch;
    return SC_HIP_SUCCESS;
}

extern "C" int sc_hip_kyber_create(sc_hip_ctx_t *ctx, const sc_hip_kyber_params_t *params, sc_hip_kyber_t **out)
{
    if (!ctx || !params || !out) return fail("kyber_create: null argument");
    static_assert(sizeof(KyP) == sizeof(sc_hip_kyber_params_t), "parameter struct layout");
    KyP P; memcpy(&P, params, sizeof P);
    if (P.n != 256) return fail("kyber_create: n must be 256");
    if (P.k < 1 || P.k > 8 || P.eta < 1 || P.eta > 8 || P.d_u < 1 || P.d_u > 16 || P.d_v < 1 || P.d_v > 16 || P.q_norm < 0 || P.q_norm > 31)
        return fail("kyber_create: parameters out of range");
    sc_hip_plan *plan = nullptr;
    if (sc_hip_plan_create(ctx, P.q, P.n, &plan) != SC_HIP_SUCCESS) return SC_HIP_FAILURE;
    sc_hip_kyber *h = new sc_hip_kyber;
    memset(h, 0, sizeof *h);
    h->ctx = ctx; h->plan = plan; h->P = P;
    *out = h;
    return SC_HIP_SUCCESS;
}

extern "C" void sc_hip_kyber_destroy(sc_hip_kyber_t *h)
{
    if (!h) return;
    void *ptrs[] = { h->d_As, h->d_Ss, h->d_A, h->d_t, h->d_s, h->r, h->e1, h->e2, h->u0, h->v0, h->yh, h->d_gstate, h->d_gtail, h->krd, h->noise, h->mbuf, h->ku, h->kv, h->pk_gen, h->pk_A, h->pk_t, h->pk_s };
    for (void *p : ptrs) if (p) (void)sc::wipe_free(p);
    if (h->side) (void)hipStreamDestroy(h->side);
    if (h->ev_fork) (void)hipEventDestroy(h->ev_fork);
    if (h->ev_join) (void)hipEventDestroy(h->ev_join);
    sc_hip_plan_destroy(h->plan);
    delete h;
}

// natural-order NTT-domain polynomials (any representative) -> internal order, Montgomery form; rows/cols: transpose on the way
static int ky_load(sc_hip_kyber *h, const int32_t *host, size_t npoly, uint32_t **dst, int rows, int cols)
{
    const size_t bytes = npoly * N * 4;
    HIPCHK(hipSetDevice(h->ctx->device));
    DevBuf tb;
    if (tb.get(bytes)) return SC_HIP_FAILURE;
    int32_t *tmp = tb.as<int32_t>();
    if (!*dst) HIPCHK(hipMalloc((void **)dst, bytes));
    HIPCHK(sc::copy_in(tmp, host, bytes));
    hipLaunchKernelGGL(dil_to_internal_mont, dim3((unsigned)((npoly * N + 255) / 256)), dim3(256), 0, nullptr, tmp, *dst, npoly, h->plan->k[1], h->plan->r2, h->plan->r2p, rows, cols);
    return (hipDeviceSynchronize() == hipSuccess) ? SC_HIP_SUCCESS : fail("kyber key load: kernel failed");
}

// t_ntt: NTT(t) as the public key stores it (kyber_kem.c:636-650); Agen: the k*k matrix polynomials in the order the
// CSPRNG seeded with rho generates them (module_lwe.c:519-547) - encryption uses them transposed (:700-720)
extern "C" int sc_hip_kyber_set_pubkey(sc_hip_kyber_t *h, const int32_t *t_ntt, const int32_t *Agen)
{
    if (!h || !t_ntt || !Agen) return fail("kyber_set_pubkey: null argument");
    const int k = h->P.k;
    if (ky_load(h, Agen, (size_t)k * k, &h->d_A, k, k)) return SC_HIP_FAILURE;       // A_enc[i][j] = G[j*k + i]
    if (ky_load(h, t_ntt, (size_t)k, &h->d_t, 0, 0)) return SC_HIP_FAILURE;
    {
        // [A^T ; t] as one (k + 1) x k matrix in plain form with Shoup companions: ky_cpa_enc_core's products for this key are one dil_yaw_kernel launch
        const size_t np = (size_t)(k + 1) * k, words = np * N;
        std::vector<int32_t> stack(words);
        for (int i = 0; i < k; i++)
            for (int j = 0; j < k; j++) memcpy(&stack[((size_t)i * k + j) * N], Agen + ((size_t)j * k + i) * N, N * 4);      // A_enc[i][j] = G[j*k + i]
        memcpy(&stack[(size_t)k * k * N], t_ntt, (size_t)k * N * 4);
        DevBuf tb;
        if (tb.get(words * 4)) return SC_HIP_FAILURE;
        if (!h->d_As) HIPCHK(hipMalloc((void **)&h->d_As, 2 * words * 4));
        HIPCHK(sc::copy_in(tb.as<int32_t>(), stack.data(), words * 4));
        hipLaunchKernelGGL(dil_to_internal_shoup, dim3((unsigned)((words + 255) / 256)), dim3(256), 0, nullptr, tb.as<int32_t>(), h->d_As, h->d_As + words, np, h->plan->k[1]);
        if (hipDeviceSynchronize() != hipSuccess) return fail("kyber_set_pubkey: kernel failed");
    }
    h->have_pub = true;
    return SC_HIP_SUCCESS;
}

extern "C" int sc_hip_kyber_set_privkey(sc_hip_kyber_t *h, const int32_t *s_ntt)
{
    if (!h || !s_ntt) return fail("kyber_set_privkey: null argument");
    if (ky_load(h, s_ntt, (size_t)h->P.k, &h->d_s, 0, 0)) return SC_HIP_FAILURE;
    {
        const size_t words = (size_t)h->P.k * N;
        DevBuf tb;
        if (tb.get(words * 4)) return SC_HIP_FAILURE;
        if (!h->d_Ss) HIPCHK(hipMalloc((void **)&h->d_Ss, 2 * words * 4));
        HIPCHK(sc::copy_in(tb.as<int32_t>(), s_ntt, words * 4));
        hipLaunchKernelGGL(dil_to_internal_shoup, dim3((unsigned)((words + 255) / 256)), dim3(256), 0, nullptr, tb.as<int32_t>(), h->d_Ss, h->d_Ss + words, (size_t)h->P.k, h->plan->k[1]);
        const bool ok = hipDeviceSynchronize() == hipSuccess;
        (void)hipMemset(tb.as<int32_t>(), 0, words * 4);                   // the staging copy of the secret key
        if (!ok) return fail("kyber_set_privkey: kernel failed");
    }
    h->have_priv = true;
    return SC_HIP_SUCCESS;
}

extern "C" size_t sc_hip_kyber_noise_bytes(const sc_hip_kyber_t *h) { return h ? (size_t)(2 * h->P.k + 1) * N * h->P.eta / 4 : 0; }

static int ky_cpa_enc_core(sc_hip_kyber *h, const KyKeys &K, size_t batch, const uint8_t *d_m, const uint8_t *d_noise, size_t noise_stride, int32_t *d_u, int32_t *d_v, void *stream);
static int ky_cpa_dec_core(sc_hip_kyber *h, const KyKeys &K, size_t batch, const int32_t *d_u, const int32_t *d_v, uint8_t *d_m, void *stream);

extern "C" int sc_hip_kyber_cpa_enc(sc_hip_kyber_t *h, size_t batch, const uint8_t *d_m, const uint8_t *d_noise, size_t noise_stride,
                                    int32_t *d_u, int32_t *d_v, void *stream)
{
    if (!h) return fail("kyber_cpa_enc: null handle");
    if (!h->have_pub) return fail("kyber_cpa_enc: no public key loaded");
    if ((reinterpret_cast<uintptr_t>(d_u) | reinterpret_cast<uintptr_t>(d_v)) & 15) return fail("kyber_cpa_enc: u and v must be 16-byte aligned");
    const KyKeys K = { h->d_A, h->d_t, h->d_s, 0, 0, nullptr, false };
    return ky_cpa_enc_core(h, K, batch, d_m, d_noise, noise_stride, d_u, d_v, stream);
}

static int ky_cpa_enc_core(sc_hip_kyber *h, const KyKeys &K, size_t batch, const uint8_t *d_m, const uint8_t *d_noise, size_t noise_stride, int32_t *d_u, int32_t *d_v, void *stream)
{
    if (batch == 0) return SC_HIP_SUCCESS;
    if (batch > 0x7fffffffu / 4096) return fail("kyber_cpa_enc: batch too large");
    if (!d_m || !d_noise || !d_u || !d_v) return fail("kyber_cpa_enc: null device pointer");
    if (noise_stride < sc_hip_kyber_noise_bytes(h)) return fail("kyber_cpa_enc: noise_stride smaller than sc_hip_kyber_noise_bytes()");
    HIPCHK(hipSetDevice(h->ctx->device));
    if (ky_ensure(h, batch)) return SC_HIP_FAILURE;
    hipStream_t s = (hipStream_t)stream;
    const KyP &P = h->P;
    const uint32_t B = (uint32_t)batch;
    const int k = P.k;
    hipLaunchKernelGGL(ky_binomial_kernel, ky_grid((size_t)((2 * k + 1) * N / 4), batch), dim3(256), 0, s, d_noise, noise_stride, B, h->r, h->e1, h->e2, P);
    HIPCHK(hipGetLastError());
    const size_t kb = batch * (size_t)k * N * 4;
    bool fused = false;
    if (K.A == h->d_A && K.t == h->d_t && K.sA == 0 && !K.nat && h->d_As && k >= 2 && k <= 5) {
        // the handle's own key: NTT(r) -> [A^T ; t] NTT(r) -> INTT in one launch, NTT(r) never leaving the registers (the three separate kernels wrote it out
        // and read it back k + 1 times: 16 KB of traffic per operation against 7, 350 us against ~150 at batch 65536)
        DilNtt a = mlwe_base_args(h->plan, k + 1, k, 1 << 20);
        a.in = h->r; a.key = h->d_As; a.keyp = h->d_As + (size_t)(k + 1) * k * N; a.w = h->u0; a.w1 = nullptr; a.w_chunked = 0;
        fused = mlwe_launch_yaw(h->ctx, a, B, s) == SC_HIP_SUCCESS;
    }
    if (fused) {
        hipLaunchKernelGGL(ky_enc_finish_kernel, ky_grid((size_t)(k + 1) * N / 4, batch), dim3(256), 0, s, h->u0, (size_t)(k + 1) * N, h->u0 + (size_t)k * N, (size_t)(k + 1) * N,
                           h->e1, h->e2, d_m, B, d_u, d_v, P);
    } else {
        DilNtt a = mlwe_base_args(h->plan, k, k, 1 << 20);
        a.in = h->r; a.out_int = h->yh; a.yh = h->yh;
        if (K.nat ? mlwe_launch<DOP_KEYPREP>(h->ctx, a, B, k, s) : mlwe_launch<DOP_FWD_INT>(h->ctx, a, B, k, s)) return SC_HIP_FAILURE;
        a.key = K.A; a.key_item_stride = K.sA; a.w = h->u0; a.w1 = nullptr; a.key_nat = K.nat ? 2 : 0;
        if (K.ready) HIPCHK(hipStreamWaitEvent(s, K.ready, 0));
        if (mlwe_launch<DOP_MATVEC>(h->ctx, a, B, k, s)) return SC_HIP_FAILURE;
        a.key = K.t; a.key_item_stride = K.st; a.w = h->v0; a.kk = 1; a.key_nat = K.nat ? 1 : 0;
        if (mlwe_launch<DOP_MATVEC>(h->ctx, a, B, 1, s)) return SC_HIP_FAILURE;
        hipLaunchKernelGGL(ky_enc_finish_kernel, ky_grid((size_t)(k + 1) * N / 4, batch), dim3(256), 0, s, h->u0, (size_t)k * N, h->v0, (size_t)N, h->e1, h->e2, d_m, B, d_u, d_v, P);
        HIPCHK(hipMemsetAsync(h->yh, 0, kb, s));
    }
    HIPCHK(hipGetLastError());
    // the ephemeral secret r and the noise do not outlive the call (kyber_kem.c:1014, module_lwe.c:1397-1400)
    HIPCHK(hipMemsetAsync(h->r, 0, kb, s)); HIPCHK(hipMemsetAsync(h->e1, 0, kb / 4, s)); HIPCHK(hipMemsetAsync(h->e2, 0, batch * N, s));      // (e1, e2: bytes)
    return SC_HIP_SUCCESS;
}

extern "C" int sc_hip_kyber_cpa_dec(sc_hip_kyber_t *h, size_t batch, const int32_t *d_u, const int32_t *d_v, uint8_t *d_m, void *stream)
{
    if (!h) return fail("kyber_cpa_dec: null handle");
    if (!h->have_priv) return fail("kyber_cpa_dec: no private key loaded");
    const KyKeys K = { h->d_A, h->d_t, h->d_s, 0, 0, nullptr, false };
    return ky_cpa_dec_core(h, K, batch, d_u, d_v, d_m, stream);
}

static int ky_cpa_dec_core(sc_hip_kyber *h, const KyKeys &K, size_t batch, const int32_t *d_u, const int32_t *d_v, uint8_t *d_m, void *stream)
{
    if (batch == 0) return SC_HIP_SUCCESS;
    if (batch > 0x7fffffffu / 4096) return fail("kyber_cpa_dec: batch too large");
    if (!d_m || !d_u || !d_v) return fail("kyber_cpa_dec: null device pointer");
    HIPCHK(hipSetDevice(h->ctx->device));
    if (ky_ensure(h, batch)) return SC_HIP_FAILURE;
    hipStream_t s = (hipStream_t)stream;
    const KyP &P = h->P;
    const uint32_t B = (uint32_t)batch;
    const int k = P.k;
    const size_t nu = batch * (size_t)k * N;
    hipLaunchKernelGGL(ky_decompress_kernel, dim3((unsigned)((nu + 255) / 256)), dim3(256), 0, s, d_u, h->r, nu, P.d_u, P.q);
    HIPCHK(hipGetLastError());
    bool fused = false;
    if (K.s == h->d_s && K.st == 0 && !K.nat && h->d_Ss && k >= 2 && k <= 5) {     // the handle's own key: NTT(u) -> s . NTT(u) -> INTT in one launch (as ky_cpa_enc_core)
        DilNtt a = mlwe_base_args(h->plan, 1, k, 1 << 20);
        a.in = h->r; a.key = h->d_Ss; a.keyp = h->d_Ss + (size_t)k * N; a.w = h->v0; a.w1 = nullptr; a.w_chunked = 0;
        fused = mlwe_launch_yaw(h->ctx, a, B, s) == SC_HIP_SUCCESS;
    }
    if (!fused) {
        DilNtt a = mlwe_base_args(h->plan, 1, k, 1 << 20);
        a.in = h->r; a.out_int = h->yh; a.yh = h->yh;
        if (K.nat ? mlwe_launch<DOP_KEYPREP>(h->ctx, a, B, k, s) : mlwe_launch<DOP_FWD_INT>(h->ctx, a, B, k, s)) return SC_HIP_FAILURE;
        a.key = K.s; a.key_item_stride = K.st; a.w = h->v0; a.w1 = nullptr; a.key_nat = K.nat ? 1 : 0;
        if (mlwe_launch<DOP_MATVEC>(h->ctx, a, B, 1, s)) return SC_HIP_FAILURE;
    }
    hipLaunchKernelGGL(ky_dec_finish_kernel, dim3((unsigned)((batch * 32 + 255) / 256)), dim3(256), 0, s, d_v, h->v0, d_m, B, P);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemsetAsync(h->v0, 0, batch * N * 4, s));                   // s . u: with v it is the plaintext
    return SC_HIP_SUCCESS;
}


// ---- KEM level --------------------------------------------------------------------------------------------------------------

static size_t ky_noise_stride(const sc_hip_kyber *h) { return (sc_hip_kyber_noise_bytes(h) + 15) & ~(size_t)15; }

static int ky_kem_ensure(sc_hip_kyber *h, size_t batch)
{
    if (batch <= h->kem_cap) return SC_HIP_SUCCESS;
    void *old[] = { h->krd, h->noise, h->mbuf, h->ku, h->kv };
    for (void *p : old) if (p) (void)sc::wipe_free(p);
    h->krd = h->noise = h->mbuf = nullptr; h->ku = h->kv = nullptr; h->kem_cap = 0;
    const size_t kn = (size_t)h->P.k * N;
    HIPCHK(hipMalloc((void **)&h->krd, batch * 96)); HIPCHK(hipMalloc((void **)&h->noise, batch * ky_noise_stride(h))); HIPCHK(hipMalloc((void **)&h->mbuf, batch * 32));
    HIPCHK(hipMalloc((void **)&h->ku, batch * kn * 4)); HIPCHK(hipMalloc((void **)&h->kv, batch * N * 4));
    h->kem_cap = batch;
    return SC_HIP_SUCCESS;
}

// K-bar | r | d, the noise stream and the recovered message do not outlive an encapsulation / decapsulation (kyber_kem.c:1014)
static void ky_kem_wipe(sc_hip_kyber *h, size_t batch, hipStream_t s)
{
    (void)hipMemsetAsync(h->krd, 0, batch * 96, s);
    (void)hipMemsetAsync(h->noise, 0, batch * ky_noise_stride(h), s);
    (void)hipMemsetAsync(h->mbuf, 0, batch * 32, s);
}

// t: the public key's coefficient-domain vector as the reference stores it (first k*n words of sc->pubkey->key, kyber_kem.c:636-650),
// rho: its 32-byte seed; both only enter the G hash
extern "C" int sc_hip_kyber_set_kem_pubkey(sc_hip_kyber_t *h, const int32_t *t, const uint8_t *rho)
{
    if (!h || !t || !rho) return fail("kyber_set_kem_pubkey: null argument");
    HIPCHK(hipSetDevice(h->ctx->device));
    const size_t kn = (size_t)h->P.k * N, len = 32 + 2 * kn, full = len / 72 * 72;
    std::vector<uint8_t> pre(len);
    memcpy(pre.data(), rho, 32);
    for (size_t i = 0; i < kn; i++) { pre[32 + 2 * i] = (uint8_t)(t[i] & 0xFF); pre[33 + 2 * i] = (uint8_t)(t[i] >> 8); }
    DevBuf pb;
    if (pb.get(len)) return SC_HIP_FAILURE;
    uint8_t *d_pre = pb.as<uint8_t>();
    if (!h->d_gstate) HIPCHK(hipMalloc((void **)&h->d_gstate, 25 * 8));
    if (h->d_gtail) { (void)sc::wipe_free(h->d_gtail); h->d_gtail = nullptr; }
    HIPCHK(hipMalloc((void **)&h->d_gtail, len - full + 8));
    HIPCHK(sc::copy_in(d_pre, pre.data(), len));
    HIPCHK(sc::copy_in(h->d_gtail, pre.data() + full, len - full));
    h->gtail_len = (uint32_t)(len - full);
    hipLaunchKernelGGL(ky_gprefix_kernel, dim3(1), dim3(KHT), 0, nullptr, d_pre, (uint32_t)full, h->d_gstate);
    const int rc = (hipDeviceSynchronize() == hipSuccess) ? SC_HIP_SUCCESS : fail("kyber_set_kem_pubkey: kernel failed");
    h->have_kem = rc == SC_HIP_SUCCESS;
    return rc;
}

extern "C" size_t sc_hip_kyber_ct_bytes(const sc_hip_kyber_t *h)
{ return h ? ((size_t)h->P.k * N * h->P.d_u + (size_t)N * h->P.d_v) / 8 + 32 : 0; }

static int ky_kem_check(sc_hip_kyber *h, size_t batch, const char *who)
{
    if (!h) return fail(std::string(who) + ": null handle");
    if (!h->have_pub || !h->have_kem) return fail(std::string(who) + ": public key (set_pubkey + set_kem_pubkey) not loaded");
    if (((h->P.k * N * h->P.d_u + N * h->P.d_v) & 7) != 0) return fail(std::string(who) + ": ciphertext body is not a whole number of bytes");
    if (batch > 0x7fffffffu / 4096) return fail(std::string(who) + ": batch too large");
    return SC_HIP_SUCCESS;
}

// two lanes per message while the one-lane hash kernels would not give every SIMD a wavefront
static bool ky_two_lanes(const sc_hip_kyber *h, size_t batch) { return batch <= (size_t)h->ctx->cus * 4 * 64; }

// G -> (K_bar, r, d), noise; CPA encryption of m under r; into h->ku, h->kv
// (rho, tk non-null: a public key per message, K its staged operands)
static int ky_kem_core(sc_hip_kyber *h, size_t batch, const uint8_t *d_m, hipStream_t s, const KyKeys *K = nullptr, const uint8_t *rho = nullptr, const int32_t *tk = nullptr)
{
    const uint32_t B = (uint32_t)batch;
    const size_t ns = ky_noise_stride(h);
    const uint32_t nblk = (uint32_t)(sc_hip_kyber_noise_bytes(h) / 16);
    const int kn = h->P.k * N;
    const dim3 g2((2 * B + KHT - 1) / KHT), g1((B + KHT - 1) / KHT);
    if (rho) {
        if (ky_two_lanes(h, batch)) hipLaunchKernelGGL(ky_g2_kernel<true>, g2, dim3(KHT), 0, s, B, d_m, nullptr, nullptr, 0u, h->krd, h->noise, ns, nblk, sc::drbg_const(), rho, tk, kn);
        else hipLaunchKernelGGL(ky_g_kernel<true>, g1, dim3(KHT), 0, s, B, d_m, nullptr, nullptr, 0u, h->krd, h->noise, ns, nblk, sc::drbg_const(), rho, tk, kn);
    } else {
        if (ky_two_lanes(h, batch)) hipLaunchKernelGGL(ky_g2_kernel<false>, g2, dim3(KHT), 0, s, B, d_m, h->d_gstate, h->d_gtail, h->gtail_len, h->krd, h->noise, ns, nblk, sc::drbg_const(), nullptr, nullptr, kn);
        else hipLaunchKernelGGL(ky_g_kernel<false>, g1, dim3(KHT), 0, s, B, d_m, h->d_gstate, h->d_gtail, h->gtail_len, h->krd, h->noise, ns, nblk, sc::drbg_const(), nullptr, nullptr, kn);
    }
    HIPCHK(hipGetLastError());
    if (K) return ky_cpa_enc_core(h, *K, batch, d_m, h->noise, ns, h->ku, h->kv, s);
    return sc_hip_kyber_cpa_enc(h, batch, d_m, h->noise, ns, h->ku, h->kv, s);
}

// replaces: kyber_kem_encapsulation (kyber_kem.c:855-928); d_m [batch][32] = the bytes prng_mem(sc->prng_ctx[0], m, 32) returns
static int ky_encaps_core(sc_hip_kyber *h, size_t batch, const uint8_t *d_m, uint8_t *d_ct, size_t ct_stride, uint8_t *d_keys, void *stream, const KyKeys *K, const uint8_t *rho, const int32_t *tk);
static int ky_decaps_core(sc_hip_kyber *h, size_t batch, const uint8_t *d_ct, size_t ct_stride, uint8_t *d_keys, void *stream, const KyKeys *K, const uint8_t *rho, const int32_t *tk);

extern "C" int sc_hip_kyber_kem_encaps(sc_hip_kyber_t *h, size_t batch, const uint8_t *d_m, uint8_t *d_ct, size_t ct_stride, uint8_t *d_keys, void *stream)
{
    if (ky_kem_check(h, batch, "kyber_kem_encaps")) return SC_HIP_FAILURE;
    return ky_encaps_core(h, batch, d_m, d_ct, ct_stride, d_keys, stream, nullptr, nullptr, nullptr);
}

static int ky_encaps_core(sc_hip_kyber *h, size_t batch, const uint8_t *d_m, uint8_t *d_ct, size_t ct_stride, uint8_t *d_keys, void *stream, const KyKeys *K, const uint8_t *rho, const int32_t *tk)
{
    if (batch == 0) return SC_HIP_SUCCESS;
    if (!d_m || !d_ct || !d_keys) return fail("kyber_kem_encaps: null buffer");
    if (ct_stride < sc_hip_kyber_ct_bytes(h)) return fail("kyber_kem_encaps: ct_stride smaller than sc_hip_kyber_ct_bytes()");
    if ((sc_hip_kyber_noise_bytes(h) & 15) != 0) return fail("kyber_kem_encaps: noise length is not a multiple of 16");
    HIPCHK(hipSetDevice(h->ctx->device));
    if (ky_kem_ensure(h, batch)) return SC_HIP_FAILURE;
    hipStream_t s = (hipStream_t)stream;
    const uint32_t B = (uint32_t)batch;
    if (ky_kem_core(h, batch, d_m, s, K, rho, tk)) return SC_HIP_FAILURE;
    if (ky_two_lanes(h, batch)) hipLaunchKernelGGL(ky_h2_kernel, dim3((2 * B + KHT - 1) / KHT), dim3(KHT), 0, s, B, h->krd, h->ku, h->kv, h->krd + 64, (size_t)96, d_keys, h->P);
    else hipLaunchKernelGGL(ky_h_kernel, dim3((B + KHT - 1) / KHT), dim3(KHT), 0, s, B, h->krd, h->ku, h->kv, h->krd + 64, (size_t)96, d_keys, h->P);
    hipLaunchKernelGGL(ky_pack_kernel, ky_grid((size_t)((h->P.k + 1) * N / 8 + 32), batch), dim3(256), 0, s, B, h->ku, h->kv, h->krd, d_ct, ct_stride, h->P);
    HIPCHK(hipGetLastError());
    ky_kem_wipe(h, batch, s);
    return SC_HIP_SUCCESS;
}

// replaces: kyber_kem_decapsulation (kyber_kem.c:930-1040)
extern "C" int sc_hip_kyber_kem_decaps(sc_hip_kyber_t *h, size_t batch, const uint8_t *d_ct, size_t ct_stride, uint8_t *d_keys, void *stream)
{
    if (ky_kem_check(h, batch, "kyber_kem_decaps")) return SC_HIP_FAILURE;
    if (!h->have_priv) return fail("kyber_kem_decaps: no private key loaded");
    return ky_decaps_core(h, batch, d_ct, ct_stride, d_keys, stream, nullptr, nullptr, nullptr);
}

static int ky_decaps_core(sc_hip_kyber *h, size_t batch, const uint8_t *d_ct, size_t ct_stride, uint8_t *d_keys, void *stream, const KyKeys *K, const uint8_t *rho, const int32_t *tk)
{
    if (batch == 0) return SC_HIP_SUCCESS;
    if (!d_ct || !d_keys) return fail("kyber_kem_decaps: null buffer");
    if (ct_stride < sc_hip_kyber_ct_bytes(h)) return fail("kyber_kem_decaps: ct_stride smaller than sc_hip_kyber_ct_bytes()");
    HIPCHK(hipSetDevice(h->ctx->device));
    if (ky_kem_ensure(h, batch)) return SC_HIP_FAILURE;
    hipStream_t s = (hipStream_t)stream;
    const uint32_t B = (uint32_t)batch;
    hipLaunchKernelGGL(ky_unpack_kernel, ky_grid((size_t)((h->P.k + 1) * N / 8), batch), dim3(256), 0, s, B, d_ct, ct_stride, h->ku, h->kv, h->P);
    HIPCHK(hipGetLastError());
    if (K ? ky_cpa_dec_core(h, *K, batch, h->ku, h->kv, h->mbuf, s) : sc_hip_kyber_cpa_dec(h, batch, h->ku, h->kv, h->mbuf, s)) return SC_HIP_FAILURE;
    if (ky_kem_core(h, batch, h->mbuf, s, K, rho, tk)) return SC_HIP_FAILURE;
    const size_t body = sc_hip_kyber_ct_bytes(h) - 32;
    if (ky_two_lanes(h, batch)) hipLaunchKernelGGL(ky_h2_kernel, dim3((2 * B + KHT - 1) / KHT), dim3(KHT), 0, s, B, h->krd, h->ku, h->kv, d_ct + body, ct_stride, d_keys, h->P);
    else hipLaunchKernelGGL(ky_h_kernel, dim3((B + KHT - 1) / KHT), dim3(KHT), 0, s, B, h->krd, h->ku, h->kv, d_ct + body, ct_stride, d_keys, h->P);
    HIPCHK(hipGetLastError());
    ky_kem_wipe(h, batch, s);
    return SC_HIP_SUCCESS;
}


// ---- a key per message ------------------------------------------------------------------------------------------------------

// stage the batch's keys: every matrix from its rho (transposed for encryption, module_lwe.c:700-720), NTT(t) and NTT(s) into the
// kernels' internal Montgomery form
static int ky_stage_keys(sc_hip_kyber *h, size_t batch, const int32_t *d_t_ntt, const uint8_t *d_rho, const int32_t *d_s_ntt, hipStream_t s, KyKeys *K)
{
    const KyP &P = h->P;
    const int k = P.k;
    const size_t kk = (size_t)k * k, kn = (size_t)k * N;
    if (batch > h->keys_cap) {
        if (h->pk_gen) (void)hipFree(h->pk_gen);                         // (public: the expanded matrices)
        h->pk_gen = nullptr; h->keys_cap = 0;
        HIPCHK(hipMalloc((void **)&h->pk_gen, batch * kk * N * 4));
        h->keys_cap = batch;
    }
    // the matrix of every key from its rho; nothing is converted or copied: the products read the expansion's output and the key pair's own t_ntt / s_ntt
    // arrays in place (natural order, plain form - DilNtt::key_nat)
    // The expansion (cipher-bound, ~0.6 ms at 64 K keys) needs nothing but rho and is needed by nothing before the encryption's first product - behind
    // the hash G(pk, m) and the noise expansion, which are chains bound by their own length: it runs on a stream of the handle's own.
    if (!h->side) {
        HIPCHK(hipStreamCreateWithFlags(&h->side, hipStreamNonBlocking));
        HIPCHK(hipEventCreateWithFlags(&h->ev_fork, hipEventDisableTiming));
        HIPCHK(hipEventCreateWithFlags(&h->ev_join, hipEventDisableTiming));
    }
    HIPCHK(hipEventRecord(h->ev_fork, s));
    HIPCHK(hipStreamWaitEvent(h->side, h->ev_fork, 0));
    if (sc_hip_mlwe_expand_matrix(h->ctx, batch, d_rho, 32, 32, k * k, P.q, P.q_bits, h->pk_gen, h->side)) return SC_HIP_FAILURE;
    HIPCHK(hipEventRecord(h->ev_join, h->side));
    K->A = reinterpret_cast<const uint32_t *>(h->pk_gen); K->t = reinterpret_cast<const uint32_t *>(d_t_ntt); K->s = reinterpret_cast<const uint32_t *>(d_s_ntt);
    K->sA = kk * N; K->st = kn; K->nat = true; K->ready = h->ev_join;
    return SC_HIP_SUCCESS;
}

static int ky_keys_check(sc_hip_kyber *h, size_t batch, const char *who)
{
    if (!h) return fail(std::string(who) + ": null handle");
    if (((h->P.k * N * h->P.d_u + N * h->P.d_v) & 7) != 0) return fail(std::string(who) + ": ciphertext body is not a whole number of bytes");
    if (batch > 0x7fffffffu / 4096 / (size_t)h->P.k) return fail(std::string(who) + ": batch too large");
    return SC_HIP_SUCCESS;
}

// replaces: safecrypto_encapsulation -> kyber_kem_encapsulation (kyber_kem.c:855-928) for `batch` values EACH under its own public
// key, in the layouts sc_hip_kyber_keygen writes: d_t, d_t_ntt int32 [batch][k][n], d_rho [batch][32]
extern "C" int sc_hip_kyber_kem_encaps_keys(sc_hip_kyber_t *h, size_t batch, const int32_t *d_t, const int32_t *d_t_ntt, const uint8_t *d_rho,
                                            const uint8_t *d_m, uint8_t *d_ct, size_t ct_stride, uint8_t *d_keys, void *stream)
{
    if (ky_keys_check(h, batch, "kyber_kem_encaps_keys")) return SC_HIP_FAILURE;
    if (batch == 0) return SC_HIP_SUCCESS;
    if (!d_t || !d_t_ntt || !d_rho) return fail("kyber_kem_encaps_keys: null key buffer");
    HIPCHK(hipSetDevice(h->ctx->device));
    KyKeys K;
    if (ky_stage_keys(h, batch, d_t_ntt, d_rho, nullptr, (hipStream_t)stream, &K)) return SC_HIP_FAILURE;
    const int rc = ky_encaps_core(h, batch, d_m, d_ct, ct_stride, d_keys, stream, &K, d_rho, d_t);
    (void)hipStreamWaitEvent((hipStream_t)stream, h->ev_join, 0);     // (again, in case the core returned before its first product)
    return rc;
}

// replaces: safecrypto_decapsulation -> kyber_kem_decapsulation (kyber_kem.c:930-1040) for `batch` ciphertexts EACH under its own
// key pair (d_s_ntt int32 [batch][k][n] besides the public parts)
extern "C" int sc_hip_kyber_kem_decaps_keys(sc_hip_kyber_t *h, size_t batch, const int32_t *d_s_ntt, const int32_t *d_t, const int32_t *d_t_ntt, const uint8_t *d_rho,
                                            const uint8_t *d_ct, size_t ct_stride, uint8_t *d_keys, void *stream)
{
    if (ky_keys_check(h, batch, "kyber_kem_decaps_keys")) return SC_HIP_FAILURE;
    if (batch == 0) return SC_HIP_SUCCESS;
    if (!d_s_ntt || !d_t || !d_t_ntt || !d_rho) return fail("kyber_kem_decaps_keys: null key buffer");
    HIPCHK(hipSetDevice(h->ctx->device));
    KyKeys K;
    if (ky_stage_keys(h, batch, d_t_ntt, d_rho, d_s_ntt, (hipStream_t)stream, &K)) return SC_HIP_FAILURE;
    const int rc = ky_decaps_core(h, batch, d_ct, ct_stride, d_keys, stream, &K, d_rho, d_t);      // (no copy of the secret keys is made: they are read in place)
    (void)hipStreamWaitEvent((hipStream_t)stream, h->ev_join, 0);
    return rc;
}


// ---- key generation ---------------------------------------------------------------------------------------------------------

// replaces: safecrypto_keygen -> kyber_kem_keygen (kyber_kem.c:575-670) over kyber_cpa_keygen (module_lwe.c:1189-1266) for `batch`
// key pairs, key i drawing what the reference draws from sc->prng_ctx[0] out of a CTR-DRBG seeded with d_seeds[i] (rho, the noise
// of s and e, z - in that order; the matrix comes from a second generator seeded with rho, as in the reference).  Outputs in the
// layouts the key pair uses: s_ntt, t, t_ntt int32 [batch][k][n]; rho, z [batch][32].
extern "C" int sc_hip_kyber_keygen(sc_hip_kyber_t *h, size_t batch, const uint8_t *d_seeds, size_t seed_stride, size_t seed_len,
                                   int32_t *d_s_ntt, int32_t *d_t, int32_t *d_t_ntt, uint8_t *d_rho, uint8_t *d_z, void *stream)
{
    if (!h) return fail("kyber_keygen: null handle");
    if (batch == 0) return SC_HIP_SUCCESS;
    if (!d_seeds || !d_s_ntt || !d_t || !d_t_ntt || !d_rho || !d_z) return fail("kyber_keygen: null buffer");
    if (batch > 0x7fffffffu / 4096) return fail("kyber_keygen: batch too large");
    HIPCHK(hipSetDevice(h->ctx->device));
    if (ky_ensure(h, batch)) return SC_HIP_FAILURE;
    hipStream_t s = (hipStream_t)stream;
    const KyP &P = h->P;
    const uint32_t B = (uint32_t)batch;
    const int k = P.k;
    const size_t kn = (size_t)k * N, noise = (size_t)2 * k * P.eta * 64, slen = 64 + noise + 64, npoly = batch * (size_t)k * k;
    DevBuf sb, ab, am;
    if (sb.get(batch * slen) || ab.get(npoly * N * 4) || am.get(npoly * N * 4)) return SC_HIP_FAILURE;
    uint8_t *strm = sb.as<uint8_t>();
    if (sc_hip_drbg_mem(h->ctx, batch, d_seeds, seed_stride, seed_len, strm, slen, slen, s)) return SC_HIP_FAILURE;
    hipLaunchKernelGGL(ky_kg_slices_kernel, dim3((unsigned)((batch * 64 + 255) / 256)), dim3(256), 0, s, strm, slen, 64 + noise, B, d_rho, d_z);
    const size_t ng = batch * (size_t)(2 * k * N / 4);
    hipLaunchKernelGGL(ky_kg_binomial_kernel, dim3((unsigned)((ng + 255) / 256)), dim3(256), 0, s, strm, slen, (size_t)64, B, h->r, h->e1, P);
    HIPCHK(hipGetLastError());
    // every key its own matrix: expand from rho, bring into the kernels' internal Montgomery form (no transposition: t = A s)
    if (sc_hip_mlwe_expand_matrix(h->ctx, batch, d_rho, 32, 32, k * k, P.q, P.q_bits, ab.as<int32_t>(), s)) return SC_HIP_FAILURE;
    hipLaunchKernelGGL(dil_to_internal_mont, dim3((unsigned)((npoly * N + 255) / 256)), dim3(256), 0, s, ab.as<int32_t>(), am.as<uint32_t>(), npoly, h->plan->k[1], h->plan->r2, h->plan->r2p, 0, 0);
    HIPCHK(hipGetLastError());
    DilNtt a = mlwe_base_args(h->plan, k, k, 1 << 20);
    a.in = h->r; a.out_int = h->yh; a.yh = h->yh;
    if (mlwe_launch<DOP_FWD_INT>(h->ctx, a, B, k, s)) return SC_HIP_FAILURE;
    a.key = am.as<uint32_t>(); a.key_item_stride = (size_t)k * k * N; a.w = h->u0; a.w1 = nullptr;
    if (mlwe_launch<DOP_MATVEC>(h->ctx, a, B, k, s)) return SC_HIP_FAILURE;
    hipLaunchKernelGGL(ky_kg_t_kernel, dim3((unsigned)((batch * kn + 255) / 256)), dim3(256), 0, s, h->u0, h->e1, d_t, batch * kn, P);
    HIPCHK(hipGetLastError());
    // the stored NTT forms are in the reference's natural order, centred (kyber_kem.c:648-651, module_lwe.c:1256)
    if (sc_hip_ntt_fwd(h->plan, batch * (size_t)k, h->r, d_s_ntt, s) || sc_hip_center(h->plan, batch * kn, d_s_ntt, d_s_ntt, s)) return SC_HIP_FAILURE;
    if (sc_hip_ntt_fwd(h->plan, batch * (size_t)k, d_t, d_t_ntt, s) || sc_hip_center(h->plan, batch * kn, d_t_ntt, d_t_ntt, s)) return SC_HIP_FAILURE;
    // s, e and NTT(s) in the handle's scratch do not outlive the call (the encryption path, whose e1 / e2 are bytes, wipes only what it wrote)
    HIPCHK(hipMemsetAsync(h->r, 0, batch * kn * 4, s)); HIPCHK(hipMemsetAsync(h->e1, 0, batch * kn * 4, s)); HIPCHK(hipMemsetAsync(h->yh, 0, batch * kn * 4, s));
    HIPCHK(hipStreamSynchronize(s));                                   // the scoped temporaries are released on return
    return SC_HIP_SUCCESS;
}

// ---- host-buffer convenience (the shape safecrypto_encapsulation / safecrypto_decapsulation have, for a batch) ----

extern "C" int sc_hip_kyber_kem_encaps_host(sc_hip_kyber_t *h, size_t batch, const uint8_t *h_m, uint8_t *h_ct, size_t ct_stride, uint8_t *h_keys)
{
    if (!h) return fail("kyber_kem_encaps_host: null handle");
    if (batch == 0) return SC_HIP_SUCCESS;
    if (!h_m || !h_ct || !h_keys) return fail("kyber_kem_encaps_host: null buffer");
    HIPCHK(hipSetDevice(h->ctx->device));
    DevBuf m, ct, keys;
    if (m.get(batch * 32) || ct.get(batch * ct_stride + 8) || keys.get(batch * 32)) return SC_HIP_FAILURE;
    HIPCHK(sc::copy_in(m.p, h_m, batch * 32));
    HIPCHK(hipMemset(ct.p, 0, batch * ct_stride));
    const int rc = sc_hip_kyber_kem_encaps(h, batch, m.as<uint8_t>(), ct.as<uint8_t>(), ct_stride, keys.as<uint8_t>(), nullptr);
    if (rc != SC_HIP_SUCCESS) return rc;
    HIPCHK(hipStreamSynchronize(nullptr));
    HIPCHK(sc::copy_out(h_ct, ct.p, batch * ct_stride));
    HIPCHK(sc::copy_out(h_keys, keys.p, batch * 32));
    return SC_HIP_SUCCESS;
}

extern "C" int sc_hip_kyber_kem_decaps_host(sc_hip_kyber_t *h, size_t batch, const uint8_t *h_ct, size_t ct_stride, uint8_t *h_keys)
{
    if (!h) return fail("kyber_kem_decaps_host: null handle");
    if (batch == 0) return SC_HIP_SUCCESS;
    if (!h_ct || !h_keys) return fail("kyber_kem_decaps_host: null buffer");
    HIPCHK(hipSetDevice(h->ctx->device));
    DevBuf ct, keys;
    if (ct.get(batch * ct_stride + 8) || keys.get(batch * 32)) return SC_HIP_FAILURE;
    HIPCHK(sc::copy_in(ct.p, h_ct, batch * ct_stride));
    const int rc = sc_hip_kyber_kem_decaps(h, batch, ct.as<uint8_t>(), ct_stride, keys.as<uint8_t>(), nullptr);
    if (rc != SC_HIP_SUCCESS) return rc;
    HIPCHK(hipStreamSynchronize(nullptr));
    HIPCHK(sc::copy_out(h_keys, keys.p, batch * 32));
    return SC_HIP_SUCCESS;
}
