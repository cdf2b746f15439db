/*
 * sc_oracle_kyber.c - CPU restatement of the reference's Kyber (round 1, q = 7681) CPA encryption / decryption.
 * TEST INFRASTRUCTURE ONLY (see sc_oracle.h).  Pinned: tests/test_oracle_kyber.py reproduces the reference's
 * ciphertexts coefficient for coefficient (tests/golden/kyber_ref.txt, made by oracle/ref_driver.c).
 * Paths are relative to the reference tree.  Inputs that the reference derives with its AES-CTR-DRBG (the matrix
 * from rho, the noise bytes from the seed r) are INPUTS here (the fixture carries them).
 */
#include "sc_oracle.h"
#include <stdlib.h>
#include <string.h>

typedef struct { int n, q, q_bits; uint32_t q_inv; int q_norm, k, eta, d_u, d_v, d_t, set, pad[5]; } kyp_t;

/* module_lwe.c:380-412 binomial_rand_sample_csprng: 4 coefficients from eta bytes (bit b of every byte summed) */
static void binomial(const uint8_t *bytes, int eta, int32_t *s, size_t count)
{
    for (size_t g = 0; g < count / 4; g++) {
        uint32_t a[8] = { 0 };
        for (int e = 0; e < eta; e++) for (int b = 0; b < 8; b++) a[b] += bytes[g * eta + e] & (1u << b);
        s[4 * g + 0] = (int32_t)(a[0]) - (int32_t)(a[1] >> 1);
        s[4 * g + 1] = (int32_t)(a[2] >> 2) - (int32_t)(a[3] >> 3);
        s[4 * g + 2] = (int32_t)(a[4] >> 4) - (int32_t)(a[5] >> 5);
        s[4 * g + 3] = (int32_t)(a[6] >> 6) - (int32_t)(a[7] >> 7);
    }
}

/* module_lwe.c:43-56 mlwe_compress, literally (the (UINT64)(SINT32) conversion wraps for negative inputs) */
static void compress(int32_t *v, size_t len, int d, uint32_t q, uint32_t q_inv, int q_norm)
{
    const uint32_t rnd = q >> 1, mask = (1u << d) - 1;
    for (size_t i = 0; i < len; i++) {
        int32_t t = (int32_t)((uint32_t)v[i] << d);
        t = (int32_t)((uint32_t)t + rnd);
        t = (int32_t)(((uint64_t)(int64_t)t * (uint64_t)q_inv) >> (32 + q_norm));
        v[i] = (int32_t)((uint32_t)t & mask);
    }
}
/* module_lwe.c:58-69 */
static void decompress(int32_t *v, size_t len, int d, uint32_t q) { for (size_t i = 0; i < len; i++) v[i] = (int32_t)(v[i] * (int32_t)q) >> d; }

/* t[i] = INTT( sum_j A(i,j) o NTT(y[j]) ), A(i,j) = G[j*k+i] (create_rand_product_16_csprng, TRANSPOSED, module_lwe.c:700-720)
 * or G[i*cols+j]; ntt_y receives NTT(y) (ENABLE_OVERWRITE) */
static void matvec(int32_t *out, const int32_t *G, int rows, int cols, int transposed, int32_t *yh, int n, int32_t q, const int32_t *w, const int32_t *r)
{
    int32_t *acc = malloc(sizeof(int32_t) * (size_t)n);
    for (int i = 0; i < rows; i++) {
        for (int e = 0; e < n; e++) {
            int64_t s = 0;
            for (int j = 0; j < cols; j++) {
                const int32_t a = transposed ? G[((size_t)j * rows + i) * n + e] : G[((size_t)i * cols + j) * n + e];
                s += (int64_t)yh[j * n + e] * a % q;
            }
            acc[e] = (int32_t)(s % q);
        }
        orc_inv_ntt(out + i * n, n, q, acc, w, r);
    }
    free(acc);
}

/* kyber_cpa_enc, module_lwe.c:1268-1401 (t in the NTT domain).  noise = (2k+1)*n*eta/4 bytes. u: [k][n], v: [n] compressed. */
int orc_kyber_cpa_enc(const int32_t *par, const int32_t *t_ntt, const int32_t *Agen, const uint8_t *m, const uint8_t *noise, int32_t *u, int32_t *v)
{
    kyp_t P; memcpy(&P, par, sizeof P);
    const int n = P.n, k = P.k, q = P.q;
    int32_t *w = malloc(sizeof(int32_t) * (size_t)n * (3 * k + 3)), *r = w + n, *re = r + n, *e1 = re + k * n, *e2 = e1 + k * n;
    if (orc_make_tables(q, n, w, r)) { free(w); return 1; }
    const size_t per = (size_t)k * n * P.eta / 4;
    binomial(noise, P.eta, re, (size_t)k * n);
    binomial(noise + per, P.eta, e1, (size_t)k * n);
    binomial(noise + 2 * per, P.eta, e2, (size_t)n);
    for (int j = 0; j < k; j++) { orc_fwd_ntt(re + j * n, n, q, re + j * n, w); orc_normalize(re + j * n, (size_t)n, q); }
    matvec(u, Agen, k, k, 1, re, n, q, w, r);
    for (int i = 0; i < k * n; i++) u[i] += e1[i];
    {
        int32_t *acc = malloc(sizeof(int32_t) * (size_t)n);
        for (int e = 0; e < n; e++) {
            int64_t s = 0;
            for (int i = 0; i < k; i++) { int64_t tt = t_ntt[i * n + e] % q; if (tt < 0) tt += q; s += (int64_t)re[i * n + e] * tt % q; }
            acc[e] = (int32_t)(s % q);
        }
        orc_inv_ntt(v, n, q, acc, w, r);
        free(acc);
    }
    for (int i = 0; i < n / 8; i++) for (int b = 0; b < 8; b++) v[8 * i + b] += (m[i] & (0x80 >> b)) ? (q >> 1) : 0;
    for (int i = 0; i < n; i++) v[i] += e2[i];
    compress(u, (size_t)k * n, P.d_u, (uint32_t)q, P.q_inv, P.q_norm);
    compress(v, (size_t)n, P.d_v, (uint32_t)q, P.q_inv, P.q_norm);
    free(w);
    return 0;
}

/* kyber_cpa_dec, module_lwe.c:1415-1481 (s in the NTT domain); u, v are not modified */
int orc_kyber_cpa_dec(const int32_t *par, const int32_t *s_ntt, const int32_t *u_in, const int32_t *v_in, uint8_t *m)
{
    kyp_t P; memcpy(&P, par, sizeof P);
    const int n = P.n, k = P.k, q = P.q;
    int32_t *w = malloc(sizeof(int32_t) * (size_t)n * (k + 5)), *r = w + n, *u = r + n, *v = u + k * n, *acc = v + n, *x = acc + n;
    if (orc_make_tables(q, n, w, r)) { free(w); return 1; }
    memcpy(u, u_in, sizeof(int32_t) * (size_t)k * n); memcpy(v, v_in, sizeof(int32_t) * (size_t)n);
    decompress(u, (size_t)k * n, P.d_u, (uint32_t)q);
    decompress(v, (size_t)n, P.d_v, (uint32_t)q);
    for (int i = 0; i < k; i++) { orc_fwd_ntt(u + i * n, n, q, u + i * n, w); orc_normalize(u + i * n, (size_t)n, q); }
    for (int e = 0; e < n; e++) {
        int64_t s = 0;
        for (int i = 0; i < k; i++) { int64_t ss = s_ntt[i * n + e] % q; if (ss < 0) ss += q; s += (int64_t)u[i * n + e] * ss % q; }
        acc[e] = (int32_t)(s % q);
    }
    orc_inv_ntt(x, n, q, acc, w, r);
    for (int i = 0; i < n; i++) v[i] -= x[i];
    compress(v, (size_t)n, 1, (uint32_t)q, P.q_inv, P.q_norm);
    for (int i = 0; i < n / 8; i++) { uint8_t b = 0; for (int j = 0; j < 8; j++) b |= (uint8_t)(v[8 * i + j] << (7 - j)); m[i] = b; }
    free(w);
    return 0;
}


/* ---- KEM level: kyber_kem_encapsulation / kyber_kem_decapsulation (src/schemes/kem/kyber/kyber_kem.c:690-800, 855-1040) ----
 * Build options of the reference: hash = SHA3-512 (its permutation), oracle bytes and noise through the AES CTR-DRBG
 * (KYBER_KEM_USE_CSPRNG_SAM, no KYBER_KEM_USE_H_FUNC_XOF), NTT(t) and NTT(s) stored in the keys. */

/* kem_g_function (:757-800) + kyber_oracle_bytes (:745-755): SHA3-512(rho | t as 16-bit LE | m) -> md; a CSPRNG seeded with
 * md[0..32) yields 96 bytes = K_bar | r | d */
static void kem_g(const int32_t *par, const uint8_t *rho, const int32_t *t, const uint8_t *m, uint8_t *krd)
{
    const int n = par[0], k = par[5];
    const size_t len = 32 + (size_t)k * n * 2 + 32;
    uint8_t *b = malloc(len), *p = b, md[64];
    memcpy(p, rho, 32); p += 32;
    for (int i = 0; i < k * n; i++) { *p++ = (uint8_t)(t[i] & 0xFF); *p++ = (uint8_t)(t[i] >> 8); }
    memcpy(p, m, 32);
    orc_sha3(64, b, len, md);
    orc_drbg_t *d = orc_drbg_create(md, 32);
    orc_drbg_mem(d, krd, 96);
    orc_drbg_destroy(d);
    free(b);
}

/* kem_h_function (:690-743): SHA3-512(K_bar | u, v as 16-bit LE | d) */
static void kem_h(const int32_t *par, const uint8_t *kbar, const int32_t *u, const int32_t *v, const uint8_t *dd, uint8_t *md)
{
    const int n = par[0], k = par[5];
    const size_t len = 32 + (size_t)(k + 1) * n * 2 + 32;
    uint8_t *b = malloc(len), *p = b;
    memcpy(p, kbar, 32); p += 32;
    for (int i = 0; i < k * n; i++) { *p++ = (uint8_t)(u[i] & 0xFF); *p++ = (uint8_t)(u[i] >> 8); }
    for (int i = 0; i < n; i++) { *p++ = (uint8_t)(v[i] & 0xFF); *p++ = (uint8_t)(v[i] >> 8); }
    memcpy(p, dd, 32);
    orc_sha3(64, b, len, md);
    free(b);
}

/* the bytes kyber_cpa_enc draws from create_csprng(sc, r, 32): one prng_mem(eta*64) per polynomial of r, e1, e2 */
static void noise_from_seed(const int32_t *par, const uint8_t *r, uint8_t *noise)
{
    const int n = par[0], k = par[5], eta = par[6];
    orc_drbg_t *d = orc_drbg_create(r, 32);
    for (int i = 0; i < 2 * k + 1; i++) orc_drbg_mem(d, noise + (size_t)i * eta * (n / 4), eta * (n / 4));
    orc_drbg_destroy(d);
}

size_t orc_kyber_ct_bytes(const int32_t *par) { return ((size_t)par[5] * par[0] * par[7] + (size_t)par[0] * par[8]) / 8 + 32; }

/* t = the public key's coefficient-domain polynomial vector as stored (first k*n words of sc->pubkey->key); m = the 32 bytes
 * the reference draws with prng_mem(sc->prng_ctx[0]).  ct: u (d_u bits each) | v (d_v bits) | d, MSB-first; key: 32 bytes. */
int orc_kyber_kem_enc(const int32_t *par, const int32_t *t, const int32_t *t_ntt, const uint8_t *rho, const int32_t *Agen, const uint8_t *m,
                      uint8_t *ct, uint8_t *key)
{
    const int n = par[0], k = par[5], du = par[7], dv = par[8];
    uint8_t krd[96], md[64], noise[8192];
    int32_t *u = malloc(sizeof(int32_t) * (size_t)(k + 1) * n), *v = u + (size_t)k * n;
    kem_g(par, rho, t, m, krd);
    noise_from_seed(par, krd + 32, noise);
    if (orc_kyber_cpa_enc(par, t_ntt, Agen, m, noise, u, v)) { free(u); return 1; }
    kem_h(par, krd, u, v, krd + 64, md);
    size_t bp = 0;
    memset(ct, 0, orc_kyber_ct_bytes(par));
    for (int i = 0; i < k * n; i++) for (int b = du - 1; b >= 0; b--, bp++) ct[bp >> 3] |= (uint8_t)((((uint32_t)u[i] >> b) & 1u) << (7 - (bp & 7)));
    for (int i = 0; i < n; i++) for (int b = dv - 1; b >= 0; b--, bp++) ct[bp >> 3] |= (uint8_t)((((uint32_t)v[i] >> b) & 1u) << (7 - (bp & 7)));
    memcpy(ct + (bp >> 3), krd + 64, 32);
    memcpy(key, md, 32);
    free(u);
    return 0;
}

/* kyber_kem_decapsulation (:930-1040): m' = dec(u, v); (K', r', d') = G(rho, t, m'); (u', v') = enc(m', r'); key = H(K', u', v', d)
 * with d taken from the ciphertext (this version neither compares the re-encryption nor substitutes z) */
int orc_kyber_kem_dec(const int32_t *par, const int32_t *s_ntt, const int32_t *t, const int32_t *t_ntt, const uint8_t *rho, const int32_t *Agen,
                      const uint8_t *ct, uint8_t *key)
{
    const int n = par[0], k = par[5], du = par[7], dv = par[8];
    uint8_t krd[96], md[64], noise[8192], m[32];
    int32_t *u = malloc(sizeof(int32_t) * (size_t)(k + 1) * n), *v = u + (size_t)k * n;
    size_t bp = 0;
    for (int i = 0; i < k * n; i++) { uint32_t x = 0; for (int b = 0; b < du; b++, bp++) x = (x << 1) | ((ct[bp >> 3] >> (7 - (bp & 7))) & 1u); u[i] = (int32_t)x; }
    for (int i = 0; i < n; i++) { uint32_t x = 0; for (int b = 0; b < dv; b++, bp++) x = (x << 1) | ((ct[bp >> 3] >> (7 - (bp & 7))) & 1u); v[i] = (int32_t)x; }
    const uint8_t *dd = ct + (bp >> 3);
    if (orc_kyber_cpa_dec(par, s_ntt, u, v, m)) { free(u); return 1; }
    kem_g(par, rho, t, m, krd);
    noise_from_seed(par, krd + 32, noise);
    if (orc_kyber_cpa_enc(par, t_ntt, Agen, m, noise, u, v)) { free(u); return 1; }
    kem_h(par, krd, u, v, dd, md);
    memcpy(key, md, 32);
    free(u);
    return 0;
}


/* kyber_kem_keygen (kyber_kem.c:575-670) over kyber_cpa_keygen (module_lwe.c:1189-1266), with the context CSPRNG seeded by `seed`:
 * stream = rho block (64 bytes, 32 used) | k noise blocks for s | k noise blocks for e | z block (64 bytes, 32 used).
 * Outputs as the key pair stores them: s_ntt = centre(NTT(s)); t = normalise(decompress(compress(centre(A s + e), d_t)));
 * t_ntt = centre(NTT(t)); rho; z. */
int orc_kyber_keygen(const int32_t *par, const uint8_t *seed, size_t seed_len, int32_t *s_ntt, int32_t *t, int32_t *t_ntt, uint8_t *rho, uint8_t *z)
{
    kyp_t P; memcpy(&P, par, sizeof P);
    const int n = P.n, k = P.k, q = P.q;
    int32_t *w = malloc(sizeof(int32_t) * (size_t)n * (2 + 2 * k + k * k)), *r = w + n, *sh = r + n, *e = sh + k * n, *G = e + k * n;
    uint8_t blk[64], noise[1024];
    if (orc_make_tables(q, n, w, r)) { free(w); return 1; }
    orc_drbg_t *d = orc_drbg_create(seed, seed_len);
    orc_drbg_mem(d, blk, 32); memcpy(rho, blk, 32);
    for (int j = 0; j < k; j++) { orc_drbg_mem(d, noise, P.eta * 64); binomial(noise, P.eta, sh + j * n, (size_t)n); }
    for (int j = 0; j < k; j++) { orc_drbg_mem(d, noise, P.eta * 64); binomial(noise, P.eta, e + j * n, (size_t)n); }
    orc_expand_matrix(rho, 32, k * k, q, P.q_bits, G);
    for (int j = 0; j < k; j++) { orc_fwd_ntt(sh + j * n, n, q, sh + j * n, w); orc_normalize(sh + j * n, (size_t)n, q); }
    matvec(t, G, k, k, 0, sh, n, q, w, r);
    for (int i = 0; i < k * n; i++) { s_ntt[i] = sh[i]; t[i] += e[i]; }
    orc_center(s_ntt, (size_t)k * n, q);
    orc_center(t, (size_t)k * n, q);
    compress(t, (size_t)k * n, P.d_t, (uint32_t)q, P.q_inv, P.q_norm);
    decompress(t, (size_t)k * n, P.d_t, (uint32_t)q);
    orc_normalize(t, (size_t)k * n, q);
    for (int j = 0; j < k; j++) orc_fwd_ntt(t_ntt + j * n, n, q, t + j * n, w);
    orc_center(t_ntt, (size_t)k * n, q);
    orc_drbg_mem(d, blk, 32); memcpy(z, blk, 32);
    orc_drbg_destroy(d);
    free(w);
    return 0;
}
