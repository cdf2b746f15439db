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
