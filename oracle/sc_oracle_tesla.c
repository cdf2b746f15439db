/*
 * sc_oracle_tesla.c - CPU restatement of the reference's ring-TESLA sign / verify.  TEST INFRASTRUCTURE ONLY.
 *
 * Follows src/schemes/sig/ring_tesla/ring_tesla.c:542-595 (test_w, test_rejection), :752-910 (sign), :922-1040 (verify) and
 * src/schemes/sig/ring_tesla/random_oracle.c (poly_rounding, oracle, random_stream = Salsa20 keyed with the digest, populate_c,
 * f_function), raw signature coding.  Randomness: n prng_32 words per attempt from sc->prng_ctx[0] (the tape).
 *
 * Behaviour kept as it is: poly_rounding computes (p - modn(p)) >> d with modn over the ring modulus q, which is 0 for every
 * centred / normalised coefficient - so the oracle hashes 8n zero bytes followed by the message, and verification does not
 * depend on w1, w2 (a signature whose z is altered within the rejection bound still verifies).  Parity, not repair.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include "sc_oracle.h"
#include "sc_oracle_keccak.h"

typedef struct { int n, n_bits, q, q_bits, b, b_bits, omega, d, bound, u, set, pad; } teslap_t;

#define ROTL(x, k) (((x) << (k)) | ((x) >> (32 - (k))))
/* salsa20_core of src/utils/crypto/salsa/salsa20.c:46-94: 10 double rounds, feed-forward - and the result stored with
 * SC_BIG_ENDIAN_32_COPY, i.e. every output word BYTE-SWAPPED on a little-endian host (not the Salsa20 specification's order) */
static void salsa20_core(const uint32_t in[16], uint8_t out[64])
{
    uint32_t x[16];
    memcpy(x, in, 64);
    for (int i = 0; i < 10; i++) {
        x[ 4] ^= ROTL(x[ 0] + x[12],  7); x[ 8] ^= ROTL(x[ 4] + x[ 0],  9); x[12] ^= ROTL(x[ 8] + x[ 4], 13); x[ 0] ^= ROTL(x[12] + x[ 8], 18);
        x[ 9] ^= ROTL(x[ 5] + x[ 1],  7); x[13] ^= ROTL(x[ 9] + x[ 5],  9); x[ 1] ^= ROTL(x[13] + x[ 9], 13); x[ 5] ^= ROTL(x[ 1] + x[13], 18);
        x[14] ^= ROTL(x[10] + x[ 6],  7); x[ 2] ^= ROTL(x[14] + x[10],  9); x[ 6] ^= ROTL(x[ 2] + x[14], 13); x[10] ^= ROTL(x[ 6] + x[ 2], 18);
        x[ 3] ^= ROTL(x[15] + x[11],  7); x[ 7] ^= ROTL(x[ 3] + x[15],  9); x[11] ^= ROTL(x[ 7] + x[ 3], 13); x[15] ^= ROTL(x[11] + x[ 7], 18);
        x[ 1] ^= ROTL(x[ 0] + x[ 3],  7); x[ 2] ^= ROTL(x[ 1] + x[ 0],  9); x[ 3] ^= ROTL(x[ 2] + x[ 1], 13); x[ 0] ^= ROTL(x[ 3] + x[ 2], 18);
        x[ 6] ^= ROTL(x[ 5] + x[ 4],  7); x[ 7] ^= ROTL(x[ 6] + x[ 5],  9); x[ 4] ^= ROTL(x[ 7] + x[ 6], 13); x[ 5] ^= ROTL(x[ 4] + x[ 7], 18);
        x[11] ^= ROTL(x[10] + x[ 9],  7); x[ 8] ^= ROTL(x[11] + x[10],  9); x[ 9] ^= ROTL(x[ 8] + x[11], 13); x[10] ^= ROTL(x[ 9] + x[ 8], 18);
        x[12] ^= ROTL(x[15] + x[14],  7); x[13] ^= ROTL(x[12] + x[15],  9); x[14] ^= ROTL(x[13] + x[12], 13); x[15] ^= ROTL(x[14] + x[13], 18);
    }
    for (int i = 0; i < 16; i++) { const uint32_t v = x[i] + in[i]; out[4 * i] = (uint8_t)(v >> 24); out[4 * i + 1] = (uint8_t)(v >> 16); out[4 * i + 2] = (uint8_t)(v >> 8); out[4 * i + 3] = (uint8_t)v; }
}

static uint32_t le32(const uint8_t *p) { return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24); }

/* random_oracle.c:68-134 (800 bytes, nonce 1..8, 64-bit block counter) + populate_c (:136-160) */
static void f_function(const teslap_t *P, const uint8_t *md, int32_t *c)
{
    static const uint8_t sigma[16] = "expand 32-byte k";
    uint8_t r[832], in[16] = { 1, 2, 3, 4, 5, 6, 7, 8, 0, 0, 0, 0, 0, 0, 0, 0 };
    for (int off = 0; off < 800; off += 64) {
        uint32_t st[16];
        st[0] = le32(sigma); st[5] = le32(sigma + 4); st[10] = le32(sigma + 8); st[15] = le32(sigma + 12);
        for (int i = 0; i < 4; i++) { st[1 + i] = le32(md + 4 * i); st[11 + i] = le32(md + 16 + 4 * i); st[6 + i] = le32(in + 4 * i); }
        salsa20_core(st, r + off);
        uint32_t u = 1;
        for (int i = 8; i < 16; i++) { u += in[i]; in[i] = (uint8_t)u; u >>= 8; }
    }
    uint8_t *used = calloc((size_t)P->n, 1);
    int cnt = 0, i = 0;
    while (i < P->omega) {
        int pos = (r[cnt] << 8) | r[cnt + 1];
        cnt += 2;
        pos &= P->n - 1;
        if (!used[pos]) { used[pos] = 1; c[i++] = pos; }
    }
    free(used);
}

/* random_oracle.c:28-66: SHA3-512 (the reference's permutation) over the big-endian words of the two "rounded" polynomials, then m */
static void tesla_oracle(const teslap_t *P, const int32_t *v1, const int32_t *v2, const uint8_t *m, size_t mlen, uint8_t *md)
{
    shake_t h;
    sha3_init(&h, 64);
    for (int k = 0; k < 2; k++) {
        const int32_t *v = k ? v2 : v1;
        for (int i = 0; i < P->n; i++) {
            const int32_t t = orc_modn(v[i], P->q);
            const int32_t rr = (v[i] - t) >> P->d;
            const uint8_t be[4] = { (uint8_t)((uint32_t)rr >> 24), (uint8_t)((uint32_t)rr >> 16), (uint8_t)((uint32_t)rr >> 8), (uint8_t)rr };
            shake_absorb(&h, be, 4);
        }
    }
    shake_absorb(&h, m, mlen);
    sha3_final(&h, md, 64);
}

static int test_w(const teslap_t *P, const int32_t *w)           /* ring_tesla.c:542-578 */
{
    const int32_t d_mask = (1 << P->d) - 1, d_sub = 1 << (P->d - 1), right = d_sub - P->bound;
    for (int i = 0; i < P->n; i++) {
        int32_t val = orc_modn(w[i], P->q);
        if (val < 0) val += P->q;
        int32_t left = (val & d_mask) - d_sub + 1;
        if (left < 0) left = -left;
        if (left > right) return 1;
    }
    return 0;
}

static int test_rejection(const teslap_t *P, const int32_t *z)  /* :580-595 */
{
    const int32_t th = P->b - P->u;
    for (int i = P->n; i--;) if (z[i] < -th || z[i] > th) return 1;
    return 0;
}

size_t orc_tesla_sig_bytes(const int32_t *par) { return ((size_t)par[0] * par[3] + 7) / 8 + 64; }

/* par = {n, n_bits, q, q_bits, b, b_bits, omega, d, bound, u, set, 0}; a1, a2: the reference's public constants (NTT domain);
 * s, e1, e2: the private key (SINT16).  Returns 0 and the signature (z packed q_bits-bit signed MSB-first, then the 64-byte
 * digest), the attempts and the words consumed; 2 = tape exhausted. */
int orc_tesla_sign(const int32_t *par, const int32_t *a1, const int32_t *a2, const int16_t *s, const int16_t *e1, const int16_t *e2,
                   const uint8_t *msg, size_t mlen, const uint32_t *tape, size_t tape_len, uint8_t *sig, size_t *siglen, int32_t *trials, size_t *consumed)
{
    teslap_t P; memcpy(&P, par, sizeof P);
    const int n = P.n, q = P.q;
    int32_t *w = malloc(sizeof(int32_t) * (size_t)n * 9), *r = w + n, *v1 = r + n, *v2 = v1 + n, *y = v2 + n, *z = y + n, *yh = z + n, *t32 = yh + n, *c = t32 + n;
    uint8_t md[64];
    size_t pos = 0;
    int rc = 2;
    if (orc_make_tables(q, n, w, r)) { free(w); return 1; }
    for (int iter = 0; iter < 9999; iter++) {
        if (pos + (size_t)n > tape_len) break;
        (*trials)++;
        for (int i = 0; i < n; i++) y[i] = (int32_t)(tape[pos++] & ((1u << (P.b_bits + 1)) - 1)) - P.b;
        orc_fwd_ntt(yh, n, q, y, w);
        orc_mul_pointwise(v1, n, q, yh, a1); orc_inv_ntt(v1, n, q, v1, w, r);
        orc_mul_pointwise(v2, n, q, yh, a2); orc_inv_ntt(v2, n, q, v2, w, r);
        orc_center(v1, (size_t)n, q); orc_center(v2, (size_t)n, q);
        tesla_oracle(&P, v1, v2, msg, mlen, md);
        f_function(&P, md, c);
        for (int i = 0; i < n; i++) t32[i] = e1[i];
        orc_mult_sparse(z, n, P.omega, t32, c);
        for (int i = 0; i < n; i++) v1[i] -= z[i];
        orc_normalize(v1, (size_t)n, q);
        if (test_w(&P, v1)) continue;
        for (int i = 0; i < n; i++) t32[i] = e2[i];
        orc_mult_sparse(z, n, P.omega, t32, c);
        for (int i = 0; i < n; i++) v2[i] -= z[i];
        orc_normalize(v2, (size_t)n, q);
        if (test_w(&P, v2)) continue;
        for (int i = 0; i < n; i++) t32[i] = s[i];
        orc_mult_sparse(z, n, P.omega, t32, c);
        for (int i = 0; i < n; i++) z[i] += y[i];
        orc_center(z, (size_t)n, q);
        if (test_rejection(&P, z)) continue;
        size_t bp = 0;
        memset(sig, 0, orc_tesla_sig_bytes(par));
        for (int i = 0; i < n; i++) for (int b = P.q_bits - 1; b >= 0; b--, bp++) sig[bp >> 3] |= (uint8_t)((((uint32_t)z[i] >> b) & 1u) << (7 - (bp & 7)));
        memcpy(sig + (bp + 7) / 8, md, 64);
        *siglen = (bp + 7) / 8 + 64;
        rc = 0;
        break;
    }
    *consumed = pos;
    free(w);
    return rc;
}

/* ring_tesla.c:922-1040.  t1, t2: the public key.  Returns 0 for a valid signature. */
int orc_tesla_verify(const int32_t *par, const int32_t *a1, const int32_t *a2, const int32_t *t1, const int32_t *t2,
                     const uint8_t *msg, size_t mlen, const uint8_t *sig, size_t siglen)
{
    teslap_t P; memcpy(&P, par, sizeof P);
    const int n = P.n, q = P.q;
    if (siglen < orc_tesla_sig_bytes(par)) return 1;
    int32_t *w = malloc(sizeof(int32_t) * (size_t)n * 7), *r = w + n, *z = r + n, *zh = z + n, *w1 = zh + n, *w2 = w1 + n, *t = w2 + n;
    int32_t c[64];
    uint8_t md[64];
    int rc = 1;
    if (orc_make_tables(q, n, w, r)) goto done;
    size_t bp = 0;
    for (int i = 0; i < n; i++) { uint32_t v = 0; for (int b = 0; b < P.q_bits; b++, bp++) v = (v << 1) | ((sig[bp >> 3] >> (7 - (bp & 7))) & 1u); z[i] = (int32_t)(v << (32 - P.q_bits)) >> (32 - P.q_bits); }
    const uint8_t *sigmd = sig + (bp + 7) / 8;
    if (test_rejection(&P, z)) goto done;
    f_function(&P, sigmd, c);
    orc_fwd_ntt(zh, n, q, z, w);
    orc_mul_pointwise(w1, n, q, zh, a1); orc_inv_ntt(w1, n, q, w1, w, r);
    orc_mult_sparse(t, n, P.omega, t1, c);
    for (int i = 0; i < n; i++) w1[i] -= t[i];
    orc_normalize(w1, (size_t)n, q);
    orc_mul_pointwise(w2, n, q, zh, a2); orc_inv_ntt(w2, n, q, w2, w, r);
    orc_mult_sparse(t, n, P.omega, t2, c);
    for (int i = 0; i < n; i++) w2[i] -= t[i];
    orc_normalize(w2, (size_t)n, q);
    tesla_oracle(&P, w1, w2, msg, mlen, md);
    rc = memcmp(md, sigmd, 64) ? 1 : 0;
done:
    free(w);
    return rc;
}
