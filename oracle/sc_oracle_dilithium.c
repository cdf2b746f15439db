/*
 * sc_oracle_dilithium.c - CPU restatement of the reference's Dilithium sign / verify path.
 * TEST INFRASTRUCTURE ONLY (see sc_oracle.h).  Pinned: tests/test_oracle_dilithium.py reproduces the
 * reference's signatures byte for byte (tests/golden/dilithium_ref.txt, made by oracle/ref_driver.c
 * from the reference build).  Paths are relative to the reference tree.
 *
 * The matrix A is an INPUT (NTT domain, row-major [k][l][n]) - the reference re-derives it from rho
 * with its AES-CTR-DRBG on every call (module_lwe.c:519-547,914-940); the fixture carries the
 * matrix the reference drew, so no DRBG is restated here.
 */
#include "sc_oracle.h"
#include "sc_oracle_keccak.h"
#include <stdlib.h>
#include <string.h>

/* SHAKE: sc_oracle_keccak.c (the reference's permutation, not Keccak-f[1600]) */

/* ------------------------------------------------------------- bit packer ---- */
/* src/utils/entropy/packer.c:299-351: an MSB-first bit stream (the 64-bit scratch word is stored big
 * endian), values masked to `bits`; flush pads the last byte with zero bits. */
typedef struct { uint8_t *buf; size_t bitpos; } bitw_t;
static void bw_put(bitw_t *w, uint32_t v, int bits)
{
    for (int b = bits - 1; b >= 0; b--) {
        if ((w->bitpos & 7) == 0) w->buf[w->bitpos >> 3] = 0;
        w->buf[w->bitpos >> 3] |= (uint8_t)(((v >> b) & 1u) << (7 - (w->bitpos & 7)));
        w->bitpos++;
    }
}
typedef struct { const uint8_t *buf; size_t bitpos, bitlen; int err; } bitr_t;
static uint32_t br_get(bitr_t *r, int bits)
{
    uint32_t v = 0;
    for (int b = 0; b < bits; b++) {
        if (r->bitpos >= r->bitlen) { r->err = 1; return 0; }
        v = (v << 1) | ((r->buf[r->bitpos >> 3] >> (7 - (r->bitpos & 7))) & 1u);
        r->bitpos++;
    }
    return v;
}

/* --------------------------------------------------------------- helpers ---- */

/* module_lwe.c:89-104 decompose(t1,t0,t,d): t0 = t mod+- 2^d mapped to [0,q), t1 = (t - t0)/2^d */
void orc_dil_decompose(int32_t *t1, int32_t *t0, const int32_t *t, size_t len, int d, int32_t q)
{
    const int32_t mask = (1 << d) - 1, sign = 1 << d, thresh = 1 << (d - 1);
    for (size_t i = 0; i < len; i++) {
        int32_t x = t[i] & mask;
        x -= (x > thresh) * sign;
        t1[i] = (t[i] - x) >> d;
        t0[i] = (x >= 0) ? x : x + q;
    }
}

/* module_lwe.h:52-66,110-136 round_alpha with alpha = 2*gamma_2, Barrett k = 30, m = floor(2^30/alpha)
 * (dilithium.c:196-204, ntt.c:142-146) */
static uint32_t round_alpha(uint32_t a, int32_t *a0, int32_t alpha, int32_t q)
{
    const int64_t m = (1 << 30) / alpha;
    int64_t t = ((int64_t)a * m) >> 30, c = (int64_t)a - t * alpha;
    if (alpha <= c) c -= alpha;
    *a0 = (int32_t)c;
    if (*a0 > (alpha >> 1)) *a0 -= alpha;
    uint32_t a1 = a - (uint32_t)*a0;
    if (a1 == (uint32_t)(q - 1)) { a1 = 0; *a0 -= 1; }
    else {
        int64_t t2 = ((int64_t)a1 * m) >> 30, c2 = (int64_t)a1 - t2 * alpha;
        if (alpha <= c2) t2++;
        a1 = (uint32_t)t2;
    }
    return a1;
}

/* dilithium.c:1185-1200 */
static int check_norm_inf(const int32_t *v, size_t len, int32_t q, int32_t b)
{
    const int32_t lower_half = (q - 1) >> 1, upper_b = q - b;
    int bad = 0;
    for (size_t i = 0; i < len; i++) {
        const int smaller = v[i] <= lower_half;
        bad |= (smaller && (v[i] >= b)) || (!smaller && (v[i] <= upper_b));
    }
    return bad;
}

/* module_lwe.c:201-244 expand_mask */
static void expand_mask(const uint8_t *K, const uint8_t *mu, uint16_t kappa, int32_t gamma1, int32_t q, size_t limit, int32_t *y)
{
    const uint8_t kb[2] = { (uint8_t)(kappa >> 8), (uint8_t)(kappa & 0xFF) };
    const int32_t thresh = 2 * gamma1 - 2, add = q + gamma1 - 1;
    shake_t x; shake_init(&x, 256);
    shake_absorb(&x, mu, 48); shake_absorb(&x, K, 32); shake_absorb(&x, kb, 2); shake_final(&x);
    size_t j = 0;
    while (j < limit) {
        uint8_t s[5];
        shake_squeeze(&x, s, 5);
        const uint32_t s0 = (((uint32_t)s[2] & 0xF) << 16) | ((uint32_t)s[1] << 8) | s[0];
        const uint32_t s1 = ((uint32_t)s[4] << 12) | ((uint32_t)s[3] << 4) | ((uint32_t)s[2] >> 4);
        y[j] = add - (int32_t)s0; j += ((s0 - (uint32_t)thresh) >> 31);
        if (j == limit) break;
        y[j] = add - (int32_t)s1; j += ((s1 - (uint32_t)thresh) >> 31);
    }
}

/* dilithium.c:1011-1036 h_function_deterministic (SHAKE128 variant) + module_lwe.c:1115-1141 kyber_oracle_core */
static void h_function(int32_t *c, const uint8_t *mu, const uint8_t *w1, size_t w1len, int n, int weight)
{
    const int nwb = (weight + 7) >> 3;
    uint8_t signs[8 + 64 + 8];
    shake_t x; shake_init(&x, 128);
    shake_absorb(&x, mu, 48); shake_absorb(&x, w1, w1len); shake_final(&x);
    shake_squeeze(&x, signs, (size_t)(nwb + weight));
    for (int i = 0; i < n - weight; i++) c[i] = 0;
    uint8_t mask = 1;
    for (int i = n - weight, j = 0, kk = nwb; i < n; i++, kk++) {
        size_t b = signs[kk];
        while (b > (size_t)i) b >>= 1;
        c[i] = c[b];
        c[b] = (signs[j] & mask) ? -1 : 1;
        mask = (uint8_t)(mask << 1);
        if (mask == 0) { mask = 1; j++; }
    }
}

/* module_lwe.c:588-667, NOT_TRANSPOSED, matrix given: w[i] = INTT(sum_j A[i][j] o NTT(y[j])), canonical */
static void matvec(int32_t *w_out, const int32_t *A, const int32_t *y, int n, int k, int l, int32_t q, const int32_t *w, const int32_t *r)
{
    int32_t *yh = malloc(sizeof(int32_t) * (size_t)n * (l + 1)), *acc = yh + (size_t)l * n;
    for (int j = 0; j < l; j++) { orc_fwd_ntt(yh + j * n, n, q, y + j * n, w); orc_normalize(yh + j * n, (size_t)n, q); }
    for (int i = 0; i < k; i++) {
        for (int e = 0; e < n; e++) {
            int64_t s = 0;
            for (int j = 0; j < l; j++) s += (int64_t)yh[j * n + e] * A[((size_t)i * l + j) * n + e] % q;
            acc[e] = (int32_t)(s % q);
        }
        orc_inv_ntt(w_out + i * n, n, q, acc, w, r);
    }
    free(yh);
}

/* c * a mod (x^n+1, q) for c in {-1,0,1}^n: the value of sparse_mul_mod_ring / sparse_mul_mod_q_ring
 * (dilithium.c:1314-1383) as an element of Z_q, canonical */
static void sparse_mul(int32_t *out, const int32_t *a, const int32_t *c, int n, int32_t q)
{
    int64_t *t = calloc((size_t)2 * n, sizeof(int64_t));
    for (int j = 0; j < n; j++) {
        if (c[j] == 0) continue;
        for (int e = 0; e < n; e++) t[j + e] += (c[j] < 0) ? -(int64_t)a[e] : (int64_t)a[e];
    }
    for (int j = 0; j < n; j++) { int64_t v = (t[j] - t[j + n]) % q; if (v < 0) v += q; out[j] = (int32_t)v; }
    free(t);
}

/* dilithium.c:1258-1284 make_hint */
static int make_hint(int32_t *h, const int32_t *r, const int32_t *z, size_t len, int32_t alpha, int32_t q)
{
    int sum = 0;
    for (size_t i = 0; i < len; i++) {
        int32_t add = r[i] + z[i], t;
        add -= (add >= q) * q;
        add += (add < 0) * q;
        const uint32_t r1 = round_alpha((uint32_t)r[i], &t, alpha, q), r0 = round_alpha((uint32_t)add, &t, alpha, q);
        h[i] = r1 != r0;
        sum += h[i];
    }
    return sum;
}

typedef struct { int n, q, q_bits, k, l, d, weight, gamma1, gamma1_bits, gamma2, eta, z_bits, beta, omega, omega_bits, set; } dilp_t;

/* ------------------------------------------------------------------ sign ---- */
/* dilithium.c:1397-1958 (SC_SCHEME_SIG_DILITHIUM, USE_DETERMINISTIC_DILITHIUM, sparse multiplier build).
 * par: the 16 ints of the fixture's *_params record.  Returns 0 and the packed signature. */
int orc_dilithium_sign(const int32_t *par, const int32_t *s1, const int32_t *s2, const int32_t *t,
                       const uint8_t *K, const uint8_t *tr, const int32_t *Ahat,
                       const uint8_t *msg, size_t mlen, uint8_t *sig, size_t *siglen, int32_t *trials)
{
    dilp_t P; memcpy(&P, par, sizeof P);
    const int n = P.n, k = P.k, l = P.l, q = P.q, alpha = 2 * P.gamma2;
    int32_t *w_t = malloc(sizeof(int32_t) * (size_t)n * 2), *r_t = w_t + n;
    if (orc_make_tables(q, n, w_t, r_t)) { free(w_t); return 1; }
    int32_t *buf = malloc(sizeof(int32_t) * (size_t)n * (7 * k + 3 * l + 2));
    int32_t *t1 = buf, *t0 = t1 + k * n, *w = t0 + k * n, *h = w + k * n, *ct0 = h + k * n, *wcs2 = ct0 + k * n, *tmpk = wcs2 + k * n;
    int32_t *y = tmpk + k * n, *z = y + l * n, *c = z + l * n;   /* + n spare */
    uint8_t *w1b = malloc((size_t)k * n);
    uint8_t mu[48];
    orc_dil_decompose(t1, t0, t, (size_t)k * n, P.d, q);
    { shake_t x; shake_init(&x, 256); shake_absorb(&x, tr, 48); shake_absorb(&x, msg, mlen); shake_final(&x); shake_squeeze(&x, mu, 48); }
    uint16_t kappa = 0;
    int ntr = 0, num_ones = 0;
    for (;;) {
        ntr++;
        expand_mask(K, mu, kappa, P.gamma1, q, (size_t)l * n, y);
        matvec(w, Ahat, y, n, k, l, q, w_t, r_t);
        for (int i = 0; i < k * n; i++) { int32_t r1; w1b[i] = (uint8_t)round_alpha((uint32_t)w[i], &r1, alpha, q); }
        h_function(c, mu, w1b, (size_t)k * n, n, P.weight);
        /* z = y + c*s1, canonical; check 1 */
        for (int i = 0; i < l; i++) sparse_mul(z + i * n, s1 + i * n, c, n, q);
        for (int i = 0; i < l * n; i++) z[i] = (int32_t)(((int64_t)z[i] + y[i]) % q);
        if (check_norm_inf(z, (size_t)l * n, q, P.gamma1 - P.beta)) { kappa++; continue; }
        /* check 2: low bits of w - c*s2 */
        for (int i = 0; i < k; i++) sparse_mul(tmpk + i * n, s2 + i * n, c, n, q);
        for (int i = 0; i < k * n; i++) { int32_t v = w[i] - tmpk[i]; wcs2[i] = v < 0 ? v + q : v; }
        for (int i = 0; i < k * n; i++) { int32_t r1; round_alpha((uint32_t)wcs2[i], &r1, alpha, q); ct0[i] = r1 < 0 ? q + r1 : r1; }
        if (check_norm_inf(ct0, (size_t)k * n, q, P.gamma2 - P.beta)) { kappa++; continue; }
        /* check 3: c*t0 */
        for (int i = 0; i < k; i++) sparse_mul(ct0 + i * n, t0 + i * n, c, n, q);
        if (check_norm_inf(ct0, (size_t)k * n, q, P.gamma2 - P.beta)) { kappa++; continue; }
        /* hint (dilithium.c:1712-1725): r = w - cs2 + ct0, z = -ct0 (poly_32_mod_negate maps 0 -> q) */
        for (int i = 0; i < k * n; i++) { wcs2[i] = (int32_t)(((int64_t)wcs2[i] + ct0[i]) % q); ct0[i] = q - ct0[i]; }
        num_ones = make_hint(h, wcs2, ct0, (size_t)k * n, alpha, q);
        if (num_ones > P.omega) { kappa++; continue; }
        break;
    }
    orc_center(z, (size_t)l * n, q);
    /* packing, dilithium.c:1903-1951 */
    const int h_bits = 8 + ((k + 1) >> 1);
    bitw_t bw = { sig, 0 };
    for (int i = 0; i < l * n; i++) bw_put(&bw, (uint32_t)z[i], P.z_bits);
    bw_put(&bw, (uint32_t)num_ones, P.omega_bits);
    for (int i = 0, left = num_ones; left; i++) if (h[i]) { bw_put(&bw, (uint32_t)i, h_bits); left--; }
    for (int i = 0; i < n; i++) bw_put(&bw, (uint32_t)c[i], 2);
    *siglen = (bw.bitpos + 7) >> 3;
    if (trials) *trials = ntr;
    free(w1b); free(buf); free(w_t);
    return 0;
}

/* Test support for the CPU wave emulation (tests/emu/): what the signer's kernels in front of the finishing kernel leave per attempt - the mask y
 * (as expand_mask writes it: q + a value in (-gamma_1, gamma_1)), w = A y canonical and the challenge c - for attempts kappa0 .. kappa0 + count - 1 of one
 * message, by the same routines orc_dilithium_sign runs (dilithium.c:1555-1650). */
int orc_dilithium_attempts(const int32_t *par, const uint8_t *K, const uint8_t *tr, const int32_t *Ahat, const uint8_t *msg, size_t mlen,
                           uint32_t kappa0, uint32_t count, int32_t *y_out, int32_t *w_out, int8_t *c_out)
{
    dilp_t P; memcpy(&P, par, sizeof P);
    const int n = P.n, k = P.k, l = P.l, q = P.q, alpha = 2 * P.gamma2;
    int32_t *w_t = malloc(sizeof(int32_t) * (size_t)n * 3), *r_t = w_t + n, *c = r_t + n;
    if (orc_make_tables(q, n, w_t, r_t)) { free(w_t); return 1; }
    uint8_t *w1b = malloc((size_t)k * n);
    uint8_t mu[48];
    { shake_t x; shake_init(&x, 256); shake_absorb(&x, tr, 48); shake_absorb(&x, msg, mlen); shake_final(&x); shake_squeeze(&x, mu, 48); }
    for (uint32_t a = 0; a < count; a++) {
        int32_t *y = y_out + (size_t)a * l * n, *w = w_out + (size_t)a * k * n;
        expand_mask(K, mu, (uint16_t)(kappa0 + a), P.gamma1, q, (size_t)l * n, y);
        matvec(w, Ahat, y, n, k, l, q, w_t, r_t);
        for (int i = 0; i < k * n; i++) { int32_t r1; w1b[i] = (uint8_t)round_alpha((uint32_t)w[i], &r1, alpha, q); }
        h_function(c, mu, w1b, (size_t)k * n, n, P.weight);
        for (int i = 0; i < n; i++) c_out[(size_t)a * n + i] = (int8_t)c[i];
    }
    free(w1b); free(w_t);
    return 0;
}

/* module_lwe.c:140-180 collision_resistant_hash_t1: SHAKE256( pack(rho as 32x8 bits, t1 as k*n x bits) ) -> 48 B */
void orc_dilithium_crh_t1(const uint8_t *rho, const int32_t *t1, size_t len, int bits, uint8_t *out48)
{
    uint8_t *m = calloc(32 + (len * (size_t)bits + 7) / 8 + 8, 1);
    bitw_t bw = { m, 0 };
    for (int i = 0; i < 32; i++) bw_put(&bw, rho[i], 8);
    for (size_t i = 0; i < len; i++) bw_put(&bw, (uint32_t)t1[i], bits);
    orc_shake(256, m, (bw.bitpos + 7) >> 3, out48, 48);
    free(m);
}

/* ---------------------------------------------------------------- verify ---- */
/* dilithium.c:2002-2240.  Returns 0 (SC_FUNC_SUCCESS) for a valid signature, 1 otherwise. */
int orc_dilithium_verify(const int32_t *par, const int32_t *t1, const uint8_t *rho, const int32_t *Ahat,
                         const uint8_t *msg, size_t mlen, const uint8_t *sig, size_t siglen)
{
    dilp_t P; memcpy(&P, par, sizeof P);
    const int n = P.n, k = P.k, l = P.l, q = P.q, alpha = 2 * P.gamma2;
    int rc = 1;
    int32_t *w_t = malloc(sizeof(int32_t) * (size_t)n * 2), *r_t = w_t + n;
    if (orc_make_tables(q, n, w_t, r_t)) { free(w_t); return 1; }
    int32_t *buf = calloc((size_t)n * (3 * k + l + 2), sizeof(int32_t));
    int32_t *z = buf, *h = z + l * n, *w = h + k * n, *t0 = w + k * n, *c = t0 + k * n, *c2 = c + n;
    uint8_t *w1b = malloc((size_t)k * n);
    bitr_t br = { sig, 0, siglen * 8, 0 };
    for (int i = 0; i < l * n; i++) {
        uint32_t v = br_get(&br, P.z_bits);
        int32_t sv = (int32_t)(v << (32 - P.z_bits)) >> (32 - P.z_bits);       /* signed raw coefficient */
        z[i] = sv < 0 ? sv + q : sv;
    }
    const int h_bits = 8 + ((k + 1) >> 1);
    uint32_t num_ones = br_get(&br, P.omega_bits);
    for (uint32_t i = 0; i < num_ones && !br.err; i++) { uint32_t pos = br_get(&br, h_bits); if (pos < (uint32_t)(k * n)) h[pos] = 1; else br.err = 1; }
    for (int i = 0; i < n; i++) { uint32_t v = br_get(&br, 2); c[i] = (v == 3) ? -1 : (int32_t)v; }
    if (br.err) goto done;
    uint8_t mu[48];
    orc_dilithium_crh_t1(rho, t1, (size_t)k * n, P.q_bits - P.d, mu);
    { shake_t x; shake_init(&x, 256); shake_absorb(&x, mu, 48); shake_absorb(&x, msg, mlen); shake_final(&x); shake_squeeze(&x, mu, 48); }
    if (check_norm_inf(z, (size_t)l * n, q, P.gamma1 - P.beta)) goto done;
    { int ones = 0; for (int i = 0; i < k * n; i++) ones += h[i]; if (ones > P.omega) goto done; }
    matvec(w, Ahat, z, n, k, l, q, w_t, r_t);
    for (int i = 0; i < k * n; i++) t0[i] = t1[i] << P.d;
    for (int i = 0; i < k; i++) sparse_mul(t0 + i * n, t0 + i * n, c, n, q);
    for (int i = 0; i < k * n; i++) { int32_t v = w[i] - t0[i]; t0[i] = v < 0 ? v + q : v; }
    /* use_hint, dilithium.c:1286-1312 */
    {
        const int32_t m = (q - 1) / alpha, m_mask = m - 1;
        for (int i = 0; i < k * n; i++) {
            int32_t a0; int32_t a1 = (int32_t)round_alpha((uint32_t)t0[i], &a0, alpha, q);
            if (h[i] == 1) { if (a0 > 0) a1++; else a1--; }
            if (a1 < 0) a1 += m;
            w1b[i] = (uint8_t)(a1 & m_mask);
        }
    }
    h_function(c2, mu, w1b, (size_t)k * n, n, P.weight);
    rc = memcmp(c, c2, sizeof(int32_t) * (size_t)n) ? 1 : 0;
done:
    free(w1b); free(buf); free(w_t);
    return rc;
}


/* ------------------------------------------------------------------- keygen ---- */
/* uniform_rand_sample_small_csprng, src/utils/arith/module_lwe.c:280-310: the loop draws 512-byte blocks for i = 0, 1024, ... <
 * m*n and writes 1024 samples per block - past the end of s for m*n = 1280 or 1536, as the reference does into its own key
 * buffer (the overflow lands in memory that is rewritten afterwards); what survives is the first m*n samples of the blocks.
 * Two samples per byte: (b & mask) and ((b >> 4) & mask), halved when > 2 eta, stored as eta - value. */
static void sample_small(orc_drbg_t *d, int eta, int eta_bits, int32_t *s, size_t count)
{
    const uint32_t mask = (1u << (eta_bits + 1)) - 1;
    size_t o = 0;
    for (size_t i = 0; i < count; i += 1024) {
        uint8_t r[512];
        orc_drbg_mem(d, r, 512);
        for (int j = 0; j < 512; j++) {
            uint32_t b = r[j];
            for (int h = 0; h < 2; h++) {
                uint32_t t = b & mask;
                if (t > 2u * (uint32_t)eta) t >>= 1;
                if (o < count) s[o] = eta - (int32_t)t;
                o++;
                b >>= 4;
            }
        }
    }
}

/* dilithium_keygen (src/schemes/sig/dilithium/dilithium.c:752-935) with the context CSPRNG seeded by `seed`: stream = rho block |
 * K block | blocks of s1 | blocks of s2; A from a second generator seeded with rho; t = normalise(A s1 + s2); t1 = high part
 * (decompose, d bits); tr = CRH(rho, t1).  eta_bits is dilithium_set_t::eta_bits. */
int orc_dilithium_keygen(const int32_t *par, int eta_bits, const uint8_t *seed, size_t seed_len,
                         int32_t *s1, int32_t *s2, int32_t *t, uint8_t *rho, uint8_t *K, uint8_t *tr, int32_t *t1)
{
    dilp_t P; memcpy(&P, par, sizeof P);
    const int n = P.n, k = P.k, l = P.l, q = P.q;
    int32_t *w_t = malloc(sizeof(int32_t) * (size_t)n * (2 + k * l + k)), *r_t = w_t + n, *A = r_t + n, *t0 = A + (size_t)k * l * n;
    uint8_t blk[64];
    if (orc_make_tables(q, n, w_t, r_t)) { free(w_t); return 1; }
    orc_drbg_t *d = orc_drbg_create(seed, seed_len);
    orc_drbg_mem(d, blk, 32); memcpy(rho, blk, 32);
    orc_drbg_mem(d, blk, 32); memcpy(K, blk, 32);
    sample_small(d, P.eta, eta_bits, s1, (size_t)l * n);
    sample_small(d, P.eta, eta_bits, s2, (size_t)k * n);
    orc_drbg_destroy(d);
    orc_expand_matrix(rho, 32, k * l, q, P.q_bits, A);
    matvec(t, A, s1, n, k, l, q, w_t, r_t);
    for (int i = 0; i < k * n; i++) t[i] += s2[i];
    orc_normalize(t, (size_t)k * n, q);
    orc_dil_decompose(t1, t0, t, (size_t)k * n, P.d, q);
    orc_dilithium_crh_t1(rho, t1, (size_t)k * n, P.q_bits - P.d, tr);
    free(w_t);
    return 0;
}
