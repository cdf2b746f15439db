/*
 * sc_oracle_rlwe.c - CPU restatement of the reference's Ring-LWE encryption scheme.  TEST INFRASTRUCTURE ONLY.
 *
 * Follows src/schemes/enc/rlwe_enc/rlwe_enc.c: a_gen / r1_gen / r2_gen (:205-290), rlwe_enc_keygen (:425-493),
 * rlwe_enc_encrypt (:511-631), rlwe_enc_decrypt (:633-707), parameter sets rlwe_enc_params.c:23-41, raw coding
 * (SC_ENTROPY_NONE: q_bits per coefficient, MSB first).  The reference is built with RLWE_ENC_ENABLE_NTT_TRANSMISSION
 * (rlwe_enc.h:24): c1 = a o NTT(e1) + NTT(e2) is transmitted in the NTT domain and decryption does not transform it.
 *
 * Randomness ("tape" = the prng_32 word stream of sc->prng_ctx[0]): a_gen n/2 words (two 16-bit halves each), r1_gen n Gaussian
 * samples (one prng_64 = two words, high word first, prng.c:963-978, through the 64-bit CDF sampler, gaussian_cdf.c:639-659),
 * r2_gen n/16 words (16 two-bit fields each); encryption 3n Gaussian samples e1 | e2 | e3.
 *
 * Pinned byte-for-byte against tests/golden/rlwe_ref.txt (made by oracle/ref_driver.c from the reference built in oracle/_ref).
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include "sc_oracle.h"

typedef struct { int set, q, q_bits, n, n_bits, m_scale, o_scale_0, o_scale_1; float sig, tail; int pad[2]; } rlwep_t;

static uint64_t t64(const uint32_t *tape, size_t *pos) { uint64_t hi = tape[(*pos)++]; return (hi << 32) | tape[(*pos)++]; }

size_t orc_rlwe_kg_words(const int32_t *par) { return (size_t)par[3] / 2 + 2 * (size_t)par[3] + (size_t)par[3] / 16; }
size_t orc_rlwe_enc_words(const int32_t *par) { return 6 * (size_t)par[3]; }
size_t orc_rlwe_ct_bytes(const int32_t *par) { return (2 * (size_t)par[3] * par[2] + 7) / 8; }

static void put_bits(uint8_t *buf, size_t *bp, uint32_t v, int bits)
{
    for (int b = bits - 1; b >= 0; b--, (*bp)++) buf[*bp >> 3] |= (uint8_t)(((v >> b) & 1u) << (7 - (*bp & 7)));
}
static uint32_t get_bits(const uint8_t *buf, size_t *bp, int bits)
{
    uint32_t v = 0;
    for (int b = 0; b < bits; b++, (*bp)++) v = (v << 1) | ((buf[*bp >> 3] >> (7 - (*bp & 7))) & 1u);
    return v;
}

/* rlwe_enc.c:425-493.  cdf: the sampler's table (orc_cdt_create(13.0, sigma)).  Outputs as the reference stores them: SINT16,
 * NTT domain, canonical.  Returns 0, or 2 when the tape is too short. */
int orc_rlwe_keygen(const int32_t *par, const uint64_t *cdf, int cdf_size, const uint32_t *tape, size_t tape_len, int16_t *a16, int16_t *p16, int16_t *r2_16)
{
    rlwep_t P; memcpy(&P, par, sizeof P);
    const int n = P.n, q = P.q;
    if (tape_len < orc_rlwe_kg_words(par)) return 2;
    int32_t *w = malloc(sizeof(int32_t) * (size_t)n * 5), *r = w + n, *a = r + n, *r1 = a + n, *r2 = r1 + n;
    if (orc_make_tables(q, n, w, r)) { free(w); return 1; }
    size_t pos = 0;
    for (int i = 0; i < n / 2; i++) {                                          /* a_gen :205-225 */
        const uint32_t x = tape[pos++];
        a[2 * i] = (int32_t)(x & 0xffff) % q; a[2 * i + 1] = (int32_t)(x >> 16) % q;
    }
    orc_fwd_ntt(a, n, q, a, w); orc_normalize(a, (size_t)n, q);
    for (int i = 0; i < n; i++) r1[i] = orc_cdt_sample(t64(tape, &pos), cdf, cdf_size);   /* r1_gen :227-244 */
    orc_fwd_ntt(r1, n, q, r1, w); orc_normalize(r1, (size_t)n, q);
    for (int i = 0; i < n;) {                                                  /* r2_gen :246-285: +-1 / 0 from two-bit fields */
        uint32_t x = tape[pos++];
        for (int j = 0; j < 16; j++) {
            int32_t bit = x & 1, sign = (x >> 1) & 1;
            if (sign == 1 && bit == 1) bit = q - 1;
            r2[i++] = bit;
            x >>= 2;
        }
    }
    orc_fwd_ntt(r2, n, q, r2, w); orc_normalize(r2, (size_t)n, q);
    for (int i = 0; i < n; i++) { a16[i] = (int16_t)a[i]; r2_16[i] = (int16_t)r2[i]; }
    orc_mul_pointwise(a, n, q, a, r2);                                         /* p = r1 - a r2 :475-482 */
    for (int i = 0; i < n; i++) r1[i] -= a[i];
    orc_normalize(r1, (size_t)n, q);
    for (int i = 0; i < n; i++) p16[i] = (int16_t)r1[i];
    free(w);
    return 0;
}

/* rlwe_enc.c:511-631.  msg: n/8 bytes (bit 7 of byte 0 first).  ct: orc_rlwe_ct_bytes bytes. */
int orc_rlwe_encrypt(const int32_t *par, const uint64_t *cdf, int cdf_size, const int16_t *a16, const int16_t *p16, const uint8_t *msg,
                     const uint32_t *tape, size_t tape_len, uint8_t *ct)
{
    rlwep_t P; memcpy(&P, par, sizeof P);
    const int n = P.n, q = P.q;
    if (tape_len < orc_rlwe_enc_words(par)) return 2;
    int32_t *w = malloc(sizeof(int32_t) * (size_t)n * 9), *r = w + n, *e1 = r + n, *e2 = e1 + n, *e3 = e2 + n, *c1 = e3 + n, *c2 = c1 + n, *k = c2 + n, *em = k + n;
    if (orc_make_tables(q, n, w, r)) { free(w); return 1; }
    size_t pos = 0;
    for (int i = 0; i < n; i++) em[i] = ((msg[i >> 3] >> (7 - (i & 7))) & 1) * P.m_scale;
    for (int i = 0; i < n; i++) e1[i] = orc_cdt_sample(t64(tape, &pos), cdf, cdf_size);
    for (int i = 0; i < n; i++) e2[i] = orc_cdt_sample(t64(tape, &pos), cdf, cdf_size);
    for (int i = 0; i < n; i++) e3[i] = orc_cdt_sample(t64(tape, &pos), cdf, cdf_size);
    orc_fwd_ntt(e1, n, q, e1, w);
    orc_fwd_ntt(e2, n, q, e2, w);
    for (int i = 0; i < n; i++) k[i] = a16[i];
    orc_mul_pointwise(c1, n, q, e1, k);
    for (int i = 0; i < n; i++) c1[i] += e2[i];
    orc_normalize(c1, (size_t)n, q);                                           /* NTT domain (RLWE_ENC_ENABLE_NTT_TRANSMISSION) */
    for (int i = 0; i < n; i++) k[i] = p16[i];
    orc_mul_pointwise(c2, n, q, e1, k);
    orc_inv_ntt(c2, n, q, c2, w, r);
    for (int i = 0; i < n; i++) c2[i] += em[i] + e3[i];
    orc_normalize(c2, (size_t)n, q);
    memset(ct, 0, orc_rlwe_ct_bytes(par));
    size_t bp = 0;
    for (int i = 0; i < n; i++) put_bits(ct, &bp, (uint32_t)c1[i], P.q_bits);
    for (int i = 0; i < n; i++) put_bits(ct, &bp, (uint32_t)c2[i], P.q_bits);
    free(w);
    return 0;
}

/* rlwe_enc.c:633-707.  msg: n/8 bytes out. */
int orc_rlwe_decrypt(const int32_t *par, const int16_t *r2_16, const uint8_t *ct, size_t ctlen, uint8_t *msg)
{
    rlwep_t P; memcpy(&P, par, sizeof P);
    const int n = P.n, q = P.q;
    if (ctlen < orc_rlwe_ct_bytes(par)) return 1;
    int32_t *w = malloc(sizeof(int32_t) * (size_t)n * 5), *r = w + n, *c1 = r + n, *c2 = c1 + n, *k = c2 + n;
    if (orc_make_tables(q, n, w, r)) { free(w); return 1; }
    size_t bp = 0;
    for (int i = 0; i < n; i++) c1[i] = (int32_t)get_bits(ct, &bp, P.q_bits);
    for (int i = 0; i < n; i++) c2[i] = (int32_t)get_bits(ct, &bp, P.q_bits);
    for (int i = 0; i < n; i++) k[i] = r2_16[i];
    orc_mul_pointwise(c1, n, q, c1, k);
    orc_inv_ntt(c1, n, q, c1, w, r);
    for (int i = 0; i < n; i++) c2[i] += c1[i];
    orc_normalize(c2, (size_t)n, q);
    memset(msg, 0, (size_t)n / 8);
    for (int i = 0; i < n; i++)
        if (c2[i] > P.o_scale_0 && c2[i] < P.o_scale_1) msg[i >> 3] |= (uint8_t)(1u << (7 - (i & 7)));
    free(w);
    return 0;
}
