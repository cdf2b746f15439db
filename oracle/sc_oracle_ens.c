/*
 * sc_oracle_ens.c - CPU restatement of the reference's ENS-KEM encapsulation / decapsulation.  TEST INFRASTRUCTURE ONLY.
 *
 * Follows src/schemes/kem/ens/ens_kem.c: generate_ciphertext (:280-318), ens_kem_encapsulation (:785-865), ens_kem_decapsulation
 * (:867-960), reverse_and_pack_inv_g (:191-211, the layout of the key's 1/g mod 2), poly_32_uniform_rand
 * (src/utils/arith/poly_32.c:132-156), z2_conv_mod2 (src/utils/arith/poly_z2.c:354-376) with sc_arr_rotl_32 (sc_math.c:894-925),
 * raw coding (q_bits per coefficient, MSB first).  Keys come from the reference's keygen as data: h, g SINT16 (NTT domain, centred),
 * inv_g n/32 words.  Randomness: the prng_32 word stream of sc->prng_ctx[0] (r, then e).
 *
 * Pinned byte-for-byte against tests/golden/ens_ref.txt (made by oracle/ref_driver.c from the reference built in oracle/_ref).
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include "sc_oracle.h"

/* par = {set, q, q_bits, n, n_bits, 0, 0, coeff_rnd[13]} */
size_t orc_ens_ct_bytes(const int32_t *par) { return ((size_t)par[3] * par[2] + 7) / 8; }
size_t orc_ens_key_bytes(const int32_t *par) { return (size_t)par[3] / 8; }

static int uniform12(const uint32_t *tape, size_t len, size_t *pos, int32_t *v, int n, const int32_t *c, int c_len)
{
    for (int i = 0; i < n; i++) v[i] = 0;
    for (int j = 0; j < c_len; j++) {
        int i = 0;
        while (i < c[j]) {
            if (*pos >= len) return 1;
            const uint32_t r = tape[(*pos)++];
            const uint32_t idx = (r >> 1) & (uint32_t)(n - 1);
            if (v[idx] == 0) { v[idx] = (r & 1) ? j - c_len : c_len - j; i++; }
        }
    }
    return 0;
}

/* ens_kem.c:785-865: 0 = ok (ct, key written, *consumed = words used), 2 = tape exhausted */
int orc_ens_encaps(const int32_t *par, const int16_t *h16, const uint32_t *tape, size_t tape_len, uint8_t *ct, uint8_t *key, size_t *consumed)
{
    const int q = par[1], q_bits = par[2], n = par[3];
    int32_t *w = malloc(sizeof(int32_t) * (size_t)n * 6), *rt = w + n, *r = rt + n, *e = r + n, *t = e + n, *h = t + n;
    size_t pos = 0;
    int rc = 2;
    if (orc_make_tables(q, n, w, rt)) { free(w); return 1; }
    if (uniform12(tape, tape_len, &pos, r, n, par + 7, 12)) goto done;
    if (uniform12(tape, tape_len, &pos, e, n, par + 7, 12)) goto done;
    for (int i = 0; i < n; i++) h[i] = h16[i];
    orc_polymul_ntt(t, n, q, r, h, w, rt);                                       /* fwd, pointwise with h, inv (:294-302); canonical */
    for (int i = 0; i < n; i++) { if (t[i] > q / 2) t[i] -= q; t[i] = 2 * t[i] + e[i]; }
    orc_normalize(t, (size_t)n, q);
    memset(ct, 0, orc_ens_ct_bytes(par));
    size_t bp = 0;
    for (int i = 0; i < n; i++) for (int b = q_bits - 1; b >= 0; b--, bp++) ct[bp >> 3] |= (uint8_t)((((uint32_t)t[i] >> b) & 1u) << (7 - (bp & 7)));
    memset(key, 0, (size_t)n / 8);
    for (int i = 0; i < n; i++) key[i >> 3] |= (uint8_t)((e[i] & 1) << (7 - (i & 7)));
    rc = 0;
done:
    *consumed = pos;
    free(w);
    return rc;
}

/* ens_kem.c:867-960.  inv_g: n/32 words as the key holds them. */
int orc_ens_decaps(const int32_t *par, const int16_t *g16, const uint32_t *inv_g, const uint8_t *ct, size_t ctlen, uint8_t *key)
{
    const int q = par[1], q_bits = par[2], n = par[3], W = n / 32;
    if (ctlen < orc_ens_ct_bytes(par)) return 1;
    int32_t *w = malloc(sizeof(int32_t) * (size_t)n * 4), *rt = w + n, *t = rt + n, *g = t + n;
    uint32_t x[64], b[64], out[64];
    if (orc_make_tables(q, n, w, rt)) { free(w); return 1; }
    size_t bp = 0;
    for (int i = 0; i < n; i++) { uint32_t v = 0; for (int k = 0; k < q_bits; k++, bp++) v = (v << 1) | ((ct[bp >> 3] >> (7 - (bp & 7))) & 1u); t[i] = (int32_t)v; }
    for (int i = 0; i < n; i++) g[i] = g16[i];
    orc_polymul_ntt(t, n, q, t, g, w, rt);
    memset(x, 0, sizeof x);
    for (int i = 0; i < n; i++) { int32_t v = t[i] > (q >> 1) ? t[i] - q : t[i]; x[i >> 5] |= (uint32_t)(v & 1) << (31 - (i & 31)); }
    /* z2_conv_mod2: bit i of the result is the parity of a AND (b_rev rotated right by i + 1 positions as one MSB-first bit string) */
    memcpy(b, inv_g, sizeof(uint32_t) * (size_t)W);
    for (int i = 0; i < n; i++) {
        uint32_t carry = b[W - 1], acc = 0;
        for (int j = 0; j < W; j++) { const uint32_t nw = (carry << 31) | (b[j] >> 1); carry = b[j]; b[j] = nw; }
        for (int j = 0; j < W; j++) acc ^= x[j] & b[j];
        acc ^= acc >> 16; acc ^= acc >> 8; acc ^= acc >> 4; acc &= 0xF;
        acc = (0x6996u >> acc) & 1u;
        if ((i & 31) == 0) out[i >> 5] = acc << 31; else out[i >> 5] |= acc << (31 - (i & 31));
    }
    for (int j = 0; j < W; j++) { key[4 * j] = (uint8_t)(out[j] >> 24); key[4 * j + 1] = (uint8_t)(out[j] >> 16); key[4 * j + 2] = (uint8_t)(out[j] >> 8); key[4 * j + 3] = (uint8_t)out[j]; }
    free(w);
    return 0;
}
