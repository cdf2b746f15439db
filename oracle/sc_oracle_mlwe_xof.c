/*
 * sc_oracle_mlwe_xof.c - CPU oracle, TEST INFRASTRUCTURE ONLY (see sc_oracle.h): the XOF-driven forms of the module-LWE helpers.
 *
 * A plain-C restatement of /root/reference/src/utils/arith/module_lwe.c:330-367 (uniform_rand_sample_xof, uniform_rand_sample_small_xof),
 * :449-512 (binomial_rand_sample_xof, the unconstrained-RAM branch the library build compiles), :550-560 (uniform_random_ring_q_xof) and
 * :751-912 (create_rand_product_32_xof / _16_xof) on SHAKE128 / SHAKE256 prepared as the scheme code does (xof_init, xof_absorb(seed), xof_final:
 * src/schemes/sig/dilithium/dilithium.c:866-870).  Pinned by tests/golden/mlwe_xof_ref.txt: the reference's own functions called directly by
 * oracle/ref_driver.c (gen_mlwe_xof) - no scheme of the library as compiled reaches them, so there is no scheme-level vector to pin against.
 *
 * Every function takes the stream position to start from (`skip` bytes of the squeezed stream are discarded first) and returns the position after
 * the call: calls that the reference would make one after the other on one xof context chain through it.
 *
 * Two properties of the reference's code that the restatement keeps because "identical results" includes them:
 *   - uniform_rand_sample_small_xof writes floor(m n / 1024) * 1024 values (its loop runs (m n) >> 10 times over blocks of 512 bytes): for
 *     m n = 1280 the last 256 entries of s are NOT written (module_lwe.c:355);
 *   - binomial_rand_sample_xof restarts its output pointer at s for every block of 1024 values (the inner `SINT32 *out = s` shadows the outer one,
 *     module_lwe.c:452-456): after the call s[0..1024) holds the LAST block's values and the rest of s is not written, while the stream has moved
 *     by (m n >> 10) * eta * 256 bytes.
 */
#include <stdint.h>
#include <stddef.h>
#include <string.h>
#include <stdlib.h>
#include "sc_oracle.h"
#include "sc_oracle_keccak.h"

static void xof_start(shake_t *c, int shake_bits, const uint8_t *seed, size_t seed_len, size_t skip)
{
    shake_init(c, shake_bits);
    shake_absorb(c, seed, seed_len);
    shake_final(c);
    uint8_t junk[256];
    while (skip) { size_t m = skip < sizeof junk ? skip : sizeof junk; shake_squeeze(c, junk, m); skip -= m; }
}

/* uniform_random_ring_q_xof, npoly times in a row: n 32-bit little-endian words a polynomial, masked to q_bits, minus q when >= q */
size_t orc_ring_q_xof(int shake_bits, const uint8_t *seed, size_t seed_len, size_t skip, int npoly, int n, int32_t q, int q_bits, int32_t *out)
{
    shake_t c;
    xof_start(&c, shake_bits, seed, seed_len, skip);
    const int32_t mask = (int32_t)((1u << q_bits) - 1u);
    for (int p = 0; p < npoly; p++)
        for (int i = 0; i < n; i++) {
            uint8_t b[4];
            shake_squeeze(&c, b, 4);
            int32_t v = (int32_t)((uint32_t)b[0] | ((uint32_t)b[1] << 8) | ((uint32_t)b[2] << 16) | ((uint32_t)b[3] << 24)) & mask;
            v -= (v >= q) * q;
            out[(size_t)p * n + i] = v;
        }
    return skip + (size_t)npoly * n * 4;
}

/* uniform_rand_sample_xof: a 32-bit word per value, masked to bits + 1 bits, halved up to twice while above 2 eta, s = eta - value */
size_t orc_uniform_sample_xof(int shake_bits, const uint8_t *seed, size_t seed_len, size_t skip, int32_t eta, int bits, int32_t *s, size_t count)
{
    shake_t c;
    xof_start(&c, shake_bits, seed, seed_len, skip);
    const int32_t mask = (int32_t)((1u << (bits + 1)) - 1u);
    for (size_t i = 0; i < count; i++) {
        uint8_t b[4];
        shake_squeeze(&c, b, 4);
        int32_t t = (int32_t)((uint32_t)b[0] | ((uint32_t)b[1] << 8) | ((uint32_t)b[2] << 16) | ((uint32_t)b[3] << 24)) & mask;
        t >>= (t > 2 * eta);
        t >>= (t > 2 * eta);
        s[i] = eta - t;
    }
    return skip + count * 4;
}

/* uniform_rand_sample_small_xof: blocks of 512 bytes -> 1024 values, two a byte (low nibble first), one conditional halving; only whole blocks are written */
size_t orc_uniform_sample_small_xof(int shake_bits, const uint8_t *seed, size_t seed_len, size_t skip, int32_t eta, int bits, int32_t *s, size_t count)
{
    shake_t c;
    xof_start(&c, shake_bits, seed, seed_len, skip);
    const int32_t mask = (int32_t)((1u << (bits + 1)) - 1u);
    const size_t blocks = count >> 10;
    for (size_t i = 0; i < blocks; i++) {
        uint8_t r[512];
        shake_squeeze(&c, r, 512);
        for (int j = 0; j < 512; j++) {
            int32_t t = r[j] & mask;
            if (t > 2 * eta) t >>= 1;
            *s++ = eta - t;
            t = (r[j] >> 4) & mask;
            if (t > 2 * eta) t >>= 1;
            *s++ = eta - t;
        }
    }
    return skip + blocks * 512;
}

/* binomial_rand_sample_xof: blocks of eta * 256 bytes -> 1024 values (4 a group of eta bytes: bit pairs summed over the eta bytes, differences), every
 * block written at s[0..1024) */
size_t orc_binomial_sample_xof(int shake_bits, const uint8_t *seed, size_t seed_len, size_t skip, int32_t eta, int32_t *s, size_t count)
{
    shake_t c;
    xof_start(&c, shake_bits, seed, seed_len, skip);
    const size_t blocks = count >> 10;
    if (eta < 1 || eta > 5) return skip;
    for (size_t i = 0; i < blocks; i++) {
        uint8_t t[5 * 256];
        shake_squeeze(&c, t, (size_t)eta * 256);
        const uint8_t *p = t;
        int32_t *out = s;
        for (int j = 0; j < 256; j++) {
            uint32_t a[8] = { 0, 0, 0, 0, 0, 0, 0, 0 };
            for (int e = 0; e < eta; e++, p++)
                for (int b = 0; b < 8; b++) a[b] += *p & (1u << b);                       /* the bit stays at its position: sums of 2^b multiples */
            *out++ = (int32_t)(a[0]) - (int32_t)(a[1] >> 1);
            *out++ = (int32_t)(a[2] >> 2) - (int32_t)(a[3] >> 3);
            *out++ = (int32_t)(a[4] >> 4) - (int32_t)(a[5] >> 5);
            *out++ = (int32_t)(a[6] >> 6) - (int32_t)(a[7] >> 7);
        }
    }
    return skip + blocks * (size_t)eta * 256;
}

/* create_rand_product_32_xof / _16_xof (module_lwe.c:751-912): t = A y, the matrix drawn from the stream in the order the loops visit it - row by row
 * (transposed == 0: the (i, j) entry is draw i l + j) or, transposed, for j over l and i over k (draw j k + i multiplies NTT(y[j]) into t[i]).  y has l
 * polynomials, t has k; t comes back canonical.
 * `width` (16 or 32) selects which of the two functions is restated.  They differ in the width of the twiddle tables - and in one statement of the
 * transposed branch: the _16 form accumulates, add_32(t_i, n, t_i, out) (:875), the _32 form has add_32(t_i, n, out, c) (:796; the same text in the
 * CSPRNG form, :629) - the sum of the CURRENT term and the matrix entry itself REPLACES t_i, so for l > 1 the _32 form's transposed result is
 * INTT(A[l-1][i] o (NTT(y[l-1]) + 1)), not (A^T y)_i.  No caller of the reference reaches that branch (Dilithium, the only user of the 32-bit form,
 * never asks for the transpose: dilithium.c:879,1592,1845,2147); width == 32 with transposed != 0 restates it as it is, so that the fixture is
 * explained to the last word - the engine serves the _16 form's A^T y for every modulus and says so (include/sc_hip.h). */
size_t orc_rand_product_xof(int shake_bits, const uint8_t *seed, size_t seed_len, size_t skip, int32_t q, int q_bits, int k, int l, int transposed,
                            const int32_t *y, int32_t *t, int n, const int32_t *w, const int32_t *r, int width)
{
    int32_t *A = malloc(sizeof(int32_t) * (size_t)n * k * l), *yh = malloc(sizeof(int32_t) * (size_t)n * l);
    const size_t pos = orc_ring_q_xof(shake_bits, seed, seed_len, skip, k * l, n, q, q_bits, A);
    for (int j = 0; j < l; j++) {
        memcpy(yh + (size_t)j * n, y + (size_t)j * n, sizeof(int32_t) * n);
        orc_fwd_ntt(yh + (size_t)j * n, n, q, yh + (size_t)j * n, w);
    }
    for (int i = 0; i < k; i++) {
        int64_t acc[1024];
        for (int x = 0; x < n; x++) acc[x] = 0;
        for (int j = 0; j < l; j++) {
            const int32_t *a = A + (size_t)(transposed ? j * k + i : i * l + j) * n;
            const int replace = transposed && width == 32 && j > 0;                         /* module_lwe.c:796 */
            for (int x = 0; x < n; x++) {
                int64_t yy = yh[(size_t)j * n + x] % q; if (yy < 0) yy += q;
                acc[x] = replace ? (yy * a[x] % q + a[x]) % q : (acc[x] + yy * a[x]) % q;
            }
        }
        for (int x = 0; x < n; x++) t[(size_t)i * n + x] = (int32_t)acc[x];
        orc_inv_ntt(t + (size_t)i * n, n, q, t + (size_t)i * n, w, r);
        orc_normalize(t + (size_t)i * n, (size_t)n, q);
    }
    free(A); free(yh);
    return pos;
}
