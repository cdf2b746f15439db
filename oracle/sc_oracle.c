/*
 * sc_oracle.c - CPU restatement of libsafecrypto's polynomial hot path.
 * TEST INFRASTRUCTURE ONLY - see sc_oracle.h for the rules and the pinning status.
 *
 * All paths below are relative to the reference tree (libsafecrypto 0.2.0-79).
 */
#include "sc_oracle.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------ tables ---- */

static int64_t mulmod64(int64_t a, int64_t b, int64_t q) { return (int64_t)(((__int128)a * b) % q); }

/* src/utils/arith/roots_of_unity.c:67-88: first m in 2..q-2 with m^n == q-1 (so m is a
 * primitive 2n-th root of unity); the loop multiplies n times rather than using a ladder. */
int32_t orc_find_root(int32_t q, int n)
{
    for (int64_t m = 2; m < (int64_t)q - 1; m++) {
        int64_t p = 1;
        for (int l = 1; l <= n; l++) p = mulmod64(p, m, q);
        if (p == (int64_t)q - 1) return (int32_t)m;
    }
    return 0;
}

/* roots_of_unity.c:90-104 (xgcd(n,q) -> |x|) and :141-173 (fwd[i]=g^i, inv[i]=inv_p*g^i).
 * With q == 1 (mod n) the Bezout coefficient of smallest magnitude is x = -(q-1)/n, so
 * inv_p = (q-1)/n == -n^{-1} (mod q); the sign is undone by ntt32_flip_generic. */
int orc_make_tables(int32_t q, int n, int32_t *w, int32_t *r)
{
    int32_t g = orc_find_root(q, n);
    if (g == 0) return 1;
    /* extended Euclid on (n, q), tracking only the coefficient of n */
    int64_t r0 = n, r1 = q, s0 = 1, s1 = 0;
    while (r1 != 0) {
        int64_t t = r0 / r1, tmp;
        tmp = r0 - t * r1; r0 = r1; r1 = tmp;
        tmp = s0 - t * s1; s0 = s1; s1 = tmp;
    }
    if (r0 != 1) return 1;
    int64_t inv_p = s0 < 0 ? -s0 : s0;
    w[0] = 1;
    for (int i = 1; i < n; i++) w[i] = (int32_t)mulmod64(w[i - 1], g, q);
    r[0] = (int32_t)inv_p;
    for (int i = 1; i < n; i++) r[i] = (int32_t)mulmod64(r[i - 1], g, q);
    return 0;
}

/* ------------------------------------------------------------- reduction ---- */

/* ntt_template.c.in:700-701, 786-787 (the _ENABLE_REFERENCE branch): C remainder, sign
 * follows the dividend, |result| < q.  The other backends return different
 * representatives of the same residue; parity is on canonical values. */
int32_t orc_modn(int32_t x, int32_t q) { return x % q; }
int32_t orc_muln(int32_t x, int32_t y, int32_t q) { return (int32_t)(((int64_t)x * (int64_t)y) % q); }

/* ------------------------------------------------------------ transforms ---- */

/* ntt_template.c.in:956-1007 generic tail: v[i] = muln(t[i], u[i]) for i = n-1..0 */
void orc_mul_pointwise(int32_t *v, int n, int32_t q, const int32_t *t, const int32_t *u)
{
    for (int i = n; i--;) v[i] = orc_muln(t[i], u[i], q);
}

/* ntt.c:470-569: in-place bit-reversal permutation (the LUTs enumerate the swap pairs
 * (i, rev(i)) with i < rev(i); the generic loops compute the same pairs). */
void orc_inverse_shuffle(int32_t *v, int n)
{
    int bits = 0;
    while ((1 << bits) < n) bits++;
    for (int i = 0; i < n; i++) {
        int j = 0;
        for (int b = 0; b < bits; b++) j |= ((i >> b) & 1) << (bits - 1 - b);
        if (i < j) { int32_t t = v[i]; v[i] = v[j]; v[j] = t; }
    }
}

/* ntt_template.c.in:1144-1222 (_NTT_MK1): radix-2 decimation in time on bit-reversed
 * input; the stage of half-size i uses twiddle w[j * (n/i)] (l starts at n and halves as
 * i doubles, :1160); j = 0 butterflies skip the multiply and only reduce (:1208-1213);
 * sums and differences are stored unreduced. */
void orc_fft(int32_t *v, int n, int32_t q, const int32_t *w)
{
    for (int i = 1, l = n; i < n; i <<= 1, l >>= 1) {
        int i_next = i + i;
        for (int k = 0; k < n; k += i_next) {
            int32_t x = orc_modn(v[k + i], q);
            v[k + i] = v[k] - x;
            v[k] = v[k] + x;
        }
        for (int j = 1; j < i; j++) {
            int32_t y = w[j * l];
            for (int k = j; k < n; k += i_next) {
                int32_t x = orc_muln(v[k + i], y, q);
                v[k + i] = v[k] - x;
                v[k] = v[k] + x;
            }
        }
    }
}

/* ntt_template.c.in:1541-1552: twist by w, bit-reverse, fft.
 * Result: V[k] == sum_i t[i] g^{i(2k+1)} (mod q), natural order, NOT canonical. */
void orc_fwd_ntt(int32_t *v, int n, int32_t q, const int32_t *t, const int32_t *w)
{
    orc_mul_pointwise(v, n, q, t, w);
    orc_inverse_shuffle(v, n);
    orc_fft(v, n, q, w);
}

/* ntt.c:572-604: a[0] = -a[0]; swap a[i] <-> a[n-i] for i = 1..(n-1)/2; then one
 * conditional +q and one conditional -q per coefficient -> canonical [0,q). */
static void orc_flip(int32_t *v, int n, int32_t q)
{
    int limit = (n - 1) >> 1;
    for (int i = 1, j = n - 1; i <= limit; i++, j--) { int32_t x = v[i]; v[i] = v[j]; v[j] = x; }
    v[0] = -v[0];
    for (int i = n; i--;) {
        int32_t x = v[i];
        if (x < 0) x += q;
        if (x >= q) x -= q;
        v[i] = x;
    }
}

/* ntt_template.c.in:1567-1589: copy, bit-reverse, the SAME forward fft, pointwise by r, flip */
void orc_inv_ntt(int32_t *v, int n, int32_t q, const int32_t *t, const int32_t *w, const int32_t *r)
{
    if (v != t) memmove(v, t, sizeof(int32_t) * (size_t)n);
    orc_inverse_shuffle(v, n);
    orc_fft(v, n, q, w);
    orc_mul_pointwise(v, n, q, v, r);
    orc_flip(v, n, q);
}

/* ntt_template.c.in:1841-1901: modn then +q if negative, -q if >= q */
void orc_normalize(int32_t *v, size_t len, int32_t q)
{
    for (size_t i = len; i--;) {
        int32_t x = orc_modn(v[i], q);
        if (x < 0) x += q;
        if (x >= q) x -= q;
        v[i] = x;
    }
}

/* ntt_template.c.in:1777-1839: modn then fold into [-(q-1)/2, (q-1)/2] */
void orc_center(int32_t *v, size_t len, int32_t q)
{
    int32_t q2 = (q - 1) >> 1;
    for (size_t i = len; i--;) {
        int32_t x = orc_modn(v[i], q);
        while (x < -q2) x += q;
        while (x > q2) x -= q;
        v[i] = x;
    }
}

void orc_polymul(int32_t *c, int n, int32_t q, const int32_t *a, const int32_t *b, const int32_t *w, const int32_t *r)
{
    int32_t *fa = malloc(sizeof(int32_t) * (size_t)n * 2), *fb = fa + n;
    orc_fwd_ntt(fa, n, q, a, w);
    orc_fwd_ntt(fb, n, q, b, w);
    orc_mul_pointwise(fa, n, q, fa, fb);
    orc_inv_ntt(c, n, q, fa, w, r);
    free(fa);
}

void orc_polymul_ntt(int32_t *c, int n, int32_t q, const int32_t *a, const int32_t *bhat, const int32_t *w, const int32_t *r)
{
    int32_t *fa = malloc(sizeof(int32_t) * (size_t)n);
    orc_fwd_ntt(fa, n, q, a, w);
    orc_mul_pointwise(fa, n, q, fa, bhat);
    orc_inv_ntt(c, n, q, fa, w, r);
    free(fa);
}

/* ntt.c:381-400: v = t * sum_k x^{pos[k]} in Z[x]/(x^n+1), exact integers (no reduction) */
void orc_mult_sparse(int32_t *v, int n, int omega, const int32_t *t, const int32_t *pos)
{
    for (int i = n; i--;) v[i] = 0;
    for (int i = 0; i < omega; i++) {
        int p = pos[i];
        for (int j = 0; j < p; j++) v[j] += t[j + n - p];
        for (int j = p; j < n; j++) v[j] -= t[j - p];
    }
}

/* poly_32.c:48-62 */
void orc_poly_add(int32_t *out, int n, const int32_t *a, const int32_t *b) { for (int i = n; i--;) out[i] = a[i] + b[i]; }
void orc_poly_sub(int32_t *out, int n, const int32_t *a, const int32_t *b) { for (int i = n; i--;) out[i] = a[i] - b[i]; }

/* ntt_template.c.in:1689-1720: square-and-multiply, LSB first, through muln (sqrn = muln(x,x)) */
int32_t orc_pwr(int32_t x, int32_t e, int32_t q)
{
    int32_t y = (e & 1) ? x : 1;
    e >>= 1;
    while (e > 0) {
        x = orc_muln(x, x, q);
        if (e & 1) y = orc_muln(x, y, q);
        e >>= 1;
    }
    return y;
}

/* ntt_template.c.in:1722-1747: in place x -> x^(q-2); stops with SC_FUNC_FAILURE (1) at the first x = 0 mod q,
 * leaving the earlier entries inverted and the rest untouched */
int orc_invert(int32_t *v, size_t n, int32_t q)
{
    for (size_t i = 0; i < n; i++) {
        int32_t x = orc_modn(v[i], q);
        if (x == 0) return 1;
        v[i] = orc_pwr(x, q - 2, q);
    }
    return 0;
}

/* ntt_template.c.in:1749-1775: num[i] = muln(num[i], den[i]^(q-2)); same early exit */
int orc_div(int32_t *num, const int32_t *den, size_t n, int32_t q)
{
    for (size_t i = 0; i < n; i++) {
        int32_t x = orc_modn(den[i], q);
        if (x == 0) return 1;
        num[i] = orc_muln(num[i], orc_pwr(x, q - 2, q), q);
    }
    return 0;
}

/* poly_32.c:38-82,149-165 and vectors.c:25-60 (SINT32 arithmetic wraps) */
void orc_poly_mul_scalar(int32_t *p, size_t n, int32_t s) { for (size_t i = n; i--;) p[i] = (int32_t)((uint32_t)p[i] * (uint32_t)s); }
void orc_poly_add_scalar(int32_t *p, size_t n, int32_t s) { if (n > 0) p[0] = (int32_t)((uint32_t)p[0] + (uint32_t)s); }
void orc_poly_mod_negate(int32_t *out, size_t n, int32_t q, const int32_t *in) { for (size_t i = n; i--;) out[i] = q - in[i]; }
void orc_poly_add_single(int32_t *out, size_t n, const int32_t *in) { for (size_t i = n; i--;) out[i] = (int32_t)((uint32_t)out[i] + (uint32_t)in[i]); }
void orc_poly_sub_single(int32_t *out, size_t n, const int32_t *in) { for (size_t i = n; i--;) out[i] = (int32_t)((uint32_t)out[i] - (uint32_t)in[i]); }
int32_t orc_poly_cmp_not_equal(const int32_t *a, const int32_t *b, size_t n) { int32_t ne = 0; for (size_t i = n; i--;) ne |= a[i] ^ b[i]; return ne; }
int32_t orc_vec_absmax(const int32_t *v, size_t n)
{
    int32_t max = 0;
    for (size_t i = 0; i < n; i++) { if (v[i] > max) max = v[i]; if (-v[i] > max) max = -v[i]; }
    return max;
}
int32_t orc_vec_scalar(const int32_t *t, const int32_t *u, size_t n)
{
    uint32_t sum = 0;
    for (size_t i = 0; i < n; i++) sum += (uint32_t)t[i] * (uint32_t)u[i];
    return (int32_t)sum;
}

/* --------------------------------------------------------------- sampler ---- */

/* sc_math.c sc_ceil_log2 applied to (UINT32)(tail*sigma) (gaussian_cdf.c:562): the number
 * of bits needed so that 2^bits >= the value */
int orc_cdt_bits(float tail, float sigma)
{
    float prod = tail * sigma;             /* FLOAT product as in the reference */
    uint32_t v = (uint32_t)prod;
    int bits = 0;
    while (((uint64_t)1 << bits) < v) bits++;
    return bits;
}

/* gaussian_cdf.c:555-610: d = 2/sqrt(2 pi) * 2^64 / sigma ; cdf[0]=0 ; cdf[i] = (UINT64)s
 * with s_1 = d/2, s_{i+1} = s_i + d*expl(-i^2/(2 sigma^2)) in long double; stop early if
 * the cast yields 0; the rest (always including the last entry) is all-ones. */
int orc_cdt_create(float tail, float sigma_f, uint64_t *cdf, int size)
{
    (void)tail;
    /* SC_M_2_SQRTPIl * SC_M_SQRT1_2l, include/safecrypto_types.h long-double constants */
    const long double two_sqrtpi = 1.128379167095512573896158903121545172L;
    const long double sqrt1_2    = 0.707106781186547524400844362104849039L;
    long double sigma = sigma_f;
    long double d = two_sqrtpi * sqrt1_2 * 18446744073709551616.0L / sigma;
    /* sigma is a FLOAT in the reference, so sigma*sigma is a float product (:595) */
    float sigma_sq = sigma_f * sigma_f;
    long double e = -0.5L / sigma_sq;
    long double s = 0.5L * d;
    int i;
    cdf[0] = 0;
    for (i = 1; i < size - 1; i++) {
        cdf[i] = (uint64_t)s;
        if (cdf[i] == 0) break;
        s += d * expl(e * ((long double)(i * i)));
    }
    for (; i < size; i++) cdf[i] = 0xFFFFFFFFFFFFFFFFULL;
    return 0;
}

/* the table of a sampler created with BLINDING_SAMPLES (gaussian_cdf.c:586-589): sigma *= SC_M_SQRT1_2l (a FLOAT) before the recurrence, the size
 * (taken from the unscaled tail * sigma, :562) unchanged */
int orc_cdt_create_blinding(float tail, float sigma_f, uint64_t *cdf, int size)
{
    float scaled = sigma_f;
    scaled *= 0.707106781186547524400844362104849039L;
    return orc_cdt_create(tail, scaled, cdf, size);
}

/* gaussian_cdf.c:536-553 + :639-659 (k == 0): a = largest index reached by the halving
 * search with "cdf[b] < x"; the sign is the LSB of the same uniform word. */
int32_t orc_cdt_sample(uint64_t x, const uint64_t *cdf, int size)
{
    uint32_t a = 0;
    for (uint32_t st = (uint32_t)size >> 1; st > 0; st >>= 1) {
        uint32_t b = a + st;
        if (b < (uint32_t)size && cdf[b] < x) a = b;
    }
    return (x & 1) ? (int32_t)a : -(int32_t)a;
}

/* the sampler at 32-bit precision: gaussian_cdf_create_32 (gaussian_cdf.c:683-740) - the same recurrence with FLOAT (float) accumulators: d, e and the
 * running sum s are floats, the constants doubles, expl() long double, every assignment rounds to float - and gaussian_cdf_sample_32 / binary_search_32
 * (:661-681, 765-774): one 32-bit word per sample, "entry < word", the sign from the word's bit 0 */
int orc_cdt32_create(float tail, float sigma, uint32_t *cdf, int size)
{
    (void)tail;
    float s, d, e;
    d = 1.12837916709551257390 * 0.70710678118654752440 * 4294967296.0 / sigma;   /* M_2_SQRTPI * M_SQRT1_2 * 4294967296.0 / sigma */
    e = -0.5L / (sigma * sigma);
    s = 0.5L * d;
    int i;
    cdf[0] = 0;
    for (i = 1; i < size - 1; i++) {
        cdf[i] = s;
        if (cdf[i] == 0) break;
        s += d * expl(e * ((float)(i * i)));
    }
    for (; i < size; i++) cdf[i] = 0xFFFFFFFFu;
    return 0;
}
int32_t orc_cdt32_sample(uint32_t x, const uint32_t *cdf, int size)
{
    uint32_t a = 0;
    for (uint32_t st = (uint32_t)size >> 1; st > 0; st >>= 1) {
        uint32_t b = a + st;
        if (b < (uint32_t)size && cdf[b] < x) a = b;
    }
    return (x & 1) ? (int32_t)a : -(int32_t)a;
}
void orc_cdt32_sample_vector(int32_t *v, size_t n, const uint32_t *x, const uint32_t *cdf, int size, int32_t centre)
{
    for (size_t i = 0; i < n; i++) v[i] = orc_cdt32_sample(x[i], cdf, size) + centre;
}

/* sampling.c:211-228 with discard == 0: sequential fill, v[i] = sample + centre */
void orc_cdt_sample_vector(int32_t *v, size_t n, const uint64_t *x, const uint64_t *cdf, int size, int32_t centre)
{
    for (size_t i = 0; i < n; i++) v[i] = orc_cdt_sample(x[i], cdf, size) + centre;
}

/* The side-channel variants of the vector fillers (sampling.c:68-228), restated over the word tape the CSPRNG would deliver (prng_32 takes one pool
 * word, prng_64 two, high word first: prng.c:963-993).  blinding: 0 NORMAL_SAMPLES (sample_vector_32), 1 BLINDING_SAMPLES, 2 SHUFFLE_SAMPLES
 * (safecrypto_private.h:154-158); discard: 0 or SCA_PATTERN_SAMPLE_DISCARD_LO / MD / HI = 2 / 4 / 6 (:164-167), thresholds 2^28 / 2^30 / 2^31
 * (set_threshold, sampling.c:85-92).  Returns the words consumed, or -1 when the tape runs out.  Test infrastructure only. */
typedef struct { const uint32_t *w; size_t n, pos; int dry; } orc_tape_t;
static uint32_t tape32(orc_tape_t *t) { if (t->pos >= t->n) { t->dry = 1; return 0; } return t->w[t->pos++]; }
static uint64_t tape64(orc_tape_t *t) { const uint64_t hi = tape32(t); return (hi << 32) | tape32(t); }
static size_t orc_rand_range(orc_tape_t *t, size_t x)                       /* sampling.c:68-83 */
{
    const size_t rem = 0xFFFFFFFFu % x;
    for (;;) {
        const size_t y = tape32(t);
        if (t->dry) return 0;
        if (y >= (0xFFFFFFFFu - rem)) continue;
        return y % x;
    }
}
static int orc_discard(orc_tape_t *t, uint32_t thresh) { return thresh ? (tape32(t) < thresh) : 0; }   /* :95-105 */
static void orc_shuffle_fill(orc_tape_t *t, int32_t *v, size_t n, int32_t centre, uint32_t thresh, const uint64_t *cdf, int size)   /* :127-145 */
{
    v[0] = orc_cdt_sample(tape64(t), cdf, size);
    for (size_t i = 1; i < n && !t->dry; i++) {
        const size_t j = orc_rand_range(t, i);
        const int32_t cond = (int32_t)((-(int32_t)(i ^ j) >> 31) & 1);
        v[i] = cond * v[j] + (1 ^ cond) * v[i];
        v[j] = orc_cdt_sample(tape64(t), cdf, size) + centre;
        i -= (size_t)orc_discard(t, thresh);
    }
}
long orc_sample_vector_sca(int32_t *v, size_t n, int32_t centre, int blinding, uint32_t discard, const uint32_t *tape, size_t tape_words,
                           const uint64_t *cdf, int size)
{
    orc_tape_t t = { tape, tape_words, 0, 0 };
    const uint32_t thresh = discard == 2 ? 1u << 28 : discard == 4 ? 1u << 30 : discard == 6 ? 1u << 31 : 0;
    if (n == 0) return 0;
    if (blinding == 0) {                                                       /* :211-228 */
        for (size_t i = 0; i < n && !t.dry; i++) {
            v[i] = orc_cdt_sample(tape64(&t), cdf, size) + centre;
            i -= (size_t)orc_discard(&t, thresh);
        }
    } else {
        orc_shuffle_fill(&t, v, n, centre, thresh, cdf, size);
        if (blinding == 1)                                                     /* :168-191; the closing swap loop exchanges v[i] with itself */
            for (size_t i = 0; i < n && !t.dry; i++) v[i] -= orc_cdt_sample(tape64(&t), cdf, size);
    }
    return t.dry ? -1 : (long)t.pos;
}

/* ---------------------------------------------------------------- batches ---- */

void orc_batch_fwd_ntt(int32_t *out, const int32_t *in, size_t batch, int n, int32_t q, const int32_t *w)
{
    for (size_t b = 0; b < batch; b++) {
        orc_fwd_ntt(out + b * n, n, q, in + b * n, w);
        orc_normalize(out + b * n, (size_t)n, q);
    }
}

void orc_batch_inv_ntt(int32_t *out, const int32_t *in, size_t batch, int n, int32_t q, const int32_t *w, const int32_t *r)
{
    for (size_t b = 0; b < batch; b++) orc_inv_ntt(out + b * n, n, q, in + b * n, w, r);
}

void orc_batch_polymul_ntt(int32_t *out, const int32_t *a, const int32_t *bhat, size_t batch, int n, int32_t q,
                           const int32_t *w, const int32_t *r)
{
    int32_t *fa = malloc(sizeof(int32_t) * (size_t)n);
    for (size_t b = 0; b < batch; b++) {
        orc_fwd_ntt(fa, n, q, a + b * n, w);
        orc_mul_pointwise(fa, n, q, fa, bhat + b * n);
        orc_inv_ntt(out + b * n, n, q, fa, w, r);
    }
    free(fa);
}

/* poly_32.c:132-156 on a word tape: returns 0 and the words consumed, 2 when the tape runs out */
int orc_poly_uniform(int32_t *v, int n, const uint16_t *c, int c_len, const uint32_t *tape, size_t tape_len, size_t *consumed)
{
    size_t pos = 0;
    for (int i = 0; i < n; i++) v[i] = 0;
    for (int j = 0; j < c_len; j++) {
        int i = 0;
        while (i < c[j]) {
            if (pos >= tape_len) { *consumed = pos; return 2; }
            const uint32_t r = tape[pos++];
            const uint32_t idx = (r >> 1) & (uint32_t)(n - 1);
            if (v[idx] == 0) { v[idx] = (r & 1) ? j - c_len : c_len - j; i++; }
        }
    }
    *consumed = pos;
    return 0;
}

/* gaussian_cdf_sample_128 / _192 (src/utils/sampling/gaussian_cdf.c:468-509) over binary_search_128 / _192 (:134-166) and compare_ge_prec (:110-132), for 64-bit limbs:
 * x = W words in the order the sampler draws them (limb 0 first), tab = [size][W] limbs, least significant first.  compare_ge_prec is restated as it is written -
 * a loop from the least to the most significant limb with retval = !(a < b) | (equal & retval) - not as what it amounts to (the top limb's ">=" alone:
 * tests/test_oracle_cdt_hp.py shows the two agree and that a full-width comparison does not). */
static int orc_compare_ge_prec(const uint64_t *x, const uint64_t *y, int words)
{
    uint64_t retval = 1;
    for (int i = 0; i < words; i++) {
        const uint64_t a = x[i], b = y[i];
        const uint64_t equal = !(a ^ b);
        const uint64_t x_lt_y = a < b;
        retval = !x_lt_y | (equal & retval);
    }
    return (int)retval;
}
int32_t orc_cdt_hp_sample(const uint64_t *x, const uint64_t *tab, int size, int words)
{
    uint32_t a = 0;
    for (uint32_t st = (uint32_t)size >> 1; st > 0; st >>= 1) {
        const uint32_t b = a + st;
        if (b < (uint32_t)size && orc_compare_ge_prec(x, tab + (size_t)b * words, words)) a = b;
    }
    return (x[0] & 1) ? (int32_t)a : -(int32_t)a;
}
void orc_cdt_hp_sample_vector(int32_t *v, size_t n, const uint64_t *x, const uint64_t *tab, int size, int words, int32_t centre)
{
    for (size_t i = 0; i < n; i++) v[i] = orc_cdt_hp_sample(x + i * (size_t)words, tab, size, words) + centre;
}
