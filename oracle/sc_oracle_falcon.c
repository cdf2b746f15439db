/*
 * sc_oracle_falcon.c - CPU restatement of the reference's Falcon signing path.  TEST INFRASTRUCTURE ONLY
 * (checker for tests/, smoke() and bench.py's cpu_baseline leg; the product never links or calls it).
 *
 * Follows, for the binary rings (sets 0 and 2, x^n + 1):
 *   src/utils/arith/gpv.c:2961-3090   gaussian_sample_with_tree  (ter == 0 branch)
 *   src/utils/arith/gpv.c:2663-2714   gaussian_lattice_sample_fft (fast Fourier sampling over the LDL tree)
 *   src/utils/arith/falcon_fft.c:203-378, 380-424, 466-483, 543-553, 694-785
 *                                      FFT / iFFT / add / sub / mul_fft / mulconst / split_fft / merge_fft
 *   src/schemes/sig/falcon/falcon.c:511-521, 534-591, 808-935, 955-1040
 *                                      check_norm_bd, oracle_xof + h_function_xof, sign, verify
 * IEEE-754 double arithmetic, one rounding per operation, in the reference's operation order (the reference
 * is built without FMA contraction; this file must be compiled with -ffp-contract=off).
 *
 * The reference's root table fpr_gm_tab (falcon_fft.h, 1024 complex constants) is not copied: gm_init()
 * regenerates it - exp(i*pi*j/2048) in binary128 from a Taylor series and repeated multiplication, rounded to
 * double - and tests/test_oracle_falcon.py pins the result indirectly (the fixtures' s1/s2 only come out
 * bit-exact when every constant is).
 *
 * Randomness: every tree leaf builds a throw-away 64-bit CDF sampler (tail 10, sigma = (float)leaf) on
 * sc->prng_ctx[0] and draws two samples, i.e. two prng_64 = four 32-bit pool words, high word first
 * (src/utils/crypto/prng.c:963-976).  The tape is that word stream.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include "sc_oracle.h"
#include "sc_oracle_keccak.h"

typedef __float128 q_t;

static double gm[2048];
static int gm_ready;

static unsigned brev(unsigned x, int bits)
{
    unsigned r = 0;
    for (int b = 0; b < bits; b++) r |= ((x >> b) & 1u) << (bits - 1 - b);
    return r;
}

/* GM[k], k in [m, 2m) with m = 2^logm, is w^rev(k) for w = exp(i*pi/N): the angle is pi*(2*rev_logm(k-m)+1)/(2m).
 * Entry 0 is (0,0) and entry 1 is written as (-0.0, 1.0) in the reference's table. */
static void gm_init(void)
{
    if (gm_ready) return;
    const q_t pi = (q_t)3.141592653589793116 + (q_t)1.2246467991473532e-16 + (q_t)(-2.9947698097183397e-33);
    const q_t x = pi / 2048, x2 = x * x;
    q_t ts = x, tc = 1, s1 = x, c1 = 1;
    for (int k = 1; k < 16; k++) {
        tc = -tc * x2 / ((2 * k - 1) * (2 * k)); c1 += tc;
        ts = -ts * x2 / ((2 * k) * (2 * k + 1)); s1 += ts;
    }
    static q_t cr[2049], ci[2049];
    cr[0] = 1; ci[0] = 0;
    for (int j = 1; j <= 2048; j++) { cr[j] = cr[j - 1] * c1 - ci[j - 1] * s1; ci[j] = cr[j - 1] * s1 + ci[j - 1] * c1; }
    gm[0] = 0.0; gm[1] = 0.0;
    for (int logm = 0; logm < 10; logm++) {
        const unsigned m = 1u << logm;
        for (unsigned i1 = 0; i1 < m; i1++) {
            const unsigned k = m + i1, j = (2 * brev(i1, logm) + 1) * (2048 / (2 * m));
            gm[2 * k] = (double)cr[j]; gm[2 * k + 1] = (double)ci[j];
        }
    }
    gm[2] = -0.0; gm[3] = 1.0;
    gm_ready = 1;
}

const double *orc_falcon_gm(void) { gm_init(); return gm; }

#define CMUL(dr, di, ar, ai, br, bi) do { double ar_ = (ar), ai_ = (ai), br_ = (br), bi_ = (bi); \
    double r_ = (ar_ * br_) - (ai_ * bi_), i_ = (ar_ * bi_) + (ai_ * br_); (dr) = r_; (di) = i_; } while (0)

/* falcon_fft.c:203-282 */
void orc_falcon_fft(double *f, unsigned logn)
{
    gm_init();
    const size_t n = (size_t)1 << logn, hn = n >> 1;
    size_t t = hn, m = 2;
    for (unsigned u = 1; u < logn; u++, m <<= 1) {
        const size_t ht = t >> 1, hm = m >> 1;
        for (size_t i1 = 0, j1 = 0; i1 < hm; i1++, j1 += t) {
            const double sr = gm[(m + i1) << 1], si = gm[((m + i1) << 1) + 1];
            for (size_t j = j1; j < j1 + ht; j++) {
                double xr = f[j], xi = f[j + hn], yr = f[j + ht], yi = f[j + ht + hn];
                CMUL(yr, yi, yr, yi, sr, si);
                f[j] = xr + yr; f[j + hn] = xi + yi;
                f[j + ht] = xr - yr; f[j + ht + hn] = xi - yi;
            }
        }
        t = ht;
    }
}

/* falcon_fft.c:284-378 */
void orc_falcon_ifft(double *f, unsigned logn)
{
    gm_init();
    const size_t n = (size_t)1 << logn, hn = n >> 1;
    size_t t = 1, m = n;
    for (unsigned u = logn; u > 1; u--) {
        const size_t hm = m >> 1, dt = t << 1;
        for (size_t i1 = 0, j1 = 0; j1 < hn; i1++, j1 += dt) {
            const double sr = gm[(hm + i1) << 1], si = -gm[((hm + i1) << 1) + 1];
            for (size_t j = j1; j < j1 + t; j++) {
                double xr = f[j], xi = f[j + hn], yr = f[j + t], yi = f[j + t + hn];
                f[j] = xr + yr; f[j + hn] = xi + yi;
                xr = xr - yr; xi = xi - yi;
                CMUL(f[j + t], f[j + t + hn], xr, xi, sr, si);
            }
        }
        t = dt; m = hm;
    }
    if (logn > 0) {
        const double ni = ldexp(2, -(int)logn);
        for (size_t u = 0; u < n; u++) f[u] = f[u] * ni;
    }
}

static void poly_mul_fft(double *a, const double *b, unsigned logn)           /* falcon_fft.c:466-483 */
{
    const size_t hn = ((size_t)1 << logn) >> 1;
    for (size_t u = 0; u < hn; u++) CMUL(a[u], a[u + hn], a[u], a[u + hn], b[u], b[u + hn]);
}

static void split_fft(double *f0, double *f1, const double *f, unsigned logn)   /* falcon_fft.c:694-739 */
{
    const size_t n = (size_t)1 << logn, hn = n >> 1, qn = hn >> 1;
    f0[0] = f[0]; f1[0] = f[hn];
    for (size_t u = 0; u < qn; u++) {
        const double ar = f[2 * u], ai = f[2 * u + hn], br = f[2 * u + 1], bi = f[2 * u + 1 + hn];
        double tr = ar + br, ti = ai + bi;
        f0[u] = tr * 0.5; f0[u + qn] = ti * 0.5;
        tr = ar - br; ti = ai - bi;
        CMUL(tr, ti, tr, ti, gm[(u + hn) << 1], -gm[((u + hn) << 1) + 1]);
        f1[u] = tr * 0.5; f1[u + qn] = ti * 0.5;
    }
}

static void merge_fft(double *f, const double *f0, const double *f1, unsigned logn)  /* falcon_fft.c:742-785 */
{
    const size_t n = (size_t)1 << logn, hn = n >> 1, qn = hn >> 1;
    f[0] = f0[0]; f[hn] = f1[0];
    for (size_t u = 0; u < qn; u++) {
        const double ar = f0[u], ai = f0[u + qn];
        double br, bi;
        CMUL(br, bi, f1[u], f1[u + qn], gm[(u + hn) << 1], gm[((u + hn) << 1) + 1]);
        f[2 * u] = ar + br; f[2 * u + hn] = ai + bi;
        f[2 * u + 1] = ar - br; f[2 * u + 1 + hn] = ai - bi;
    }
}

typedef struct { const uint32_t *w; size_t len, pos; int underrun; } ftape_t;
static uint64_t f64w(ftape_t *t)
{
    if (t->pos + 2 > t->len) { t->underrun = 1; return 0; }
    const uint64_t hi = t->w[t->pos], lo = t->w[t->pos + 1];
    t->pos += 2;
    return (hi << 32) | lo;
}

static unsigned treesize(unsigned logn) { return (logn + 1) << logn; }             /* falcon_ldl.h:56-71 */

/* gpv.c:2663-2714 */
static void ffsample(double *z0, double *z1, const double *tree, const double *t0, const double *t1, unsigned logn,
                     double *tmp, ftape_t *tape)
{
    const size_t n = (size_t)1 << logn;
    if (n == 1) {
        const float sigma = (float)tree[0];
        const int size = 1 << orc_cdt_bits(10.0f, sigma);
        uint64_t *cdf = malloc(sizeof(uint64_t) * (size_t)(size < 2 ? 2 : size));
        orc_cdt_create(10.0f, sigma, cdf, size);
        z0[0] = floor(t0[0]) + orc_cdt_sample(f64w(tape), cdf, size);
        z1[0] = floor(t1[0]) + orc_cdt_sample(f64w(tape), cdf, size);
        free(cdf);
        return;
    }
    const size_t hn = n >> 1;
    const double *tree0 = tree + n, *tree1 = tree + n + treesize(logn - 1);
    split_fft(z1, z1 + hn, t1, logn);
    ffsample(tmp, tmp + hn, tree1, z1, z1 + hn, logn - 1, tmp + n, tape);
    merge_fft(z1, tmp, tmp + hn, logn);
    memcpy(tmp, t1, n * sizeof *t1);
    for (size_t u = 0; u < n; u++) tmp[u] = tmp[u] - z1[u];
    poly_mul_fft(tmp, tree, logn);
    for (size_t u = 0; u < n; u++) tmp[u] = tmp[u] + t0[u];
    split_fft(z0, z0 + hn, tmp, logn);
    ffsample(tmp, tmp + hn, tree0, z0, z0 + hn, logn - 1, tmp + n, tape);
    merge_fft(z0, tmp, tmp + hn, logn);
}

/* gpv.c:2961-3090, ter == 0.  sk = b00 | b01 | b10 | b11 | tree as load_skey leaves it (falcon_keygen.h:107-136).
 * Returns 0, or 2 when the tape is too short (4n words are needed). */
int orc_falcon_sample(const double *sk, int n, int q, int logn, const int32_t *c, const uint32_t *tape, size_t tape_len,
                      int32_t *s1, int32_t *s2)
{
    gm_init();
    const double *b00 = sk, *b01 = sk + n, *b10 = sk + 2 * (size_t)n, *b11 = sk + 3 * (size_t)n, *tree = sk + 4 * (size_t)n;
    double *c0 = malloc(sizeof(double) * 11 * (size_t)n), *c1 = c0 + n, *tmp = c1 + n, *z0 = tmp + 7 * (size_t)n, *z1 = z0 + n;
    ftape_t tp = { tape, tape_len, 0, 0 };
    for (int i = 0; i < n; i++) c0[i] = c[i];
    const double ni = 1.0 / (double)(long)q;
    orc_falcon_fft(c0, (unsigned)logn);
    memcpy(c1, c0, sizeof(double) * (size_t)n);
    poly_mul_fft(c1, b01, (unsigned)logn);
    for (int i = 0; i < n; i++) c1[i] = c1[i] * -ni;
    poly_mul_fft(c0, b11, (unsigned)logn);
    for (int i = 0; i < n; i++) c0[i] = c0[i] * ni;
    ffsample(z0, z1, tree, c0, c1, (unsigned)logn, tmp, &tp);
    memcpy(c0, z0, sizeof(double) * (size_t)n);
    memcpy(c1, z1, sizeof(double) * (size_t)n);
    poly_mul_fft(z0, b00, (unsigned)logn);
    poly_mul_fft(z1, b10, (unsigned)logn);
    for (int i = 0; i < n; i++) z0[i] = z0[i] + z1[i];
    memcpy(z1, c0, sizeof(double) * (size_t)n);
    poly_mul_fft(z1, b01, (unsigned)logn);
    memcpy(c0, z0, sizeof(double) * (size_t)n);
    poly_mul_fft(c1, b11, (unsigned)logn);
    for (int i = 0; i < n; i++) c1[i] = c1[i] + z1[i];
    orc_falcon_ifft(c0, (unsigned)logn);
    orc_falcon_ifft(c1, (unsigned)logn);
    if (s1) for (int i = 0; i < n; i++) s1[i] = (int32_t)(c[i] - llrint(c0[i]));
    for (int i = 0; i < n; i++) s2[i] = (int32_t)-llrint(c1[i]);
    free(c0);
    return tp.underrun ? 2 : 0;
}

/* falcon.c:534-591: SHAKE128 (the reference's permutation) over the message, then n 32-bit little-endian words;
 * y = word & (2^q_bits - 1); y -= ((SINT32)(q - y) >> 31) * q - which ADDS q when y > q (the product is taken
 * in UINT32), so c ranges over [0, 2^q_bits + q).  Restated as it behaves. */
void orc_falcon_hash_to_ring(int32_t *c, int n, int q, int q_bits, const uint8_t *msg, size_t mlen)
{
    shake_t sh;
    shake_init(&sh, 128);
    shake_absorb(&sh, msg, mlen);
    shake_final(&sh);
    const uint32_t mask = (1u << q_bits) - 1;
    for (int i = 0; i < n; i++) {
        uint8_t b[4];
        shake_squeeze(&sh, b, 4);
        uint32_t y = ((uint32_t)b[0] | ((uint32_t)b[1] << 8) | ((uint32_t)b[2] << 16) | ((uint32_t)b[3] << 24)) & mask;
        y -= (uint32_t)((int32_t)((uint32_t)q - y) >> 31) * (uint32_t)q;
        c[i] = (int32_t)y;
    }
}

/* falcon.c:511-521: integer squares, FLOAT accumulation from the top index down, sqrtf, ">= bd" fails */
static int norm_fails(float bd, const int32_t *s1, const int32_t *s2, int n)
{
    float norm = 0;
    for (int i = n; i--;) norm += s1[i] * s1[i] + s2[i] * s2[i];
    norm = sqrtf(norm);
    return norm >= bd;
}

/* falcon.c:808-935.  par = the fixture's *_params record (n, q, q_bits, n_bits, set, bd as float bits).
 * Returns 0 = SC_FUNC_SUCCESS with the packed s2 ((q_bits-2)-bit two's complement, MSB first), 1 = norm bound
 * exceeded (the reference gives up, it does not retry), 2 = tape too short. */
int orc_falcon_sign(const int32_t *par, const double *sk, const uint8_t *msg, size_t mlen, const uint32_t *tape, size_t tape_len,
                    uint8_t *sig, size_t *siglen)
{
    const int n = par[0], q = par[1], q_bits = par[2], logn = par[3];
    float bd; memcpy(&bd, &par[5], 4);
    int32_t *c = malloc(sizeof(int32_t) * 3 * (size_t)n), *s1 = c + n, *s2 = s1 + n;
    int rc;
    orc_falcon_hash_to_ring(c, n, q, q_bits, msg, mlen);
    if ((rc = orc_falcon_sample(sk, n, q, logn, c, tape, tape_len, s1, s2)) != 0) goto done;
    orc_center(s2, (size_t)n, q);
    if (norm_fails(bd, s1, s2, n)) { rc = 1; goto done; }
    {
        const int bits = q_bits - 2;
        size_t bp = 0;
        memset(sig, 0, ((size_t)n * (size_t)bits + 7) / 8);
        for (int i = 0; i < n; i++)
            for (int b = bits - 1; b >= 0; b--, bp++)
                sig[bp >> 3] |= (uint8_t)((((uint32_t)s2[i] >> b) & 1u) << (7 - (bp & 7)));
        *siglen = (bp + 7) / 8;
    }
    rc = 0;
done:
    free(c);
    return rc;
}

/* falcon.c:955-1040.  h = the public key in the coefficient domain.  Returns 0 for a valid signature. */
int orc_falcon_verify(const int32_t *par, const int16_t *h, const uint8_t *msg, size_t mlen, const uint8_t *sig, size_t siglen)
{
    const int n = par[0], q = par[1], q_bits = par[2], bits = q_bits - 2;
    float bd; memcpy(&bd, &par[5], 4);
    if (siglen * 8 < (size_t)n * (size_t)bits) return 1;
    int32_t *w = malloc(sizeof(int32_t) * 6 * (size_t)n), *r = w + n, *c = r + n, *s1 = c + n, *s2 = s1 + n, *h32 = s2 + n;
    int rc = 1;
    if (orc_make_tables(q, n, w, r)) goto done;
    size_t bp = 0;
    for (int i = 0; i < n; i++) {
        uint32_t v = 0;
        for (int b = 0; b < bits; b++, bp++) v = (v << 1) | ((sig[bp >> 3] >> (7 - (bp & 7))) & 1u);
        s2[i] = (int32_t)(v << (32 - bits)) >> (32 - bits);
    }
    orc_center(s2, (size_t)n, q);
    orc_falcon_hash_to_ring(c, n, q, q_bits, msg, mlen);
    for (int i = 0; i < n; i++) h32[i] = h[i];
    orc_normalize(h32, (size_t)n, q);
    for (int i = 0; i < n; i++) s1[i] = s2[i];
    orc_normalize(s1, (size_t)n, q);
    orc_polymul(s1, n, q, s1, h32, w, r);
    for (int i = 0; i < n; i++) s1[i] = c[i] - s1[i];
    orc_center(s1, (size_t)n, q);
    rc = norm_fails(bd, s1, s2, n) ? 1 : 0;
done:
    free(w);
    return rc;
}

/* per-leaf sigma in visiting order (tree1 subtree first, then tree0): what a batched engine tabulates at key load */
static void leaves_rec(const double *tree, unsigned logn, float *out, size_t *k)
{
    if (logn == 0) { out[(*k)++] = (float)tree[0]; return; }
    const size_t n = (size_t)1 << logn;
    leaves_rec(tree + n + treesize(logn - 1), logn - 1, out, k);
    leaves_rec(tree + n, logn - 1, out, k);
}
void orc_falcon_leaf_sigmas(const double *sk, int n, int logn, float *out)
{
    size_t k = 0;
    leaves_rec(sk + 4 * (size_t)n, (unsigned)logn, out, &k);
}

/* ---- DLP-IBE extract (src/schemes/ibe/dlp/dlp_ibe.c:874-1043 with DLP_IBE_USE_ENHANCED_EXTRACT = 1, H_NTT_OPTIMISATION, XOF oracle) ----
 * par = {n, q, q_bits, n_bits, set, ubits, 0, 0}; tree = sc->dlp_ibe->master_tree (load_skey layout, as Falcon's sk).
 * id_function (:805-816): SHAKE128(id) -> n little-endian words, masked to q_bits, minus q when >= q; the result is taken as an NTT-domain
 * vector (inv_ntt_32_32_large, :939); gaussian_sample_with_tree(..., s1 = NULL, s2 = v) (:981); center_32; ubits-bit signed raw coding. */
int orc_dlp_extract(const int32_t *par, const double *tree, const uint8_t *id, size_t idlen, const uint32_t *tape, size_t tape_len, uint8_t *sk)
{
    const int n = par[0], q = par[1], q_bits = par[2], logn = par[3], ubits = par[5];
    int32_t *w = malloc(sizeof(int32_t) * (size_t)n * 5), *r = w + n, *c = r + n, *s1 = c + n, *v = s1 + n;
    if (orc_make_tables(q, n, w, r)) { free(w); return 1; }
    shake_t sh;
    shake_init(&sh, 128);
    shake_absorb(&sh, id, idlen);
    shake_final(&sh);
    const uint32_t mask = (1u << q_bits) - 1;
    for (int i = 0; i < n; i++) {
        uint8_t b[4];
        shake_squeeze(&sh, b, 4);
        int32_t y = (int32_t)(((uint32_t)b[0] | ((uint32_t)b[1] << 8) | ((uint32_t)b[2] << 16) | ((uint32_t)b[3] << 24)) & mask);
        y -= (y >= q) * q;
        c[i] = y;
    }
    orc_inv_ntt(c, n, q, c, w, r);
    const int rc = orc_falcon_sample(tree, n, q, logn, c, tape, tape_len, s1, v);
    if (rc) { free(w); return rc; }
    orc_center(v, (size_t)n, q);
    const size_t bytes = ((size_t)n * ubits + 7) / 8;
    memset(sk, 0, bytes);
    size_t bp = 0;
    for (int i = 0; i < n; i++) for (int b = ubits - 1; b >= 0; b--, bp++) sk[bp >> 3] |= (uint8_t)((((uint32_t)v[i] >> b) & 1u) << (7 - (bp & 7)));
    free(w);
    return 0;
}

/* ---- DLP-IBE encrypt / decrypt (dlp_ibe.c:1140-1370, 1372-1520; SC_IBE_MESSAGE_LENGTH_N, XOF oracles, H_NTT_OPTIMISATION) ----
 * par = {n, q, q_bits, n_bits, set, ubits, m_scale, l}.  h_ntt: the second half of the public key (NTT(h), SINT32); uk_ntt: the NTT of the
 * user key as dlp_ibe_secret_key keeps it (centred, :1100-1104).  Randomness (tape = prng_32 word stream of sc->prng_ctx[0]): n/32 words of
 * key bits (LSB first), then n prng_var(6) draws (prng.c:1017-1048 reservoir): bit 0 / 1 = |e1| / sign, 2 / 3 = e2, 4 / 5 = e3.
 * Ciphertext: u (q_bits, signed) | v >> l (q_bits - l, signed) | message XOR SHAKE128(k as n bytes)[0, n/8), MSB first. */
typedef struct { const uint32_t *w; size_t len, pos; uint32_t var_buf; int var_bits; int underrun; } dtape_t;
static uint32_t d32(dtape_t *t) { if (t->pos >= t->len) { t->underrun = 1; return 0; } return t->w[t->pos++]; }
static uint32_t dvar(dtape_t *t, int n)
{
    uint32_t ret = t->var_buf;
    if (t->var_bits < n) {
        const int bits = n - t->var_bits;
        ret <<= bits;
        t->var_buf = d32(t);
        ret |= t->var_buf & ((1u << bits) - 1u);
        t->var_buf >>= bits;
        t->var_bits = 32 - bits;
    } else {
        t->var_buf >>= n;
        t->var_bits -= n;
    }
    return ret & ((1u << n) - 1u);
}

static void dlp_id(int32_t *c, int n, int q, int q_bits, const uint8_t *id, size_t idlen)
{
    shake_t sh;
    shake_init(&sh, 128);
    shake_absorb(&sh, id, idlen);
    shake_final(&sh);
    const uint32_t mask = (1u << q_bits) - 1;
    for (int i = 0; i < n; i++) {
        uint8_t b[4];
        shake_squeeze(&sh, b, 4);
        int32_t y = (int32_t)(((uint32_t)b[0] | ((uint32_t)b[1] << 8) | ((uint32_t)b[2] << 16) | ((uint32_t)b[3] << 24)) & mask);
        c[i] = y - (y >= q) * q;
    }
}

static void dlp_pad(uint8_t *k, int n)                         /* k_function_xof, dlp_ibe.c:850-862: k (n bytes of 0 / 1) -> n/8 bytes in place */
{
    shake_t sh;
    shake_init(&sh, 128);
    shake_absorb(&sh, k, (size_t)n);
    shake_final(&sh);
    shake_squeeze(&sh, k, (size_t)n / 8);
}

size_t orc_dlp_ct_bytes(const int32_t *par) { return ((size_t)par[0] * (size_t)(2 * par[2] - par[7]) + 7) / 8 + (size_t)par[0] / 8; }

int orc_dlp_encrypt(const int32_t *par, const int32_t *h_ntt, const uint8_t *id, size_t idlen, const uint8_t *msg, const uint32_t *tape, size_t tape_len, uint8_t *ct)
{
    const int n = par[0], q = par[1], q_bits = par[2], m_scale = par[6], l = par[7], q2 = q >> 1;
    int32_t *w = malloc(sizeof(int32_t) * (size_t)n * 9), *r = w + n, *c = r + n, *e1 = c + n, *e2 = e1 + n, *e3 = e2 + n, *u = e3 + n, *v = u + n, *ek = v + n;
    uint8_t *k = malloc((size_t)n);
    dtape_t T = { tape, tape_len, 0, 0, 0, 0 };
    int rc = 2;
    if (orc_make_tables(q, n, w, r)) { free(w); free(k); return 1; }
    dlp_id(c, n, q, q_bits, id, idlen);
    for (int i = 0; i < n; i += 32) { uint32_t x = d32(&T); for (int j = 0; j < 32; j++) { k[i + j] = x & 1; x >>= 1; } }
    for (int i = 0; i < n; i++) ek[i] = (k[i] & 1) * m_scale;
    for (int i = 0; i < n; i++) {
        uint32_t b = dvar(&T, 6);
        e1[i] = (int32_t)(b & 1); b >>= 1; if (b & 1) e1[i] = -e1[i]; b >>= 1;
        e2[i] = (int32_t)(b & 1); b >>= 1; if (b & 1) e2[i] = -e2[i]; b >>= 1;
        e3[i] = (int32_t)(b & 1); b >>= 1; if (b & 1) e3[i] = -e3[i];
    }
    if (T.underrun) goto done;
    orc_polymul_ntt(u, n, q, e3, h_ntt, w, r);                                    /* e3 * h, canonical */
    orc_polymul_ntt(v, n, q, e3, c, w, r);                                        /* e3 * H(id) (H(id) is an NTT-domain vector) */
    for (int i = 0; i < n; i++) { u[i] += q2 + e1[i]; v[i] += q2 + e2[i] + ek[i]; }
    orc_normalize(u, (size_t)n, q); orc_normalize(v, (size_t)n, q);
    for (int i = 0; i < n; i++) { u[i] -= q2; v[i] -= q2; v[i] >>= l; }
    dlp_pad(k, n);
    memset(ct, 0, orc_dlp_ct_bytes(par));
    size_t bp = 0;
    for (int i = 0; i < n; i++) for (int b = q_bits - 1; b >= 0; b--, bp++) ct[bp >> 3] |= (uint8_t)((((uint32_t)u[i] >> b) & 1u) << (7 - (bp & 7)));
    for (int i = 0; i < n; i++) for (int b = q_bits - l - 1; b >= 0; b--, bp++) ct[bp >> 3] |= (uint8_t)((((uint32_t)v[i] >> b) & 1u) << (7 - (bp & 7)));
    for (int i = 0; i < n / 8; i++) ct[(bp >> 3) + i] = (uint8_t)(k[i] ^ msg[i]);
    rc = 0;
done:
    free(w); free(k);
    return rc;
}

int orc_dlp_decrypt(const int32_t *par, const int32_t *uk_ntt, const uint8_t *ct, size_t ctlen, uint8_t *msg)
{
    const int n = par[0], q = par[1], q_bits = par[2], l = par[7], q2 = q >> 1, q4 = q >> 2, vb = q_bits - l;
    if (ctlen < orc_dlp_ct_bytes(par)) return 1;
    int32_t *w = malloc(sizeof(int32_t) * (size_t)n * 4), *r = w + n, *u = r + n, *v = u + n;
    uint8_t *k = malloc((size_t)n);
    if (orc_make_tables(q, n, w, r)) { free(w); free(k); return 1; }
    size_t bp = 0;
    for (int i = 0; i < n; i++) { uint32_t x = 0; for (int b = 0; b < q_bits; b++, bp++) x = (x << 1) | ((ct[bp >> 3] >> (7 - (bp & 7))) & 1u); u[i] = (int32_t)(x << (32 - q_bits)) >> (32 - q_bits); }
    for (int i = 0; i < n; i++) { uint32_t x = 0; for (int b = 0; b < vb; b++, bp++) x = (x << 1) | ((ct[bp >> 3] >> (7 - (bp & 7))) & 1u); v[i] = ((int32_t)(x << (32 - vb)) >> (32 - vb)) << l; }
    orc_polymul_ntt(u, n, q, u, uk_ntt, w, r);
    for (int i = 0; i < n; i++) v[i] -= u[i];
    orc_normalize(v, (size_t)n, q);
    for (int i = 0; i < n; i++) { const int32_t x = v[i] + q4; k[i] = (x >= q2 && x < q) ? 1 : 0; }
    dlp_pad(k, n);
    for (int i = 0; i < n / 8; i++) msg[i] = (uint8_t)(ct[(bp >> 3) + i] ^ k[i]);
    free(w); free(k);
    return 0;
}

/* ---- ENS / DLP signature verification (src/schemes/sig/ens_dlp/ens_dlp_sig.c:1564-1665, without message recovery) ----
 * par = {n, q, q_bits, n_bits, set, bd as float bits, 0, 0}; h: the public key (SINT16, coefficient domain).  s1 = the signature's n
 * q_bits-bit signed fields (not centred); t = H(m) by SHAKE128 with q SUBTRACTED when >= q (:728-742); s2 = centre(t - h s1);
 * ||(s1, s2)|| in single precision from the top index down (:1536-1562).  0 = valid. */
int orc_ensdlp_verify(const int32_t *par, const int16_t *h, const uint8_t *msg, size_t mlen, const uint8_t *sig, size_t siglen)
{
    const int n = par[0], q = par[1], q_bits = par[2];
    float bd; memcpy(&bd, &par[5], 4);
    if (siglen * 8 < (size_t)n * (size_t)q_bits) return 1;
    int32_t *w = malloc(sizeof(int32_t) * 7 * (size_t)n), *r = w + n, *c = r + n, *s1 = c + n, *s2 = s1 + n, *h32 = s2 + n, *t = h32 + n;
    int rc = 1;
    if (orc_make_tables(q, n, w, r)) goto done;
    size_t bp = 0;
    for (int i = 0; i < n; i++) {
        uint32_t v = 0;
        for (int b = 0; b < q_bits; b++, bp++) v = (v << 1) | ((sig[bp >> 3] >> (7 - (bp & 7))) & 1u);
        s1[i] = (int32_t)(v << (32 - q_bits)) >> (32 - q_bits);
    }
    dlp_id(c, n, q, q_bits, msg, mlen);
    for (int i = 0; i < n; i++) { h32[i] = h[i]; t[i] = s1[i]; }
    orc_normalize(h32, (size_t)n, q); orc_normalize(t, (size_t)n, q);
    orc_polymul(t, n, q, t, h32, w, r);
    for (int i = 0; i < n; i++) s2[i] = c[i] - t[i];
    orc_center(s2, (size_t)n, q);
    rc = norm_fails(bd, s1, s2, n) ? 1 : 0;
done:
    free(w);
    return rc;
}

/* ---- ENS / DLP signing (src/schemes/sig/ens_dlp/ens_dlp_sig.c:1171-1315, without message recovery) --------------------------------------
 * The signer is Klein's sampler over the 2n x 2n basis B = [[A(g), -A(f)], [A(G), -A(F)]] and its SINGLE-precision Gram-Schmidt
 * orthogonalisation (GSO_TYPE = FLOAT, ens_dlp_sig.h:26-37), both kept in the context by create_gpv_matrices (:364-421). */

/* gpv_expand_basis + poly_limb_anticirculant (gpv.c:687-760): row i of A(p) is x^i p mod (x^n + 1).  The scratch matrix is allocated (zeroed) once
 * and only the entries of coefficients up to deg(p) are written, so a polynomial whose top coefficients are zero leaves the PREVIOUS polynomial's
 * entries standing there (order g, f, G, F) - restated as it behaves. */
int orc_ensdlp_expand_basis(int n, const int32_t *fgFG, int32_t *b)
{
    const size_t N = (size_t)n, N2 = 2 * N;
    int32_t *A = calloc(N * N, sizeof(int32_t));
    const int32_t *polys[4] = { fgFG + N, fgFG, fgFG + 3 * N, fgFG + 2 * N };          /* g, f, G, F */
    for (int m = 0; m < 4; m++) {
        const int32_t *p = polys[m];
        size_t deg = N - 1;
        while (p[deg] == 0 && deg > 0) deg--;                                           /* poly_32_degree, poly_32.c:176-188 */
        for (size_t i = 0; i < N; i++) {
            for (size_t j = i; j <= deg + i && j < N; j++) A[i * N + j] = p[j - i];
            for (size_t j = 0; j + N <= deg + i; j++) A[i * N + j] = -p[j + N - i];
        }
        const size_t r0 = (m >= 2) ? N : 0, c0 = (m & 1) ? N : 0;
        for (size_t i = 0; i < N; i++)
            for (size_t j = 0; j < N; j++) b[(r0 + i) * N2 + c0 + j] = (m & 1) ? -A[i * N + j] : A[i * N + j];
    }
    free(A);
    return 0;
}

static float dotf(const float *a, const float *b, size_t n) { float d = 0; for (size_t k = 0; k < n; k++) d += a[k] * b[k]; return d; }

/* modified_gram_schmidt_fast_flt (gpv.c:1021-1117) and gpv_precompute_inv_flt (:977-984): b_gs is 2n x 2n, inv_norm 2n */
void orc_ensdlp_gso(int n, int q, const int32_t *fgFG, float *b_gs, float *inv_norm)
{
    const size_t N = (size_t)n, N2 = 2 * N;
    const int32_t *f = fgFG, *g = fgFG + N;
    float *v = malloc(sizeof(float) * 2 * N2), *v1 = v + N2;
    float C_k, D_k;
    for (size_t i = 0; i < N; i++) { b_gs[i] = (float)g[i]; b_gs[N + i] = (float)-f[i]; }
    for (int half = 0; half < 2; half++) {
        const size_t base = half ? N : 0;
        if (half) {
            const float D_k_inv = 1 / D_k;
            for (size_t i = 0; i < N; i++) {
                b_gs[N * N2 + N + i] = b_gs[(N - 1) * N2 + N - 1 - i] * (float)q * D_k_inv;
                b_gs[N * N2 + i] = -b_gs[(N - 1) * N2 + N2 - 1 - i] * (float)q * D_k_inv;
            }
        }
        const float *r0 = b_gs + base * N2;
        for (size_t i = 0; i + 1 < N; i++) { v[i] = r0[i + 1]; v[N + i] = r0[i + N + 1]; }
        v[N - 1] = -r0[0];
        v[N2 - 1] = -r0[N];
        memcpy(v1, v, sizeof(float) * N2);
        C_k = dotf(r0, v1, N2);
        D_k = half ? dotf(r0, r0, N2) : dotf(v, v, N2);
        for (size_t i = base + 1; i < base + N; i++) {
            const float aux = C_k / D_k;
            float *ri = b_gs + i * N2;
            const float *rp = ri - N2;
            ri[0] = -rp[N - 1] + aux * v[N - 1];
            ri[N] = -rp[N2 - 1] + aux * v[N2 - 1];
            for (size_t j = 1; j < N; j++) {
                ri[j] = rp[j - 1] - aux * v[j - 1];
                ri[N + j] = rp[N + j - 1] - aux * v[N + j - 1];
            }
            for (size_t j = 0; j < N2; j++) v[j] -= aux * rp[j];
            const float C_ko = C_k, D_ko = D_k;
            C_k = dotf(ri, v1, N2);
            D_k = D_ko - C_ko * C_ko / D_ko;
        }
    }
    for (size_t j = 0; j < N2; j++) {
        const double sqr = dotf(b_gs + j * N2, b_gs + j * N2, N2);
        inv_norm[j] = (float)(1.0 / sqrt(sqr));
    }
    free(v);
}

/* gaussian_lattice_sample_flt (gpv.c:2409-2492) with flags = 0: from the last basis row down, d = <ci, b_gs[j]> / ||b_gs[j]||^2 (a
 * sequential single-precision sum over SINT64 -> FLOAT conversions), a FRESH 64-bit CDT sampler of sigma_j = s_f / ||b_gs[j]|| (tail 10) per row,
 * z = sample + (SINT32)d (truncation, not floor), ci -= z b[j].  Then s1 = centre(ci[n, 2n)), n q_bits-bit signed fields.
 * tape = the prng_32 word stream of sc->prng_ctx[0] (one prng_64 per row: 4n words).  Returns 0, 2 if the tape is too short. */
int orc_ensdlp_sign(const int32_t *par, const int32_t *b, const float *b_gs, const float *inv_norm, const uint8_t *msg, size_t mlen,
                    const uint32_t *tape, size_t tape_len, uint8_t *sig)
{
    const int n = par[0], q = par[1], q_bits = par[2];
    const size_t N = (size_t)n, N2 = 2 * N;
    int32_t *c = malloc(sizeof(int32_t) * N);
    int64_t *ci = malloc(sizeof(int64_t) * N2);
    dlp_id(c, n, q, q_bits, msg, mlen);
    for (size_t j = 0; j < N; j++) { ci[j] = c[j]; ci[N + j] = 0; }
    const float sig0 = (float)(2.0L / inv_norm[0]);                                  /* ens_dlp_sig.c:1253 */
    const double s_f = sig0;
    ftape_t tp = { tape, tape_len, 0, 0 };
    uint64_t cdf[4096];
    for (size_t j = N2; j--;) {
        const float *row = b_gs + j * N2;
        float dot = 0;
        for (size_t k = 0; k < N2; k++) dot += (float)ci[k] * row[k];
        const float d = dot * inv_norm[j] * inv_norm[j];
        const float sg = (float)(s_f * inv_norm[j]);
        const int size = 1 << orc_cdt_bits(10.0f, sg);
        if (size > 4096) { free(c); free(ci); return 1; }
        orc_cdt_create(10.0f, sg, cdf, size);
        const int64_t z = (int64_t)orc_cdt_sample(f64w(&tp), cdf, size) + (int32_t)d;
        const int32_t *brow = b + j * N2;
        for (size_t i = N2; i--;) ci[i] -= z * brow[i];
    }
    for (size_t j = 0; j < N; j++) c[j] = (int32_t)ci[N + j];
    orc_center(c, N, q);
    const size_t bytes = (N * (size_t)q_bits + 7) / 8;
    memset(sig, 0, bytes);
    size_t bp = 0;
    for (size_t i = 0; i < N; i++) for (int bb = q_bits - 1; bb >= 0; bb--, bp++) sig[bp >> 3] |= (uint8_t)((((uint32_t)c[i] >> bb) & 1u) << (7 - (bp & 7)));
    free(c); free(ci);
    return tp.underrun ? 2 : 0;
}

/* ---- ENS / DLP signatures WITH MESSAGE RECOVERY (src/schemes/sig/ens_dlp/ens_dlp_sig.c:1319-1501 sign, 1663-1826 verify; the XOF forms of the
 * random oracles, which is what the default build compiles: ens_dlp_sig.h:31) ------------------------------------------------------------------
 * par = {n, q, q_bits, n_bits, set, bd as float bits, k_recovery, 0}.  The message m = m1 || m2: the first (n - k)(q_bits - 1) bits (m1) ride in
 * the target t1 = F(t2) + m1 of the lattice point, t2 = H'(m) = k coefficients; the signature is s1 | s2 (n q_bits-bit signed fields each) and the
 * signer leaves m2 in the message buffer. */

/* H'(m): SHAKE128 over m, zero-extended to min_len bytes when shorter (oracle_recovery_xof, :686-697; SC_MALLOC zeroes) / as it is (oracle_xof,
 * :660-665), then k 32-bit words masked to q_bits with q subtracted when >= q (h_function_xof, :728-742) */
static void rec_hash(int32_t *out, int k, int q, int q_bits, const uint8_t *m, size_t mlen, size_t min_len)
{
    shake_t sh;
    shake_init(&sh, 128);
    shake_absorb(&sh, m, mlen);
    for (size_t i = mlen; i < min_len; i++) { const uint8_t z = 0; shake_absorb(&sh, &z, 1); }
    shake_final(&sh);
    const uint32_t mask = (1u << q_bits) - 1;
    for (int i = 0; i < k; i++) {
        uint8_t b[4];
        shake_squeeze(&sh, b, 4);
        int32_t y = (int32_t)(((uint32_t)b[0] | ((uint32_t)b[1] << 8) | ((uint32_t)b[2] << 16) | ((uint32_t)b[3] << 24)) & mask);
        out[i] = y - (y >= q) * q;
    }
}

/* F (f_function, :746-792, XOF form): SHAKE128 over the k words of t2 as they lie in memory, squeezed a block (168 bytes) at a time, two bytes per
 * output, first byte high, masked to q_bits - 1 bits */
static void rec_f(int32_t *out, int count, int q_bits, const int32_t *in, int k)
{
    shake_t sh;
    shake_init(&sh, 128);
    for (int i = 0; i < k; i++) {
        const uint32_t v = (uint32_t)in[i];
        const uint8_t b[4] = { (uint8_t)v, (uint8_t)(v >> 8), (uint8_t)(v >> 16), (uint8_t)(v >> 24) };
        shake_absorb(&sh, b, 4);
    }
    shake_final(&sh);
    const int32_t mask = (1 << (q_bits - 1)) - 1;
    for (int i = 0; i < count; i++) {
        uint8_t b[2];
        shake_squeeze(&sh, b, 2);
        out[i] = (((int32_t)b[0] << 8) | (int32_t)b[1]) & mask;
    }
}

/* bit `pos` (MSB first) of a byte string zero-extended past its end: what the reference's packer hands out (packer.c:265-297; its buffer is zeroed
 * and a read past it leaves the bits it had) */
static inline uint32_t bit_at(const uint8_t *m, size_t len, size_t pos) { return (pos >> 3) < len ? (m[pos >> 3] >> (7 - (pos & 7))) & 1u : 0u; }

size_t orc_ensdlp_rec_m1_bits(const int32_t *par) { return (size_t)(par[0] - par[6]) * (size_t)(par[2] - 1); }

/* ens_dlp_sig_sign_recovery.  m (mlen bytes) in; sig (2 n q_bits / 8 bytes) and m2 (at most mlen bytes; *m2len) out.  0 = signed, 2 = tape too short */
int orc_ensdlp_sign_recovery(const int32_t *par, const int32_t *b, const float *b_gs, const float *inv_norm, const uint8_t *msg, size_t mlen,
                             const uint32_t *tape, size_t tape_len, uint8_t *sig, uint8_t *m2, size_t *m2len)
{
    const int n = par[0], q = par[1], q_bits = par[2], k = par[6];
    const size_t N = (size_t)n, N2 = 2 * N;
    const size_t m1_bits = orc_ensdlp_rec_m1_bits(par), m1_bytes = (m1_bits + 7) >> 3;
    int32_t *c = malloc(sizeof(int32_t) * N);
    int64_t *ci = malloc(sizeof(int64_t) * N2);
    rec_hash(c + n - k, k, q, q_bits, msg, mlen, m1_bytes);
    rec_f(c, n - k, q_bits, c + n - k, k);
    {   /* add_recovery_msg (:797-832): the first min(m1_bytes, mlen) bytes, q_bits - 1 bits per coefficient, no reduction */
        const size_t avail = m1_bytes < mlen ? m1_bytes : mlen;
        size_t bp = 0;
        for (int i = 0; i < n - k; i++) {
            uint32_t v = 0;
            for (int bb = 0; bb < q_bits - 1; bb++, bp++) v = (v << 1) | bit_at(msg, avail, bp);
            c[i] += (int32_t)v;
        }
    }
    for (size_t j = 0; j < N; j++) { ci[j] = c[j]; ci[N + j] = 0; }
    const float sig0 = (float)(2.0L / inv_norm[0]);                                  /* :1404 */
    const double s_f = sig0;
    ftape_t tp = { tape, tape_len, 0, 0 };
    uint64_t cdf[4096];
    for (size_t j = N2; j--;) {                                                      /* gaussian_lattice_sample_flt, gpv.c:2409-2492, as in orc_ensdlp_sign */
        const float *row = b_gs + j * N2;
        float dot = 0;
        for (size_t kk = 0; kk < N2; kk++) dot += (float)ci[kk] * row[kk];
        const float d = dot * inv_norm[j] * inv_norm[j];
        const float sg = (float)(s_f * inv_norm[j]);
        const int size = 1 << orc_cdt_bits(10.0f, sg);
        if (size > 4096) { free(c); free(ci); return 1; }
        orc_cdt_create(10.0f, sg, cdf, size);
        const int64_t z = (int64_t)orc_cdt_sample(f64w(&tp), cdf, size) + (int32_t)d;
        const int32_t *brow = b + j * N2;
        for (size_t i = N2; i--;) ci[i] -= z * brow[i];
    }
    /* s1 = centre(ci[n, 2n)), s2 = centre(ci[0, n)) (gpv.c:2480-2489, :1418, :1441), one packer stream */
    const size_t bytes = (2 * N * (size_t)q_bits + 7) / 8;
    memset(sig, 0, bytes);
    size_t bp = 0;
    for (int half = 0; half < 2; half++) {
        for (size_t j = 0; j < N; j++) c[j] = (int32_t)ci[(half ? 0 : N) + j];
        orc_center(c, N, q);
        for (size_t i = 0; i < N; i++) for (int bb = q_bits - 1; bb >= 0; bb--, bp++) sig[bp >> 3] |= (uint8_t)((((uint32_t)c[i] >> bb) & 1u) << (7 - (bp & 7)));
    }
    /* m2 (:1455-1479): what lies behind the m1 bits, shifted up to a byte boundary - byte for byte as the reference's loop leaves it (its last byte
     * carries no bits of its own when the shift is not zero; nothing at all is kept of a message that ends inside the last m1 byte) */
    if (mlen <= m1_bytes) *m2len = 0;
    else {
        const unsigned shiftu = (unsigned)(m1_bits & 7), shiftl = 8 - shiftu;
        const uint32_t low = (1u << shiftu) - 1;                                     /* sc_bit_reverse_8 of it: the top shiftu bits */
        uint32_t maskl = 0;
        for (int i = 0; i < 8; i++) if (low & (1u << i)) maskl |= 0x80u >> i;
        const uint32_t masku = maskl ^ 0xFF;
        const uint8_t *ptr = msg + (m1_bits >> 3);
        uint8_t *w = m2;
        uint8_t byte = *ptr++;
        uint32_t msb = (byte & masku) << shiftu, lsb;
        while (ptr < msg + mlen) {
            byte = *ptr++;
            lsb = (byte & maskl) >> shiftl;
            *w++ = (uint8_t)(msb | lsb);
            msb = (byte & masku) << shiftu;
        }
        lsb = (byte & maskl) << shiftl;
        *w++ = (uint8_t)(msb | lsb);
        *m2len = (size_t)(w - m2);
    }
    free(c); free(ci);
    return tp.underrun ? 2 : 0;
}

/* ens_dlp_sig_verify_recovery.  buf holds m2 (*len bytes) followed by whatever the caller has there, cap bytes in all (the reference rebuilds the
 * message in place in the caller's buffer); on return buf / *len = the rebuilt message.  0 = valid. */
int orc_ensdlp_verify_recovery(const int32_t *par, const int16_t *h, uint8_t *buf, size_t *len, size_t cap, const uint8_t *sig, size_t siglen)
{
    const int n = par[0], q = par[1], q_bits = par[2], k = par[6];
    float bd; memcpy(&bd, &par[5], 4);
    const size_t m1_bits = orc_ensdlp_rec_m1_bits(par), m1_bytes = (m1_bits + 7) >> 3;
    if (siglen * 8 < 2 * (size_t)n * (size_t)q_bits) return 1;
    const size_t m2_len = *len, m2_bits = m2_len << 3;
    if (cap < m1_bytes + m2_len + 1) return 1;
    int32_t *s1 = malloc(sizeof(int32_t) * 4 * (size_t)n), *s2 = s1 + n, *t = s2 + n, *fo = t + n;
    int rc = 1;
    size_t bp = 0;
    for (int half = 0; half < 2; half++)
        for (int i = 0; i < n; i++) {
            uint32_t v = 0;
            for (int bb = 0; bb < q_bits; bb++, bp++) v = (v << 1) | ((sig[bp >> 3] >> (7 - (bp & 7))) & 1u);
            (half ? s2 : s1)[i] = (int32_t)(v << (32 - q_bits)) >> (32 - q_bits);
        }
    if (norm_fails(bd, s1, s2, n)) goto done;                                        /* :1726-1729, on the fields as decoded */
    /* m2 back to where it sat in m (:1733-1759) */
    *len = m1_bytes;
    if (m2_bits) {
        const unsigned shiftu = (unsigned)(m1_bits & 7), shiftl = 8 - shiftu;
        const uint32_t masku = (1u << shiftl) - 1, maskl = masku ^ 0xFF;
        const uint8_t *ptr = buf + m2_len - (shiftu != 0);
        uint8_t *w = buf + ((m1_bits + m2_bits) >> 3) - (shiftu != 0);
        uint8_t byte = *ptr--;
        uint32_t lsb = (byte >> shiftu) & masku, msb;
        while (ptr >= buf) {
            byte = *ptr--;
            msb = ((uint32_t)byte << shiftl) & maskl;
            *w-- = (uint8_t)(msb | lsb);
            lsb = (byte >> shiftu) & masku;
        }
        *w = (uint8_t)lsb;
        *len = (m1_bits + m2_bits) >> 3;
    }
    /* t = s1 h + s2 (small_mul_mod_ring, :892-912: a schoolbook product in 32-bit words that wrap), normalised */
    for (int j = 0; j < n; j++) {
        uint32_t acc = 0;
        for (int kk = 0; kk < n; kk++) {
            const uint32_t p = (uint32_t)s1[kk] * (uint32_t)(int32_t)h[(j - kk) & (n - 1)];
            acc += (kk <= j) ? p : 0u - p;
        }
        t[j] = (int32_t)(acc + (uint32_t)s2[j]);
    }
    orc_normalize(t, (size_t)n, q);
    rec_f(fo, n - k, q_bits, t + n - k, k);
    for (int i = 0; i < n - k; i++) t[i] -= fo[i];
    orc_normalize(t, (size_t)n, q);
    {   /* recover_msg (:835-890): every m1 coefficient must fit q_bits - 1 bits; the bits go to the front of the buffer, the last byte shared with m2 */
        const uint32_t bound = (1u << (q_bits - 1)) - 1;
        for (int i = 0; i < n - k; i++) if ((uint32_t)t[i] > bound) goto done;
        uint8_t *tmp = calloc(m1_bytes + 4, 1);
        size_t wp = 0;
        for (int i = 0; i < n - k; i++) for (int bb = q_bits - 2; bb >= 0; bb--, wp++) tmp[wp >> 3] |= (uint8_t)((((uint32_t)t[i] >> bb) & 1u) << (7 - (wp & 7)));
        const unsigned offset = (unsigned)(m1_bits & 7);
        const size_t last = m1_bytes - 1;
        memcpy(buf, tmp, last);
        if (offset == 0) buf[last] = tmp[last];
        else {
            const uint8_t mask = (uint8_t)((((1u << (8 - offset)) - 1) ^ 0xFF));
            buf[last] = (uint8_t)((buf[last] & (mask ^ 0xFF)) | (tmp[last] & mask));
        }
        free(tmp);
    }
    rec_hash(fo, k, q, q_bits, buf, *len, 0);
    rc = 0;
    for (int i = 0; i < k; i++) if (fo[i] != t[n - k + i]) rc = 1;
done:
    free(s1);
    return rc;
}
