/*
 * sc_oracle_bliss.c - CPU restatement of the reference's BLISS-B sign / verify path.
 * TEST INFRASTRUCTURE ONLY (see sc_oracle.h).  Pinned: tests/test_oracle_bliss.py reproduces the reference's
 * signatures byte for byte from (key, message, random tape) fixtures (tests/golden/bliss_ref.txt, made by
 * oracle/ref_driver.c from the reference build).  Paths are relative to the reference tree.
 *
 * Randomness: bliss_b_sign draws everything from one CSPRNG context as a stream of 32-bit pool words
 * (src/utils/crypto/prng.c:963-1048): prng_64 = two words (high first), prng_var(n) = a 32-bit reservoir
 * refilled by prng_32, prng_bit = prng_var(1), prng_double = prng_var(27), prng_var(26).  The stream ("tape")
 * is an input here; the consumption rules are restated below.
 */
#include "sc_oracle.h"
#include "sc_oracle_keccak.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>

typedef struct { const uint32_t *w; size_t len, pos; uint32_t var_buf; int var_bits; int underrun; } tape_t;

static uint32_t t32(tape_t *t) { if (t->pos >= t->len) { t->underrun = 1; return 0; } return t->w[t->pos++]; }
static uint64_t t64(tape_t *t) { uint64_t hi = t32(t); return (hi << 32) | t32(t); }                     /* prng.c:963-976 */
static uint32_t tvar(tape_t *t, int n)                                                                       /* prng.c:1017-1048 */
{
    uint32_t mask = (n >= 32) ? 0xFFFFFFFFu : ((1u << n) - 1u), ret = t->var_buf;
    if (t->var_bits < n) {
        const int bits = n - t->var_bits;
        ret <<= bits;
        t->var_buf = t32(t);
        ret |= t->var_buf & ((1u << bits) - 1u);
        t->var_buf >>= bits;
        t->var_bits = 32 - bits;
    } else {
        t->var_buf >>= n;
        t->var_bits -= n;
    }
    return ret & mask;
}
static double tdouble(tape_t *t) { uint32_t a = tvar(t, 27), b = tvar(t, 26); return (a * 67108864.0 + b) * 1.11022302462516e-16; }   /* prng.c:1010-1015 */

typedef struct { int n, q, q_bits, n_bits, d, p, kappa, b_inf, b_l2, z1_bits, z2_bits, set; float sigma, m, tail; int pad; } blissp_t;

/* bliss_b.c:58-64 */
static int32_t barrett(int64_t x, int k, int64_t m, int64_t q) { int64_t t = (x * m) >> k, c = x - t * q; if (q <= c) c -= q; return (int32_t)c; }

/* bliss_b.c:552-620: SHA3-512(m || z as little-endian int16 || r_be16), r = 0,1,.., 9-bit indices from big-endian
 * byte pairs, first kappa distinct ones */
static int bliss_oracle(int32_t *c_idx, int kappa, const uint8_t *m, size_t mlen, const int16_t *z, int n, int mask)
{
    uint8_t *fl = calloc((size_t)n, 1), md[64];
    int cnt = 0;
    if (mlen == 0) { free(fl); return 1; }
    for (int r = 0; r < 65536; r++) {
        shake_t h; sha3_init(&h, 64);
        shake_absorb(&h, m, mlen);
        shake_absorb(&h, (const uint8_t *)z, 2 * (size_t)n);
        const uint8_t t[2] = { (uint8_t)(r >> 8), (uint8_t)(r & 0xFF) };
        shake_absorb(&h, t, 2);
        sha3_final(&h, md, 64);
        for (int i = 0; i < 64; i += 2) {
            const int idx = ((((int)md[i]) << 8) + md[i + 1]) & mask;
            if (!fl[idx]) {
                c_idx[cnt++] = idx;
                if (cnt == kappa) { free(fl); return 0; }
                fl[idx] = 1;
            }
        }
    }
    free(fl);
    return 1;
}

/* bliss_b.c:622-679 */
static void greedy_sc(const int16_t *f, const int16_t *g, int n, const int32_t *c_idx, int kappa, int32_t *x, int32_t *y)
{
    memset(x, 0, sizeof(int32_t) * (size_t)n); memset(y, 0, sizeof(int32_t) * (size_t)n);
    for (int k = 0; k < kappa; k++) {
        const int i = c_idx[k];
        int32_t sgn = 0;
        for (int j = 0; j < n - i; j++) sgn += f[j] * x[i + j] + g[j] * y[i + j];
        for (int j = n - i; j < n; j++) sgn -= f[j] * x[i + j - n] + g[j] * y[i + j - n];
        if (sgn > 0) {
            for (int j = 0; j < n - i; j++) { x[i + j] -= f[j]; y[i + j] -= g[j]; }
            for (int j = n - i; j < n; j++) { x[i + j - n] += f[j]; y[i + j - n] += g[j]; }
        } else {
            for (int j = 0; j < n - i; j++) { x[i + j] += f[j]; y[i + j] += g[j]; }
            for (int j = n - i; j < n; j++) { x[i + j - n] -= f[j]; y[i + j - n] -= g[j]; }
        }
    }
}

/* bliss_b.c:1010-1043 */
static int check_norms(const blissp_t *P, const int32_t *t, const int16_t *z)
{
    int32_t max = 0, mz = 0, nt = 0, nz = 0;
    for (int i = 0; i < P->n; i++) { if (t[i] > max) max = t[i]; if (-t[i] > max) max = -t[i]; }
    if (max > P->b_inf) return 1;
    for (int i = 0; i < P->n; i++) { if (z[i] > mz) mz = z[i]; if (-z[i] > mz) mz = -z[i]; }
    if ((mz << P->d) > P->b_inf) return 1;
    for (int i = 0; i < P->n; i++) { nt += t[i] * t[i]; nz += z[i] * z[i]; }
    if (nt + (nz << (2 * P->d)) > P->b_l2) return 1;
    return 0;
}

typedef struct { uint8_t *buf; size_t bitpos; } bw_t;
static void bw_put(bw_t *w, uint32_t v, int bits)
{
    for (int b = bits - 1; b >= 0; b--) {
        if ((w->bitpos & 7) == 0) w->buf[w->bitpos >> 3] = 0;
        w->buf[w->bitpos >> 3] |= (uint8_t)(((v >> b) & 1u) << (7 - (w->bitpos & 7)));
        w->bitpos++;
    }
}

/* bliss_b.c:1266-1535 (raw signature coding, 64-bit CDT sampler, no blinding / discard).  par = the 16 ints of the
 * fixture's *_params record; f, g, a = the key as the reference stores it (SINT16; a in the NTT domain);
 * tape = the CSPRNG's 32-bit word stream.  Returns 0, the packed signature, the attempts taken and the words consumed;
 * 2 = the tape ran out before a signature was accepted. */
static int bliss_sign_core(const int32_t *par, const int16_t *f, const int16_t *g, const int16_t *a, const uint8_t *msg, size_t mlen,
                           const uint32_t *tape, size_t tape_len, uint8_t *sig, size_t *siglen, int32_t *trials, size_t *consumed,
                           int hp_words, const uint64_t *hp_table, int hp_size);
int orc_bliss_sign(const int32_t *par, const int16_t *f, const int16_t *g, const int16_t *a, const uint8_t *msg, size_t mlen,
                   const uint32_t *tape, size_t tape_len, uint8_t *sig, size_t *siglen, int32_t *trials, size_t *consumed)
{ return bliss_sign_core(par, f, g, a, msg, mlen, tape, tape_len, sig, siglen, trials, consumed, 0, NULL, 0); }
/* the same with the sampler at 128- / 192-bit precision (SC_FLAG_0_SAMPLE_128BIT / _192BIT; gaussian_cdf_sample_128 / _192, gaussian_cdf.c:468-509): hp_words = 2 / 3
 * prng_64 draws a sample over the caller's table [hp_size][hp_words] - pinned by tests/golden/bliss_hp_ref.txt */
int orc_bliss_sign_hp(const int32_t *par, const int16_t *f, const int16_t *g, const int16_t *a, const uint8_t *msg, size_t mlen,
                      const uint32_t *tape, size_t tape_len, int hp_words, const uint64_t *hp_table, int hp_size,
                      uint8_t *sig, size_t *siglen, int32_t *trials, size_t *consumed)
{
    if (hp_words < 2 || hp_words > 3 || !hp_table || hp_size < 2) return 1;
    return bliss_sign_core(par, f, g, a, msg, mlen, tape, tape_len, sig, siglen, trials, consumed, hp_words, hp_table, hp_size);
}
static int bliss_sign_core(const int32_t *par, const int16_t *f, const int16_t *g, const int16_t *a, const uint8_t *msg, size_t mlen,
                           const uint32_t *tape, size_t tape_len, uint8_t *sig, size_t *siglen, int32_t *trials, size_t *consumed,
                           int hp_words, const uint64_t *hp_table, int hp_size)
{
    blissp_t P; memcpy(&P, par, sizeof P);
    const int n = P.n, q = P.q;
    int32_t *wt = malloc(sizeof(int32_t) * (size_t)n * 9), *rt = wt + n, *a32 = rt + n, *t = a32 + n, *u = t + n, *v = u + n, *x = v + n, *y = x + n, *uz = y + n;
    int16_t *z = malloc(sizeof(int16_t) * (size_t)n);
    int32_t c_idx[64];
    if (orc_make_tables(q, n, wt, rt)) return 1;
    for (int i = 0; i < n; i++) a32[i] = a[i];
    const int bits = orc_cdt_bits(P.tail, P.sigma), size = 1 << bits;
    uint64_t *cdf = malloc(sizeof(uint64_t) * (size_t)size);
    orc_cdt_create(P.tail, P.sigma, cdf, size);
    tape_t T = { tape, tape_len, 0, 0, 0, 0 };
    const double inv_sig2 = 1.0 / ((double)P.sigma * P.sigma);
    const int k2q = 30, kp = 30;
    const int64_t m2q = (1 << 30) / (2 * q), mp = (1 << 30) / P.p;     /* barrett_init on ntt_2q / ntt_p (bliss_b.c:278-280) */
    int rc = 2, iter;
    (void)uz;
    for (iter = 0; iter < 99999 && !T.underrun; iter++) {
        if (iter & 1) { int32_t *tmp = t; t = u; u = tmp; }
        else {
            if (hp_words) {
                for (int i = 0; i < 2 * n; i++) {
                    uint64_t x[3];
                    for (int j = 0; j < hp_words; j++) x[j] = t64(&T);
                    (i < n ? t : u)[i < n ? i : i - n] = orc_cdt_hp_sample(x, hp_table, hp_size, hp_words);
                }
            } else {
                for (int i = 0; i < n; i++) t[i] = orc_cdt_sample(t64(&T), cdf, size);
                for (int i = 0; i < n; i++) u[i] = orc_cdt_sample(t64(&T), cdf, size);
            }
        }
        if (T.underrun) break;
        orc_polymul_ntt(v, n, q, t, a32, wt, rt);                       /* fwd_ntt, mul_32_pointwise_16, inv_ntt: canonical */
        /* round_and_drop, bliss_b.c:681-703 */
        for (int i = n; i--;) {
            int32_t tmp = v[i];
            tmp += (tmp & 1) * q;
            tmp = barrett((int64_t)tmp + u[i], k2q, m2q, 2 * q);
            tmp += (tmp < 0) * 2 * q;
            v[i] = tmp;
            z[i] = (int16_t)barrett((tmp + (1 << (P.d - 1))) >> P.d, kp, mp, P.p);
        }
        if (bliss_oracle(c_idx, P.kappa, msg, mlen, z, n, (1 << P.n_bits) - 1)) continue;
        greedy_sc(f, g, n, c_idx, P.kappa, x, y);
        if (tvar(&T, 1)) { for (int i = 0; i < n; i++) { t[i] -= x[i]; u[i] -= y[i]; } }
        else             { for (int i = 0; i < n; i++) { t[i] += x[i]; u[i] += y[i]; } }
        int32_t num = 0, den = 0;
        for (int i = 0; i < n; i++) { num += x[i] * x[i] + y[i] * y[i]; den += t[i] * x[i] + u[i] * y[i]; }
        const double thresh = 1.0 / (P.m * exp(-0.5 * inv_sig2 * num) * cosh(inv_sig2 * den));
        const double rr = tdouble(&T);
        if (T.underrun) break;
        if (rr > thresh) continue;
        /* signature_gen, bliss_b.c:1045-1075 (the Barrett multiplier is built from q, not p - kept as is) */
        {
            const int32_t kk = 30, mm = (1 << kk) / q, q2 = 2 * q, p2 = P.p >> 1;
            for (int i = n; i--;) {
                int32_t tmp = v[i] - u[i];
                tmp += (tmp < 0) * q2;
                tmp -= (tmp >= q2) * q2;
                tmp = ((tmp + (1 << (P.d - 1))) >> P.d);
                tmp = tmp - ((tmp * mm) >> kk) * P.p;
                tmp -= (P.p <= tmp) * P.p;
                tmp = z[i] - tmp;
                tmp += (tmp <= -p2) * P.p;
                tmp -= (tmp > p2) * P.p;
                z[i] = (int16_t)tmp;
            }
        }
        if (check_norms(&P, t, z)) continue;
        bw_t bw = { sig, 0 };
        for (int i = 0; i < n; i++) bw_put(&bw, (uint32_t)t[i], P.z1_bits);
        for (int i = 0; i < n; i++) bw_put(&bw, (uint32_t)(int32_t)z[i], P.z2_bits);
        for (int i = 0; i < P.kappa; i++) bw_put(&bw, (uint32_t)c_idx[i], P.n_bits);
        *siglen = (bw.bitpos + 7) >> 3;
        rc = 0;
        iter++;
        break;
    }
    if (trials) *trials = iter;
    if (consumed) *consumed = T.pos;
    free(cdf); free(z); free(wt);
    return rc;
}

/* bliss_b.c:1583-1740.  Returns 0 (SC_FUNC_SUCCESS) for a valid signature, 1 otherwise. */
int orc_bliss_verify(const int32_t *par, const int16_t *a, const uint8_t *msg, size_t mlen, const uint8_t *sig, size_t siglen)
{
    blissp_t P; memcpy(&P, par, sizeof P);
    const int n = P.n, q = P.q;
    if (siglen * 8 < (size_t)(n * (P.z1_bits + P.z2_bits) + P.kappa * P.n_bits)) return 1;
    int32_t *wt = malloc(sizeof(int32_t) * (size_t)n * 5), *rt = wt + n, *a32 = rt + n, *t = a32 + n, *v = t + n;
    int16_t *z = malloc(sizeof(int16_t) * (size_t)n);
    int32_t c_idx[64], my_idx[64];
    int rc = 1;
    if (orc_make_tables(q, n, wt, rt)) return 1;
    for (int i = 0; i < n; i++) a32[i] = a[i];
    size_t bp = 0;
#define GET(dst, bits) do { uint32_t v_ = 0; for (int b_ = 0; b_ < (bits); b_++, bp++) v_ = (v_ << 1) | ((sig[bp >> 3] >> (7 - (bp & 7))) & 1u); (dst) = v_; } while (0)
    for (int i = 0; i < n; i++) { uint32_t r; GET(r, P.z1_bits); t[i] = (int32_t)(r << (32 - P.z1_bits)) >> (32 - P.z1_bits); }
    for (int i = 0; i < n; i++) { uint32_t r; GET(r, P.z2_bits); z[i] = (int16_t)((int32_t)(r << (32 - P.z2_bits)) >> (32 - P.z2_bits)); }
    for (int i = 0; i < P.kappa; i++) { uint32_t r; GET(r, P.n_bits); c_idx[i] = (int32_t)r; }
#undef GET
    if (check_norms(&P, t, z)) goto done;
    orc_polymul_ntt(v, n, q, t, a32, wt, rt);
    for (int i = n; i--;) v[i] += (v[i] & 1) * q;
    for (int i = 0; i < P.kappa; i++) v[c_idx[i]] = barrett((int64_t)v[c_idx[i]] + q, 30, (1 << 30) / (2 * q), 2 * q);
    for (int i = n; i--;) {
        int32_t tmp = ((v[i] + (1 << (P.d - 1))) >> P.d) + z[i];
        tmp -= (tmp >= P.p) * P.p;
        tmp += (tmp < 0) * P.p;
        z[i] = (int16_t)tmp;
    }
    if (bliss_oracle(my_idx, P.kappa, msg, mlen, z, n, (1 << P.n_bits) - 1)) goto done;
    rc = memcmp(my_idx, c_idx, sizeof(int32_t) * (size_t)P.kappa) ? 1 : 0;
done:
    free(z); free(wt);
    return rc;
}

/* ---- SC_ENTROPY_HUFFMAN_STATIC coding of the signature (bliss_b.c:1447-1461 / 1641-1657 with sc->coding_signature.type = HUFFMAN_STATIC) ----
 * entropy_poly_encode_32 / _16, signed (src/utils/entropy/entropy.c:70-112, 147-190): beta = max(bits - 7, 0) low bits of |v| raw, |v| >> beta
 * through a static prefix code (entropy_huffman.c:22-57, 155-190; huffman.c:386-403), then a sign bit unless v = 0; the kappa indices stay raw
 * n_bits fields; MSB-first bit stream, flushed to whole bytes (packer.c:299-351).  The code tables are data of the reference
 * (src/utils/entropy/huffman.h) and come in as (code, bits) per symbol: table 1 serves t (z1_bits), table 2 serves z (z2_bits).
 * Both directions are pure format conversions between the raw signature orc_bliss_sign writes and the coded one. */
static int huff_put(bw_t *w, int32_t v, int bits, const uint32_t *code, const uint16_t *len, uint32_t depth)
{
    const int beta = bits > 7 ? bits - 7 : 0;
    const uint32_t a = (uint32_t)(v < 0 ? -v : v);
    if (beta) bw_put(w, a & ((1u << beta) - 1u), beta);
    if ((a >> beta) >= depth) return 1;                                   /* encode_huffman: SC_OUT_OF_BOUNDS */
    bw_put(w, code[a >> beta], len[a >> beta]);
    if (v != 0) bw_put(w, v < 0, 1);
    return 0;
}
typedef struct { const uint8_t *buf; size_t bits, pos; int over; } br_t;
static uint32_t br_get(br_t *r, int bits)
{
    uint32_t v = 0;
    for (int b = 0; b < bits; b++, r->pos++) {
        if (r->pos >= r->bits) { r->over = 1; v <<= 1; continue; }
        v = (v << 1) | ((r->buf[r->pos >> 3] >> (7 - (r->pos & 7))) & 1u);
    }
    return v;
}
static int32_t huff_get(br_t *r, int bits, const uint32_t *code, const uint16_t *len, uint32_t depth)
{
    const int beta = bits > 7 ? bits - 7 : 0;
    uint32_t a = beta ? br_get(r, beta) : 0, acc = 0;
    int have = 0;
    for (;;) {                                                            /* decode_huffman (huffman.c:405-445): walk until a codeword matches */
        acc = (acc << 1) | br_get(r, 1); have++;
        if (r->over || have > 32) { r->over = 1; return 0; }
        uint32_t s;
        for (s = 0; s < depth; s++) if (len[s] == have && code[s] == acc) break;
        if (s < depth) { a |= s << beta; break; }
    }
    if (a != 0 && br_get(r, 1)) return -(int32_t)a;
    return (int32_t)a;
}
int orc_bliss_sig_to_huffman(const int32_t *par, const uint8_t *raw, size_t rawlen, const uint32_t *c1, const uint16_t *l1, uint32_t d1,
                             const uint32_t *c2, const uint16_t *l2, uint32_t d2, uint8_t *out, size_t *outlen)
{
    blissp_t P; memcpy(&P, par, sizeof P);
    if (rawlen * 8 < (size_t)(P.n * (P.z1_bits + P.z2_bits) + P.kappa * P.n_bits)) return 1;
    br_t r = { raw, rawlen * 8, 0, 0 };
    bw_t w = { out, 0 };
    int bad = 0;
    for (int i = 0; i < P.n; i++) { const uint32_t v = br_get(&r, P.z1_bits); bad |= huff_put(&w, (int32_t)(v << (32 - P.z1_bits)) >> (32 - P.z1_bits), P.z1_bits, c1, l1, d1); }
    for (int i = 0; i < P.n; i++) { const uint32_t v = br_get(&r, P.z2_bits); bad |= huff_put(&w, (int32_t)(v << (32 - P.z2_bits)) >> (32 - P.z2_bits), P.z2_bits, c2, l2, d2); }
    for (int i = 0; i < P.kappa; i++) bw_put(&w, br_get(&r, P.n_bits), P.n_bits);
    *outlen = (w.bitpos + 7) >> 3;
    return bad;
}
/* returns 1 when the coded stream ends early (the reference's pack_decode of the indices then fails, bliss_b.c:1646-1651) */
int orc_bliss_sig_from_huffman(const int32_t *par, const uint8_t *in, size_t inlen, const uint32_t *c1, const uint16_t *l1, uint32_t d1,
                               const uint32_t *c2, const uint16_t *l2, uint32_t d2, uint8_t *raw, size_t *rawlen)
{
    blissp_t P; memcpy(&P, par, sizeof P);
    br_t r = { in, inlen * 8, 0, 0 };
    bw_t w = { raw, 0 };
    for (int i = 0; i < P.n; i++) bw_put(&w, (uint32_t)huff_get(&r, P.z1_bits, c1, l1, d1), P.z1_bits);
    for (int i = 0; i < P.n; i++) bw_put(&w, (uint32_t)huff_get(&r, P.z2_bits, c2, l2, d2), P.z2_bits);
    for (int i = 0; i < P.kappa; i++) bw_put(&w, br_get(&r, P.n_bits), P.n_bits);
    *rawlen = (w.bitpos + 7) >> 3;
    return r.over;
}

/* The same coding for a signature that is ONE polynomial of n signed `bits`-bit fields followed by `tail` raw bytes in the same bit stream: Falcon's s2
 * (falcon.c:892-903), the ENS / DLP signatures' s1 (ens_dlp_sig.c:1275-1287), ring-TESLA's z + 64 digest bytes (ring_tesla.c:880-890). */
int orc_huff_row_encode(const uint8_t *raw, size_t rawlen, int n, int bits, int tail, const uint32_t *code, const uint16_t *len, uint32_t depth, uint8_t *out, size_t *outlen)
{
    if (rawlen * 8 < (size_t)n * (size_t)bits + 8 * (size_t)tail) return 1;
    br_t r = { raw, rawlen * 8, 0, 0 };
    bw_t w = { out, 0 };
    int bad = 0;
    for (int i = 0; i < n; i++) { const uint32_t v = br_get(&r, bits); bad |= huff_put(&w, (int32_t)(v << (32 - bits)) >> (32 - bits), bits, code, len, depth); }
    for (int i = 0; i < tail; i++) bw_put(&w, br_get(&r, 8), 8);
    *outlen = (w.bitpos + 7) >> 3;
    return bad;
}
int orc_huff_row_decode(const uint8_t *in, size_t inlen, int n, int bits, int tail, const uint32_t *code, const uint16_t *len, uint32_t depth, uint8_t *raw, size_t *rawlen)
{
    br_t r = { in, inlen * 8, 0, 0 };
    bw_t w = { raw, 0 };
    for (int i = 0; i < n; i++) bw_put(&w, (uint32_t)huff_get(&r, bits, code, len, depth), bits);
    for (int i = 0; i < tail; i++) bw_put(&w, br_get(&r, 8), 8);
    *rawlen = (w.bitpos + 7) >> 3;
    return r.over;
}

/* ---- the binary arithmetic coder (SC_ENTROPY_BAC): src/utils/entropy/bac.c ------------------------------------------------------------------
 * A coefficient is `bits` binary decisions, most significant first; the probability of a 0 at a node of that binary tree is dist[prefix | 1 << k]
 * (64-bit fixed point), the coder a 64-bit range coder with bit-wise renormalisation and byte output with carry propagation.  A coded polynomial is a
 * byte-aligned 2-byte big-endian length followed by that many bytes. */
/* gauss_freq_bac_64, bac.c:83-122: P(lower half | node) of exp(-x^2 / 2 sig^2) over x in [-n/2, n/2), long double arithmetic (host dependent in its
 * last bits like the CDT tables: compare against the reference on the same host, or take the table as data) */
void orc_bac_dist(uint64_t *dist, float sig, size_t n)
{
    const long double s2 = (long double)(-0.5 / (sig * sig));           /* as written there with FLOAT sig: float * float, -0.5 (a double) / float -> double, kept in a LONGDOUBLE */
    for (size_t i = 0; i < n; i++) dist[i] = 0;
    for (size_t i = n >> 1; i >= 1; i >>= 1) {
        for (size_t j = 0; j < n; j += i + i) {
            long double a = 0.0, b = 0.0;
            for (size_t k = 0; k < i; k++) {
                int32_t x = (int32_t)(j + k) - (int32_t)(n >> 1);
                a += expl(s2 * (long double)(x * x));
                x = (int32_t)(i + j + k) - (int32_t)(n >> 1);
                b += expl(s2 * (long double)(x * x));
            }
            a = a / (a + b);
            uint64_t r = (uint64_t)(0x1.p64L * a);
            if (r < 4) r = (a > 0.5) ? (uint64_t)-4 : 4;
            dist[j + i] = r;
        }
    }
}
static uint64_t bac_mul64hi(uint64_t x, uint64_t y) { return (uint64_t)(((unsigned __int128)x * (unsigned __int128)y) >> 64); }
/* bac_encode_64_32 / _16, bac.c:162-258 (the two differ in the input type only).  out: a byte-aligned position; returns the bytes written, header included. */
size_t orc_bac_encode(uint8_t *out, const int32_t *in, size_t inlen, const uint64_t *dist, int bits, int32_t offset)
{
    uint8_t *buffer = out + 2;
    uint64_t b = 0, l = ~0ull;
    uint32_t data = 0;
    int ocnt = 0;
    size_t optr = 0;
#define BAC_BYTE(val) do { uint32_t ob_ = (val); buffer[optr] = (uint8_t)ob_; for (ptrdiff_t i_ = (ptrdiff_t)optr - 1; ob_ >= 0x100 && i_ >= 0; i_--) { ob_ >>= 8; ob_ += buffer[i_]; buffer[i_] = (uint8_t)ob_; } optr++; } while (0)
    for (size_t ip = 0; ip < inlen; ip++) {
        const uint32_t iwrd = (uint32_t)(offset + in[ip]);
        for (int icnt = bits; icnt--;) {
            uint64_t c = dist[(iwrd & (0xFFFFFFFEu << icnt)) | (1u << icnt)];
            c = bac_mul64hi(l, c);
            if (0 == ((iwrd >> icnt) & 1)) l = c;
            else { b += c; l -= c; if (b < c) data++; }
            if (l) while (l < 0x8000000000000000ull) {
                data <<= 1; data |= (uint32_t)(b >> 63) & 1; ocnt++;
                if (ocnt >= 8) { BAC_BYTE(data); ocnt = 0; data = 0; }
                b <<= 1; l <<= 1;
            }
        }
    }
    while (ocnt < 8) { data = (data << 1) ^ (uint32_t)(b >> 63); b <<= 1; ocnt++; }
    BAC_BYTE(data);
    while (b != 0) { buffer[optr++] = (uint8_t)(b >> 56); b <<= 8; }
#undef BAC_BYTE
    out[0] = (uint8_t)(optr >> 8); out[1] = (uint8_t)optr;
    return optr + 2;
}
/* bac_decode_64_32 / _16, bac.c:260-334.  in: a byte-aligned position with `avail` bytes behind it (a read past them yields zero bits, as the packer's
 * failed read does, and sets *over); *used: the bytes consumed.  The shift "(ibyt >> icnt)" runs with icnt < 0 once the stated length is exhausted -
 * undefined in C, a shift by icnt mod 32 on the x86 the reference runs on: restated as such. */
void orc_bac_decode(const uint8_t *in, size_t avail, size_t *used, int *over, int32_t *out, size_t outlen, const uint64_t *dist, int bits, int32_t offset)
{
    size_t rp = 0;
#define BAC_RD() (rp < avail ? in[rp++] : (*over = 1, rp++, (uint8_t)0))
    uint32_t length = (uint32_t)BAC_RD() << 8; length |= BAC_RD();
    uint64_t v = 0;
    for (int i = 0; i < 8; i++) v = (v << 8) | BAC_RD();
    uint64_t b = 0, l = ~0ull;
    uint32_t ibyt = 0; int icnt = 0; int32_t iptr = 8;
    for (size_t op = 0; op < outlen; op++) {
        uint32_t owrd = 0;
        for (int ocnt = bits; ocnt--;) {
            uint64_t c = dist[(owrd & (0xFFFFFFFEu << ocnt)) | (1u << ocnt)];
            c = bac_mul64hi(l, c);
            if (v - b < c) l = c;
            else { b += c; l -= c; owrd |= 1u << ocnt; }
            while (l < 0x8000000000000000ull) {
                icnt--;
                if (icnt < 0 && iptr < (int32_t)length) { ibyt = BAC_RD(); iptr++; icnt = 7; }
                v <<= 1; v += (ibyt >> (icnt & 31)) & 1;
                b <<= 1; l <<= 1;
            }
        }
        out[op] = (int32_t)(owrd - (uint32_t)offset);
    }
#undef BAC_RD
    *used = rp;
}
/* a raw BLISS-B signature (t: n fields of z1_bits, z: n of z2_bits, kappa indices of n_bits, MSB first) <-> the same under SC_FLAG_0_ENTROPY_BAC
 * (bliss_b.c:1453-1463, 1650-1657: t through distribution 2, z through distribution 3, the indices raw behind them) */
int orc_bliss_sig_to_bac(const int32_t *par, const uint8_t *raw, size_t rawlen, const uint64_t *dist_t, const uint64_t *dist_z, uint8_t *out, size_t *outlen)
{
    blissp_t P; memcpy(&P, par, sizeof P);
    if (rawlen * 8 < (size_t)(P.n * (P.z1_bits + P.z2_bits) + P.kappa * P.n_bits)) return 1;
    br_t r = { raw, rawlen * 8, 0, 0 };
    int32_t v[1024];
    for (int i = 0; i < P.n; i++) { const uint32_t f = br_get(&r, P.z1_bits); v[i] = (int32_t)(f << (32 - P.z1_bits)) >> (32 - P.z1_bits); }
    size_t at = orc_bac_encode(out, v, (size_t)P.n, dist_t, P.z1_bits, 1 << (P.z1_bits - 1));
    for (int i = 0; i < P.n; i++) { const uint32_t f = br_get(&r, P.z2_bits); v[i] = (int32_t)(f << (32 - P.z2_bits)) >> (32 - P.z2_bits); }
    at += orc_bac_encode(out + at, v, (size_t)P.n, dist_z, P.z2_bits, 1 << (P.z2_bits - 1));
    bw_t w = { out + at, 0 };
    const size_t tailbytes = ((size_t)P.kappa * P.n_bits + 7) / 8;
    memset(out + at, 0, tailbytes);
    for (int i = 0; i < P.kappa; i++) bw_put(&w, br_get(&r, P.n_bits), P.n_bits);
    *outlen = at + tailbytes;
    return 0;
}
/* returns 1 when the stream ends early */
int orc_bliss_sig_from_bac(const int32_t *par, const uint8_t *in, size_t inlen, const uint64_t *dist_t, const uint64_t *dist_z, uint8_t *raw, size_t *rawlen)
{
    blissp_t P; memcpy(&P, par, sizeof P);
    int32_t v[1024];
    int over = 0; size_t used = 0, at = 0;
    const size_t rawbytes = ((size_t)P.n * (P.z1_bits + P.z2_bits) + (size_t)P.kappa * P.n_bits + 7) / 8;
    memset(raw, 0, rawbytes);
    bw_t w = { raw, 0 };
    orc_bac_decode(in, inlen, &used, &over, v, (size_t)P.n, dist_t, P.z1_bits, 1 << (P.z1_bits - 1));
    for (int i = 0; i < P.n; i++) bw_put(&w, (uint32_t)v[i], P.z1_bits);
    at = used;
    orc_bac_decode(in + (at < inlen ? at : inlen), at < inlen ? inlen - at : 0, &used, &over, v, (size_t)P.n, dist_z, P.z2_bits, 1 << (P.z2_bits - 1));
    for (int i = 0; i < P.n; i++) bw_put(&w, (uint32_t)v[i], P.z2_bits);
    at += used;
    br_t r = { in + (at < inlen ? at : inlen), (at < inlen ? inlen - at : 0) * 8, 0, 0 };
    for (int i = 0; i < P.kappa; i++) bw_put(&w, br_get(&r, P.n_bits), P.n_bits);
    *rawlen = rawbytes;
    return over | r.over;
}

/* poly_32_uniform_rand (src/utils/arith/poly_32.c:132-156) on the tape: c[j] coefficients of value +-(c_len - j), one prng_32 word
 * per placement attempt (bit 0: sign, the bits above: position), occupied positions are drawn again.  Returns 1 on underrun. */
static int uniform_rand_tape(tape_t *T, int32_t *v, int n, const int *c, int c_len)
{
    for (int i = 0; i < n; i++) v[i] = 0;
    for (int j = 0; j < c_len; j++) {
        int i = 0;
        while (i < c[j]) {
            const uint32_t r = t32(T);
            if (T->underrun) return 1;
            const uint32_t idx = (r >> 1) & (uint32_t)(n - 1);
            if (v[idx] == 0) { v[idx] = (r & 1) ? j - c_len : c_len - j; i++; }
        }
    }
    return 0;
}

/* bliss_b_keygen, bliss_b.c:1124-1264 (no sparse multiplier, single thread): g <- uniform, g = 2g + 1 (at coefficient 0), t = NTT(g);
 * f <- uniform until NTT(f) has no zero coefficient; a = normalise(t o (-NTT(f)^-1)).  par as for signing (n = par[0], q = par[1]);
 * nz0, nz1 = bliss_set_t::nz.  Outputs as the reference stores them (SINT16; a in the NTT domain).  Returns 0, the words consumed and the
 * number of f drawn; 2 = tape exhausted. */
int orc_bliss_keygen(const int32_t *par, int nz0, int nz1, const uint32_t *tape, size_t tape_len, int16_t *f16, int16_t *g16, int16_t *a16,
                     size_t *consumed, int32_t *attempts)
{
    const int n = par[0], q = par[1];
    const int c[2] = { nz0, nz1 };
    int32_t *w = malloc(sizeof(int32_t) * (size_t)n * 6), *r = w + n, *g = r + n, *f = g + n, *t = f + n, *u = t + n;
    tape_t T = { tape, tape_len, 0, 0, 0, 0 };
    int rc = 2;
    *attempts = 0;
    if (orc_make_tables(q, n, w, r)) { free(w); return 1; }
    if (uniform_rand_tape(&T, g, n, c, 2)) goto done;
    for (int i = 0; i < n; i++) g[i] *= 2;
    g[0] += 1;
    for (int i = 0; i < n; i++) g16[i] = (int16_t)g[i];                     /* the key stores g after the in-place 2g + 1 (:1175-1176, 1232-1235) */
    orc_fwd_ntt(t, n, q, g, w); orc_normalize(t, (size_t)n, q);
    for (;;) {
        if (uniform_rand_tape(&T, f, n, c, 2)) goto done;
        (*attempts)++;
        orc_fwd_ntt(u, n, q, f, w);
        if (orc_invert(u, (size_t)n, q) == 0) break;
    }
    for (int i = 0; i < n; i++) u[i] = -u[i];
    orc_mul_pointwise(u, n, q, t, u);
    orc_normalize(u, (size_t)n, q);
    for (int i = 0; i < n; i++) { f16[i] = (int16_t)f[i]; a16[i] = (int16_t)u[i]; }
    rc = 0;
done:
    *consumed = T.pos;
    free(w);
    return rc;
}
