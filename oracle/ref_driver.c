/*
 * ref_driver.c - golden-vector generator.  TEST INFRASTRUCTURE ONLY.
 *
 * Our own source, linked against the reference library built by build_ref.sh
 * (oracle/_ref/libsafecrypto_ref.so, compiled from /root/reference where it lies).
 * It calls the reference through its public function-pointer tables exactly as the
 * reference's schemes do (utils_arith_ntt() src/utils/arith/arith.c:360,
 * create_sampler()/get_vector_32() src/utils/sampling/sampling.c:425,556) and dumps
 * seeded inputs + the reference's outputs as small text fixtures for tests/golden/.
 *
 * Usage: ref_driver <ntt|cdt|cdt32|dilithium|bliss|bliss_huff|bliss_bac|kyber|falcon|ops|drbg|tesla|rlwe|ens|dlp|ensdlp|ensdlp_rec|mlwe_xof|cdt_hp|bliss_hp|all> <outdir>
 *        ref_driver bench <q> <n> <seconds>      (single-thread CPU baseline of the reference)
 *        ref_driver bench_kyber <set> <seconds>  (single-thread safecrypto_encapsulation / safecrypto_decapsulation)
 *        ref_driver bench_ens <set> <seconds>    (the same for ENS-KEM)
 *        ref_driver bench_dlp <set> <seconds>    (single-thread safecrypto_ibe_extract, DLP-IBE)
 *        ref_driver bench_rlwe <set> <seconds>   (single-thread safecrypto_public_encrypt / safecrypto_private_decrypt, RLWE-Enc)
 *        ref_driver bench_dil|bench_falcon|bench_bliss <set> <seconds>
 *        ref_driver falcon_keys <set> <nkeys> <outfile>
 *                                                (single-thread safecrypto_sign / safecrypto_verify, 64-byte messages)
 *
 * Fixture format (one file per topic): repeated records
 *     @ <name> <count>\n  followed by <count> whitespace separated decimal integers.
 */
/* table gates the scheme *_params.h headers set for the library build (e.g. ring_tesla_params.h:20-21) */
#define NTT_NEEDS_7681
#define NTT_NEEDS_8380417
#define NTT_NEEDS_12289
#define NTT_NEEDS_8399873
#define NTT_NEEDS_51750913
#define NTT_NEEDS_4206593
#define NTT_NEEDS_16813057
#define NTT_NEEDS_134348801
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <stdint.h>

#include "safecrypto.h"
#include "safecrypto_private.h"
#include "utils/arith/arith.h"
#include "utils/arith/ntt.h"
#include "utils/arith/roots_of_unity.h"
#include "utils/arith/ntt_tables.h"
#include "utils/sampling/sampling.h"
#include "utils/crypto/prng.h"

/* SC_REF_SALT=<n> in the environment (0 / unset: the committed fixtures): every seeded stream of the driver - the messages' and inputs' generator, the entropy the
 * contexts draw their keys from, the per-case CSPRNG seeds - is perturbed by it, so that `ref_driver all <dir>` writes an independently seeded set of fixtures of the
 * same shape: other keys, other messages, other tapes (tests/test_oracle_fresh_fixtures.py holds the oracle to such a set as well) */
static uint64_t g_salt;
static uint64_t sm_state;
static uint64_t splitmix64(void)
{
    uint64_t z = (sm_state += 0x9E3779B97F4A7C15ULL) + g_salt * 0xD1342543DE82EF95ULL;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
}

static void rec_i32(FILE *f, const char *name, const int32_t *v, size_t n)
{
    fprintf(f, "@ %s %zu\n", name, n);
    for (size_t i = 0; i < n; i++) fprintf(f, "%d%c", v[i], ((i & 15) == 15 || i + 1 == n) ? '\n' : ' ');
}
static void rec_i16(FILE *f, const char *name, const int16_t *v, size_t n)
{
    fprintf(f, "@ %s %zu\n", name, n);
    for (size_t i = 0; i < n; i++) fprintf(f, "%d%c", v[i], ((i & 15) == 15 || i + 1 == n) ? '\n' : ' ');
}
static void rec_u64(FILE *f, const char *name, const uint64_t *v, size_t n)
{
    fprintf(f, "@ %s %zu\n", name, n);
    for (size_t i = 0; i < n; i++) fprintf(f, "%llu%c", (unsigned long long)v[i], ((i & 7) == 7 || i + 1 == n) ? '\n' : ' ');
}

/* ------------------------------------------------------------------ NTT ---- */

typedef struct {
    int32_t q; int n; int is16;
    const void *w, *r;
} ntt_cfg_t;

static void ntt_case(FILE *f, const ntt_cfg_t *c, safecrypto_ntt_e backend, const char *bname, int emit_inputs)
{
    const utils_arith_ntt_t *ntt = utils_arith_ntt(backend);
    ntt_params_t p;
    char name[128];
    int n = c->n, q = c->q;
    const int NP = 4;             /* polynomials per case */
    int32_t *a = malloc(sizeof(int32_t) * n * NP), *b = malloc(sizeof(int32_t) * n * NP);
    int32_t *fa = malloc(sizeof(int32_t) * n), *fb = malloc(sizeof(int32_t) * n);
    int32_t *out = malloc(sizeof(int32_t) * n * NP);

    init_reduce(&p, n, q);

    /* inputs: poly 0 = ramp a[i]=i (SURVEY sanity vector), 1 = small signed {-12..12}
     * (mirrors small_rand_dist_32, src/unit/unit_ntt.c:501-524), 2,3 = uniform [0,q) */
    sm_state = 0x5afec0de00000000ULL + (uint64_t)q * 4096 + n;
    for (int i = 0; i < n; i++) { a[i] = i; b[i] = (n - 1 - i); }
    for (int i = 0; i < n; i++) { a[n + i] = (int)(splitmix64() % 25) - 12; b[n + i] = (int)(splitmix64() % 25) - 12; }
    for (int k = 2; k < NP; k++)
        for (int i = 0; i < n; i++) { a[k * n + i] = (int)(splitmix64() % (uint64_t)q); b[k * n + i] = (int)(splitmix64() % (uint64_t)q); }
    if (emit_inputs) {
        snprintf(name, sizeof name, "q%d_n%d_a", q, n); rec_i32(f, name, a, (size_t)n * NP);
        snprintf(name, sizeof name, "q%d_n%d_b", q, n); rec_i32(f, name, b, (size_t)n * NP);
    }

    /* forward NTT then normalize_32 -> canonical representative (SURVEY 8(a) contract) */
    for (int k = 0; k < NP; k++) {
        if (c->is16) ntt->fwd_ntt_32_16(out + k * n, &p, a + k * n, (const SINT16 *)c->w);
        else         ntt->fwd_ntt_32_32(out + k * n, &p, a + k * n, (const SINT32 *)c->w);
        ntt->normalize_32(out + k * n, n, &p);
    }
    snprintf(name, sizeof name, "q%d_n%d_%s_fwd", q, n, bname); rec_i32(f, name, out, (size_t)n * NP);

    /* inverse of the canonical forward output: must return a mod q (canonical) */
    for (int k = 0; k < NP; k++) {
        int32_t *t = out + k * n;
        if (c->is16) ntt->inv_ntt_32_16(t, &p, t, (const SINT16 *)c->w, (const SINT16 *)c->r);
        else         ntt->inv_ntt_32_32(t, &p, t, (const SINT32 *)c->w, (const SINT32 *)c->r);
    }
    snprintf(name, sizeof name, "q%d_n%d_%s_inv", q, n, bname); rec_i32(f, name, out, (size_t)n * NP);

    /* negacyclic product a*b through fwd, fwd, pointwise, inv exactly as the schemes do */
    for (int k = 0; k < NP; k++) {
        if (c->is16) { ntt->fwd_ntt_32_16(fa, &p, a + k * n, (const SINT16 *)c->w); ntt->fwd_ntt_32_16(fb, &p, b + k * n, (const SINT16 *)c->w); }
        else         { ntt->fwd_ntt_32_32(fa, &p, a + k * n, (const SINT32 *)c->w); ntt->fwd_ntt_32_32(fb, &p, b + k * n, (const SINT32 *)c->w); }
        ntt->mul_32_pointwise(fa, &p, fa, fb);
        if (c->is16) ntt->inv_ntt_32_16(out + k * n, &p, fa, (const SINT16 *)c->w, (const SINT16 *)c->r);
        else         ntt->inv_ntt_32_32(out + k * n, &p, fa, (const SINT32 *)c->w, (const SINT32 *)c->r);
    }
    snprintf(name, sizeof name, "q%d_n%d_%s_mul", q, n, bname); rec_i32(f, name, out, (size_t)n * NP);

    /* center_32 of the uniform inputs */
    memcpy(out, a, sizeof(int32_t) * n * NP);
    ntt->center_32(out, (size_t)n * NP, &p);
    snprintf(name, sizeof name, "q%d_n%d_%s_center", q, n, bname); rec_i32(f, name, out, (size_t)n * NP);

    free(a); free(b); free(fa); free(fb); free(out);
}

static int gen_ntt(const char *dir)
{
    char path[512];
    snprintf(path, sizeof path, "%s/ntt_ref.txt", dir);
    FILE *f = fopen(path, "w");
    if (!f) return 1;
    const ntt_cfg_t cfgs[] = {
        { 7681,    256,  1, w7681_n256,     r7681_n256 },
        { 8380417, 256,  0, w8380417_n256,  r8380417_n256 },
        { 12289,   512,  1, w12289_n512,    r12289_n512 },
        { 12289,   1024, 1, w12289_n1024,   r12289_n1024 },
        { 8399873, 512,  0, w8399873_n512,  r8399873_n512 },
    };
    for (size_t i = 0; i < sizeof cfgs / sizeof cfgs[0]; i++) {
        const ntt_cfg_t *c = &cfgs[i];
        char name[64];
        /* the reference's generated twiddle tables (build_tools/ntt_table_gen/main.c:205-297) */
        snprintf(name, sizeof name, "q%d_n%d_w", c->q, c->n);
        if (c->is16) rec_i16(f, name, (const int16_t *)c->w, c->n); else rec_i32(f, name, (const int32_t *)c->w, c->n);
        snprintf(name, sizeof name, "q%d_n%d_r", c->q, c->n);
        if (c->is16) rec_i16(f, name, (const int16_t *)c->r, c->n); else rec_i32(f, name, (const int32_t *)c->r, c->n);
        ntt_case(f, c, SC_NTT_FLOATING_POINT, "fp", 1);
        ntt_case(f, c, SC_NTT_REFERENCE, "ref", 0);
        if (c->q == 7681)    ntt_case(f, c, SC_NTT_SOLINAS_7681, "sol", 0);
        if (c->q == 8380417) ntt_case(f, c, SC_NTT_SOLINAS_8380417, "sol", 0);
    }
    /* scalar known answers, src/unit/unit_ntt.c:166-320 */
    {
        const utils_arith_ntt_t *ntt = utils_arith_ntt(SC_NTT_FLOATING_POINT);
        ntt_params_t p; init_reduce(&p, 1024, 12289);
        int32_t v[4];
        v[0] = ntt->muln_32(1234, 5678, &p);
        v[1] = ntt->muln_32(-1234, 5678, &p);
        v[2] = ntt->muln_32(12288, 12288, &p);
        v[3] = ntt->modn_32(12289 * 12289 - 1, &p);
        rec_i32(f, "scalar_q12289_fp", v, 4);
    }
    fclose(f);
    return 0;
}

/* the `_large` transform variants (modn after every addition, ntt_template.c.in:1246-1339,1484-1613) that ENS / DLP / ring-TESLA-1
 * use with the 26-bit modulus 51750913 (the only large-modulus tables the library build contains): forward (normalised), inverse and product vectors for two polynomials per (q, n) */
static int gen_ntt_large(const char *dir)
{
    char path[512], name[96];
    snprintf(path, sizeof path, "%s/ntt_large_ref.txt", dir);
    FILE *f = fopen(path, "w");
    if (!f) return 1;
    /* every modulus the `_large` transforms are used with: ring-TESLA-1 / DLP-IBE 2, 5 (26 bits), DLP-IBE 0, 1 (23 bits), 6 (25 bits), 7 (28 bits) */
    const struct { int32_t q; int n; const SINT32 *w, *r; } cf[6] = {
        { 51750913, 512, w51750913_n512, r51750913_n512 }, { 51750913, 1024, w51750913_n1024, r51750913_n1024 },
        { 4206593, 512, w4206593_n512, r4206593_n512 }, { 4206593, 1024, w4206593_n1024, r4206593_n1024 },
        { 16813057, 512, w16813057_n512, r16813057_n512 }, { 134348801, 1024, w134348801_n1024, r134348801_n1024 } };
    const utils_arith_ntt_t *ntt = utils_arith_ntt(SC_NTT_FLOATING_POINT);
    for (int c = 0; c < 6; c++) {
        const int q = cf[c].q, n = cf[c].n, NP = 2;
        ntt_params_t p; init_reduce(&p, n, q);
        int32_t *a = malloc(sizeof(int32_t) * n * NP), *b = malloc(sizeof(int32_t) * n * NP), *o = malloc(sizeof(int32_t) * n * NP), *fa = malloc(sizeof(int32_t) * n), *fb = malloc(sizeof(int32_t) * n);
        sm_state = 0x1a46e000ULL + (uint64_t)q + n;
        for (int i = 0; i < n; i++) { a[i] = (int)(splitmix64() % (uint64_t)q); b[i] = (int)(splitmix64() % (uint64_t)q); }
        for (int i = 0; i < n; i++) { a[n + i] = (int)(splitmix64() % 2001) - 1000; b[n + i] = (int)(splitmix64() % 25) - 12; }
        snprintf(name, sizeof name, "q%d_n%d_w1", q, n); rec_i32(f, name, cf[c].w + 1, 1);             /* the root the tables were built from */
        snprintf(name, sizeof name, "q%d_n%d_a", q, n); rec_i32(f, name, a, (size_t)n * NP);
        snprintf(name, sizeof name, "q%d_n%d_b", q, n); rec_i32(f, name, b, (size_t)n * NP);
        for (int k = 0; k < NP; k++) { ntt->fwd_ntt_32_32_large(o + k * n, &p, a + k * n, cf[c].w); ntt->normalize_32(o + k * n, n, &p); }
        snprintf(name, sizeof name, "q%d_n%d_fwd", q, n); rec_i32(f, name, o, (size_t)n * NP);
        for (int k = 0; k < NP; k++) { ntt->inv_ntt_32_32_large(o + k * n, &p, o + k * n, cf[c].w, cf[c].r); ntt->normalize_32(o + k * n, n, &p); }
        snprintf(name, sizeof name, "q%d_n%d_inv", q, n); rec_i32(f, name, o, (size_t)n * NP);
        for (int k = 0; k < NP; k++) {
            ntt->fwd_ntt_32_32_large(fa, &p, a + k * n, cf[c].w); ntt->fwd_ntt_32_32_large(fb, &p, b + k * n, cf[c].w);
            ntt->mul_32_pointwise(fa, &p, fa, fb);
            ntt->inv_ntt_32_32_large(o + k * n, &p, fa, cf[c].w, cf[c].r); ntt->normalize_32(o + k * n, n, &p);
        }
        snprintf(name, sizeof name, "q%d_n%d_mul", q, n); rec_i32(f, name, o, (size_t)n * NP);
        free(a); free(b); free(o); free(fa); free(fb);
    }
    fclose(f);
    return 0;
}
/* ------------------------------------------------- reproducible contexts ----
 * Every scheme context of a fixture generator draws its entropy from a callback (safecrypto_entropy_callback + SC_FLAG_1_CSPRNG_USE_CALLBACK_RANDOM,
 * include/safecrypto.h:108,281; the flag is read at safecrypto.c:440-442 and the CSPRNG instances then seed and reseed through
 * get_entropy_callback, prng_get_func.c:51-55) that returns a fixed stream per (generator, context number): `ref_driver all <dir>` writes the same
 * bytes every time, keys included.  The stream is splitmix64 on a state of its own (the messages' generator sm_state is not touched). */
static uint64_t ent_state, ent_domain;
static unsigned ent_serial;
static void fixed_entropy(size_t n, UINT8 *data)
{
    for (size_t i = 0; i < n; i++) {
        uint64_t z = (ent_state += 0x9E3779B97F4A7C15ULL);
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
        data[i] = (UINT8)((z ^ (z >> 31)) >> 24);
    }
}
static void fixed_domain(const char *name)
{
    uint64_t h = 0xcbf29ce484222325ULL;                     /* FNV-1a of the generator's name */
    for (const char *c = name; *c; c++) h = (h ^ (uint8_t)*c) * 0x100000001b3ULL;
    ent_domain = h; ent_serial = 0;
    safecrypto_entropy_callback(fixed_entropy);
}
static safecrypto_t *create_fixed(sc_scheme_e scheme, SINT32 set, const UINT32 *flags)
{
    UINT32 fl[2] = { flags[0] | SC_FLAG_MORE, ((flags[0] & SC_FLAG_MORE) ? flags[1] : 0) | SC_FLAG_1_CSPRNG_USE_CALLBACK_RANDOM };
    ent_state = ent_domain ^ ((uint64_t)(++ent_serial) * 0xD6E8FEB86659FD93ULL) ^ (g_salt * 0xA0761D6478BD642FULL);
    return safecrypto_create(scheme, set, fl);
}

/* ------------------------------------------------------------------ CDT ---- */

typedef struct __attribute__((packed)) { UINT64 *cdf; SINT32 cdf_size; } cdf64_view_t; /* src/utils/sampling/gaussian_cdf.c:74-81 */

static prng_ctx_t *seeded_prng(const uint8_t *seed, size_t len)
{
    /* the seeding pattern of src/unit/unit_module_lwe.c:44-47 */
    prng_ctx_t *ctx = prng_create(SC_ENTROPY_USER_PROVIDED, SC_PRNG_AES_CTR_DRBG, SC_PRNG_THREADING_NONE, 0x00100000);
    prng_set_entropy(ctx, seed, len);
    prng_init(ctx, NULL, 0);
    return ctx;
}
/* for the cases whose seed is a constant of this file (not recorded in the fixture): the salt goes into a copy */
static prng_ctx_t *const_seeded_prng(const uint8_t *seed, size_t len)
{
    uint8_t salted[64];
    if (!g_salt || len > sizeof salted) return seeded_prng(seed, len);
    memcpy(salted, seed, len);
    for (size_t i = 0; i < len && i < 8; i++) salted[i] ^= (uint8_t)((g_salt * 0x9E3779B97F4A7C15ULL) >> (8 * i));
    return seeded_prng(salted, len);
}

static int cdt_case(FILE *f, float tail, float sigma, int nsamp, const char *tag)
{
    static const uint8_t seed[32] = { 3, 1, 4, 1, 5, 9, 2, 6, 5, 3, 5, 8, 9, 7, 9, 3, 2, 3, 8, 4, 6, 2, 6, 4, 3, 3, 8, 3, 2, 7, 9, 5 };
    char name[96];
    prng_ctx_t *p1 = const_seeded_prng(seed, 32), *p2 = const_seeded_prng(seed, 32);
    utils_sampling_t *s = create_sampler(CDF_GAUSSIAN_SAMPLING, SAMPLING_64BIT, NORMAL_SAMPLES, 512,
                                         SAMPLING_DISABLE_BOOTSTRAP, p1, tail, sigma);
    if (!s) return 1;
    const cdf64_view_t *g = (const cdf64_view_t *)s->gauss;
    snprintf(name, sizeof name, "%s_cdf", tag);
    rec_u64(f, name, (const uint64_t *)g->cdf, (size_t)g->cdf_size);

    /* the twin PRNG (same seed) yields the uniform words the sampler is about to consume:
     * gaussian_cdf_sample_64 takes exactly one prng_64 per sample (gaussian_cdf.c:639-659) */
    uint64_t *x = malloc(sizeof(uint64_t) * nsamp);
    int32_t *v = malloc(sizeof(int32_t) * nsamp);
    for (int i = 0; i < nsamp; i++) x[i] = prng_64(p2);
    get_vector_32(s, v, nsamp, 0);
    snprintf(name, sizeof name, "%s_uniform", tag); rec_u64(f, name, x, nsamp);
    snprintf(name, sizeof name, "%s_samples", tag); rec_i32(f, name, v, nsamp);
    free(x); free(v);
    destroy_sampler(&s);
    prng_destroy(p1); prng_destroy(p2);
    return 0;
}

/* the same at 32-bit precision (SAMPLING_32BIT: gaussian_cdf_create_32 / gaussian_cdf_sample_32, one prng_32 per sample) */
typedef struct __attribute__((packed)) { UINT32 *cdf; SINT32 cdf_size; } cdf32_view_t;  /* src/utils/sampling/gaussian_cdf.c:84-92 */
static int cdt32_case(FILE *f, float tail, float sigma, int nsamp, const char *tag)
{
    static const uint8_t seed[32] = { 2, 7, 1, 8, 2, 8, 1, 8, 2, 8, 4, 5, 9, 0, 4, 5, 2, 3, 5, 3, 6, 0, 2, 8, 7, 4, 7, 1, 3, 5, 2, 6 };
    char name[96];
    prng_ctx_t *p1 = const_seeded_prng(seed, 32), *p2 = const_seeded_prng(seed, 32);
    utils_sampling_t *s = create_sampler(CDF_GAUSSIAN_SAMPLING, SAMPLING_32BIT, NORMAL_SAMPLES, 512, SAMPLING_DISABLE_BOOTSTRAP, p1, tail, sigma);
    if (!s) return 1;
    const cdf32_view_t *g = (const cdf32_view_t *)s->gauss;
    snprintf(name, sizeof name, "%s_tab32", tag); rec_i32(f, name, (const int32_t *)g->cdf, (size_t)g->cdf_size);
    int32_t *x = malloc(sizeof(int32_t) * nsamp), *v = malloc(sizeof(int32_t) * nsamp);
    for (int i = 0; i < nsamp; i++) x[i] = (int32_t)prng_32(p2);
    get_vector_32(s, v, nsamp, 0);
    snprintf(name, sizeof name, "%s_words32", tag); rec_i32(f, name, x, nsamp);
    snprintf(name, sizeof name, "%s_samples", tag); rec_i32(f, name, v, nsamp);
    free(x); free(v);
    destroy_sampler(&s);
    prng_destroy(p1); prng_destroy(p2);
    return 0;
}
static int gen_cdt32(const char *dir)
{
    char path[512];
    snprintf(path, sizeof path, "%s/cdt32_ref.txt", dir);
    FILE *f = fopen(path, "w");
    if (!f) return 1;
    int rc = 0;
    rc |= cdt32_case(f, 13.42f, 215.0f, 2048, "bliss1_s215_t13.42");
    rc |= cdt32_case(f, 12.0f, 100.0f, 1024, "unit_s100_t12");
    rc |= cdt32_case(f, 10.0f, 1.7f, 512, "leaf_s1.7_t10");
    fclose(f);
    return rc;
}

/* The samplers at 128- and 192-bit precision (SAMPLING_128BIT / _192BIT: gaussian_cdf_sample_128 / _192 over binary_search_128 / _192 and compare_ge_prec,
 * src/utils/sampling/gaussian_cdf.c:110-184,468-509).  Their TABLES are MPFR arithmetic the hand build does not have (sc_mpf_exp and sc_mpf_get_pi are empty
 * under USE_SAFECRYPTO_FLOAT_MP, sc_mpf.c:99-106,822-828): create_sampler returns a context whose table is noise.  The SEARCH is plain C and is what a
 * batched engine serves - so the table of the context the reference made is REPLACED here by a table of known content and the reference's own
 * get_vector_32 (-> gaussian_cdf_sample_128 / _192) is run over it: the fixture pins the search, the limb order of the draws and the sign rule, on
 *   most significant limb = the 64-bit table of the same (tail, sigma) (gaussian_cdf_create_64: monotone, ends in a run of equal entries),
 *   lower limbs            = noise (so that it shows whether they take part in the comparison),
 * with a handful of entries set equal, in the most significant limb, to words the sampler is about to draw (ties: "x >= entry", not "entry < x") and
 * the limbs below them set just above / just below the drawn ones. */
typedef struct __attribute__((packed)) { UINT64 *cdf_256, *cdf_192, *cdf_128; SINT32 cdf_size; } cdfhp_view_t;   /* gaussian_cdf.c:59-68 */
static int cdt_hp_case(FILE *f, int prec, float tail, float sigma, int nsamp, const char *tag)
{
    static const uint8_t seed[32] = { 1, 4, 1, 4, 2, 1, 3, 5, 6, 2, 3, 7, 3, 0, 9, 5, 0, 4, 8, 8, 0, 1, 6, 8, 8, 7, 2, 4, 2, 0, 9, 6 };
    char name[96];
    uint8_t sd[32];
    memcpy(sd, seed, 32); sd[0] = (uint8_t)prec; sd[1] = (uint8_t)nsamp;
    const int W = prec / 64;
    prng_ctx_t *p1 = const_seeded_prng(sd, 32), *p2 = const_seeded_prng(sd, 32), *p3 = const_seeded_prng(sd, 32);
    utils_sampling_t *s64 = create_sampler(CDF_GAUSSIAN_SAMPLING, SAMPLING_64BIT, NORMAL_SAMPLES, 512, SAMPLING_DISABLE_BOOTSTRAP, p3, tail, sigma);
    utils_sampling_t *s = create_sampler(CDF_GAUSSIAN_SAMPLING, prec == 128 ? SAMPLING_128BIT : SAMPLING_192BIT, NORMAL_SAMPLES, 512,
                                         SAMPLING_DISABLE_BOOTSTRAP, p1, tail, sigma);
    if (!s || !s64) return 1;
    const cdf64_view_t *g64 = (const cdf64_view_t *)s64->gauss;
    const cdfhp_view_t *g = (const cdfhp_view_t *)s->gauss;
    const int size = g->cdf_size;
    if (size != g64->cdf_size || size < 2) { fprintf(stderr, "cdt_hp: table sizes differ (%d, %d)\n", size, g64->cdf_size); return 1; }
    UINT64 *tab = prec == 128 ? g->cdf_128 : g->cdf_192;                       /* [size][W] limbs, least significant first */
    /* the words the sampler is about to draw: W prng_64 a sample, limb 0 first (gaussian_cdf.c:482-483,498-500) */
    uint64_t *x = malloc(sizeof(uint64_t) * (size_t)nsamp * W);
    for (int i = 0; i < nsamp * W; i++) x[i] = prng_64(p2);
    sm_state = 0x6870000000000000ULL + (uint64_t)prec * 65536 + (uint64_t)size;
    for (int b = 0; b < size; b++) {
        for (int j = 0; j < W - 1; j++) tab[(size_t)b * W + j] = splitmix64();
        tab[(size_t)b * W + W - 1] = g64->cdf[b];
    }
    /* ties: for every 16th sample (while the table stays monotone) the entry just above the word takes the word's most significant limb */
    int ties = 0;
    for (int i = 0; i < nsamp; i += 16) {
        const uint64_t top = x[(size_t)i * W + W - 1];
        int a = 0;
        for (int st = size >> 1; st > 0; st >>= 1) { const int b = a + st; if (b < size && tab[(size_t)b * W + W - 1] < top) a = b; }
        const int b = a + 1;
        if (b >= size || tab[(size_t)b * W + W - 1] == top) continue;
        if (b + 1 < size && tab[(size_t)(b + 1) * W + W - 1] < top) continue;
        tab[(size_t)b * W + W - 1] = top;
        /* the limbs below: alternately just above and just below the word's (a full-width comparison would decide differently in the two cases) */
        for (int j = 0; j < W - 1; j++) tab[(size_t)b * W + j] = x[(size_t)i * W + j] + ((ties & 1) ? (uint64_t)-1 : 1);
        ties++;
    }
    int32_t par[4] = { prec, size, nsamp, ties };
    snprintf(name, sizeof name, "%s_par", tag); rec_i32(f, name, par, 4);
    snprintf(name, sizeof name, "%s_table", tag); rec_u64(f, name, tab, (size_t)size * W);
    snprintf(name, sizeof name, "%s_words", tag); rec_u64(f, name, x, (size_t)nsamp * W);
    int32_t *v = malloc(sizeof(int32_t) * nsamp);
    get_vector_32(s, v, nsamp, 0);
    snprintf(name, sizeof name, "%s_samples", tag); rec_i32(f, name, v, nsamp);
    free(x); free(v);
    destroy_sampler(&s); destroy_sampler(&s64);
    prng_destroy(p1); prng_destroy(p2); prng_destroy(p3);
    return 0;
}
static int gen_cdt_hp(const char *dir)
{
    char path[512];
    snprintf(path, sizeof path, "%s/cdt_hp_ref.txt", dir);
    FILE *f = fopen(path, "w");
    if (!f) return 1;
    int rc = 0;
    rc |= cdt_hp_case(f, 128, 13.42f, 215.0f, 2048, "hp128_bliss1_s215_t13.42");      /* what the reference's BLISS-B harness asks for: func_alg_bliss_b.c:118 */
    rc |= cdt_hp_case(f, 192, 13.42f, 215.0f, 1024, "hp192_bliss1_s215_t13.42");
    rc |= cdt_hp_case(f, 128, 12.0f, 3.33f, 512, "hp128_unit_s3.33_t12");              /* src/unit/unit_sampling.c:44 */
    rc |= cdt_hp_case(f, 192, 10.0f, 1.7f, 512, "hp192_leaf_s1.7_t10");
    fclose(f);
    return rc;
}

static int gen_cdt(const char *dir)
{
    char path[512];
    snprintf(path, sizeof path, "%s/cdt_ref.txt", dir);
    FILE *f = fopen(path, "w");
    if (!f) return 1;
    int rc = 0;
    rc |= cdt_case(f, 13.42f, 215.0f, 2048, "bliss1_s215_t13.42");     /* bliss_b.c:342-344, bliss_params.c:35-43 */
    rc |= cdt_case(f, 12.0f, 100.0f, 1024, "unit_s100_t12");           /* src/unit/unit_sampling.c:194-222 */
    rc |= cdt_case(f, 10.0f, 1.7f, 512, "leaf_s1.7_t10");              /* falcon-like leaf sampler, gpv.c:2674-2676 */
    fclose(f);
    return rc;
}

/* The vector fillers behind get_vector_32 in their side-channel variants (src/utils/sampling/sampling.c:107-228): NORMAL / SHUFFLE / BLINDING samples
 * (create_sampler's `blinding`, safecrypto.c:492-497) x sample discarding off / LO / MD / HI (set_discard, bliss_b.c:350).  One seeded CSPRNG per case;
 * its twin (same seed) yields the pool words the filler is about to consume - prng_32 and prng_64 read the same word pool (prng.c:95-143,963-993). */
static int sampling_case(FILE *f, int blinding, unsigned discard, int n, int centre, const char *tag)
{
    static const uint8_t seed[32] = { 1, 6, 1, 8, 0, 3, 3, 9, 8, 8, 7, 4, 9, 8, 9, 4, 8, 4, 8, 2, 0, 4, 5, 8, 6, 8, 3, 4, 3, 6, 5, 6 };
    char name[96];
    uint8_t sd[32];
    memcpy(sd, seed, 32); sd[0] = (uint8_t)(blinding * 16 + discard); sd[1] = (uint8_t)n;
    prng_ctx_t *p1 = const_seeded_prng(sd, 32), *p2 = const_seeded_prng(sd, 32);
    utils_sampling_t *s = create_sampler(CDF_GAUSSIAN_SAMPLING, SAMPLING_64BIT, (sample_blinding_e)blinding, 512,
                                         SAMPLING_DISABLE_BOOTSTRAP, p1, 13.42f, 215.0f);
    if (!s) return 1;
    set_discard(s, discard);
    const size_t words = 16 * (size_t)n + 64;
    int32_t *tape = malloc(sizeof(int32_t) * words), *v = malloc(sizeof(int32_t) * n);
    for (size_t i = 0; i < words; i++) tape[i] = (int32_t)prng_32(p2);
    get_vector_32(s, v, n, centre);
    snprintf(name, sizeof name, "%s_par", tag);
    { int32_t par[4] = { blinding, (int32_t)discard, n, centre }; rec_i32(f, name, par, 4); }
    snprintf(name, sizeof name, "%s_tape", tag); rec_i32(f, name, tape, words);
    snprintf(name, sizeof name, "%s_v", tag); rec_i32(f, name, v, n);
    free(tape); free(v);
    destroy_sampler(&s);
    prng_destroy(p1); prng_destroy(p2);
    return 0;
}
static int gen_sampling(const char *dir)
{
    char path[512], tag[64];
    snprintf(path, sizeof path, "%s/sampling_ref.txt", dir);
    FILE *f = fopen(path, "w");
    if (!f) return 1;
    int rc = 0, k = 0;
    static const int bl[3] = { NORMAL_SAMPLES, SHUFFLE_SAMPLES, BLINDING_SAMPLES };
    static const unsigned dc[4] = { 0, SCA_PATTERN_SAMPLE_DISCARD_LO, SCA_PATTERN_SAMPLE_DISCARD_MD, SCA_PATTERN_SAMPLE_DISCARD_HI };
    for (int b = 0; b < 3; b++) for (int d = 0; d < 4; d++) {
        snprintf(tag, sizeof tag, "sca%d", k++);
        rc |= sampling_case(f, bl[b], dc[d], b == 1 && d == 3 ? 96 : 128, d == 2 ? 37 : 0, tag);
    }
    { int32_t cnt = k; rec_i32(f, "sca_cases", &cnt, 1); }
    fclose(f);
    return rc;
}

/* ------------------------------------------------------------ Dilithium ---- */

#include "schemes/sig/dilithium/dilithium.h"
#include "schemes/sig/dilithium/dilithium_params.h"
#include "utils/arith/module_lwe.h"

static void rec_u8(FILE *f, const char *name, const uint8_t *v, size_t n)
{
    fprintf(f, "@ %s %zu\n", name, n);
    for (size_t i = 0; i < n; i++) fprintf(f, "%u%c", v[i], ((i & 31) == 31 || i + 1 == n) ? '\n' : ' ');
    if (n == 0) fprintf(f, "\n");
}

/* One key pair of parameter set `set` (3 = "Dilithium-III": k=6,l=5, dilithium_params.c:56-64) made by the
 * reference's keygen; the expanded matrix A (NTT domain, row-major i then j) exactly as
 * create_rand_product_32_csprng regenerates it from rho (module_lwe.c:519-547,641-660, create_csprng :914-940);
 * signatures of counter-derived messages by safecrypto_sign (deterministic: USE_DETERMINISTIC_DILITHIUM) and the
 * verdict of safecrypto_verify on the genuine and on a tampered signature. */
static int dilithium_case(FILE *f, int set, int nmsg)
{
    UINT32 flags[2] = { SC_FLAG_NONE | SC_FLAG_MORE, SC_FLAG_NONE };
    char name[96], tag[32];
    snprintf(tag, sizeof tag, "dil%d", set);
    safecrypto_t *sc = create_fixed(SC_SCHEME_SIG_DILITHIUM, set, flags);
    if (!sc) return 1;
    if (SC_FUNC_SUCCESS != safecrypto_keygen(sc)) return 1;
    const dilithium_set_t *P = sc->dilithium->params;
    const int n = P->n, k = P->k, l = P->l;
    int32_t par[16] = { (int32_t)P->n, (int32_t)P->q, (int32_t)P->q_bits, (int32_t)P->k, (int32_t)P->l, (int32_t)P->d,
                        (int32_t)P->weight_of_c, (int32_t)P->gamma_1, (int32_t)P->gamma_1_bits, (int32_t)P->gamma_2,
                        (int32_t)P->eta, (int32_t)P->z_bits, (int32_t)P->beta, (int32_t)P->omega, (int32_t)P->omega_bits, set };
    snprintf(name, sizeof name, "%s_params", tag); rec_i32(f, name, par, 16);
    const SINT32 *priv = (const SINT32 *)sc->privkey->key;
    const SINT32 *pub = (const SINT32 *)sc->pubkey->key;
    snprintf(name, sizeof name, "%s_s1", tag); rec_i32(f, name, priv, (size_t)l * n);
    snprintf(name, sizeof name, "%s_s2", tag); rec_i32(f, name, priv + l * n, (size_t)k * n);
    snprintf(name, sizeof name, "%s_t", tag);  rec_i32(f, name, priv + (l + k) * n, (size_t)k * n);
    const uint8_t *tail = (const uint8_t *)(priv + (NUM_DILITHIUM_PRIVKEY_K * k + l) * n);
    snprintf(name, sizeof name, "%s_rho", tag); rec_u8(f, name, tail, 32);
    snprintf(name, sizeof name, "%s_K", tag);   rec_u8(f, name, tail + 32, 32);
    snprintf(name, sizeof name, "%s_tr", tag);  rec_u8(f, name, tail + 64, 48);
    snprintf(name, sizeof name, "%s_t1", tag);  rec_i32(f, name, pub, (size_t)k * n);
    snprintf(name, sizeof name, "%s_pub_rho", tag); rec_u8(f, name, (const uint8_t *)(pub + k * n), 32);

    /* the matrix, in the order the reference draws it */
    int32_t *A = malloc(sizeof(int32_t) * (size_t)k * l * n);
    prng_ctx_t *csprng = create_csprng(sc, tail, 32);
    for (int i = 0; i < k * l; i++) uniform_random_ring_q_csprng(csprng, A + (size_t)i * n, n, P->q, P->q_bits);
    prng_destroy(csprng);
    snprintf(name, sizeof name, "%s_Ahat", tag); rec_i32(f, name, A, (size_t)k * l * n);
    free(A);

    sm_state = 0xd111000000000000ULL + set;
    int32_t *trials = malloc(sizeof(int32_t) * nmsg), *verdicts = malloc(sizeof(int32_t) * nmsg * 2);
    for (int m = 0; m < nmsg; m++) {
        uint8_t msg[96];
        size_t mlen = (m == 0) ? 0 : (m == 1) ? 1 : (m == 2) ? 96 : 64;     /* empty / tiny / long / the bench's 64 B */
        for (size_t i = 0; i < mlen; i++) msg[i] = (uint8_t)splitmix64();
        uint8_t *sig = NULL; size_t siglen = 0;
        size_t before = sc->stats.sig_num_trials;
        if (SC_FUNC_SUCCESS != safecrypto_sign(sc, msg, mlen, &sig, &siglen)) return 1;
        trials[m] = (int32_t)(sc->stats.sig_num_trials - before);
        snprintf(name, sizeof name, "%s_msg%d", tag, m); rec_u8(f, name, msg, mlen);
        snprintf(name, sizeof name, "%s_sig%d", tag, m); rec_u8(f, name, sig, siglen);
        verdicts[2 * m] = safecrypto_verify(sc, msg, mlen, sig, siglen);
        sig[siglen / 2] ^= 0x10;                                             /* func_alg_dilithium.c:200-222 style bit flip */
        verdicts[2 * m + 1] = safecrypto_verify(sc, msg, mlen, sig, siglen);
        free(sig);
    }
    snprintf(name, sizeof name, "%s_trials", tag); rec_i32(f, name, trials, nmsg);
    snprintf(name, sizeof name, "%s_verify", tag); rec_i32(f, name, verdicts, (size_t)nmsg * 2);
    free(trials); free(verdicts);
    { int32_t eb = (int32_t)P->eta_bits; snprintf(name, sizeof name, "%s_eta_bits", tag); rec_i32(f, name, &eb, 1); }
    safecrypto_destroy(sc);
    /* key generation (dilithium.c:752-935) from a seeded context CSPRNG: every component of the key pair */
    for (int kg = 0; kg < 2; kg++) {
        uint8_t seed[32];
        for (int i = 0; i < 32; i++) seed[i] = (uint8_t)splitmix64();
        safecrypto_t *s2 = create_fixed(SC_SCHEME_SIG_DILITHIUM, set, flags);
        if (!s2) return 1;
        prng_ctx_t *o2 = s2->prng_ctx[0], *mine = seeded_prng(seed, 32);
        s2->prng_ctx[0] = mine;
        if (SC_FUNC_SUCCESS != safecrypto_keygen(s2)) return 1;
        const SINT32 *pv = (const SINT32 *)s2->privkey->key, *pb = (const SINT32 *)s2->pubkey->key;
        const uint8_t *tl = (const uint8_t *)(pv + (NUM_DILITHIUM_PRIVKEY_K * k + l) * n);
        snprintf(name, sizeof name, "%s_kg_seed%d", tag, kg); rec_u8(f, name, seed, 32);
        snprintf(name, sizeof name, "%s_kg_s1_%d", tag, kg); rec_i32(f, name, pv, (size_t)l * n);
        snprintf(name, sizeof name, "%s_kg_s2_%d", tag, kg); rec_i32(f, name, pv + l * n, (size_t)k * n);
        snprintf(name, sizeof name, "%s_kg_t%d", tag, kg); rec_i32(f, name, pv + (l + k) * n, (size_t)k * n);
        snprintf(name, sizeof name, "%s_kg_rho%d", tag, kg); rec_u8(f, name, tl, 32);
        snprintf(name, sizeof name, "%s_kg_K%d", tag, kg); rec_u8(f, name, tl + 32, 32);
        snprintf(name, sizeof name, "%s_kg_tr%d", tag, kg); rec_u8(f, name, tl + 64, 48);
        snprintf(name, sizeof name, "%s_kg_t1_%d", tag, kg); rec_i32(f, name, pb, (size_t)k * n);
        s2->prng_ctx[0] = o2;
        prng_destroy(mine);
        safecrypto_destroy(s2);
    }
    return 0;
}

static int gen_dilithium(const char *dir)
{
    char path[512];
    snprintf(path, sizeof path, "%s/dilithium_ref.txt", dir);
    FILE *f = fopen(path, "w");
    if (!f) return 1;
    int rc = dilithium_case(f, 3, 12);
    rc |= dilithium_case(f, 1, 4);
    rc |= dilithium_case(f, 0, 3);                                     /* the remaining sets: (k, l) = (3, 2) and (5, 4) */
    rc |= dilithium_case(f, 2, 3);
    fclose(f);
    return rc;
}

/* --------------------------------------------------------------- BLISS-B ---- */

#include "schemes/sig/bliss_b/bliss_b.h"
#include "schemes/sig/bliss_b/bliss_params.h"

typedef struct __attribute__((packed)) { UINT64 *cdf; SINT32 cdf_size; SINT32 k; SINT32 use_kl; prng_ctx_t *prng_ctx; } cdf64_full_view_t; /* gaussian_cdf.c:74-81 */

/* BLISS-B signing draws ALL its randomness from sc->prng_ctx[0] as a stream of 32-bit pool words
 * (prng_64 = two words, prng_var = a bit reservoir refilled by prng_32; src/utils/crypto/prng.c:963-1048).
 * For each message the context's CSPRNG is replaced by a freshly seeded one and a twin with the same seed
 * is read with prng_32 only: that is the "random tape" the signature is a pure function of. */
static int bliss_case(FILE *f, int set, int nmsg, int huffman)
{
    /* huffman: sc->coding_signature.type = SC_ENTROPY_HUFFMAN_STATIC (bliss_b.c:197-200) - the same signing path, the signature's t and z coded by
     * entropy_poly_encode_32 / _16 (src/utils/entropy/entropy.c:70-112, entropy_huffman.c) instead of as raw fields */
    /* huffman == 2: SC_ENTROPY_BAC - t and z through the arithmetic coder (src/utils/entropy/bac.c) with the two distributions bliss_b.c:397-398 builds,
     * which are recorded as data (gauss_freq_bac_64 is long double arithmetic) */
    UINT32 flags[2] = { SC_FLAG_0_SAMPLE_64BIT | SC_FLAG_MORE | (huffman == 1 ? SC_FLAG_0_ENTROPY_HUFFMAN : huffman == 2 ? SC_FLAG_0_ENTROPY_BAC : 0), SC_FLAG_NONE };
    char name[96], tag[32];
    snprintf(tag, sizeof tag, huffman == 1 ? "blissh%d" : huffman == 2 ? "blissb%d" : "bliss%d", set);
    safecrypto_t *sc = create_fixed(SC_SCHEME_SIG_BLISS, set, flags);
    if (!sc) return 1;
    if (SC_FUNC_SUCCESS != safecrypto_keygen(sc)) return 1;
    const bliss_set_t *P = sc->bliss->params;
    const int n = P->n;
    int32_t par[16] = { P->n, P->q, P->q_bits, P->n_bits, P->d, P->p, P->kappa, P->b_inf, (int32_t)P->b_l2, P->z1_bits, P->z2_bits, set, 0, 0, 0, 0 };
    float fl[3] = { P->sig, P->m, 13.42f };
    memcpy(&par[12], fl, 12);                                   /* sigma, M, tail as raw float bits */
    snprintf(name, sizeof name, "%s_params", tag); rec_i32(f, name, par, 16);
    snprintf(name, sizeof name, "%s_f", tag); rec_i16(f, name, (const int16_t *)sc->privkey->key, n);
    snprintf(name, sizeof name, "%s_g", tag); rec_i16(f, name, (const int16_t *)sc->privkey->key + n, n);
    snprintf(name, sizeof name, "%s_a", tag); rec_i16(f, name, (const int16_t *)sc->pubkey->key, n);
    if (huffman == 2) {                                         /* 64-bit entries as pairs of 32-bit words, low word first */
        snprintf(name, sizeof name, "%s_dist_t", tag); rec_i32(f, name, (const int32_t *)sc->dist[2], (size_t)2 << P->z1_bits);
        snprintf(name, sizeof name, "%s_dist_z", tag); rec_i32(f, name, (const int32_t *)sc->dist[3], (size_t)2 << P->z2_bits);
    }
    int32_t *trials = malloc(sizeof(int32_t) * nmsg), *verdicts = malloc(sizeof(int32_t) * nmsg * 2);
    prng_ctx_t *orig = sc->prng_ctx[0];
    sm_state = 0xb1155000ULL + set + (huffman ? 0x100 * huffman : 0);
    /* candidates are drawn until nmsg are kept: the first nmsg-3 whatever they need, then three that took >= 3 attempts
     * (a second fresh pair of Gaussian vectors) */
    int kept = 0, long_ones = 0;
    for (int cand = 0; cand < 4000 && kept < nmsg; cand++) {
        uint8_t msg[96], seed[32];
        size_t mlen = (cand == 0) ? 1 : (cand == 1) ? 96 : 64;  /* oracle() rejects an empty message (bliss_b.c:571-574) */
        for (size_t i = 0; i < mlen; i++) msg[i] = (uint8_t)splitmix64();
        for (int i = 0; i < 32; i++) seed[i] = (uint8_t)splitmix64();
        prng_ctx_t *mine = seeded_prng(seed, 32), *twin = seeded_prng(seed, 32);
        sc->prng_ctx[0] = mine;
        sc->sc_gauss->prng_ctx = mine;
        ((cdf64_full_view_t *)sc->sc_gauss->gauss)->prng_ctx = mine;
        uint8_t *sig = NULL; size_t siglen = 0;
        size_t before = sc->stats.sig_num_trials;
        if (SC_FUNC_SUCCESS != safecrypto_sign(sc, msg, mlen, &sig, &siglen)) return 1;
        const int tr = (int)(sc->stats.sig_num_trials - before);
        const int keep = (kept < nmsg - 3) || (tr >= 3 && long_ones < 3);
        if (keep) {
            const int m = kept++;
            if (tr >= 3 && m >= nmsg - 3) long_ones++;
            trials[m] = tr;
            const size_t words = (size_t)(prng_get_out_bytes(mine) / 4);
            int32_t *tape = malloc(sizeof(int32_t) * words);
            for (size_t i = 0; i < words; i++) tape[i] = (int32_t)prng_32(twin);
            snprintf(name, sizeof name, "%s_msg%d", tag, m); rec_u8(f, name, msg, mlen);
            snprintf(name, sizeof name, "%s_tape%d", tag, m); rec_i32(f, name, tape, words);
            snprintf(name, sizeof name, "%s_sig%d", tag, m); rec_u8(f, name, sig, siglen);
            verdicts[2 * m] = safecrypto_verify(sc, msg, mlen, sig, siglen);
            sig[siglen / 3] ^= 0x04;
            verdicts[2 * m + 1] = safecrypto_verify(sc, msg, mlen, sig, siglen);
            free(tape);
        }
        free(sig);
        sc->prng_ctx[0] = orig; sc->sc_gauss->prng_ctx = orig; ((cdf64_full_view_t *)sc->sc_gauss->gauss)->prng_ctx = orig;
        prng_destroy(mine); prng_destroy(twin);
    }
    if (kept != nmsg) return 1;
    snprintf(name, sizeof name, "%s_trials", tag); rec_i32(f, name, trials, nmsg);
    snprintf(name, sizeof name, "%s_verify", tag); rec_i32(f, name, verdicts, (size_t)nmsg * 2);
    free(trials); free(verdicts);
    safecrypto_destroy(sc);
    return 0;
}

/* BLISS-B signing with the sampler at 128- / 192-bit precision - what the reference's own harness asks for (test/functional/func_alg_bliss_b.c:118:
 * SC_FLAG_0_SAMPLE_CDF | SC_FLAG_0_SAMPLE_128BIT).  In this build the context's table is noise (no MPFR: cdt_hp_case above), so it is replaced right after creation by
 * one of known content - most significant limbs = the 64-bit table of the set's (tail, sigma), the limbs below = noise - and recorded; everything else is bliss_case:
 * a fresh seeded generator per message, its twin read as the tape (a sample now takes precision / 64 prng_64 draws, i.e. 4 or 6 tape words). */
typedef struct __attribute__((packed)) { UINT64 *cdf_256, *cdf_192, *cdf_128; SINT32 cdf_size; SINT32 k; SINT32 use_kl; prng_ctx_t *prng_ctx; } cdfhp_full_view_t;
static int bliss_hp_case(FILE *f, int set, int nmsg, int prec)
{
    UINT32 flags[2] = { SC_FLAG_0_SAMPLE_CDF | (prec == 128 ? SC_FLAG_0_SAMPLE_128BIT : SC_FLAG_0_SAMPLE_192BIT) | SC_FLAG_MORE, SC_FLAG_NONE };
    char name[96], tag[32];
    snprintf(tag, sizeof tag, "blisshp%d_%d", prec, set);
    safecrypto_t *sc = create_fixed(SC_SCHEME_SIG_BLISS, set, flags);
    if (!sc) return 1;
    if (!sc->sc_gauss || (int)sc->sc_gauss->precision != prec) { fprintf(stderr, "bliss_hp: the context's sampler is not %d-bit\n", prec); return 1; }
    const bliss_set_t *P = sc->bliss->params;
    const int n = P->n, W = prec / 64;
    cdfhp_full_view_t *g = (cdfhp_full_view_t *)sc->sc_gauss->gauss;
    {
        static const uint8_t sd[32] = { 9, 9, 1 };
        prng_ctx_t *p3 = seeded_prng(sd, 32);
        utils_sampling_t *s64 = create_sampler(CDF_GAUSSIAN_SAMPLING, SAMPLING_64BIT, NORMAL_SAMPLES, 512, SAMPLING_DISABLE_BOOTSTRAP, p3, 13.42f, P->sig);
        if (!s64) return 1;
        const cdf64_view_t *g64 = (const cdf64_view_t *)s64->gauss;
        if (g64->cdf_size != g->cdf_size) { fprintf(stderr, "bliss_hp: table sizes differ\n"); return 1; }
        UINT64 *tab = prec == 128 ? g->cdf_128 : g->cdf_192;
        sm_state = 0xb1155480ULL + (uint64_t)prec + set;
        for (int b = 0; b < g->cdf_size; b++) {
            for (int j = 0; j < W - 1; j++) tab[(size_t)b * W + j] = splitmix64();
            tab[(size_t)b * W + W - 1] = g64->cdf[b];
        }
        snprintf(name, sizeof name, "%s_table", tag); rec_u64(f, name, tab, (size_t)g->cdf_size * W);
        destroy_sampler(&s64); prng_destroy(p3);
    }
    if (SC_FUNC_SUCCESS != safecrypto_keygen(sc)) return 1;
    int32_t par[16] = { P->n, P->q, P->q_bits, P->n_bits, P->d, P->p, P->kappa, P->b_inf, (int32_t)P->b_l2, P->z1_bits, P->z2_bits, set, 0, 0, 0, 0 };
    float fl[3] = { P->sig, P->m, 13.42f };
    memcpy(&par[12], fl, 12);
    snprintf(name, sizeof name, "%s_params", tag); rec_i32(f, name, par, 16);
    { int32_t pr[2] = { prec, g->cdf_size }; snprintf(name, sizeof name, "%s_precision", tag); rec_i32(f, name, pr, 2); }
    snprintf(name, sizeof name, "%s_f", tag); rec_i16(f, name, (const int16_t *)sc->privkey->key, n);
    snprintf(name, sizeof name, "%s_g", tag); rec_i16(f, name, (const int16_t *)sc->privkey->key + n, n);
    snprintf(name, sizeof name, "%s_a", tag); rec_i16(f, name, (const int16_t *)sc->pubkey->key, n);
    int32_t *trials = malloc(sizeof(int32_t) * nmsg), *verdicts = malloc(sizeof(int32_t) * nmsg * 2);
    prng_ctx_t *orig = sc->prng_ctx[0];
    sm_state = 0xb1155900ULL + (uint64_t)prec * 16 + set;
    int kept = 0, long_ones = 0;
    for (int cand = 0; cand < 4000 && kept < nmsg; cand++) {
        uint8_t msg[96], seed[32];
        size_t mlen = (cand == 0) ? 1 : (cand == 1) ? 96 : 64;
        for (size_t i = 0; i < mlen; i++) msg[i] = (uint8_t)splitmix64();
        for (int i = 0; i < 32; i++) seed[i] = (uint8_t)splitmix64();
        prng_ctx_t *mine = seeded_prng(seed, 32), *twin = seeded_prng(seed, 32);
        sc->prng_ctx[0] = mine; sc->sc_gauss->prng_ctx = mine; g->prng_ctx = mine;
        uint8_t *sig = NULL; size_t siglen = 0;
        size_t before = sc->stats.sig_num_trials;
        if (SC_FUNC_SUCCESS != safecrypto_sign(sc, msg, mlen, &sig, &siglen)) return 1;
        const int tr = (int)(sc->stats.sig_num_trials - before);
        const int keep = (kept < nmsg - 2) || (tr >= 3 && long_ones < 2);
        if (keep) {
            const int m = kept++;
            if (tr >= 3 && m >= nmsg - 2) long_ones++;
            trials[m] = tr;
            const size_t words = (size_t)(prng_get_out_bytes(mine) / 4);
            int32_t *tape = malloc(sizeof(int32_t) * words);
            for (size_t i = 0; i < words; i++) tape[i] = (int32_t)prng_32(twin);
            snprintf(name, sizeof name, "%s_msg%d", tag, m); rec_u8(f, name, msg, mlen);
            snprintf(name, sizeof name, "%s_tape%d", tag, m); rec_i32(f, name, tape, words);
            snprintf(name, sizeof name, "%s_sig%d", tag, m); rec_u8(f, name, sig, siglen);
            verdicts[2 * m] = safecrypto_verify(sc, msg, mlen, sig, siglen);
            sig[siglen / 3] ^= 0x04;
            verdicts[2 * m + 1] = safecrypto_verify(sc, msg, mlen, sig, siglen);
            free(tape);
        }
        free(sig);
        sc->prng_ctx[0] = orig; sc->sc_gauss->prng_ctx = orig; g->prng_ctx = orig;
        prng_destroy(mine); prng_destroy(twin);
    }
    if (kept != nmsg) return 1;
    snprintf(name, sizeof name, "%s_trials", tag); rec_i32(f, name, trials, nmsg);
    snprintf(name, sizeof name, "%s_verify", tag); rec_i32(f, name, verdicts, (size_t)nmsg * 2);
    free(trials); free(verdicts);
    safecrypto_destroy(sc);
    return 0;
}
static int gen_bliss_hp(const char *dir)
{
    char path[512];
    snprintf(path, sizeof path, "%s/bliss_hp_ref.txt", dir);
    FILE *f = fopen(path, "w");
    if (!f) return 1;
    int rc = bliss_hp_case(f, 1, 7, 128);                              /* BLISS-B-I, the harness's configuration */
    rc |= bliss_hp_case(f, 1, 3, 192);
    rc |= bliss_hp_case(f, 4, 3, 128);
    fclose(f);
    return rc;
}

/* bliss_b_keygen (bliss_b.c:1124-1264) draws g, then f until NTT(f) is invertible, from sc->prng_ctx[0] through
 * poly_32_uniform_rand (poly_32.c:132-156: one prng_32 word per placement attempt, bit 0 = sign, bits 1.. = position).  The first
 * three key pairs are kept whatever they took, then the first one whose f had to be drawn again (~4 % of key pairs). */
static int bliss_keygen_case(FILE *f, int set, int nkeep)
{
    UINT32 flags[2] = { SC_FLAG_0_SAMPLE_64BIT | SC_FLAG_MORE, SC_FLAG_NONE };
    char name[96], tag[32];
    snprintf(tag, sizeof tag, "bliss%d", set);
    safecrypto_t *sc = create_fixed(SC_SCHEME_SIG_BLISS, set, flags);
    if (!sc) return 1;
    const bliss_set_t *P = sc->bliss->params;
    const int n = P->n;
    int32_t nz[2] = { P->nz[0], P->nz[1] };
    snprintf(name, sizeof name, "%s_kg_nz", tag); rec_i32(f, name, nz, 2);
    prng_ctx_t *orig = sc->prng_ctx[0];
    sm_state = 0xb1155c60ULL + set;
    int kept = 0, have_long = 0;
    for (int cand = 0; cand < 2000 && kept < nkeep; cand++) {
        uint8_t seed[32];
        for (int i = 0; i < 32; i++) seed[i] = (uint8_t)splitmix64();
        prng_ctx_t *mine = seeded_prng(seed, 32), *twin = seeded_prng(seed, 32);
        sc->prng_ctx[0] = mine;
        if (SC_FUNC_SUCCESS != safecrypto_keygen(sc)) return 1;
        const size_t words = (size_t)(prng_get_out_bytes(mine) / 4);
        const int is_long = words > (size_t)(3 * (nz[0] + nz[1]) + 3 * (nz[0] + nz[1]) / 4);      /* three polynomials' worth of draws */
        if (kept < nkeep - 1 || (is_long && !have_long)) {
            if (is_long) have_long = 1;
            int32_t *tape = malloc(sizeof(int32_t) * words);
            for (size_t i = 0; i < words; i++) tape[i] = (int32_t)prng_32(twin);
            snprintf(name, sizeof name, "%s_kg_tape%d", tag, kept); rec_i32(f, name, tape, words);
            snprintf(name, sizeof name, "%s_kg_f%d", tag, kept); rec_i16(f, name, (const int16_t *)sc->privkey->key, n);
            snprintf(name, sizeof name, "%s_kg_g%d", tag, kept); rec_i16(f, name, (const int16_t *)sc->privkey->key + n, n);
            snprintf(name, sizeof name, "%s_kg_a%d", tag, kept); rec_i16(f, name, (const int16_t *)sc->pubkey->key, n);
            free(tape);
            kept++;
        }
        sc->prng_ctx[0] = orig;
        prng_destroy(mine); prng_destroy(twin);
    }
    safecrypto_destroy(sc);
    return kept == nkeep && have_long ? 0 : 1;
}

static int gen_bliss(const char *dir)
{
    char path[512];
    snprintf(path, sizeof path, "%s/bliss_ref.txt", dir);
    FILE *f = fopen(path, "w");
    if (!f) return 1;
    int rc = bliss_case(f, 1, 9, 0);
    rc |= bliss_keygen_case(f, 1, 4);
    rc |= bliss_case(f, 3, 4, 0);                                      /* BLISS-B-III and -IV: other kappa, sigma, d, p, bounds */
    rc |= bliss_case(f, 4, 4, 0);
    fclose(f);
    return rc;
}

/* Huffman-coded signatures (SC_FLAG_0_ENTROPY_HUFFMAN) and the static code tables the coder walks, as data: the tables are literals in
 * src/utils/entropy/huffman.h ("Generated from Matlab/Octave"), not the output of an algorithm the library contains */
#include "utils/entropy/huffman.h"
/* signatures under the arithmetic coder (SC_FLAG_0_ENTROPY_BAC), with the coder's two distributions */
static int gen_bliss_bac(const char *dir)
{
    char path[512];
    snprintf(path, sizeof path, "%s/bliss_bac_ref.txt", dir);
    FILE *f = fopen(path, "w");
    if (!f) return 1;
    int rc = bliss_case(f, 1, 6, 2);
    rc |= bliss_case(f, 4, 4, 2);                                      /* z2_bits = 4, another sigma */
    fclose(f);
    return rc;
}

static int gen_bliss_huff(const char *dir)
{
    char path[512], name[64];
    snprintf(path, sizeof path, "%s/bliss_huff_ref.txt", dir);
    FILE *f = fopen(path, "w");
    if (!f) return 1;
    const huffman_table_t *tabs[5] = { huff_table_gaussian_2, huff_table_gaussian_3, huff_table_gaussian_4, huff_table_gaussian_5, huff_table_gaussian_6 };
    for (int t = 0; t < 5; t++) {
        int32_t code[64], bits[64], meta[2] = { (int32_t)tabs[t]->depth, (int32_t)tabs[t]->max_bits };
        for (UINT32 i = 0; i < tabs[t]->depth; i++) { code[i] = (int32_t)tabs[t]->codes[i].code; bits[i] = tabs[t]->codes[i].bits; }
        snprintf(name, sizeof name, "huff%d_meta", t + 2); rec_i32(f, name, meta, 2);
        snprintf(name, sizeof name, "huff%d_code", t + 2); rec_i32(f, name, code, tabs[t]->depth);
        snprintf(name, sizeof name, "huff%d_bits", t + 2); rec_i32(f, name, bits, tabs[t]->depth);
    }
    int rc = bliss_case(f, 1, 6, 1);
    rc |= bliss_case(f, 4, 4, 1);                                      /* z2_bits = 4: the 3-bit table for z */
    rc |= bliss_case(f, 2, 3, 1);                                      /* z2_bits = 2 */
    fclose(f);
    return rc;
}

/* Signatures that are one polynomial (+ raw bytes) in both codings: the same context signs the same message from the same CSPRNG seed twice, once with
 * sc->coding_signature.type = SC_ENTROPY_NONE and once with SC_ENTROPY_HUFFMAN_STATIC (the field entropy_poly_encode_32 is handed at falcon.c:900,
 * ens_dlp_sig.c:1282, ring_tesla.c:887) - the same polynomial, raw and coded - and verifies the coded one.  bits / tail describe the raw layout. */
static int huff_rows_case(FILE *f, sc_scheme_e scheme, int set, const char *tag, int nmsg, int n, int bits, int tail)
{
    UINT32 flags[2] = { SC_FLAG_0_ENTROPY_HUFFMAN, SC_FLAG_NONE };
    char name[96];
    safecrypto_t *sc = create_fixed(scheme, set, flags);
    if (!sc) return 1;
    if (SC_FUNC_SUCCESS != safecrypto_keygen(sc)) return 1;
    int32_t par[4] = { n, bits, tail, set };
    snprintf(name, sizeof name, "%s_layout", tag); rec_i32(f, name, par, 4);
    prng_ctx_t *orig = sc->prng_ctx[0];
    sm_state = 0x68756666ULL + (uint64_t)scheme * 64 + set;
    int32_t *verdicts = malloc(sizeof(int32_t) * nmsg);
    int kept = 0;
    for (int cand = 0; cand < 8 * nmsg && kept < nmsg; cand++) {
        uint8_t msg[64], seed[32];
        for (size_t i = 0; i < sizeof msg; i++) msg[i] = (uint8_t)splitmix64();
        for (int i = 0; i < 32; i++) seed[i] = (uint8_t)splitmix64();
        uint8_t *sig[2] = { NULL, NULL }; size_t len[2] = { 0, 0 };
        int rc[2];
        for (int pass = 0; pass < 2; pass++) {
            sc->coding_signature.type = pass ? SC_ENTROPY_HUFFMAN_STATIC : SC_ENTROPY_NONE;
            prng_ctx_t *mine = seeded_prng(seed, 32);
            sc->prng_ctx[0] = mine;
            if (sc->sc_gauss) sc->sc_gauss->prng_ctx = mine;
            rc[pass] = safecrypto_sign(sc, msg, sizeof msg, &sig[pass], &len[pass]);
            sc->prng_ctx[0] = orig;
            if (sc->sc_gauss) sc->sc_gauss->prng_ctx = orig;
            prng_destroy(mine);
        }
        if (SC_FUNC_SUCCESS == rc[0] && SC_FUNC_SUCCESS == rc[1]) {                /* (Falcon fails on its norm bound now and then: another message) */
            snprintf(name, sizeof name, "%s_raw%d", tag, kept); rec_u8(f, name, sig[0], len[0]);
            snprintf(name, sizeof name, "%s_coded%d", tag, kept); rec_u8(f, name, sig[1], len[1]);
            verdicts[kept] = safecrypto_verify(sc, msg, sizeof msg, sig[1], len[1]);      /* coding type is HUFFMAN_STATIC here */
            kept++;
        }
        free(sig[0]); free(sig[1]);
    }
    snprintf(name, sizeof name, "%s_verify", tag); rec_i32(f, name, verdicts, (size_t)kept);
    free(verdicts);
    safecrypto_destroy(sc);
    return kept == nmsg ? 0 : 1;
}

static int gen_huff_rows(const char *dir)
{
    char path[512], name[64];
    snprintf(path, sizeof path, "%s/huff_rows_ref.txt", dir);
    FILE *f = fopen(path, "w");
    if (!f) return 1;
    const huffman_table_t *tabs[5] = { huff_table_gaussian_2, huff_table_gaussian_3, huff_table_gaussian_4, huff_table_gaussian_5, huff_table_gaussian_6 };
    for (int t = 0; t < 5; t++) {
        int32_t code[64], bits[64], meta[2] = { (int32_t)tabs[t]->depth, (int32_t)tabs[t]->max_bits };
        for (UINT32 i = 0; i < tabs[t]->depth; i++) { code[i] = (int32_t)tabs[t]->codes[i].code; bits[i] = tabs[t]->codes[i].bits; }
        snprintf(name, sizeof name, "huff%d_meta", t + 2); rec_i32(f, name, meta, 2);
        snprintf(name, sizeof name, "huff%d_code", t + 2); rec_i32(f, name, code, tabs[t]->depth);
        snprintf(name, sizeof name, "huff%d_bits", t + 2); rec_i32(f, name, bits, tabs[t]->depth);
    }
    int rc = huff_rows_case(f, SC_SCHEME_SIG_FALCON, 0, "falcon0", 4, 512, 12, 0);             /* q_bits - 2 */
    rc |= huff_rows_case(f, SC_SCHEME_SIG_FALCON, 2, "falcon2", 2, 1024, 12, 0);
    rc |= huff_rows_case(f, SC_SCHEME_SIG_RING_TESLA, 0, "tesla0", 4, 512, 24, 64);            /* q_bits = 24; 64 digest bytes behind z */
    /* (the ENS / DLP signatures have the same shape - s1, n fields of q_bits bits, ens_dlp_sig.c:1282 - but this build's own verifier rejects its own
     * Huffman-coded ENS signatures, so there is nothing to pin them against: not in the fixture) */
    fclose(f);
    return rc;
}

/* ----------------------------------------------------------------- Kyber ---- */

#include "schemes/kem/kyber/kyber_kem.h"
#include "schemes/kem/kyber/kyber_kem_params.h"

/* Kyber (round 1, q = 7681) CPA encryption / decryption as kyber_kem_encapsulation / _decapsulation call them
 * (kyber_kem.c:866,992): kyber_cpa_enc / kyber_cpa_dec of module_lwe.c:1268-1481 on a key made by the reference's keygen.
 * The matrix is dumped in the order the CSPRNG generates it (k*k polynomials, module_lwe.c:519-547); the noise is the
 * byte stream the (2k+1)*n/256 prng_mem(eta*64) calls of binomial_rand_sample_csprng return for seed r
 * (module_lwe.c:380-412,1306-1316), read from a twin generator. */
static int kyber_case(FILE *f, int set, int nmsg)
{
    UINT32 flags[2] = { SC_FLAG_NONE | SC_FLAG_MORE, SC_FLAG_NONE };
    char name[96], tag[32];
    snprintf(tag, sizeof tag, "kyber%d", set);
    safecrypto_t *sc = create_fixed(SC_SCHEME_KEM_KYBER, set, flags);
    if (!sc) return 1;
    if (SC_FUNC_SUCCESS != safecrypto_keygen(sc)) return 1;
    const kyber_set_t *P = sc->kyber->params;
    const int n = P->n, k = P->k, eta = P->eta;
    int32_t par[16] = { (int32_t)P->n, (int32_t)P->q, (int32_t)P->q_bits, (int32_t)P->q_inv, (int32_t)P->q_norm, (int32_t)P->k, (int32_t)P->eta,
                        (int32_t)P->d_u, (int32_t)P->d_v, (int32_t)P->d_t, set, 0, 0, 0, 0, 0 };
    snprintf(name, sizeof name, "%s_params", tag); rec_i32(f, name, par, 16);
    const SINT32 *priv = (const SINT32 *)sc->privkey->key, *pub = (const SINT32 *)sc->pubkey->key;
    const SINT32 *t_ntt = pub + k * n;
    const UINT8 *rho = (const UINT8 *)(t_ntt + k * n);
    snprintf(name, sizeof name, "%s_s_ntt", tag); rec_i32(f, name, priv, (size_t)k * n);
    snprintf(name, sizeof name, "%s_t_ntt", tag); rec_i32(f, name, t_ntt, (size_t)k * n);
    snprintf(name, sizeof name, "%s_rho", tag); rec_u8(f, name, rho, 32);
    int32_t *G = malloc(sizeof(int32_t) * (size_t)k * k * n);
    prng_ctx_t *cs = create_csprng(sc, rho, 32);
    for (int i = 0; i < k * k; i++) uniform_random_ring_q_csprng(cs, G + (size_t)i * n, n, P->q, P->q_bits);
    prng_destroy(cs);
    snprintf(name, sizeof name, "%s_Agen", tag); rec_i32(f, name, G, (size_t)k * k * n);
    free(G);
    const int nbytes = (2 * k + 1) * n * eta / 4;
    int32_t *u = malloc(sizeof(int32_t) * (size_t)(k + 1) * n * 2), *v = u + k * n, *u2 = v + n, *v2 = u2 + k * n;
    int32_t *heap = calloc((size_t)(8 * k + 8) * n, sizeof(int32_t));
    sm_state = 0x4b79626572ULL + set;
    for (int mi = 0; mi < nmsg; mi++) {
        uint8_t m[32], r[32], noise[8192], m2[32];
        for (int i = 0; i < 32; i++) { m[i] = (uint8_t)splitmix64(); r[i] = (uint8_t)splitmix64(); }
        if (mi == 0) memset(m, 0, 32);
        if (mi == 1) memset(m, 0xFF, 32);
        prng_ctx_t *tw = create_csprng(sc, r, 32);
        for (int off = 0; off < nbytes; off += eta * 64) prng_mem(tw, noise + off, eta * 64);
        prng_destroy(tw);
        if (SC_FUNC_SUCCESS != kyber_cpa_enc(sc, 1, u, v, t_ntt, 1, rho, n, k, m, r, heap)) return 1;
        snprintf(name, sizeof name, "%s_m%d", tag, mi); rec_u8(f, name, m, 32);
        snprintf(name, sizeof name, "%s_noise%d", tag, mi); rec_u8(f, name, noise, nbytes);
        snprintf(name, sizeof name, "%s_u%d", tag, mi); rec_i32(f, name, u, (size_t)k * n);
        snprintf(name, sizeof name, "%s_v%d", tag, mi); rec_i32(f, name, v, n);
        memcpy(u2, u, sizeof(int32_t) * (size_t)k * n); memcpy(v2, v, sizeof(int32_t) * n);
        if (SC_FUNC_SUCCESS != kyber_cpa_dec(sc, u2, v2, 1, priv, n, k, m2)) return 1;
        snprintf(name, sizeof name, "%s_dec%d", tag, mi); rec_u8(f, name, m2, 32);
    }
    /* KEM level (kyber_kem.c:855-1040): the context's CSPRNG is replaced by a seeded one so that m = its first prng_mem(32)
     * is known (read from a twin); ciphertext and shared key as the reference packs them, and what decapsulation returns */
    snprintf(name, sizeof name, "%s_t", tag); rec_i32(f, name, pub, (size_t)k * n);
    snprintf(name, sizeof name, "%s_z", tag); rec_u8(f, name, (const UINT8 *)(priv + 3 * (size_t)k * n), 32);
    prng_ctx_t *orig = sc->prng_ctx[0];
    for (int mi = 0; mi < nmsg; mi++) {
        uint8_t seed[32], m[32];
        for (int i = 0; i < 32; i++) seed[i] = (uint8_t)splitmix64();
        prng_ctx_t *mine = seeded_prng(seed, 32), *twin = seeded_prng(seed, 32);
        sc->prng_ctx[0] = mine;
        prng_mem(twin, m, 32);
        UINT8 *ct = NULL, *key = NULL, *key2 = NULL; size_t ctlen = 0, keylen = 0, key2len = 0;
        if (SC_FUNC_SUCCESS != safecrypto_encapsulation(sc, &ct, &ctlen, &key, &keylen)) return 1;
        if (SC_FUNC_SUCCESS != safecrypto_decapsulation(sc, ct, ctlen, &key2, &key2len)) return 1;
        snprintf(name, sizeof name, "%s_kem_m%d", tag, mi); rec_u8(f, name, m, 32);
        snprintf(name, sizeof name, "%s_kem_ct%d", tag, mi); rec_u8(f, name, ct, ctlen);
        snprintf(name, sizeof name, "%s_kem_key%d", tag, mi); rec_u8(f, name, key, keylen);
        snprintf(name, sizeof name, "%s_kem_dec%d", tag, mi); rec_u8(f, name, key2, key2len);
        ct[ctlen / 2] ^= 0x20;                                   /* a damaged ciphertext still decapsulates - to another key */
        free(key2); key2 = NULL; key2len = 0;
        if (SC_FUNC_SUCCESS != safecrypto_decapsulation(sc, ct, ctlen, &key2, &key2len)) return 1;
        snprintf(name, sizeof name, "%s_kem_decbad%d", tag, mi); rec_u8(f, name, key2, key2len);
        free(ct); free(key); free(key2);
        sc->prng_ctx[0] = orig;
        prng_destroy(mine); prng_destroy(twin);
    }
    free(u); free(heap);
    safecrypto_destroy(sc);
    /* key generation (kyber_kem.c:575-670, module_lwe.c:1189-1266) from a seeded context CSPRNG: everything the key pair holds */
    for (int kg = 0; kg < 2; kg++) {
        uint8_t seed[32];
        for (int i = 0; i < 32; i++) seed[i] = (uint8_t)splitmix64();
        safecrypto_t *s2 = create_fixed(SC_SCHEME_KEM_KYBER, set, flags);
        if (!s2) return 1;
        prng_ctx_t *o2 = s2->prng_ctx[0], *mine = seeded_prng(seed, 32);
        s2->prng_ctx[0] = mine;
        if (SC_FUNC_SUCCESS != safecrypto_keygen(s2)) return 1;
        const SINT32 *pv = (const SINT32 *)s2->privkey->key, *pb = (const SINT32 *)s2->pubkey->key;
        snprintf(name, sizeof name, "%s_kg_seed%d", tag, kg); rec_u8(f, name, seed, 32);
        snprintf(name, sizeof name, "%s_kg_s_ntt%d", tag, kg); rec_i32(f, name, pv, (size_t)k * n);
        snprintf(name, sizeof name, "%s_kg_t%d", tag, kg); rec_i32(f, name, pb, (size_t)k * n);
        snprintf(name, sizeof name, "%s_kg_t_ntt%d", tag, kg); rec_i32(f, name, pb + (size_t)k * n, (size_t)k * n);
        snprintf(name, sizeof name, "%s_kg_rho%d", tag, kg); rec_u8(f, name, (const UINT8 *)(pb + 2 * (size_t)k * n), 32);
        snprintf(name, sizeof name, "%s_kg_z%d", tag, kg); rec_u8(f, name, (const UINT8 *)(pv + 3 * (size_t)k * n), 32);
        int same = !memcmp(pv + (size_t)k * n, pb, sizeof(SINT32) * 2 * (size_t)k * n) && !memcmp((const UINT8 *)(pv + 3 * (size_t)k * n) + 32, pb + 2 * (size_t)k * n, 32);
        if (!same) return 1;                                      /* the private key repeats t | NTT(t) | rho */
        s2->prng_ctx[0] = o2;
        prng_destroy(mine);
        safecrypto_destroy(s2);
    }
    return 0;
}

/* the CTR-DRBG behind create_csprng (module_lwe.c:914-940; ctr_drbg.c, prng.c:95-143,981-993,1050-1107): byte streams of
 * prng_mem calls of assorted lengths and the word stream of prng_32 / prng_64, for seeds of several lengths */
static int gen_drbg(const char *dir)
{
    char path[512], name[96];
    snprintf(path, sizeof path, "%s/drbg_ref.txt", dir);
    FILE *f = fopen(path, "w");
    if (!f) return 1;
    static const UINT8 nonce[16] = "dilithiumcrystal";
    const int lens[5] = { 32, 32, 48, 64, 7 };
    const int32_t calls[6] = { 256, 100, 64, 1, 512, 1024 + 37 };
    rec_i32(f, "drbg_mem_calls", calls, 6);
    sm_state = 0xd4b6ULL;
    for (int c = 0; c < 5; c++) {
        uint8_t seed[64];
        for (int i = 0; i < lens[c]; i++) seed[i] = (uint8_t)splitmix64();
        if (c == 0) memset(seed, 0, 32);
        snprintf(name, sizeof name, "drbg%d_seed", c); rec_u8(f, name, seed, lens[c]);
        for (int variant = 0; variant < 2; variant++) {
            prng_ctx_t *p = prng_create(SC_ENTROPY_USER_PROVIDED, SC_PRNG_AES_CTR_DRBG, SC_PRNG_THREADING_NONE, 0x01000000);
            prng_set_entropy(p, seed, lens[c]);
            prng_init(p, nonce, 16);
            if (variant == 0) {
                uint8_t buf[4096]; size_t off = 0;
                for (int k = 0; k < 6; k++) { prng_mem(p, buf + off, calls[k]); off += calls[k]; }
                snprintf(name, sizeof name, "drbg%d_mem", c); rec_u8(f, name, buf, off);
            } else if (c == 1 || c == 4) {
                int32_t w[4200 + 16];
                for (int k = 0; k < 4200; k++) w[k] = (int32_t)prng_32(p);         /* crosses the 4096-word pool refill */
                for (int k = 0; k < 8; k++) { uint64_t x = prng_64(p); w[4200 + 2 * k] = (int32_t)(x >> 32); w[4201 + 2 * k] = (int32_t)x; }
                snprintf(name, sizeof name, "drbg%d_words", c); rec_i32(f, name, w, 4216);
            }
            prng_destroy(p);
        }
    }
    fclose(f);
    return 0;
}

/* ------------------------------------------- module-LWE, XOF-driven variants ---- */
#include "utils/crypto/xof.h"
#include "utils/arith/module_lwe.h"

/* The XOF-driven forms of the module-LWE helpers (src/utils/arith/module_lwe.c:330-367 uniform_rand_sample_xof / _small_xof, :449-512
 * binomial_rand_sample_xof, :550-560 uniform_random_ring_q_xof, :751-912 create_rand_product_32_xof / _16_xof).  No scheme of the library as it is
 * compiled reaches them (dilithium_params.h / kyber_kem_params.h define the *_USE_CSPRNG_SAM switches unconditionally), but the functions themselves are
 * compiled and exported, so they are pinned here at the function level: called directly, on a XOF prepared the way the scheme code would
 * (xof_init, xof_absorb(rho), xof_final - dilithium.c:866-870). */
static int gen_mlwe_xof(const char *dir)
{
    char path[512], name[96];
    snprintf(path, sizeof path, "%s/mlwe_xof_ref.txt", dir);
    FILE *f = fopen(path, "w");
    if (!f) return 1;
    const utils_arith_ntt_t *sc_ntt = utils_arith_ntt(SC_NTT_FLOATING_POINT);
    const utils_arith_poly_t *sc_poly = utils_arith_poly();
    /* q, q_bits, k, l, 16-bit tables?, SHAKE width, seed bytes */
    const struct { int32_t q, q_bits; int k, l, is16, shake, seedlen; } cs[5] = {
        { 8380417, 23, 6, 5, 0, 128, 32 },          /* Dilithium-III shape */
        { 8380417, 23, 3, 2, 0, 128, 32 },          /* Dilithium set 0 */
        { 7681,    13, 3, 3, 1, 128, 32 },          /* Kyber-768 shape, 16-bit tables */
        { 8380417, 23, 4, 3, 0, 256, 48 },          /* the other SHAKE width, a longer seed */
        { 7681,    13, 2, 2, 1, 256, 7 } };         /* a short seed */
    const int n = 256;
    sm_state = 0x0f0f5eedULL;
    for (int c = 0; c < 5; c++) {
        const int k = cs[c].k, l = cs[c].l, q = cs[c].q, q_bits = cs[c].q_bits;
        int32_t par[8] = { q, q_bits, k, l, cs[c].is16, cs[c].shake, cs[c].seedlen, n };
        snprintf(name, sizeof name, "xof%d_params", c); rec_i32(f, name, par, 8);
        uint8_t seed[64];
        for (int i = 0; i < cs[c].seedlen; i++) seed[i] = (uint8_t)splitmix64();
        snprintf(name, sizeof name, "xof%d_seed", c); rec_u8(f, name, seed, cs[c].seedlen);
        utils_crypto_xof_t *xof = utils_crypto_xof_create(cs[c].shake == 128 ? SC_XOF_SHAKE128 : SC_XOF_SHAKE256);
        if (!xof) { fclose(f); return 1; }
        ntt_params_t ntt; init_reduce(&ntt, n, q);
        int32_t *ring = malloc(sizeof(int32_t) * n * k * l), *y = malloc(sizeof(int32_t) * n * (k > l ? k : l)), *t = malloc(sizeof(int32_t) * n * (k > l ? k : l));
        int32_t *cbuf = malloc(sizeof(int32_t) * n * 2), *temp = malloc(sizeof(int32_t) * n * (k > l ? k : l));
        /* the matrix entries in generation order: k*l consecutive uniform_random_ring_q_xof calls on the one stream */
        xof_init(xof); xof_absorb(xof, seed, cs[c].seedlen); xof_final(xof);
        for (int p = 0; p < k * l; p++) uniform_random_ring_q_xof(xof, ring + p * n, n, q, q_bits);
        snprintf(name, sizeof name, "xof%d_ring", c); rec_i32(f, name, ring, (size_t)n * k * l);
        /* t = A y (transpose 0: y has l rows, t has k) and t = A^T y (transpose 1: the loops of :776-800 draw the matrix column by column; y has l rows, t k) */
        for (int tr = 0; tr < 2; tr++) {
            for (int i = 0; i < n * l; i++) y[i] = (c & 1) ? (int32_t)(splitmix64() % 11) - 5 : (int32_t)(splitmix64() % (uint64_t)q);
            snprintf(name, sizeof name, "xof%d_y%d", c, tr); rec_i32(f, name, y, (size_t)n * l);
            memset(t, 0, sizeof(int32_t) * n * (k > l ? k : l));
            xof_init(xof); xof_absorb(xof, seed, cs[c].seedlen); xof_final(xof);
            if (cs[c].is16)
                create_rand_product_16_xof(xof, q, q_bits, t, y, n, k, l, cbuf, temp, 0, tr, w7681_n256, r7681_n256, sc_poly, sc_ntt, &ntt);
            else
                create_rand_product_32_xof(xof, q, q_bits, t, y, n, k, l, cbuf, temp, 0, tr, w8380417_n256, r8380417_n256, sc_poly, sc_ntt, &ntt);
            snprintf(name, sizeof name, "xof%d_t%d", c, tr); rec_i32(f, name, t, (size_t)n * k);
        }
        /* the small-coefficient samplers on the same seeded stream, one after the other (the stream position carries over): the output buffer is pre-filled
         * with a marker so that what a call does NOT write shows (uniform_rand_sample_small_xof writes floor(m n / 1024) * 1024 values, :355;
         * binomial_rand_sample_xof restarts at s for every 1024-value block, :452-456) */
        const int m = k > l ? k : l, eta = (q == 7681) ? 4 : 5, bits = 3;
        int32_t *sm = malloc(sizeof(int32_t) * n * 8);
        xof_init(xof); xof_absorb(xof, seed, cs[c].seedlen); xof_final(xof);
        for (int i = 0; i < n * 8; i++) sm[i] = 99;
        uniform_rand_sample_xof(xof, q, eta, bits, sm, n, m);
        snprintf(name, sizeof name, "xof%d_unif", c); rec_i32(f, name, sm, (size_t)n * m);
        for (int i = 0; i < n * 8; i++) sm[i] = 99;
        uniform_rand_sample_small_xof(xof, q, eta, bits, sm, n, 5);                       /* 1280 values asked for */
        snprintf(name, sizeof name, "xof%d_small5", c); rec_i32(f, name, sm, (size_t)n * 5);
        for (int i = 0; i < n * 8; i++) sm[i] = 99;
        uniform_rand_sample_small_xof(xof, q, eta, bits, sm, n, 8);                       /* 2048 */
        snprintf(name, sizeof name, "xof%d_small8", c); rec_i32(f, name, sm, (size_t)n * 8);
        for (int i = 0; i < n * 8; i++) sm[i] = 99;
        binomial_rand_sample_xof(xof, q, eta > 4 ? 4 : eta, sm, n, 4);                     /* one block */
        snprintf(name, sizeof name, "xof%d_binomial4", c); rec_i32(f, name, sm, (size_t)n * 4);
        for (int i = 0; i < n * 8; i++) sm[i] = 99;
        binomial_rand_sample_xof(xof, q, 3, sm, n, 8);                                     /* two blocks */
        snprintf(name, sizeof name, "xof%d_binomial8", c); rec_i32(f, name, sm, (size_t)n * 8);
        free(sm); free(ring); free(y); free(t); free(cbuf); free(temp);
        utils_crypto_xof_destroy(xof);
    }
    fclose(f);
    return 0;
}

static int gen_kyber(const char *dir)
{
    char path[512];
    snprintf(path, sizeof path, "%s/kyber_ref.txt", dir);
    FILE *f = fopen(path, "w");
    if (!f) return 1;
    int rc = kyber_case(f, 1, 6);
    rc |= kyber_case(f, 2, 2);
    rc |= kyber_case(f, 0, 2);                                         /* k = 2 */
    fclose(f);
    return rc;
}

/* ------------------------------------------------------------- ring-TESLA ---- */
#include "schemes/sig/ring_tesla/ring_tesla.h"
#include "schemes/sig/ring_tesla/ring_tesla_params.h"

/* ring_tesla_sign (src/schemes/sig/ring_tesla/ring_tesla.c:752-910) draws n prng_32 words per attempt from sc->prng_ctx[0];
 * same tape technique as BLISS.  The public polynomials a1, a2 are constants of the reference (ring_tesla_params.h:85-88) and
 * travel as data. */
static int tesla_case(FILE *f, int set, int nmsg)
{
    UINT32 flags[2] = { SC_FLAG_NONE, SC_FLAG_NONE };
    char name[96], tag[32];
    snprintf(tag, sizeof tag, "tesla%d", set);
    safecrypto_t *sc = create_fixed(SC_SCHEME_SIG_RING_TESLA, set, flags);
    if (!sc) return 1;
    if (SC_FUNC_SUCCESS != safecrypto_keygen(sc)) return 1;
    const ring_tesla_set_t *P = sc->ring_tesla->params;
    const int n = P->n;
    int32_t par[12] = { (int32_t)P->n, (int32_t)P->n_bits, (int32_t)P->q, (int32_t)P->q_bits, (int32_t)P->b, (int32_t)P->b_bits, (int32_t)P->omega, (int32_t)P->d,
                        (int32_t)P->bound, (int32_t)P->u, set, 0 };
    snprintf(name, sizeof name, "%s_params", tag); rec_i32(f, name, par, 12);
    snprintf(name, sizeof name, "%s_a1", tag); rec_i32(f, name, set == 0 ? a1_0 : a1_1, n);
    snprintf(name, sizeof name, "%s_a2", tag); rec_i32(f, name, set == 0 ? a2_0 : a2_1, n);
    snprintf(name, sizeof name, "%s_s", tag); rec_i16(f, name, (const int16_t *)sc->privkey->key, n);
    snprintf(name, sizeof name, "%s_e1", tag); rec_i16(f, name, (const int16_t *)sc->privkey->key + n, n);
    snprintf(name, sizeof name, "%s_e2", tag); rec_i16(f, name, (const int16_t *)sc->privkey->key + 2 * n, n);
    snprintf(name, sizeof name, "%s_t1", tag); rec_i32(f, name, (const int32_t *)sc->pubkey->key, n);
    snprintf(name, sizeof name, "%s_t2", tag); rec_i32(f, name, (const int32_t *)sc->pubkey->key + n, n);
    int32_t *trials = malloc(sizeof(int32_t) * nmsg), *verdicts = malloc(sizeof(int32_t) * nmsg * 2);
    prng_ctx_t *orig = sc->prng_ctx[0];
    sm_state = 0x7e51a000ULL + set;
    for (int m = 0; m < nmsg; m++) {
        uint8_t msg[150], seed[32];
        size_t mlen = (m == 0) ? 1 : (m == 1) ? 150 : 40;
        for (size_t i = 0; i < mlen; i++) msg[i] = (uint8_t)splitmix64();
        for (int i = 0; i < 32; i++) seed[i] = (uint8_t)splitmix64();
        prng_ctx_t *mine = seeded_prng(seed, 32), *twin = seeded_prng(seed, 32);
        sc->prng_ctx[0] = mine;
        uint8_t *sig = NULL; size_t siglen = 0;
        size_t before = sc->stats.sig_num_trials;
        if (SC_FUNC_SUCCESS != safecrypto_sign(sc, msg, mlen, &sig, &siglen)) return 1;
        trials[m] = (int)(sc->stats.sig_num_trials - before);
        const size_t words = (size_t)(prng_get_out_bytes(mine) / 4);
        int32_t *tape = malloc(sizeof(int32_t) * words);
        for (size_t i = 0; i < words; i++) tape[i] = (int32_t)prng_32(twin);
        snprintf(name, sizeof name, "%s_msg%d", tag, m); rec_u8(f, name, msg, mlen);
        snprintf(name, sizeof name, "%s_tape%d", tag, m); rec_i32(f, name, tape, words);
        snprintf(name, sizeof name, "%s_sig%d", tag, m); rec_u8(f, name, sig, siglen);
        verdicts[2 * m] = safecrypto_verify(sc, msg, mlen, sig, siglen);
        sig[siglen / 3] ^= 0x08;
        verdicts[2 * m + 1] = safecrypto_verify(sc, msg, mlen, sig, siglen);
        free(tape); free(sig);
        sc->prng_ctx[0] = orig;
        prng_destroy(mine); prng_destroy(twin);
    }
    snprintf(name, sizeof name, "%s_trials", tag); rec_i32(f, name, trials, nmsg);
    snprintf(name, sizeof name, "%s_verify", tag); rec_i32(f, name, verdicts, (size_t)nmsg * 2);
    free(trials); free(verdicts);
    safecrypto_destroy(sc);
    return 0;
}

static int gen_tesla(const char *dir)
{
    char path[512];
    snprintf(path, sizeof path, "%s/tesla_ref.txt", dir);
    FILE *f = fopen(path, "w");
    if (!f) return 1;
    int rc = tesla_case(f, 0, 8);
    rc |= tesla_case(f, 1, 5);
    fclose(f);
    return rc;
}

/* ---------------------------------------------------------------- RLWE-Enc ---- */
#include "schemes/enc/rlwe_enc/rlwe_enc.h"
#include "schemes/enc/rlwe_enc/rlwe_enc_params.h"

static void rlwe_swap_prng(safecrypto_t *sc, prng_ctx_t *p)
{
    sc->prng_ctx[0] = p;
    sc->sc_gauss->prng_ctx = p;
    ((cdf64_full_view_t *)sc->sc_gauss->gauss)->prng_ctx = p;
}

/* rlwe_enc_keygen / _encrypt / _decrypt (src/schemes/enc/rlwe_enc/rlwe_enc.c:425-493, 511-631, 633-707) take all their
 * randomness from sc->prng_ctx[0] (prng_32 words for a and r2, one prng_64 per Gaussian sample through the CDF sampler): same
 * tape technique as BLISS.  Key generation: n/2 + 2n + n/16 words; encryption: 6n words.  The build has
 * RLWE_ENC_ENABLE_NTT_TRANSMISSION defined (rlwe_enc.h:24): c1 travels in the NTT domain. */
static int rlwe_case(FILE *f, int set, int nkeys, int nmsg)
{
    UINT32 flags[2] = { SC_FLAG_NONE, SC_FLAG_NONE };
    char name[96], tag[32];
    snprintf(tag, sizeof tag, "rlwe%d", set);
    safecrypto_t *sc = create_fixed(SC_SCHEME_ENC_RLWE, set, flags);
    if (!sc) return 1;
    const rlwe_set_t *P = set ? &param_rlwe_enc_1 : &param_rlwe_enc_0;       /* the config struct is private to rlwe_enc.c */
    const int n = P->n;
    int32_t par[12] = { P->set, P->q, P->q_bits, P->n, P->n_bits, P->m_scale, P->o_scale_0, P->o_scale_1, 0, 0, 0, 0 };
    float fl[2] = { P->sig, 13.0f };
    memcpy(&par[8], fl, 8);                                        /* sigma and the tail cut (rlwe_enc.c:134-136) as raw float bits */
    snprintf(name, sizeof name, "%s_params", tag); rec_i32(f, name, par, 12);
    rec_u64(f, (snprintf(name, sizeof name, "%s_cdf", tag), name), (const uint64_t *)((cdf64_full_view_t *)sc->sc_gauss->gauss)->cdf,
            (size_t)((cdf64_full_view_t *)sc->sc_gauss->gauss)->cdf_size);
    prng_ctx_t *orig = sc->prng_ctx[0];
    sm_state = 0x51e0c000ULL + set;
    for (int kg = 0; kg < nkeys; kg++) {
        uint8_t seed[32];
        for (int i = 0; i < 32; i++) seed[i] = (uint8_t)splitmix64();
        prng_ctx_t *mine = seeded_prng(seed, 32), *twin = seeded_prng(seed, 32);
        rlwe_swap_prng(sc, mine);
        if (SC_FUNC_SUCCESS != safecrypto_keygen(sc)) return 1;
        const size_t words = (size_t)(prng_get_out_bytes(mine) / 4);
        int32_t *tape = malloc(sizeof(int32_t) * words);
        for (size_t i = 0; i < words; i++) tape[i] = (int32_t)prng_32(twin);
        snprintf(name, sizeof name, "%s_kg_tape%d", tag, kg); rec_i32(f, name, tape, words);
        snprintf(name, sizeof name, "%s_kg_a%d", tag, kg); rec_i16(f, name, (const int16_t *)sc->pubkey->key, n);
        snprintf(name, sizeof name, "%s_kg_p%d", tag, kg); rec_i16(f, name, (const int16_t *)sc->pubkey->key + n, n);
        snprintf(name, sizeof name, "%s_kg_r2_%d", tag, kg); rec_i16(f, name, (const int16_t *)sc->privkey->key, n);
        free(tape);
        rlwe_swap_prng(sc, orig);
        prng_destroy(mine); prng_destroy(twin);
    }
    /* the last generated key pair stays loaded for the encryption cases */
    for (int m = 0; m < nmsg; m++) {
        uint8_t msg[128], seed[32];
        for (int i = 0; i < n / 8; i++) msg[i] = (m == 0) ? 0x00 : (m == 1) ? 0xff : (uint8_t)splitmix64();
        for (int i = 0; i < 32; i++) seed[i] = (uint8_t)splitmix64();
        prng_ctx_t *mine = seeded_prng(seed, 32), *twin = seeded_prng(seed, 32);
        rlwe_swap_prng(sc, mine);
        uint8_t *ct = NULL, *dec = NULL; size_t ctlen = 0, declen = 0;
        if (SC_FUNC_SUCCESS != safecrypto_public_encrypt(sc, (size_t)n / 8, msg, &ctlen, &ct)) return 1;
        const size_t words = (size_t)(prng_get_out_bytes(mine) / 4);
        int32_t *tape = malloc(sizeof(int32_t) * words);
        for (size_t i = 0; i < words; i++) tape[i] = (int32_t)prng_32(twin);
        if (SC_FUNC_SUCCESS != safecrypto_private_decrypt(sc, ctlen, ct, &declen, &dec)) return 1;
        snprintf(name, sizeof name, "%s_msg%d", tag, m); rec_u8(f, name, msg, (size_t)n / 8);
        snprintf(name, sizeof name, "%s_tape%d", tag, m); rec_i32(f, name, tape, words);
        snprintf(name, sizeof name, "%s_ct%d", tag, m); rec_u8(f, name, ct, ctlen);
        snprintf(name, sizeof name, "%s_dec%d", tag, m); rec_u8(f, name, dec, declen);
        free(dec); dec = NULL; declen = 0;                /* a non-zero length makes the packer expect a caller buffer (packer.c:142,153-157) */
        /* decryption is a plain function of the bytes: a ciphertext of random bytes exercises coefficients >= q */
        for (size_t i = 0; i < ctlen; i++) ct[i] = (uint8_t)splitmix64();
        if (SC_FUNC_SUCCESS != safecrypto_private_decrypt(sc, ctlen, ct, &declen, &dec)) return 1;
        snprintf(name, sizeof name, "%s_randct%d", tag, m); rec_u8(f, name, ct, ctlen);
        snprintf(name, sizeof name, "%s_randdec%d", tag, m); rec_u8(f, name, dec, declen);
        free(tape); free(ct); free(dec);
        rlwe_swap_prng(sc, orig);
        prng_destroy(mine); prng_destroy(twin);
    }
    safecrypto_destroy(sc);
    return 0;
}

static int gen_rlwe(const char *dir)
{
    char path[512];
    snprintf(path, sizeof path, "%s/rlwe_ref.txt", dir);
    FILE *f = fopen(path, "w");
    if (!f) return 1;
    int rc = rlwe_case(f, 0, 3, 6);
    rc |= rlwe_case(f, 1, 2, 4);
    fclose(f);
    return rc;
}

/* ----------------------------------------------------------------- ENS-KEM ---- */
#include "schemes/kem/ens/ens_kem.h"
#include "schemes/kem/ens/ens_kem_params.h"

/* ens_kem_encapsulation / _decapsulation (src/schemes/kem/ens/ens_kem.c:785-865, 867-960) under a key pair made by the reference's
 * keygen (data): public h and private g are SINT16 in the NTT domain, centred; 1/g mod 2 is bit-reversed and packed (:191-211).
 * Encapsulation draws r and e through poly_32_uniform_rand with the set's 12 coefficient counts from sc->prng_ctx[0]: tape technique. */
static int ens_case(FILE *f, int set, int nmsg)
{
    UINT32 flags[2] = { SC_FLAG_NONE, SC_FLAG_NONE };
    char name[96], tag[32];
    snprintf(tag, sizeof tag, "ens%d", set);
    safecrypto_t *sc = create_fixed(SC_SCHEME_KEM_ENS, set, flags);
    if (!sc) return 1;
    if (SC_FUNC_SUCCESS != safecrypto_keygen(sc)) return 1;
    const ens_kem_set_t *P = set == 0 ? &param_ens_kem_0 : set == 1 ? &param_ens_kem_1 : set == 2 ? &param_ens_kem_2 : &param_ens_kem_3;
    const int n = P->n;
    int32_t par[20] = { P->set, P->q, P->q_bits, P->n, P->n_bits, 0, 0 };
    for (int i = 0; i < 13; i++) par[7 + i] = P->coeff_rnd[i];
    snprintf(name, sizeof name, "%s_params", tag); rec_i32(f, name, par, 20);
    snprintf(name, sizeof name, "%s_h", tag); rec_i16(f, name, (const int16_t *)sc->pubkey->key, n);
    snprintf(name, sizeof name, "%s_g", tag); rec_i16(f, name, (const int16_t *)sc->privkey->key, n);
    snprintf(name, sizeof name, "%s_inv_g", tag); rec_i32(f, name, (const int32_t *)((const int16_t *)sc->privkey->key + n), (size_t)n / 32);
    prng_ctx_t *orig = sc->prng_ctx[0];
    sm_state = 0xe25c000ULL + set;
    for (int m = 0; m < nmsg; m++) {
        uint8_t seed[32];
        for (int i = 0; i < 32; i++) seed[i] = (uint8_t)splitmix64();
        prng_ctx_t *mine = seeded_prng(seed, 32), *twin = seeded_prng(seed, 32);
        sc->prng_ctx[0] = mine;
        uint8_t *ct = NULL, *key = NULL, *k2 = NULL; size_t ctlen = 0, klen = 0, k2len = 0;
        if (SC_FUNC_SUCCESS != safecrypto_encapsulation(sc, &ct, &ctlen, &key, &klen)) return 1;
        const size_t words = (size_t)(prng_get_out_bytes(mine) / 4);
        int32_t *tape = malloc(sizeof(int32_t) * words);
        for (size_t i = 0; i < words; i++) tape[i] = (int32_t)prng_32(twin);
        if (SC_FUNC_SUCCESS != safecrypto_decapsulation(sc, ct, ctlen, &k2, &k2len)) return 1;
        snprintf(name, sizeof name, "%s_tape%d", tag, m); rec_i32(f, name, tape, words);
        snprintf(name, sizeof name, "%s_ct%d", tag, m); rec_u8(f, name, ct, ctlen);
        snprintf(name, sizeof name, "%s_key%d", tag, m); rec_u8(f, name, key, klen);
        snprintf(name, sizeof name, "%s_dec%d", tag, m); rec_u8(f, name, k2, k2len);
        free(k2); k2 = NULL; k2len = 0;
        for (size_t i = 0; i < ctlen; i++) ct[i] = (uint8_t)splitmix64();         /* decapsulation is a plain function of the bytes */
        if (SC_FUNC_SUCCESS != safecrypto_decapsulation(sc, ct, ctlen, &k2, &k2len)) return 1;
        snprintf(name, sizeof name, "%s_randct%d", tag, m); rec_u8(f, name, ct, ctlen);
        snprintf(name, sizeof name, "%s_randdec%d", tag, m); rec_u8(f, name, k2, k2len);
        free(tape); free(ct); free(key); free(k2);
        sc->prng_ctx[0] = orig;
        prng_destroy(mine); prng_destroy(twin);
    }
    safecrypto_destroy(sc);
    return 0;
}

static int gen_ens(const char *dir)
{
    char path[512];
    snprintf(path, sizeof path, "%s/ens_ref.txt", dir);
    FILE *f = fopen(path, "w");
    if (!f) return 1;
    int rc = ens_case(f, 0, 5);
    rc |= ens_case(f, 2, 3);
    rc |= ens_case(f, 1, 3);                                           /* the other coefficient distributions of each ring */
    rc |= ens_case(f, 3, 3);
    fclose(f);
    return rc;
}

/* ----------------------------------------------------------------- DLP-IBE ---- */
#include "schemes/ibe/dlp/dlp_ibe.h"
#include "schemes/ibe/dlp/dlp_ibe_params.h"

static void rec_f64(FILE *f, const char *name, const double *v, size_t n);     /* defined with the Falcon records below */

/* dlp_ibe_extract (src/schemes/ibe/dlp/dlp_ibe.c:874-1043, enhanced extract): H(id) by SHAKE128 (:805-816), inv_ntt_32_32_large,
 * gaussian_sample_with_tree over the master tree (load_skey of the master key, :394-425), center_32, 17-bit signed raw coding.
 * The master key pair is the reference's (data: the tree as doubles); the sampler draws from sc->prng_ctx[0]: tape technique. */
static int dlp_case(FILE *f, int set, int nids)
{
    UINT32 flags[2] = { SC_FLAG_NONE, SC_FLAG_NONE };
    char name[96], tag[32];
    snprintf(tag, sizeof tag, "dlp%d", set);
    safecrypto_t *sc = create_fixed(SC_SCHEME_IBE_DLP, set, flags);
    if (!sc) return 1;
    FILE *devnull = fopen("/dev/null", "w");
    FILE *saved = stdout, *saved_err = stderr;
    stdout = devnull; stderr = devnull;                               /* the key generator prints its polynomials */
    const int krc = safecrypto_keygen(sc);
    stdout = saved; stderr = saved_err;
    if (SC_FUNC_SUCCESS != krc) return 1;
    const dlp_ibe_set_t *P = sc->dlp_ibe->params;
    const int n = P->n, logn = P->n_bits;
    uint32_t ub = (uint32_t)(6 * 1.17 * sqrt((DOUBLE)P->q / (DOUBLE)(2 * n)));
    int ubits = 0; while ((1u << ubits) < ub) ubits++;             /* sc_ceil_log2 */
    ubits += 8;
    int32_t par[8] = { (int32_t)P->n, (int32_t)P->q, (int32_t)P->q_bits, (int32_t)P->n_bits, set, ubits, (int32_t)P->m_scale, (int32_t)P->l };
    snprintf(name, sizeof name, "%s_params", tag); rec_i32(f, name, par, 8);
    snprintf(name, sizeof name, "%s_tree", tag); rec_f64(f, name, sc->dlp_ibe->master_tree, (size_t)(logn + 5) * n);
    snprintf(name, sizeof name, "%s_h_ntt", tag); rec_i32(f, name, (const int32_t *)sc->pubkey->key + n, n);
    prng_ctx_t *orig = sc->prng_ctx[0];
    sm_state = 0xd1b000ULL + set;
    for (int m = 0; m < nids; m++) {
        uint8_t id[80], seed[32];
        size_t idlen = (m == 0) ? 1 : (m == 1) ? 80 : 17;
        for (size_t i = 0; i < idlen; i++) id[i] = (uint8_t)splitmix64();
        for (int i = 0; i < 32; i++) seed[i] = (uint8_t)splitmix64();
        prng_ctx_t *mine = seeded_prng(seed, 32), *twin = seeded_prng(seed, 32);
        sc->prng_ctx[0] = mine;
        uint8_t *sk = NULL; size_t sklen = 0;
        if (SC_FUNC_SUCCESS != safecrypto_ibe_extract(sc, idlen, id, &sklen, &sk)) return 1;
        const size_t words = (size_t)(prng_get_out_bytes(mine) / 4);
        int32_t *tape = malloc(sizeof(int32_t) * words);
        for (size_t i = 0; i < words; i++) tape[i] = (int32_t)prng_32(twin);
        snprintf(name, sizeof name, "%s_id%d", tag, m); rec_u8(f, name, id, idlen);
        snprintf(name, sizeof name, "%s_tape%d", tag, m); rec_i32(f, name, tape, words);
        snprintf(name, sizeof name, "%s_sk%d", tag, m); rec_u8(f, name, sk, sklen);
        free(tape);
        sc->prng_ctx[0] = orig;
        prng_destroy(mine); prng_destroy(twin);
        /* dlp_ibe_encrypt to this identity (dlp_ibe.c:1140-1370: n/32 prng_32 words for the key bits, then n prng_var(6) draws for e1, e2, e3)
         * and dlp_ibe_decrypt (:1372-1520) with the user key just extracted (loaded through dlp_ibe_secret_key, which keeps its NTT) */
        if (SC_FUNC_SUCCESS != safecrypto_secret_key(sc, sklen, sk)) return 1;
        snprintf(name, sizeof name, "%s_uk_ntt%d", tag, m); rec_i32(f, name, sc->dlp_ibe->user_key + n, n);
        free(sk);
        uint8_t msg[128];
        for (int i = 0; i < n / 8; i++) msg[i] = (uint8_t)splitmix64();
        for (int i = 0; i < 32; i++) seed[i] = (uint8_t)splitmix64();
        mine = seeded_prng(seed, 32); twin = seeded_prng(seed, 32);
        sc->prng_ctx[0] = mine;
        uint8_t *ct = NULL, *dec = NULL; size_t ctlen = 0, declen = 0;
        if (SC_FUNC_SUCCESS != safecrypto_ibe_public_encrypt(sc, idlen, id, (size_t)n / 8, msg, &ctlen, &ct)) return 1;
        /* prng_var keeps unread bits in its reservoir: the twin is advanced by whole words, one more than the byte count says when bits are left */
        const size_t ewords = (size_t)((prng_get_out_bytes(mine) + 3) / 4) + 1;
        int32_t *etape = malloc(sizeof(int32_t) * ewords);
        for (size_t i = 0; i < ewords; i++) etape[i] = (int32_t)prng_32(twin);
        if (SC_FUNC_SUCCESS != safecrypto_private_decrypt(sc, ctlen, ct, &declen, &dec)) return 1;
        snprintf(name, sizeof name, "%s_msg%d", tag, m); rec_u8(f, name, msg, (size_t)n / 8);
        snprintf(name, sizeof name, "%s_etape%d", tag, m); rec_i32(f, name, etape, ewords);
        snprintf(name, sizeof name, "%s_ct%d", tag, m); rec_u8(f, name, ct, ctlen);
        snprintf(name, sizeof name, "%s_dec%d", tag, m); rec_u8(f, name, dec, declen);
        free(etape); free(ct); free(dec);
        sc->prng_ctx[0] = orig;
        prng_destroy(mine); prng_destroy(twin);
    }
    safecrypto_destroy(sc);
    fclose(devnull);
    return 0;
}

static int gen_dlp(const char *dir)
{
    char path[512];
    snprintf(path, sizeof path, "%s/dlp_ref.txt", dir);
    FILE *f = fopen(path, "w");
    if (!f) return 1;
    int rc = dlp_case(f, 0, 4);
    rc |= dlp_case(f, 1, 4);                                           /* n = 1024 */
    fclose(f);
    return rc;
}

/* ------------------------------------------------------- ENS / DLP signatures ---- */
#include "schemes/sig/ens_dlp/ens_dlp_sig.h"
#include "schemes/sig/ens_dlp/ens_dlp_sig_params.h"

/* ens_dlp_sig_sign / ens_dlp_sig_verify (src/schemes/sig/ens_dlp/ens_dlp_sig.c:1171-1315, 1564-1665, without message recovery): the fixture
 * carries the reference's key pair (h as SINT16 in the coefficient domain; f | g | F | G), what create_gpv_matrices derived from it, messages, the
 * PRNG word tape each signature consumed, the signatures, and the reference's verdicts on them and on tampered copies. */
static int ensdlp_case(FILE *f, sc_scheme_e scheme, const char *tag, int set, const ens_dlp_sig_set_t *P, int nmsg)
{
    UINT32 flags[2] = { SC_FLAG_NONE, SC_FLAG_NONE };
    char name[96];
    safecrypto_t *sc = create_fixed(scheme, set, flags);
    if (!sc) return 1;
    FILE *devnull = fopen("/dev/null", "w"), *saved = stdout, *saved_err = stderr;
    stdout = devnull; stderr = devnull;
    const int krc = safecrypto_keygen(sc);
    stdout = saved; stderr = saved_err;
    if (SC_FUNC_SUCCESS != krc) return 1;
    const int n = P->n;
    int32_t par[8] = { (int32_t)P->n, (int32_t)P->q, (int32_t)P->q_bits, (int32_t)P->n_bits, set, 0, 0, 0 };
    float bd = P->bd;
    memcpy(&par[5], &bd, 4);
    snprintf(name, sizeof name, "%s_params", tag); rec_i32(f, name, par, 8);
    snprintf(name, sizeof name, "%s_h", tag); rec_i16(f, name, (const int16_t *)sc->pubkey->key, n);
    /* the signer's inputs (ens_dlp_sig_sign, ens_dlp_sig.c:1171-1315): f | g | F | G, and what create_gpv_matrices (:364-421) derived from them -
     * all of b_gs_inv_norm and a strided sample of b and b_gs (4 n^2 entries each) as float / int bit patterns, to pin the restated expansion */
    snprintf(name, sizeof name, "%s_fgFG", tag); rec_i32(f, name, (const int32_t *)sc->privkey->key, (size_t)4 * n);
    snprintf(name, sizeof name, "%s_inv_norm", tag); rec_i32(f, name, (const int32_t *)sc->ens_dlp_sig->b_gs_inv_norm, (size_t)2 * n);
    {
        const size_t tot = (size_t)4 * n * n, cnt = (tot + 1020) / 1021;
        int32_t *smp = malloc(sizeof(int32_t) * cnt * 2);
        for (size_t i = 0, k = 0; i < tot; i += 1021, k++) {
            smp[k] = sc->ens_dlp_sig->b[i];
            memcpy(&smp[cnt + k], &((const float *)sc->ens_dlp_sig->b_gs)[i], 4);
        }
        snprintf(name, sizeof name, "%s_b_sample", tag); rec_i32(f, name, smp, cnt);
        snprintf(name, sizeof name, "%s_bgs_sample", tag); rec_i32(f, name, smp + cnt, cnt);
        free(smp);
    }
    int32_t *verdicts = malloc(sizeof(int32_t) * nmsg * 3);
    sm_state = 0xe25d000ULL + set + (scheme == SC_SCHEME_SIG_DLP ? 64 : 0);
    prng_ctx_t *orig = sc->prng_ctx[0];
    for (int m = 0; m < nmsg; m++) {
        uint8_t msg[150], seed[32];
        size_t mlen = (m == 0) ? 1 : (m == 1) ? 150 : 33;
        for (size_t i = 0; i < mlen; i++) msg[i] = (uint8_t)splitmix64();
        for (int i = 0; i < 32; i++) seed[i] = (uint8_t)splitmix64();
        uint8_t *sig = NULL; size_t siglen = 0;
        /* the Klein sampler draws one prng_64 per basis row from sc->prng_ctx[0] (gpv.c:2409-2492): tape technique */
        prng_ctx_t *mine = seeded_prng(seed, 32), *twin = seeded_prng(seed, 32);
        sc->prng_ctx[0] = mine;
        if (SC_FUNC_SUCCESS != safecrypto_sign(sc, msg, mlen, &sig, &siglen)) return 1;
        const size_t words = (size_t)(prng_get_out_bytes(mine) / 4);
        int32_t *tape = malloc(sizeof(int32_t) * words);
        for (size_t i = 0; i < words; i++) tape[i] = (int32_t)prng_32(twin);
        snprintf(name, sizeof name, "%s_tape%d", tag, m); rec_i32(f, name, tape, words);
        free(tape);
        sc->prng_ctx[0] = orig;
        prng_destroy(mine); prng_destroy(twin);
        snprintf(name, sizeof name, "%s_msg%d", tag, m); rec_u8(f, name, msg, mlen);
        snprintf(name, sizeof name, "%s_sig%d", tag, m); rec_u8(f, name, sig, siglen);
        verdicts[3 * m] = safecrypto_verify(sc, msg, mlen, sig, siglen);
        sig[siglen / 2] ^= 0x01;                                        /* a low bit of one coefficient: the norm barely moves */
        verdicts[3 * m + 1] = safecrypto_verify(sc, msg, mlen, sig, siglen);
        sig[siglen / 2] ^= 0x01; sig[3] ^= 0x80;                        /* a high bit: far outside the bound */
        verdicts[3 * m + 2] = safecrypto_verify(sc, msg, mlen, sig, siglen);
        free(sig);
    }
    snprintf(name, sizeof name, "%s_verify", tag); rec_i32(f, name, verdicts, (size_t)nmsg * 3);
    free(verdicts);
    safecrypto_destroy(sc);
    fclose(devnull);
    return 0;
}

static int gen_ensdlp(const char *dir)
{
    char path[512];
    snprintf(path, sizeof path, "%s/ensdlp_ref.txt", dir);
    FILE *f = fopen(path, "w");
    if (!f) return 1;
    int rc = ensdlp_case(f, SC_SCHEME_SIG_ENS, "enssig0", 0, &param_ens_sig_0, 5);
    rc |= ensdlp_case(f, SC_SCHEME_SIG_DLP, "dlpsig1", 1, &param_dlp_sig_1, 4);
    rc |= ensdlp_case(f, SC_SCHEME_SIG_DLP, "dlpsig0", 0, &param_dlp_sig_0, 4);
    rc |= ensdlp_case(f, SC_SCHEME_SIG_ENS, "enssig1", 1, &param_ens_sig_1, 3);
    fclose(f);
    return rc;
}

/* ens_dlp_sig_sign_recovery / ens_dlp_sig_verify_recovery (src/schemes/sig/ens_dlp/ens_dlp_sig.c:1319-1501, 1663-1826: SC_SCHEME_SIG_ENS_WITH_RECOVERY /
 * SC_SCHEME_SIG_DLP_WITH_RECOVERY): per message the fixture carries what went in (m), what came out of the signer (the PRNG word tape it consumed, the
 * signature s1 | s2, the remainder m2 the signer leaves in the message buffer), and what the reference's verifier makes of signature + m2: its verdict
 * and the message it rebuilds in a zeroed buffer - plus its verdicts on a tampered signature and on a tampered m2.  Message lengths sit on both sides
 * of the (n - k)(q_bits - 1) bits that travel inside the signature. */
static int ensdlp_rec_case(FILE *f, sc_scheme_e scheme, const char *tag, int set, const ens_dlp_sig_set_t *P)
{
    UINT32 flags[2] = { SC_FLAG_NONE, SC_FLAG_NONE };
    char name[96];
    safecrypto_t *sc = create_fixed(scheme, set, flags);
    if (!sc) return 1;
    FILE *devnull = fopen("/dev/null", "w"), *saved = stdout, *saved_err = stderr;
    stdout = devnull; stderr = devnull;
    const int krc = safecrypto_keygen(sc);
    stdout = saved; stderr = saved_err;
    if (SC_FUNC_SUCCESS != krc) return 1;
    const int n = P->n, k = (int)P->k_recovery;
    int32_t par[8] = { (int32_t)P->n, (int32_t)P->q, (int32_t)P->q_bits, (int32_t)P->n_bits, set, 0, k, 0 };
    float bd = P->bd;
    memcpy(&par[5], &bd, 4);
    snprintf(name, sizeof name, "%s_params", tag); rec_i32(f, name, par, 8);
    snprintf(name, sizeof name, "%s_h", tag); rec_i16(f, name, (const int16_t *)sc->pubkey->key, n);
    snprintf(name, sizeof name, "%s_fgFG", tag); rec_i32(f, name, (const int32_t *)sc->privkey->key, (size_t)4 * n);
    const size_t m1_bits = (size_t)(n - k) * (P->q_bits - 1), m1_bytes = (m1_bits + 7) >> 3;
    const size_t lens[8] = { 1, 8, 128, m1_bytes - 1, m1_bytes, m1_bytes + 1, m1_bytes + 2, m1_bytes + 77 };
    int32_t verdicts[8 * 3];
    sm_state = 0x5ec0e47ULL + set + (scheme == SC_SCHEME_SIG_DLP_WITH_RECOVERY ? 64 : 0);
    prng_ctx_t *orig = sc->prng_ctx[0];
    for (int m = 0; m < 8; m++) {
        const size_t cap = 2048;
        uint8_t *msg = calloc(cap, 1), *back = calloc(cap, 1), seed[32];
        size_t mlen = lens[m];
        for (size_t i = 0; i < mlen; i++) msg[i] = (uint8_t)splitmix64();
        for (int i = 0; i < 32; i++) seed[i] = (uint8_t)splitmix64();
        snprintf(name, sizeof name, "%s_msg%d", tag, m); rec_u8(f, name, msg, mlen);
        uint8_t *sig = NULL; size_t siglen = 0;
        prng_ctx_t *mine = seeded_prng(seed, 32), *twin = seeded_prng(seed, 32);
        sc->prng_ctx[0] = mine;
        uint8_t *mp = msg;
        if (SC_FUNC_SUCCESS != safecrypto_sign_with_recovery(sc, &mp, &mlen, &sig, &siglen)) return 1;
        if (mp != msg) return 1;
        const size_t words = (size_t)(prng_get_out_bytes(mine) / 4);
        int32_t *tape = malloc(sizeof(int32_t) * words);
        for (size_t i = 0; i < words; i++) tape[i] = (int32_t)prng_32(twin);
        snprintf(name, sizeof name, "%s_tape%d", tag, m); rec_i32(f, name, tape, words);
        free(tape);
        sc->prng_ctx[0] = orig;
        prng_destroy(mine); prng_destroy(twin);
        snprintf(name, sizeof name, "%s_sig%d", tag, m); rec_u8(f, name, sig, siglen);
        snprintf(name, sizeof name, "%s_m2_%d", tag, m); rec_u8(f, name, msg, mlen);          /* mlen is now the length of m2 (0: an empty record) */
        /* the verifier: signature + m2 in a zeroed buffer */
        size_t blen = mlen;
        memcpy(back, msg, mlen);
        uint8_t *bp = back;
        verdicts[3 * m] = safecrypto_verify_with_recovery(sc, &bp, &blen, sig, siglen);
        if (bp != back) return 1;
        snprintf(name, sizeof name, "%s_back%d", tag, m); rec_u8(f, name, back, blen);
        /* a tampered signature: one low bit of a coefficient in s2's half */
        sig[siglen - 9] ^= 0x04;
        memset(back, 0, cap); memcpy(back, msg, mlen); blen = mlen; bp = back;
        verdicts[3 * m + 1] = safecrypto_verify_with_recovery(sc, &bp, &blen, sig, siglen);
        sig[siglen - 9] ^= 0x04;
        /* a tampered remainder (where there is one) / a tampered high bit of s1 (where there is none) */
        memset(back, 0, cap); memcpy(back, msg, mlen); blen = mlen; bp = back;
        if (mlen) back[0] ^= 0x80; else sig[2] ^= 0x40;
        verdicts[3 * m + 2] = safecrypto_verify_with_recovery(sc, &bp, &blen, sig, siglen);
        free(sig); free(msg); free(back);
    }
    snprintf(name, sizeof name, "%s_verify", tag); rec_i32(f, name, verdicts, 24);
    safecrypto_destroy(sc);
    fclose(devnull);
    return 0;
}

static int gen_ensdlp_rec(const char *dir)
{
    char path[512];
    snprintf(path, sizeof path, "%s/ensdlp_rec_ref.txt", dir);
    FILE *f = fopen(path, "w");
    if (!f) return 1;
    int rc = ensdlp_rec_case(f, SC_SCHEME_SIG_DLP_WITH_RECOVERY, "dlprec0", 0, &param_dlp_sig_0);
    rc |= ensdlp_rec_case(f, SC_SCHEME_SIG_ENS_WITH_RECOVERY, "ensrec0", 0, &param_ens_sig_0);
    fclose(f);
    return rc;
}

/* ------------------------------------------------- small arithmetic entries ---- */
/* pwr_32 / invert_32 / div_32 / mul_32_sparse of the NTT table (ntt_template.c.in:1689-1775, ntt.c:381-422) and the
 * utils_arith_poly / utils_arith_vectors entries the schemes use (poly_32.c:26-191, vectors.c:25-60), run on seeded
 * inputs through the reference's own function-pointer tables. */
static int gen_ops(const char *dir)
{
    char path[512], name[96];
    snprintf(path, sizeof path, "%s/ops_ref.txt", dir);
    FILE *f = fopen(path, "w");
    if (!f) return 1;
    const utils_arith_ntt_t *ntt = utils_arith_ntt(SC_NTT_FLOATING_POINT);
    const utils_arith_poly_t *poly = utils_arith_poly();
    const utils_arith_vec_t *vec = utils_arith_vectors();
    const int qs[2] = { 12289, 8380417 }, ns[2] = { 512, 256 };
    for (int c = 0; c < 2; c++) {
        const int q = qs[c], n = ns[c], NP = 4;
        ntt_params_t p; init_reduce(&p, n, q);
        int32_t *a = malloc(sizeof(int32_t) * n * NP), *b = malloc(sizeof(int32_t) * n * NP), *o = malloc(sizeof(int32_t) * n * NP);
        int32_t rc[4], sc4[8];
        sm_state = 0x0b5000ULL + (uint64_t)q;
        /* poly 0,1: uniform non-zero residues; 2: signed small values (non-canonical inputs); 3: contains zeros (failure case) */
        for (int i = 0; i < n; i++) { a[i] = 1 + (int)(splitmix64() % (uint64_t)(q - 1)); b[i] = 1 + (int)(splitmix64() % (uint64_t)(q - 1)); }
        for (int i = 0; i < n; i++) { a[n + i] = 1 + (int)(splitmix64() % (uint64_t)(q - 1)); b[n + i] = (int)(splitmix64() % (uint64_t)q); }
        for (int i = 0; i < n; i++) { int v = (int)(splitmix64() % 4001) - 2000; a[2 * n + i] = v ? v : 7; b[2 * n + i] = (int)(splitmix64() % 4001) - 2000; }
        for (int i = 0; i < n; i++) { a[3 * n + i] = 1 + (int)(splitmix64() % (uint64_t)(q - 1)); b[3 * n + i] = (int)(splitmix64() % (uint64_t)q); }
        a[3 * n + n / 3] = 0; a[3 * n + n / 2] = q;        /* both are zero mod q; the first one stops the reference */
        snprintf(name, sizeof name, "q%d_n%d_a", q, n); rec_i32(f, name, a, (size_t)n * NP);
        snprintf(name, sizeof name, "q%d_n%d_b", q, n); rec_i32(f, name, b, (size_t)n * NP);
        /* invert_32: in place, stops at the first zero (earlier entries already inverted); normalised for the fixture */
        memcpy(o, a, sizeof(int32_t) * n * NP);
        for (int k = 0; k < NP; k++) { rc[k] = ntt->invert_32(o + k * n, &p, n); ntt->normalize_32(o + k * n, n, &p); }
        snprintf(name, sizeof name, "q%d_n%d_invert", q, n); rec_i32(f, name, o, (size_t)n * NP);
        snprintf(name, sizeof name, "q%d_n%d_invert_rc", q, n); rec_i32(f, name, rc, NP);
        /* div_32: num = b, den = a */
        memcpy(o, b, sizeof(int32_t) * n * NP);
        for (int k = 0; k < NP; k++) { rc[k] = ntt->div_32(o + k * n, a + k * n, &p, n); ntt->normalize_32(o + k * n, n, &p); }
        snprintf(name, sizeof name, "q%d_n%d_div", q, n); rec_i32(f, name, o, (size_t)n * NP);
        snprintf(name, sizeof name, "q%d_n%d_div_rc", q, n); rec_i32(f, name, rc, NP);
        /* pwr_32 with exponents 0, 1, 2, 65537, q-2 on poly 0 (canonical inputs) */
        const int32_t es[5] = { 0, 1, 2, 65537, q - 2 };
        for (int e = 0; e < 5; e++) {
            for (int i = 0; i < n; i++) o[i] = ntt->pwr_32(a[i], es[e], &p);
            ntt->normalize_32(o, n, &p);
            snprintf(name, sizeof name, "q%d_n%d_pwr%d", q, n, e); rec_i32(f, name, o, n);
        }
        rec_i32(f, c == 0 ? "q12289_pwr_exponents" : "q8380417_pwr_exponents", es, 5);
        /* mul_32_sparse: omega distinct positions per polynomial, t = b (signed values for poly 2) */
        const int omega = 11;
        int32_t *pos = malloc(sizeof(int32_t) * omega * NP);
        for (int k = 0; k < NP; k++)
            for (int i = 0; i < omega; i++) {
                int v, dup;
                do { v = (int)(splitmix64() % (uint64_t)n); dup = 0; for (int j = 0; j < i; j++) dup |= pos[k * omega + j] == v; } while (dup);
                pos[k * omega + i] = v;
            }
        pos[0] = 0;                                                   /* exercise position 0 */
        for (int k = 0; k < NP; k++) ntt->mul_32_sparse(o + k * n, n, omega, b + k * n, pos + k * omega);
        snprintf(name, sizeof name, "q%d_n%d_sparse_pos", q, n); rec_i32(f, name, pos, (size_t)omega * NP);
        snprintf(name, sizeof name, "q%d_n%d_sparse", q, n); rec_i32(f, name, o, (size_t)n * NP);
        free(pos);
        /* poly entries */
        memcpy(o, a, sizeof(int32_t) * n * NP); poly->mul_32_scalar(o, (size_t)n * NP, -3);
        snprintf(name, sizeof name, "q%d_n%d_mul_scalar_m3", q, n); rec_i32(f, name, o, (size_t)n * NP);
        memcpy(o, a, sizeof(int32_t) * n * NP); for (int k = 0; k < NP; k++) poly->add_32_scalar(o + k * n, n, 1000 + k);
        snprintf(name, sizeof name, "q%d_n%d_add_scalar", q, n); rec_i32(f, name, o, (size_t)n * NP);
        poly->mod_negate_32(o, (size_t)n * NP, q, b);
        snprintf(name, sizeof name, "q%d_n%d_mod_negate", q, n); rec_i32(f, name, o, (size_t)n * NP);
        memcpy(o, a, sizeof(int32_t) * n * NP); poly->add_single_32(o, (size_t)n * NP, b);
        snprintf(name, sizeof name, "q%d_n%d_add_single", q, n); rec_i32(f, name, o, (size_t)n * NP);
        memcpy(o, a, sizeof(int32_t) * n * NP); poly->sub_single_32(o, (size_t)n * NP, b);
        snprintf(name, sizeof name, "q%d_n%d_sub_single", q, n); rec_i32(f, name, o, (size_t)n * NP);
        /* cmp_not_equal: a vs a (0), a vs a with one coefficient changed, a vs b */
        memcpy(o, a, sizeof(int32_t) * n * NP); o[n + 17] ^= 0x40;
        rc[0] = poly->cmp_not_equal_32(a, a, n); rc[1] = poly->cmp_not_equal_32(a + n, o + n, n);
        rc[2] = poly->cmp_not_equal_32(a + 2 * n, b + 2 * n, n); rc[3] = poly->cmp_not_equal_32(a + 3 * n, o + 3 * n, n);
        snprintf(name, sizeof name, "q%d_n%d_cmp", q, n); rec_i32(f, name, rc, 4);
        for (int k = 0; k < NP; k++) { sc4[k] = vec->absmax_32(b + k * n, n); sc4[4 + k] = vec->scalar_32(a + k * n, b + k * n, n); }
        snprintf(name, sizeof name, "q%d_n%d_absmax_scalar", q, n); rec_i32(f, name, sc4, 8);
        free(a); free(b); free(o);
    }
    fclose(f);
    return 0;
}

/* ---------------------------------------------------------------- Falcon ---- */
#include "schemes/sig/falcon/falcon.h"
#include "schemes/sig/falcon/falcon_params.h"
#include "utils/arith/gpv.h"

static void rec_f64(FILE *f, const char *name, const double *v, size_t n)
{
    /* doubles travel as their IEEE-754 bit patterns */
    fprintf(f, "@ %s %zu\n", name, n);
    for (size_t i = 0; i < n; i++) { int64_t b; memcpy(&b, &v[i], 8); fprintf(f, "%lld%c", (long long)b, (i % 8 == 7 || i + 1 == n) ? '\n' : ' '); }
}

/* Falcon signing (src/schemes/sig/falcon/falcon.c:808-935) draws its randomness inside
 * gaussian_lattice_sample_fft (src/utils/arith/gpv.c:2663-2714): one throw-away CDF sampler per tree leaf
 * on sc->prng_ctx[0], two prng_64 per leaf.  Same tape technique as BLISS: a freshly seeded CSPRNG per
 * message and a twin read with prng_32 only.  Two kinds of record: the sampler alone on a synthetic target
 * c (s1, s2 as returned), and whole signatures (message -> packed bytes, verify verdicts). */
static int falcon_case(FILE *f, int set, int nsamp, int nmsg)
{
    UINT32 flags[2] = { SC_FLAG_NONE, SC_FLAG_NONE };
    char name[96], tag[32];
    snprintf(tag, sizeof tag, "falcon%d", set);
    safecrypto_t *sc = create_fixed(SC_SCHEME_SIG_FALCON, set, flags);
    if (!sc) return 1;
    if (SC_FUNC_SUCCESS != safecrypto_keygen(sc)) return 1;
    const falcon_set_t *P = sc->falcon->params;
    const int n = P->n, logn = P->n_bits;
    int32_t par[8] = { P->n, P->q, P->q_bits, P->n_bits, set, 0, 0, 0 };
    float bd = P->bd; memcpy(&par[5], &bd, 4);
    snprintf(name, sizeof name, "%s_params", tag); rec_i32(f, name, par, 8);
    snprintf(name, sizeof name, "%s_privkey", tag); rec_i32(f, name, (const int32_t *)sc->privkey->key, 4 * (size_t)n);  /* f g F G */
    snprintf(name, sizeof name, "%s_h", tag); rec_i16(f, name, (const int16_t *)sc->pubkey->key, n);
    snprintf(name, sizeof name, "%s_sk", tag); rec_f64(f, name, sc->falcon->sk, (size_t)(logn + 5) * n);
    prng_ctx_t *orig = sc->prng_ctx[0];
    sm_state = 0xfa1c0000ULL + set;
    int32_t *c = malloc(sizeof(int32_t) * n), *s1 = malloc(sizeof(int32_t) * n), *s2 = malloc(sizeof(int32_t) * n);
    for (int m = 0; m < nsamp; m++) {
        uint8_t seed[32];
        for (int i = 0; i < n; i++) c[i] = (int32_t)(splitmix64() % (uint64_t)P->q);
        for (int i = 0; i < 32; i++) seed[i] = (uint8_t)splitmix64();
        prng_ctx_t *mine = seeded_prng(seed, 32), *twin = seeded_prng(seed, 32);
        sc->prng_ctx[0] = mine;
        if (SC_FUNC_SUCCESS != gaussian_sample_with_tree(sc, sc->falcon->sk, 0, n, P->q, logn, c, 0, s1, s2)) return 1;
        const size_t words = (size_t)(prng_get_out_bytes(mine) / 4);
        int32_t *tape = malloc(sizeof(int32_t) * words);
        for (size_t i = 0; i < words; i++) tape[i] = (int32_t)prng_32(twin);
        snprintf(name, sizeof name, "%s_c%d", tag, m); rec_i32(f, name, c, n);
        snprintf(name, sizeof name, "%s_ctape%d", tag, m); rec_i32(f, name, tape, words);
        snprintf(name, sizeof name, "%s_s1_%d", tag, m); rec_i32(f, name, s1, n);
        snprintf(name, sizeof name, "%s_s2_%d", tag, m); rec_i32(f, name, s2, n);
        free(tape);
        sc->prng_ctx[0] = orig;
        prng_destroy(mine); prng_destroy(twin);
    }
    int32_t *verdicts = malloc(sizeof(int32_t) * nmsg * 2), *status = malloc(sizeof(int32_t) * nmsg);
    for (int m = 0; m < nmsg; m++) {
        uint8_t msg[200], seed[32];
        size_t mlen = (m == 0) ? 1 : (m == 1) ? 200 : 48;
        for (size_t i = 0; i < mlen; i++) msg[i] = (uint8_t)splitmix64();
        for (int i = 0; i < 32; i++) seed[i] = (uint8_t)splitmix64();
        prng_ctx_t *mine = seeded_prng(seed, 32), *twin = seeded_prng(seed, 32);
        sc->prng_ctx[0] = mine;
        uint8_t *sig = NULL; size_t siglen = 0;
        status[m] = safecrypto_sign(sc, msg, mlen, &sig, &siglen);
        const size_t words = (size_t)(prng_get_out_bytes(mine) / 4);
        int32_t *tape = malloc(sizeof(int32_t) * words);
        for (size_t i = 0; i < words; i++) tape[i] = (int32_t)prng_32(twin);
        snprintf(name, sizeof name, "%s_msg%d", tag, m); rec_u8(f, name, msg, mlen);
        snprintf(name, sizeof name, "%s_tape%d", tag, m); rec_i32(f, name, tape, words);
        snprintf(name, sizeof name, "%s_sig%d", tag, m); rec_u8(f, name, sig, siglen);
        verdicts[2 * m] = verdicts[2 * m + 1] = -1;
        if (SC_FUNC_SUCCESS == status[m]) {
            verdicts[2 * m] = safecrypto_verify(sc, msg, mlen, sig, siglen);
            sig[siglen / 3] ^= 0x10;
            verdicts[2 * m + 1] = safecrypto_verify(sc, msg, mlen, sig, siglen);
        }
        free(tape); free(sig);
        sc->prng_ctx[0] = orig;
        prng_destroy(mine); prng_destroy(twin);
    }
    snprintf(name, sizeof name, "%s_status", tag); rec_i32(f, name, status, nmsg);
    snprintf(name, sizeof name, "%s_verify", tag); rec_i32(f, name, verdicts, (size_t)nmsg * 2);
    free(c); free(s1); free(s2); free(verdicts); free(status);
    safecrypto_destroy(sc);
    return 0;
}

/* `nkeys` key pairs from the reference's own keygen, each with its expanded private key, its public key and one signature with the CSPRNG words
 * signing consumed: the distinct-key parity test runs this on the box it runs on (no fixture of this size is committed).
 * Usage: ref_driver falcon_keys <set> <nkeys> <outfile> */
static int falcon_keys(int set, int nkeys, const char *path)
{
    FILE *f = fopen(path, "w");
    if (!f) return 1;
    UINT32 flags[2] = { SC_FLAG_NONE, SC_FLAG_NONE };
    char name[96];
    sm_state = 0xfa1c7000ULL + (uint64_t)set;
    for (int kx = 0; kx < nkeys; kx++) {
        safecrypto_t *sc = safecrypto_create(SC_SCHEME_SIG_FALCON, set, flags);
        if (!sc) return 1;
        if (SC_FUNC_SUCCESS != safecrypto_keygen(sc)) return 1;
        const falcon_set_t *P = sc->falcon->params;
        const int n = P->n, logn = P->n_bits;
        if (kx == 0) {
            int32_t par[8] = { P->n, P->q, P->q_bits, P->n_bits, set, 0, 0, 0 };
            float bd = P->bd; memcpy(&par[5], &bd, 4);
            rec_i32(f, "params", par, 8);
        }
        snprintf(name, sizeof name, "k%d_h", kx); rec_i16(f, name, (const int16_t *)sc->pubkey->key, n);
        snprintf(name, sizeof name, "k%d_sk", kx); rec_f64(f, name, sc->falcon->sk, (size_t)(logn + 5) * n);
        uint8_t msg[64], seed[32];
        for (int i = 0; i < 64; i++) msg[i] = (uint8_t)splitmix64();
        for (int i = 0; i < 32; i++) seed[i] = (uint8_t)splitmix64();
        prng_ctx_t *orig = sc->prng_ctx[0];
        prng_ctx_t *mine = seeded_prng(seed, 32), *twin = seeded_prng(seed, 32);
        sc->prng_ctx[0] = mine;
        uint8_t *sig = NULL; size_t siglen = 0;
        int32_t st = safecrypto_sign(sc, msg, 64, &sig, &siglen);
        const size_t words = (size_t)(prng_get_out_bytes(mine) / 4);
        int32_t *tape = malloc(sizeof(int32_t) * words);
        for (size_t i = 0; i < words; i++) tape[i] = (int32_t)prng_32(twin);
        snprintf(name, sizeof name, "k%d_msg", kx); rec_u8(f, name, msg, 64);
        snprintf(name, sizeof name, "k%d_tape", kx); rec_i32(f, name, tape, words);
        snprintf(name, sizeof name, "k%d_sig", kx); rec_u8(f, name, sig, st == SC_FUNC_SUCCESS ? siglen : 0);
        snprintf(name, sizeof name, "k%d_status", kx); rec_i32(f, name, &st, 1);
        free(tape); free(sig);
        sc->prng_ctx[0] = orig;
        prng_destroy(mine); prng_destroy(twin);
        safecrypto_destroy(sc);
    }
    fclose(f);
    return 0;
}

static int gen_falcon(const char *dir)
{
    char path[512];
    snprintf(path, sizeof path, "%s/falcon_ref.txt", dir);
    FILE *f = fopen(path, "w");
    if (!f) return 1;
    int rc = falcon_case(f, 0, 2, 4) | falcon_case(f, 2, 1, 2);
    fclose(f);
    return rc;
}

/* -------------------------------------------------------------- CPU baseline ---- */

#include <time.h>
static double now_s(void) { struct timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return t.tv_sec + 1e-9 * t.tv_nsec; }

/* The reference's own code path for BASELINE config 2, single thread, default (floating
 * point) reduction backend as the schemes select it (dilithium.c:184-191, kyber_kem.c:102-110):
 * fwd_ntt -> mul_32_pointwise -> inv_ntt per polynomial, over a resident pool of polynomials. */
static int bench_ntt(int32_t q, int n, double seconds)
{
    /* the backend a scheme context gets by default from this build: floating point, or AVX2 in the -mavx2 -DHAVE_AVX2 build (dilithium.c:184-191) */
#ifdef HAVE_AVX2
    const utils_arith_ntt_t *ntt = utils_arith_ntt(SC_NTT_AVX);
#else
    const utils_arith_ntt_t *ntt = utils_arith_ntt(SC_NTT_FLOATING_POINT);
#endif
    ntt_params_t p;
    const int POOL = 1024;
    int32_t *w = malloc(sizeof(int32_t) * n), *r = malloc(sizeof(int32_t) * n);
    int16_t *w16 = malloc(sizeof(int16_t) * n), *r16 = malloc(sizeof(int16_t) * n);
    int32_t *a = malloc(sizeof(int32_t) * n * POOL), *b = malloc(sizeof(int32_t) * n * POOL), *o = malloc(sizeof(int32_t) * n * POOL);
    init_reduce(&p, n, q);
    if (roots_of_unity_s32(w, r, n, q, 0, 0) != SC_FUNC_SUCCESS) return 1;
    const int is16 = q < 32768;
    for (int i = 0; i < n; i++) { w16[i] = (int16_t)w[i]; r16[i] = (int16_t)r[i]; }
    sm_state = 1;
    for (int i = 0; i < n * POOL; i++) { a[i] = (int)(splitmix64() % (uint64_t)q); b[i] = (int)(splitmix64() % (uint64_t)q); }
    uint64_t done = 0, chk = 0;
    double t0 = now_s(), t1;
    do {
        for (int k = 0; k < POOL; k++) {
            int32_t *v = o + k * n;
            if (is16) { ntt->fwd_ntt_32_16(v, &p, a + k * n, w16); ntt->mul_32_pointwise(v, &p, v, b + k * n); ntt->inv_ntt_32_16(v, &p, v, w16, r16); }
            else      { ntt->fwd_ntt_32_32(v, &p, a + k * n, w);   ntt->mul_32_pointwise(v, &p, v, b + k * n); ntt->inv_ntt_32_32(v, &p, v, w, r); }
            chk += (uint32_t)v[0];
        }
        done += POOL;
        t1 = now_s();
    } while (t1 - t0 < seconds);
    printf("{\"q\": %d, \"n\": %d, \"triples\": %llu, \"seconds\": %.6f, \"triples_per_sec\": %.1f, \"checksum\": %llu}\n",
           q, n, (unsigned long long)done, t1 - t0, done / (t1 - t0), (unsigned long long)chk);
    return 0;
}

/* safecrypto_sign / safecrypto_verify of the reference, one key, 64-byte counter-derived messages
 * (the loop of test/functional/func_alg_dilithium.c:191-230 without the key regeneration) */
static int bench_sig(sc_scheme_e scheme, int set, double seconds)
{
    UINT32 flags[2] = { SC_FLAG_NONE | SC_FLAG_MORE, SC_FLAG_NONE };
    safecrypto_t *sc = safecrypto_create(scheme, set, flags);
    if (!sc || SC_FUNC_SUCCESS != safecrypto_keygen(sc)) return 1;
    enum { POOL = 64 };
    uint8_t msg[POOL][64], *sig[POOL]; size_t siglen[POOL];
    sm_state = 7;
    for (int i = 0; i < POOL; i++) { for (int j = 0; j < 64; j++) msg[i][j] = (uint8_t)splitmix64(); sig[i] = NULL; siglen[i] = 0; }
    uint64_t nsign = 0, nver = 0; size_t tr0 = sc->stats.sig_num_trials;
    double t0 = now_s(), t1;
    do {
        for (int i = 0; i < POOL; i++) {
            if (sig[i]) { free(sig[i]); sig[i] = NULL; siglen[i] = 0; }
            if (SC_FUNC_SUCCESS != safecrypto_sign(sc, msg[i], 64, &sig[i], &siglen[i])) return 1;
        }
        nsign += POOL; t1 = now_s();
    } while (t1 - t0 < seconds);
    const double ts = t1 - t0; const double trials = (double)(sc->stats.sig_num_trials - tr0) / (double)nsign;
    t0 = now_s();
    do {
        for (int i = 0; i < POOL; i++) if (SC_FUNC_SUCCESS != safecrypto_verify(sc, msg[i], 64, sig[i], siglen[i])) return 1;
        nver += POOL; t1 = now_s();
    } while (t1 - t0 < seconds / 4);
    printf("{\"set\": %d, \"sign_per_sec\": %.1f, \"verify_per_sec\": %.1f, \"signs\": %llu, \"trials_per_sig\": %.3f}\n",
           set, nsign / ts, nver / (t1 - t0), (unsigned long long)nsign, trials);
    return 0;
}

static int bench_kem(sc_scheme_e scheme, int set, double seconds)
{
    UINT32 flags[2] = { SC_FLAG_NONE | SC_FLAG_MORE, SC_FLAG_NONE };
    safecrypto_t *sc = safecrypto_create(scheme, set, flags);
    if (!sc || SC_FUNC_SUCCESS != safecrypto_keygen(sc)) return 1;
    enum { POOL = 64 };
    UINT8 *ct[POOL], *key[POOL]; size_t ctlen[POOL], keylen[POOL];
    memset(ct, 0, sizeof ct); memset(key, 0, sizeof key);
    uint64_t nenc = 0, ndec = 0;
    double t0 = now_s(), t1;
    do {
        for (int i = 0; i < POOL; i++) {
            free(ct[i]); free(key[i]); ct[i] = key[i] = NULL; ctlen[i] = keylen[i] = 0;
            if (SC_FUNC_SUCCESS != safecrypto_encapsulation(sc, &ct[i], &ctlen[i], &key[i], &keylen[i])) return 1;
        }
        nenc += POOL; t1 = now_s();
    } while (t1 - t0 < seconds);
    const double te = t1 - t0;
    t0 = now_s();
    do {
        for (int i = 0; i < POOL; i++) { UINT8 *k2 = NULL; size_t k2len = 0; if (SC_FUNC_SUCCESS != safecrypto_decapsulation(sc, ct[i], ctlen[i], &k2, &k2len)) return 1; free(k2); }
        ndec += POOL; t1 = now_s();
    } while (t1 - t0 < seconds / 2);
    printf("{\"set\": %d, \"encaps_per_sec\": %.1f, \"decaps_per_sec\": %.1f, \"encaps\": %llu}\n", set, nenc / te, ndec / (t1 - t0), (unsigned long long)nenc);
    return 0;
}

/* single-thread rlwe_enc_encrypt / rlwe_enc_decrypt through safecrypto_public_encrypt / safecrypto_private_decrypt */
static int bench_enc(int set, double seconds)
{
    UINT32 flags[2] = { SC_FLAG_NONE, SC_FLAG_NONE };
    safecrypto_t *sc = safecrypto_create(SC_SCHEME_ENC_RLWE, set, flags);
    if (!sc || SC_FUNC_SUCCESS != safecrypto_keygen(sc)) return 1;
    const size_t mlen = (set ? param_rlwe_enc_1.n : param_rlwe_enc_0.n) / 8;
    enum { POOL = 64 };
    UINT8 msg[128], *ct[POOL]; size_t ctlen[POOL];
    memset(ct, 0, sizeof ct);
    sm_state = 0xbe7c4ULL;
    uint64_t nenc = 0, ndec = 0;
    double t0 = now_s(), t1;
    do {
        for (int i = 0; i < POOL; i++) {
            for (size_t b = 0; b < mlen; b += 8) { uint64_t v = splitmix64(); memcpy(msg + b, &v, 8); }
            free(ct[i]); ct[i] = NULL; ctlen[i] = 0;
            if (SC_FUNC_SUCCESS != safecrypto_public_encrypt(sc, mlen, msg, &ctlen[i], &ct[i])) return 1;
        }
        nenc += POOL; t1 = now_s();
    } while (t1 - t0 < seconds);
    const double te = t1 - t0;
    t0 = now_s();
    do {
        for (int i = 0; i < POOL; i++) { UINT8 *m2 = NULL; size_t m2len = 0; if (SC_FUNC_SUCCESS != safecrypto_private_decrypt(sc, ctlen[i], ct[i], &m2len, &m2)) return 1; free(m2); }
        ndec += POOL; t1 = now_s();
    } while (t1 - t0 < seconds / 2);
    printf("{\"set\": %d, \"encrypt_per_sec\": %.1f, \"decrypt_per_sec\": %.1f, \"encryptions\": %llu}\n", set, nenc / te, ndec / (t1 - t0), (unsigned long long)nenc);
    return 0;
}

/* single-thread dlp_ibe_extract through safecrypto_ibe_extract under one master key */
static int bench_dlp(int set, double seconds)
{
    UINT32 flags[2] = { SC_FLAG_NONE, SC_FLAG_NONE };
    safecrypto_t *sc = safecrypto_create(SC_SCHEME_IBE_DLP, set, flags);
    if (!sc) return 1;
    FILE *devnull = fopen("/dev/null", "w"), *saved = stdout, *saved_err = stderr;
    stdout = devnull; stderr = devnull;
    const int krc = safecrypto_keygen(sc);
    stdout = saved; stderr = saved_err;
    if (SC_FUNC_SUCCESS != krc) return 1;
    uint64_t nex = 0;
    uint8_t id[16];
    sm_state = 0xd1bbeULL;
    double t0 = now_s(), t1;
    do {
        for (int i = 0; i < 16; i++) {
            uint64_t v = splitmix64(); memcpy(id, &v, 8); v = splitmix64(); memcpy(id + 8, &v, 8);
            UINT8 *sk = NULL; size_t sklen = 0;
            if (SC_FUNC_SUCCESS != safecrypto_ibe_extract(sc, 16, id, &sklen, &sk)) return 1;
            free(sk);
        }
        nex += 16; t1 = now_s();
    } while (t1 - t0 < seconds);
    const double tex = t1 - t0;
    /* encrypt to / decrypt for one identity */
    UINT8 *sk = NULL; size_t sklen = 0;
    if (SC_FUNC_SUCCESS != safecrypto_ibe_extract(sc, 16, id, &sklen, &sk) || SC_FUNC_SUCCESS != safecrypto_secret_key(sc, sklen, sk)) return 1;
    const size_t mlen = (set ? param_dlp_ibe_1.n : param_dlp_ibe_0.n) / 8;
    enum { POOL = 32 };
    UINT8 msg[128], *ct[POOL]; size_t ctlen[POOL];
    memset(ct, 0, sizeof ct);
    uint64_t nenc = 0, ndec = 0;
    t0 = now_s();
    do {
        for (int i = 0; i < POOL; i++) {
            for (size_t b = 0; b < mlen; b += 8) { uint64_t v = splitmix64(); memcpy(msg + b, &v, 8); }
            free(ct[i]); ct[i] = NULL; ctlen[i] = 0;
            if (SC_FUNC_SUCCESS != safecrypto_ibe_public_encrypt(sc, 16, id, mlen, msg, &ctlen[i], &ct[i])) return 1;
        }
        nenc += POOL; t1 = now_s();
    } while (t1 - t0 < seconds / 2);
    const double te = t1 - t0;
    t0 = now_s();
    do {
        for (int i = 0; i < POOL; i++) { UINT8 *m2 = NULL; size_t m2len = 0; if (SC_FUNC_SUCCESS != safecrypto_private_decrypt(sc, ctlen[i], ct[i], &m2len, &m2)) return 1; free(m2); }
        ndec += POOL; t1 = now_s();
    } while (t1 - t0 < seconds / 2);
    printf("{\"set\": %d, \"extract_per_sec\": %.1f, \"encrypt_per_sec\": %.1f, \"decrypt_per_sec\": %.1f, \"extractions\": %llu}\n", set, nex / tex, nenc / te,
           ndec / (t1 - t0), (unsigned long long)nex);
    return 0;
}

int main(int argc, char **argv)
{
    if (argc == 4 && !strcmp(argv[1], "bench_dlp")) return bench_dlp(atoi(argv[2]), atof(argv[3]));
    if (argc == 4 && !strcmp(argv[1], "bench_kyber")) return bench_kem(SC_SCHEME_KEM_KYBER, atoi(argv[2]), atof(argv[3]));
    if (argc == 4 && !strcmp(argv[1], "bench_ens")) return bench_kem(SC_SCHEME_KEM_ENS, atoi(argv[2]), atof(argv[3]));
    if (argc == 4 && !strcmp(argv[1], "bench_rlwe")) return bench_enc(atoi(argv[2]), atof(argv[3]));
    if (argc == 4 && !strcmp(argv[1], "bench_dil")) return bench_sig(SC_SCHEME_SIG_DILITHIUM, atoi(argv[2]), atof(argv[3]));
    if (argc == 4 && !strcmp(argv[1], "bench_falcon")) return bench_sig(SC_SCHEME_SIG_FALCON, atoi(argv[2]), atof(argv[3]));
    if (argc == 4 && !strcmp(argv[1], "bench_bliss")) return bench_sig(SC_SCHEME_SIG_BLISS, atoi(argv[2]), atof(argv[3]));
    if (argc == 4 && !strcmp(argv[1], "bench_enssig")) return bench_sig(SC_SCHEME_SIG_ENS, atoi(argv[2]), atof(argv[3]));
    if (argc == 4 && !strcmp(argv[1], "bench_dlpsig")) return bench_sig(SC_SCHEME_SIG_DLP, atoi(argv[2]), atof(argv[3]));
    if (argc == 4 && !strcmp(argv[1], "bench_tesla")) return bench_sig(SC_SCHEME_SIG_RING_TESLA, atoi(argv[2]), atof(argv[3]));
    if (argc == 5 && !strcmp(argv[1], "falcon_keys")) return falcon_keys(atoi(argv[2]), atoi(argv[3]), argv[4]);
    if (argc == 5 && !strcmp(argv[1], "bench")) return bench_ntt(atoi(argv[2]), atoi(argv[3]), atof(argv[4]));
    if (argc < 3) { fprintf(stderr, "usage: %s <ntt|cdt|cdt32|dilithium|bliss|bliss_huff|bliss_bac|kyber|falcon|ops|drbg|tesla|rlwe|ens|dlp|ensdlp|ensdlp_rec|mlwe_xof|cdt_hp|bliss_hp|all> <outdir>\n", argv[0]); return 2; }
    if (getenv("SC_REF_SALT")) g_salt = strtoull(getenv("SC_REF_SALT"), NULL, 0);
    int all = !strcmp(argv[1], "all"), rc = 0;
    if (all || !strcmp(argv[1], "ntt")) { fixed_domain("ntt"); rc |= gen_ntt(argv[2]); }
    if (all || !strcmp(argv[1], "ntt_large")) { fixed_domain("ntt_large"); rc |= gen_ntt_large(argv[2]); }
    if (all || !strcmp(argv[1], "cdt")) { fixed_domain("cdt"); rc |= gen_cdt(argv[2]); }
    if (all || !strcmp(argv[1], "dilithium")) { fixed_domain("dilithium"); rc |= gen_dilithium(argv[2]); }
    if (all || !strcmp(argv[1], "bliss")) { fixed_domain("bliss"); rc |= gen_bliss(argv[2]); }
    if (all || !strcmp(argv[1], "cdt32")) { fixed_domain("cdt32"); rc |= gen_cdt32(argv[2]); }
    if (all || !strcmp(argv[1], "sampling")) { fixed_domain("sampling"); rc |= gen_sampling(argv[2]); }
    if (all || !strcmp(argv[1], "bliss_huff")) { fixed_domain("bliss_huff"); rc |= gen_bliss_huff(argv[2]); }
    if (all || !strcmp(argv[1], "bliss_bac")) { fixed_domain("bliss_bac"); rc |= gen_bliss_bac(argv[2]); }
    if (all || !strcmp(argv[1], "huff_rows")) { fixed_domain("huff_rows"); rc |= gen_huff_rows(argv[2]); }
    if (all || !strcmp(argv[1], "kyber")) { fixed_domain("kyber"); rc |= gen_kyber(argv[2]); }
    if (all || !strcmp(argv[1], "falcon")) { fixed_domain("falcon"); rc |= gen_falcon(argv[2]); }
    if (all || !strcmp(argv[1], "ops")) { fixed_domain("ops"); rc |= gen_ops(argv[2]); }
    if (all || !strcmp(argv[1], "drbg")) { fixed_domain("drbg"); rc |= gen_drbg(argv[2]); }
    if (all || !strcmp(argv[1], "tesla")) { fixed_domain("tesla"); rc |= gen_tesla(argv[2]); }
    if (all || !strcmp(argv[1], "rlwe")) { fixed_domain("rlwe"); rc |= gen_rlwe(argv[2]); }
    if (all || !strcmp(argv[1], "ens")) { fixed_domain("ens"); rc |= gen_ens(argv[2]); }
    if (all || !strcmp(argv[1], "dlp")) { fixed_domain("dlp"); rc |= gen_dlp(argv[2]); }
    if (all || !strcmp(argv[1], "ensdlp")) { fixed_domain("ensdlp"); rc |= gen_ensdlp(argv[2]); }
    if (all || !strcmp(argv[1], "ensdlp_rec")) { fixed_domain("ensdlp_rec"); rc |= gen_ensdlp_rec(argv[2]); }
    if (all || !strcmp(argv[1], "mlwe_xof")) { fixed_domain("mlwe_xof"); rc |= gen_mlwe_xof(argv[2]); }
    if (all || !strcmp(argv[1], "cdt_hp")) { fixed_domain("cdt_hp"); rc |= gen_cdt_hp(argv[2]); }
    if (all || !strcmp(argv[1], "bliss_hp")) { fixed_domain("bliss_hp"); rc |= gen_bliss_hp(argv[2]); }
    return rc;
}
