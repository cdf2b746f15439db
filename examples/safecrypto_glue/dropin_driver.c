/*
 * dropin_driver.c - proof that the engine is a drop-in behind libsafecrypto's own function pointers.
 *
 * Our own source.  Built in the dev container against libsafecrypto's headers and linked with BOTH the reference library
 * (libsafecrypto_ref.so, compiled from the reference sources where they lie) and the engine (libsc_hip.so); run on the GPU box
 * by tests/test_gpu_dropin.py.  Test infrastructure: nothing in the product links it.
 *
 * Inner boundary: for each scheme a reference context signs / encapsulates with its own CPU tables, then sc_hip_glue_attach()
 * points sc->sc_ntt / sc_poly / sc_vec / sc_gauss at the engine and the SAME unmodified reference code (dilithium_sign,
 * bliss_b_sign, kyber_kem_encapsulation, ring_tesla_sign, ...) runs again on the same inputs and the same CSPRNG seed: the bytes
 * must be identical, genuine signatures must verify, tampered ones must not, and key generation must produce the same key pair.
 * Outer boundary: the safecrypto_alg_t-shaped entries of sc_hip_glue.h (whole operation on the device) against safecrypto_sign /
 * safecrypto_verify of the reference, one message and a batch.
 *
 * Output: one JSON line per check and a summary; exit status 0 iff every check passed.
 */
#define NTT_NEEDS_7681
#define NTT_NEEDS_8380417
#define NTT_NEEDS_12289
#define NTT_NEEDS_8399873
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <stdint.h>
#include <time.h>

#include "sc_hip_glue.h"
#include "utils/crypto/prng.h"
#include "utils/sampling/sampling.h"
#include "sc_hip.h"
#include "sc_hip_vtable.h"

static int g_fail = 0, g_checks = 0;
static void report(const char *name, int ok, const char *detail)
{
    g_checks++;
    if (!ok) g_fail++;
    printf("{\"check\": \"%s\", \"ok\": %s, \"detail\": \"%s\"}\n", name, ok ? "true" : "false", detail ? detail : "");
    fflush(stdout);
}
static double now(void) { struct timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return t.tv_sec + 1e-9 * t.tv_nsec; }

static uint64_t sm_state;
static uint64_t splitmix64(void)
{
    uint64_t z = (sm_state += 0x9E3779B97F4A7C15ULL);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
}

static prng_ctx_t *seeded_prng(const uint8_t *seed, size_t len)
{
    /* the seeding pattern of src/unit/unit_module_lwe.c:44-47 */
    prng_ctx_t *ctx = prng_create(SC_ENTROPY_USER_PROVIDED, SC_PRNG_AES_CTR_DRBG, SC_PRNG_THREADING_NONE, 0x00100000);
    prng_set_entropy(ctx, seed, len);
    prng_init(ctx, NULL, 0);
    return ctx;
}

/* the private struct behind a 64-bit CDF sampler's gauss pointer (gaussian_cdf.c:74-81): its PRNG follows the context's */
typedef struct __attribute__((packed)) { UINT64 *cdf; SINT32 cdf_size; SINT32 k; SINT32 use_kl; prng_ctx_t *prng_ctx; } cdf64_view_t;
/* ... and behind a 128- / 192-bit one (gauss_cdf_high_t, :59-68) */
typedef struct __attribute__((packed)) { UINT64 *cdf_256, *cdf_192, *cdf_128; SINT32 cdf_size; SINT32 k; SINT32 use_kl; prng_ctx_t *prng_ctx; } cdfhp_view_t;

static void swap_prng(safecrypto_t *sc, sc_hip_glue_t *g, prng_ctx_t *p)
{
    sc->prng_ctx[0] = p;
    if (g) { sc_hip_glue_refresh_prng(g); return; }
    if (sc->sc_gauss) {
        sc->sc_gauss->prng_ctx = p;
        if (CDF_GAUSSIAN_SAMPLING == sc->sampling && SAMPLING_64BIT == sc->sc_gauss->precision) ((cdf64_view_t *)sc->sc_gauss->gauss)->prng_ctx = p;
        if (CDF_GAUSSIAN_SAMPLING == sc->sampling && (SAMPLING_128BIT == sc->sc_gauss->precision || SAMPLING_192BIT == sc->sc_gauss->precision))
            ((cdfhp_view_t *)sc->sc_gauss->gauss)->prng_ctx = p;
    }
}

/* A context created with SC_FLAG_0_SAMPLE_128BIT / _192BIT - what the reference's own BLISS-B harness asks for (test/functional/func_alg_bliss_b.c:118) - holds a table of noise
 * in THIS build of the reference (no MPFR: sc_mpf_exp / sc_mpf_get_pi are empty, sc_mpf.c:99-106,822-828) and its safecrypto_sign would not terminate.  The driver gives such a
 * context a table of known content right after creation - most significant limbs = the 64-bit table of the same (tail, sigma), which the reference can build, the limbs below =
 * noise - so that the reference's search (gaussian_cdf_sample_128 / _192) and the engine's run over the same table, as they would over an MPFR-made one. */
static void install_hp_table(safecrypto_t *sc)
{
    if (!sc || !sc->sc_gauss || CDF_GAUSSIAN_SAMPLING != sc->sampling) return;
    const int prec = (int)sc->sc_gauss->precision;
    if (prec != 128 && prec != 192) return;
    static const uint8_t sd[32] = { 7, 7, 7 };
    prng_ctx_t *p3 = seeded_prng(sd, 32);
    utils_sampling_t *s64 = create_sampler(CDF_GAUSSIAN_SAMPLING, SAMPLING_64BIT, NORMAL_SAMPLES, 512, SAMPLING_DISABLE_BOOTSTRAP, p3, sc->sc_gauss->tail, sc->sc_gauss->sigma);
    if (!s64) { prng_destroy(p3); return; }
    const cdf64_view_t *g64 = (const cdf64_view_t *)s64->gauss;
    cdfhp_view_t *g = (cdfhp_view_t *)sc->sc_gauss->gauss;
    const int W = prec / 64;
    UINT64 *tab = 128 == prec ? g->cdf_128 : g->cdf_192;
    uint64_t keep = sm_state;
    sm_state = 0x7ab1e000ULL + (uint64_t)prec;
    for (int b = 0; b < g->cdf_size && b < g64->cdf_size; b++) {
        for (int j = 0; j < W - 1; j++) tab[(size_t)b * W + j] = splitmix64();
        tab[(size_t)b * W + W - 1] = g64->cdf[b];
    }
    sm_state = keep;
    destroy_sampler(&s64);
    prng_destroy(p3);
}

#define NMSG 6
typedef struct { uint8_t *sig[NMSG]; size_t len[NMSG]; int verdict[NMSG], verdict_bad[NMSG]; } sigset_t2;

/* signs NMSG counter-derived messages, message m under a CSPRNG seeded with seed m; verifies each and a tampered copy */
static int sign_all(safecrypto_t *sc, sc_hip_glue_t *g, uint64_t tag, sigset_t2 *out)
{
    prng_ctx_t *orig = sc->prng_ctx[0];
    sm_state = tag;
    for (int m = 0; m < NMSG; m++) {
        uint8_t msg[96], seed[32];
        const size_t mlen = (m == 0) ? 1 : (m == 1) ? 96 : 64;
        for (size_t i = 0; i < mlen; i++) msg[i] = (uint8_t)splitmix64();
        for (int i = 0; i < 32; i++) seed[i] = (uint8_t)splitmix64();
        prng_ctx_t *mine = seeded_prng(seed, 32);
        swap_prng(sc, g, mine);
        out->sig[m] = NULL; out->len[m] = 0;
        const int rc = safecrypto_sign(sc, msg, mlen, &out->sig[m], &out->len[m]);
        if (SC_FUNC_SUCCESS == rc) {
            out->verdict[m] = safecrypto_verify(sc, msg, mlen, out->sig[m], out->len[m]);
            uint8_t *bad = malloc(out->len[m]);
            memcpy(bad, out->sig[m], out->len[m]);
            bad[out->len[m] / 3] ^= 0x04;
            out->verdict_bad[m] = safecrypto_verify(sc, msg, mlen, bad, out->len[m]);
            free(bad);
        }
        swap_prng(sc, g, orig);
        prng_destroy(mine);
        if (SC_FUNC_SUCCESS != rc) return 1;
    }
    return 0;
}
static void free_all(sigset_t2 *s) { for (int m = 0; m < NMSG; m++) free(s->sig[m]); }

static void signature_scheme(const char *name, sc_scheme_e scheme, int set, UINT32 flag0, int tamper_must_fail)
{
    char label[128], detail[256];
    UINT32 flags[2] = { flag0, SC_FLAG_NONE };
    safecrypto_t *sc = safecrypto_create(scheme, set, flags);
    install_hp_table(sc);                                                  /* (only a 128- / 192-bit sampler's context gets one) */
    if (!sc || SC_FUNC_SUCCESS != safecrypto_keygen(sc)) { snprintf(label, sizeof label, "%s: create/keygen", name); report(label, 0, ""); return; }
    sigset_t2 cpu, gpu;
    memset(&cpu, 0, sizeof cpu); memset(&gpu, 0, sizeof gpu);
    double t0 = now();
    int rc = sign_all(sc, NULL, 0x5ca1ab1e00ULL + set, &cpu);
    const double t_cpu = now() - t0;
    const unsigned long e0 = sc_hip_vtable_errors();
    sc_hip_glue_t *g = sc_hip_glue_attach(sc, 0);
    snprintf(label, sizeof label, "%s: tables attached", name);
    report(label, g != NULL && rc == 0, g ? "" : sc_hip_last_error());
    if (!g || rc) { safecrypto_destroy(sc); return; }
    t0 = now();
    rc = sign_all(sc, g, 0x5ca1ab1e00ULL + set, &gpu);
    const double t_gpu = now() - t0;
    int same = (rc == 0), ok_v = 1, ok_bad = 1;
    for (int m = 0; m < NMSG && rc == 0; m++) {
        if (cpu.len[m] != gpu.len[m] || memcmp(cpu.sig[m], gpu.sig[m], cpu.len[m])) same = 0;
        if (gpu.verdict[m] != SC_FUNC_SUCCESS || cpu.verdict[m] != SC_FUNC_SUCCESS) ok_v = 0;
        if (gpu.verdict_bad[m] != cpu.verdict_bad[m]) ok_bad = 0;
        if (tamper_must_fail && gpu.verdict_bad[m] == SC_FUNC_SUCCESS) ok_bad = 0;
    }
    snprintf(detail, sizeof detail, "%d messages, %zu-byte signatures; reference code over CPU tables %.1f ms, over the engine's tables %.1f ms", NMSG, cpu.len[NMSG - 1], t_cpu * 1e3, t_gpu * 1e3);
    snprintf(label, sizeof label, "%s: signatures over the engine's tables == over the CPU tables", name); report(label, same, detail);
    snprintf(label, sizeof label, "%s: genuine signatures verify (both)", name); report(label, ok_v, "");
    snprintf(label, sizeof label, "%s: tampered signatures get the same verdict (both)", name); report(label, ok_bad, "");
    /* cross: a signature made over the engine verifies over the CPU tables */
    sc_hip_glue_detach(g);
    snprintf(label, sizeof label, "%s: no engine errors", name);
    snprintf(detail, sizeof detail, "%lu", sc_hip_vtable_errors() - e0);
    report(label, sc_hip_vtable_errors() == e0, detail);
    free_all(&cpu); free_all(&gpu);
    safecrypto_destroy(sc);
}

/* the same key pair from the same CSPRNG seed, CPU tables vs the engine's */
static void keygen_scheme(const char *name, sc_scheme_e scheme, int set, UINT32 flag0)
{
    char label[128];
    UINT32 flags[2] = { flag0, SC_FLAG_NONE };
    uint8_t seed[32];
    sm_state = 0x6b657967656eULL + set;
    for (int i = 0; i < 32; i++) seed[i] = (uint8_t)splitmix64();
    safecrypto_t *a = safecrypto_create(scheme, set, flags), *b = safecrypto_create(scheme, set, flags);
    if (!a || !b) { snprintf(label, sizeof label, "%s keygen: create", name); report(label, 0, ""); return; }
    prng_ctx_t *oa = a->prng_ctx[0], *ob = b->prng_ctx[0];
    prng_ctx_t *pa = seeded_prng(seed, 32), *pb = seeded_prng(seed, 32);
    swap_prng(a, NULL, pa);
    const int ra = safecrypto_keygen(a);
    sc_hip_glue_t *g = sc_hip_glue_attach(b, 0);
    swap_prng(b, g, pb);
    const int rb = g ? safecrypto_keygen(b) : SC_FUNC_FAILURE;
    int same = (ra == SC_FUNC_SUCCESS && rb == SC_FUNC_SUCCESS);
    if (same) same = a->privkey->len == b->privkey->len && a->pubkey->len == b->pubkey->len &&
                     !memcmp(a->privkey->key, b->privkey->key, a->privkey->len) && !memcmp(a->pubkey->key, b->pubkey->key, a->pubkey->len);
    snprintf(label, sizeof label, "%s: key pair generated over the engine's tables == over the CPU tables", name);
    report(label, same, "");
    swap_prng(a, NULL, oa); swap_prng(b, g, ob);
    if (g) sc_hip_glue_detach(g);
    prng_destroy(pa); prng_destroy(pb);
    safecrypto_destroy(a); safecrypto_destroy(b);
}

static void kem_scheme(const char *name, sc_scheme_e scheme, int set, UINT32 flag0)
{
    char label[128];
    UINT32 flags[2] = { flag0, SC_FLAG_NONE };
    safecrypto_t *sc = safecrypto_create(scheme, set, flags);
    if (!sc || SC_FUNC_SUCCESS != safecrypto_keygen(sc)) { snprintf(label, sizeof label, "%s: create/keygen", name); report(label, 0, ""); return; }
    prng_ctx_t *orig = sc->prng_ctx[0];
    int same = 1, dec = 1;
    sm_state = 0x6b656d00ULL + set;
    for (int m = 0; m < 4; m++) {
        uint8_t seed[32];
        for (int i = 0; i < 32; i++) seed[i] = (uint8_t)splitmix64();
        UINT8 *ct[2] = { NULL, NULL }, *key[2] = { NULL, NULL }, *key2[2] = { NULL, NULL };
        size_t ctlen[2] = { 0, 0 }, keylen[2] = { 0, 0 }, key2len[2] = { 0, 0 };
        for (int pass = 0; pass < 2; pass++) {
            sc_hip_glue_t *g = pass ? sc_hip_glue_attach(sc, 0) : NULL;
            if (pass && !g) { same = 0; break; }
            prng_ctx_t *mine = seeded_prng(seed, 32);
            swap_prng(sc, g, mine);
            if (SC_FUNC_SUCCESS != safecrypto_encapsulation(sc, &ct[pass], &ctlen[pass], &key[pass], &keylen[pass])) same = 0;
            else if (SC_FUNC_SUCCESS != safecrypto_decapsulation(sc, ct[pass], ctlen[pass], &key2[pass], &key2len[pass])) dec = 0;
            swap_prng(sc, g, orig);
            prng_destroy(mine);
            if (g) sc_hip_glue_detach(g);
        }
        if (!same || !dec) break;
        if (ctlen[0] != ctlen[1] || memcmp(ct[0], ct[1], ctlen[0]) || keylen[0] != keylen[1] || memcmp(key[0], key[1], keylen[0])) same = 0;
        if (key2len[1] != keylen[1] || memcmp(key2[1], key[1], keylen[1]) || key2len[0] != key2len[1] || memcmp(key2[0], key2[1], key2len[0])) dec = 0;
        for (int pass = 0; pass < 2; pass++) { free(ct[pass]); free(key[pass]); free(key2[pass]); }
    }
    snprintf(label, sizeof label, "%s: ciphertexts and shared keys over the engine's tables == over the CPU tables", name); report(label, same, "");
    snprintf(label, sizeof label, "%s: decapsulation over the engine's tables recovers the key", name); report(label, dec, "");
    safecrypto_destroy(sc);
}

/* outer boundary: the whole Dilithium operation on the device behind scheme-table-shaped entries */
static void outer_dilithium(int set)
{
    char label[160], detail[256];
    UINT32 flags[2] = { SC_FLAG_NONE, SC_FLAG_NONE };
    safecrypto_t *sc = safecrypto_create(SC_SCHEME_SIG_DILITHIUM, set, flags);
    if (!sc || SC_FUNC_SUCCESS != safecrypto_keygen(sc)) { report("outer dilithium: create/keygen", 0, ""); return; }
    enum { B = 256 };
    const size_t cap = sc_hip_glue_dilithium_sig_max(sc);
    UINT8 *msgs = malloc(B * 64), *sigs = calloc(B, cap ? cap : 1);
    UINT32 *mlens = malloc(B * 4), *siglens = calloc(B, 4), *verd = malloc(B * 4);
    sm_state = 0x0d111ULL + set;
    for (size_t i = 0; i < B * 64; i++) msgs[i] = (uint8_t)splitmix64();
    for (int i = 0; i < B; i++) mlens[i] = (i == 0) ? 0 : (i == 1) ? 1 : 64;
    int same1 = 1, ver1 = 1, alloc_ok = 1;
    /* one message at a time, both buffer conventions of the scheme table */
    for (int i = 0; i < 4; i++) {
        UINT8 *s_ref = NULL, *s_hip = NULL; size_t l_ref = 0, l_hip = 0;
        if (SC_FUNC_SUCCESS != safecrypto_sign(sc, msgs + 64 * i, mlens[i], &s_ref, &l_ref)) same1 = 0;
        if (SC_FUNC_SUCCESS != sc_hip_glue_dilithium_sign(sc, msgs + 64 * i, mlens[i], &s_hip, &l_hip)) same1 = 0;
        if (same1 && (l_ref != l_hip || memcmp(s_ref, s_hip, l_ref))) same1 = 0;
        if (same1) {
            if (SC_FUNC_SUCCESS != sc_hip_glue_dilithium_verify(sc, msgs + 64 * i, mlens[i], s_ref, l_ref)) ver1 = 0;     /* reference signature, engine verifier */
            if (SC_FUNC_SUCCESS != safecrypto_verify(sc, msgs + 64 * i, mlens[i], s_hip, l_hip)) ver1 = 0;                 /* engine signature, reference verifier */
            s_hip[l_hip / 2] ^= 0x10;
            if (SC_FUNC_SUCCESS == sc_hip_glue_dilithium_verify(sc, msgs + 64 * i, mlens[i], s_hip, l_hip)) ver1 = 0;
            UINT8 *own = malloc(cap); size_t ol = cap;                                                                     /* caller-supplied buffer */
            if (SC_FUNC_SUCCESS != sc_hip_glue_dilithium_sign(sc, msgs + 64 * i, mlens[i], &own, &ol) || ol != l_ref || memcmp(own, s_ref, ol)) alloc_ok = 0;
            ol = 8;
            if (SC_FUNC_SUCCESS == sc_hip_glue_dilithium_sign(sc, msgs + 64 * i, mlens[i], &own, &ol)) alloc_ok = 0;       /* too small: refused */
            free(own);
        }
        free(s_ref); free(s_hip);
    }
    snprintf(label, sizeof label, "outer boundary, Dilithium set %d: sc_hip_glue_dilithium_sign == safecrypto_sign", set); report(label, same1, "");
    snprintf(label, sizeof label, "outer boundary, Dilithium set %d: cross verification and tamper rejection", set); report(label, ver1, "");
    snprintf(label, sizeof label, "outer boundary, Dilithium set %d: caller-supplied and undersized buffers", set); report(label, alloc_ok, "");
    /* a batch beside safecrypto_sign */
    double t0 = now();
    int rc = sc_hip_glue_dilithium_sign_batch(sc, B, msgs, 64, mlens, sigs, cap, siglens);
    const double t_hip = now() - t0;
    int same = (rc == SC_FUNC_SUCCESS);
    t0 = now();
    for (int i = 0; i < B && same; i++) {
        UINT8 *s_ref = NULL; size_t l_ref = 0;
        if (SC_FUNC_SUCCESS != safecrypto_sign(sc, msgs + 64 * i, mlens[i], &s_ref, &l_ref) || l_ref != siglens[i] || memcmp(s_ref, sigs + (size_t)i * cap, l_ref)) same = 0;
        free(s_ref);
    }
    const double t_ref = now() - t0;
    snprintf(detail, sizeof detail, "%d signatures: engine %.1f ms (host buffers, first call), reference %.1f ms", (int)B, t_hip * 1e3, t_ref * 1e3);
    snprintf(label, sizeof label, "outer boundary, Dilithium set %d: a batch of %d == %d calls of safecrypto_sign", set, (int)B, (int)B); report(label, same, detail);
    rc = sc_hip_glue_dilithium_verify_batch(sc, B, msgs, 64, mlens, sigs, cap, siglens, verd);
    int allok = (rc == SC_FUNC_SUCCESS);
    for (int i = 0; i < B && allok; i++) if (verd[i] != 0) allok = 0;
    snprintf(label, sizeof label, "outer boundary, Dilithium set %d: the batch verifies", set); report(label, allok, "");
    sc_hip_glue_forget(sc);
    free(msgs); free(sigs); free(mlens); free(siglens); free(verd);
    safecrypto_destroy(sc);
}

/* outer boundary, the other signature schemes: sc_hip_glue_<scheme>_sign / _verify against safecrypto_sign / safecrypto_verify from the same CSPRNG state,
 * and the batched entries */
/* SC_DROPIN_BATCH=<n> shrinks the batched sections (the CPU build of the library under tests/emu runs a signature in seconds, not microseconds) */
static int batch_size(int dflt) { const char *e = getenv("SC_DROPIN_BATCH"); const int v = e ? atoi(e) : 0; return (v >= 8 && v < dflt) ? v : dflt; }
typedef SINT32 (*sign_fn)(safecrypto_t *, const UINT8 *, size_t, UINT8 **, size_t *);
typedef SINT32 (*verify_fn)(safecrypto_t *, const UINT8 *, size_t, const UINT8 *, size_t);
static void outer_signature(const char *name, sc_scheme_e scheme, int set, UINT32 flag0, sign_fn gsign, verify_fn gverify, int tamper_must_fail)
{
    char label[200], detail[256];
    UINT32 flags[2] = { flag0, SC_FLAG_NONE };
    safecrypto_t *sc = safecrypto_create(scheme, set, flags);
    install_hp_table(sc);                                                  /* (only a 128- / 192-bit sampler's context gets one) */
    if (!sc || SC_FUNC_SUCCESS != safecrypto_keygen(sc)) { snprintf(label, sizeof label, "outer %s: create/keygen", name); report(label, 0, ""); return; }
    prng_ctx_t *orig = sc->prng_ctx[0];
    int same = 1, cross = 1, bufs = 1, made = 0;
    sm_state = 0x0ce7e500ULL + (uint64_t)scheme * 16 + set;
    for (int m = 0; m < NMSG; m++) {
        uint8_t msg[96], seed[32];
        const size_t mlen = (m == 0) ? 1 : (m == 1) ? 96 : 64;
        for (size_t i = 0; i < mlen; i++) msg[i] = (uint8_t)splitmix64();
        for (int i = 0; i < 32; i++) seed[i] = (uint8_t)splitmix64();
        UINT8 *s_ref = NULL, *s_hip = NULL; size_t l_ref = 0, l_hip = 0;
        prng_ctx_t *p1 = seeded_prng(seed, 32), *p2 = seeded_prng(seed, 32);
        swap_prng(sc, NULL, p1);
        const int r_ref = safecrypto_sign(sc, msg, mlen, &s_ref, &l_ref);
        swap_prng(sc, NULL, p2);
        const int r_hip = gsign(sc, msg, mlen, &s_hip, &l_hip);
        swap_prng(sc, NULL, orig);
        prng_destroy(p1); prng_destroy(p2);
        if (r_ref != r_hip) same = 0;                                     /* Falcon: the reference fails on its norm bound, so must the engine */
        if (SC_FUNC_SUCCESS == r_ref && SC_FUNC_SUCCESS == r_hip) {
            made++;
            if (l_ref != l_hip || memcmp(s_ref, s_hip, l_ref)) same = 0;
            if (SC_FUNC_SUCCESS != gverify(sc, msg, mlen, s_ref, l_ref)) cross = 0;            /* reference signature, engine verifier */
            if (SC_FUNC_SUCCESS != safecrypto_verify(sc, msg, mlen, s_hip, l_hip)) cross = 0;   /* engine signature, reference verifier */
            s_hip[l_hip / 3] ^= 0x04;
            const int v_ref = safecrypto_verify(sc, msg, mlen, s_hip, l_hip), v_hip = gverify(sc, msg, mlen, s_hip, l_hip);
            if ((v_ref == SC_FUNC_SUCCESS) != (v_hip == SC_FUNC_SUCCESS)) cross = 0;
            if (tamper_must_fail && v_hip == SC_FUNC_SUCCESS) cross = 0;
            if (m == 2) {                                                  /* caller-supplied and undersized buffers */
                UINT8 *own = malloc(l_ref + 64); size_t ol = l_ref + 64;
                prng_ctx_t *p3 = seeded_prng(seed, 32);
                swap_prng(sc, NULL, p3);
                if (SC_FUNC_SUCCESS != gsign(sc, msg, mlen, &own, &ol) || ol != l_ref || memcmp(own, s_ref, ol)) bufs = 0;
                ol = 8;
                if (SC_FUNC_SUCCESS == gsign(sc, msg, mlen, &own, &ol)) bufs = 0;
                swap_prng(sc, NULL, orig);
                prng_destroy(p3);
                free(own);
            }
        }
        free(s_ref); free(s_hip);
    }
    snprintf(detail, sizeof detail, "%d of %d messages signed by both", made, NMSG);
    snprintf(label, sizeof label, "outer boundary, %s: sc_hip_glue sign == safecrypto_sign from the same CSPRNG state", name); report(label, same && made >= NMSG - 2, detail);
    snprintf(label, sizeof label, "outer boundary, %s: cross verification and tampered signatures", name); report(label, cross, "");
    snprintf(label, sizeof label, "outer boundary, %s: caller-supplied and undersized buffers", name); report(label, bufs, "");
    /* a batch: tapes drawn from one CSPRNG in message order; every signature verifies under the reference, message 0 equals the one-message call from the same state */
    const int B = batch_size(192);
    const size_t cap = sc_hip_glue_sig_bytes(sc), row = (cap + 15) & ~(size_t)15;
    UINT8 *msgs = malloc(B * 64), *sigs = calloc(B, row ? row : 1);
    UINT32 *mlens = malloc(B * 4), *siglens = calloc(B, 4), *st = calloc(B, 4), *verd = malloc(B * 4);
    for (size_t i = 0; i < B * 64; i++) msgs[i] = (uint8_t)splitmix64();
    for (int i = 0; i < B; i++) mlens[i] = (i == 1) ? 1 : 64;
    uint8_t seed[32];
    for (int i = 0; i < 32; i++) seed[i] = (uint8_t)splitmix64();
    const unsigned attempts = (SC_SCHEME_SIG_RING_TESLA == scheme) ? 24 : 8;
    prng_ctx_t *pb = seeded_prng(seed, 32), *p1 = seeded_prng(seed, 32);
    swap_prng(sc, NULL, pb);
    double t0 = now();
    const int rc = sc_hip_glue_sign_batch(sc, B, msgs, 64, mlens, attempts, sigs, row, siglens, st);
    const double t_hip = now() - t0;
    swap_prng(sc, NULL, p1);
    UINT8 *s0 = NULL; size_t l0 = 0;
    const int r0 = gsign(sc, msgs, mlens[0], &s0, &l0);
    swap_prng(sc, NULL, orig);
    prng_destroy(pb); prng_destroy(p1);
    int ok = (rc == SC_FUNC_SUCCESS), signed_ = 0;
    for (int i = 0; i < B && ok; i++) {
        if (st[i] != 1) continue;
        signed_++;
        if (SC_FUNC_SUCCESS != safecrypto_verify(sc, msgs + 64 * i, mlens[i], sigs + (size_t)i * row, siglens[i])) ok = 0;
    }
    if (ok && st[0] == 1 && (r0 != SC_FUNC_SUCCESS || l0 != siglens[0] || memcmp(s0, sigs, l0))) ok = 0;
    snprintf(detail, sizeof detail, "%d of %d signed (%.1f ms, host buffers), all verified by safecrypto_verify", signed_, (int)B, t_hip * 1e3);
    snprintf(label, sizeof label, "outer boundary, %s: a batch of %d through sc_hip_glue_sign_batch", name, (int)B); report(label, ok && signed_ >= B - B / 8, detail);
    int vok = (SC_FUNC_SUCCESS == sc_hip_glue_verify_batch(sc, B, msgs, 64, mlens, sigs, row, siglens, verd));
    for (int i = 0; i < B && vok; i++) if ((st[i] == 1) != (verd[i] == 0)) vok = 0;
    snprintf(label, sizeof label, "outer boundary, %s: the batch verifies through sc_hip_glue_verify_batch", name); report(label, vok, "");
    free(s0); free(msgs); free(sigs); free(mlens); free(siglens); free(st); free(verd);
    sc_hip_glue_forget(sc);
    safecrypto_destroy(sc);
}

/* outer boundary, Kyber: encapsulation / decapsulation rows; prng_mem consumption is exact, so a batch equals the same number of sequential calls */
static void outer_kyber(int set)
{
    char label[200], detail[256];
    UINT32 flags[2] = { SC_FLAG_NONE, SC_FLAG_NONE };
    safecrypto_t *sc = safecrypto_create(SC_SCHEME_KEM_KYBER, set, flags);
    if (!sc || SC_FUNC_SUCCESS != safecrypto_keygen(sc)) { report("outer kyber: create/keygen", 0, ""); return; }
    prng_ctx_t *orig = sc->prng_ctx[0];
    enum { B = 256 };
    const size_t ctb = sc_hip_glue_kyber_ct_bytes(sc), row = (ctb + 15) & ~(size_t)15;
    uint8_t seed[32];
    sm_state = 0x0c1be700ULL + set;
    for (int i = 0; i < 32; i++) seed[i] = (uint8_t)splitmix64();
    prng_ctx_t *p1 = seeded_prng(seed, 32), *p2 = seeded_prng(seed, 32), *p3 = seeded_prng(seed, 32);
    UINT8 *cts = calloc(B, row ? row : 1), *keys = calloc(B, 32), *keys2 = calloc(B, 32);
    swap_prng(sc, NULL, p1);
    double t0 = now();
    int rc = sc_hip_glue_kyber_encapsulation_batch(sc, B, cts, row, keys);
    const double t_hip = now() - t0;
    int same = (rc == SC_FUNC_SUCCESS && ctb > 0), dec = 1, one = 1;
    swap_prng(sc, NULL, p2);
    t0 = now();
    for (int i = 0; i < B && same; i++) {
        UINT8 *c = NULL, *k = NULL, *k2 = NULL; size_t cl = 0, kl = 0, k2l = 0;
        if (SC_FUNC_SUCCESS != safecrypto_encapsulation(sc, &c, &cl, &k, &kl) || cl != ctb || kl != 32 || memcmp(c, cts + (size_t)i * row, cl) || memcmp(k, keys + 32 * i, 32)) same = 0;
        if (same && i < 8) {                                              /* the reference decapsulates the engine's ciphertext; the engine the reference's */
            if (SC_FUNC_SUCCESS != safecrypto_decapsulation(sc, cts + (size_t)i * row, ctb, &k2, &k2l) || k2l != 32 || memcmp(k2, k, 32)) dec = 0;
            UINT8 *k3 = NULL; size_t k3l = 0;
            if (SC_FUNC_SUCCESS != sc_hip_glue_kyber_decapsulation(sc, c, cl, &k3, &k3l) || k3l != 32 || memcmp(k3, k, 32)) dec = 0;
            free(k3);
        }
        free(c); free(k); free(k2);
    }
    const double t_ref = now() - t0;
    swap_prng(sc, NULL, p3);
    {   /* the one-value row with both buffer conventions */
        UINT8 *c = NULL, *k = NULL; size_t cl = 0, kl = 0;
        if (SC_FUNC_SUCCESS != sc_hip_glue_kyber_encapsulation(sc, &c, &cl, &k, &kl) || cl != ctb || kl != 32 || memcmp(c, cts, cl) || memcmp(k, keys, 32)) one = 0;
        UINT8 small[8], *ps = small; size_t sl = sizeof small;
        if (one && SC_FUNC_SUCCESS == sc_hip_glue_kyber_decapsulation(sc, c, cl, &ps, &sl)) one = 0;     /* 8-byte key buffer: refused */
        free(c); free(k);
    }
    swap_prng(sc, NULL, orig);
    rc = sc_hip_glue_kyber_decapsulation_batch(sc, B, cts, row, keys2);
    if (rc != SC_FUNC_SUCCESS || memcmp(keys, keys2, 32 * B)) dec = 0;
    snprintf(detail, sizeof detail, "%d values: engine %.1f ms (host buffers, first call), reference %.1f ms", (int)B, t_hip * 1e3, t_ref * 1e3);
    snprintf(label, sizeof label, "outer boundary, Kyber set %d: a batch of %d == %d calls of safecrypto_encapsulation (ciphertexts and keys)", set, (int)B, (int)B); report(label, same, detail);
    snprintf(label, sizeof label, "outer boundary, Kyber set %d: decapsulation, both directions and the batch", set); report(label, dec, "");
    snprintf(label, sizeof label, "outer boundary, Kyber set %d: sc_hip_glue_kyber_encapsulation == the batch's first value; undersized key buffer refused", set); report(label, one, "");
    prng_destroy(p1); prng_destroy(p2); prng_destroy(p3);
    free(cts); free(keys); free(keys2);
    sc_hip_glue_forget(sc);
    safecrypto_destroy(sc);
}

/* dropin_driver [substring]: with an argument, only the sections whose name contains it */
static const char *g_only = NULL;
static int want(const char *name) { return !g_only || strstr(name, g_only) != NULL; }
#define signature_scheme(name, ...) do { if (want(name)) (signature_scheme)(name, __VA_ARGS__); } while (0)
#define kem_scheme(name, ...) do { if (want(name)) (kem_scheme)(name, __VA_ARGS__); } while (0)
#define keygen_scheme(name, ...) do { if (want(name)) (keygen_scheme)(name, __VA_ARGS__); } while (0)
#define outer_signature(name, ...) do { if (want(name)) (outer_signature)(name, __VA_ARGS__); } while (0)

int main(int argc, char **argv)
{
    if (argc > 1) g_only = argv[1];
    if (sc_hip_device_count() < 1) { printf("{\"error\": \"no HIP device: the drop-in proof needs the GPU\"}\n"); return 2; }
    const double t0 = now();
    signature_scheme("Dilithium set 3 (Dilithium-III)", SC_SCHEME_SIG_DILITHIUM, 3, SC_FLAG_NONE, 1);
    signature_scheme("Dilithium set 1", SC_SCHEME_SIG_DILITHIUM, 1, SC_FLAG_NONE, 1);
    signature_scheme("BLISS-B-I", SC_SCHEME_SIG_BLISS, 1, SC_FLAG_0_SAMPLE_64BIT, 1);
    signature_scheme("BLISS-B-IV", SC_SCHEME_SIG_BLISS, 4, SC_FLAG_0_SAMPLE_64BIT, 1);
    signature_scheme("BLISS-B-I with the 128-bit sampler", SC_SCHEME_SIG_BLISS, 1, SC_FLAG_0_SAMPLE_CDF | SC_FLAG_0_SAMPLE_128BIT, 1);
    signature_scheme("ring-TESLA set 0", SC_SCHEME_SIG_RING_TESLA, 0, SC_FLAG_NONE, 0);     /* its oracle ignores z inside the bound: SURVEY / DESIGN 4.10 */
    kem_scheme("Kyber KEM set 1", SC_SCHEME_KEM_KYBER, 1, SC_FLAG_NONE);
    kem_scheme("ENS-KEM set 0", SC_SCHEME_KEM_ENS, 0, SC_FLAG_NONE);
    keygen_scheme("Dilithium set 3", SC_SCHEME_SIG_DILITHIUM, 3, SC_FLAG_NONE);
    keygen_scheme("Kyber KEM set 1", SC_SCHEME_KEM_KYBER, 1, SC_FLAG_NONE);
    keygen_scheme("BLISS-B-I", SC_SCHEME_SIG_BLISS, 1, SC_FLAG_0_SAMPLE_64BIT);
    if (want("outer Dilithium")) outer_dilithium(3);
    outer_signature("BLISS-B-I", SC_SCHEME_SIG_BLISS, 1, SC_FLAG_0_SAMPLE_64BIT, sc_hip_glue_bliss_b_sign, sc_hip_glue_bliss_b_verify, 1);
    outer_signature("BLISS-B-I with the 128-bit sampler of the reference's own harness (func_alg_bliss_b.c:118)", SC_SCHEME_SIG_BLISS, 1, SC_FLAG_0_SAMPLE_CDF | SC_FLAG_0_SAMPLE_128BIT,
                    sc_hip_glue_bliss_b_sign, sc_hip_glue_bliss_b_verify, 1);
    outer_signature("BLISS-B-I, signatures under the arithmetic coder", SC_SCHEME_SIG_BLISS, 1, SC_FLAG_0_SAMPLE_64BIT | SC_FLAG_0_ENTROPY_BAC, sc_hip_glue_bliss_b_sign, sc_hip_glue_bliss_b_verify, 1);
    outer_signature("BLISS-B-IV, signatures in the static Huffman coding", SC_SCHEME_SIG_BLISS, 4, SC_FLAG_0_SAMPLE_64BIT | SC_FLAG_0_ENTROPY_HUFFMAN, sc_hip_glue_bliss_b_sign, sc_hip_glue_bliss_b_verify, 1);
    outer_signature("ring-TESLA set 0", SC_SCHEME_SIG_RING_TESLA, 0, SC_FLAG_NONE, sc_hip_glue_ring_tesla_sign, sc_hip_glue_ring_tesla_verify, 0);
    outer_signature("Falcon-512", SC_SCHEME_SIG_FALCON, 0, SC_FLAG_NONE, sc_hip_glue_falcon_sign, sc_hip_glue_falcon_verify, 1);
    if (want("outer Kyber")) outer_kyber(1);
    printf("{\"summary\": true, \"checks\": %d, \"failed\": %d, \"vtable_errors\": %lu, \"seconds\": %.2f}\n", g_checks, g_fail, sc_hip_vtable_errors(), now() - t0);
    return g_fail ? 1 : 0;
}
