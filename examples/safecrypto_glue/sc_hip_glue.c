/* sc_hip_glue.c - see sc_hip_glue.h.  Compiled against libsafecrypto's headers; every layout it relies on is the reference's own. */
#include <stddef.h>
#include <stdlib.h>
#include <string.h>

#include "sc_hip_glue.h"
#include "utils/arith/arith.h"
#include "utils/arith/ntt.h"
#include "utils/arith/module_lwe.h"
#include "utils/sampling/sampling.h"
#include "utils/crypto/prng.h"
#include "utils/entropy/huffman.h"
#include "schemes/sig/dilithium/dilithium.h"
#include "schemes/sig/dilithium/dilithium_params.h"
#include "schemes/sig/bliss_b/bliss_b.h"
#include "schemes/sig/bliss_b/bliss_params.h"
#include "schemes/sig/ring_tesla/ring_tesla.h"
#include "schemes/sig/ring_tesla/ring_tesla_params.h"
#include "schemes/sig/falcon/falcon.h"
#include "schemes/sig/falcon/falcon_params.h"
#include "schemes/kem/kyber/kyber_kem.h"
#include "schemes/kem/kyber/kyber_kem_params.h"

#include "sc_hip.h"
#include "sc_hip_vtable.h"

/* the mirrors must BE the reference's structs: checked where both definitions are visible */
_Static_assert(sizeof(sc_hip_utils_arith_ntt_t) == sizeof(utils_arith_ntt_t), "utils_arith_ntt_t mirror");
_Static_assert(offsetof(sc_hip_utils_arith_ntt_t, modn_32) == offsetof(utils_arith_ntt_t, modn_32), "modn_32");
_Static_assert(offsetof(sc_hip_utils_arith_ntt_t, mul_32_pointwise) == offsetof(utils_arith_ntt_t, mul_32_pointwise), "mul_32_pointwise");
_Static_assert(offsetof(sc_hip_utils_arith_ntt_t, fft_32_16) == offsetof(utils_arith_ntt_t, fft_32_16), "fft_32_16");
_Static_assert(offsetof(sc_hip_utils_arith_ntt_t, pwr_32) == offsetof(utils_arith_ntt_t, pwr_32), "pwr_32");
_Static_assert(offsetof(sc_hip_utils_arith_ntt_t, flip_32) == offsetof(utils_arith_ntt_t, flip_32), "flip_32");
_Static_assert(offsetof(sc_hip_utils_arith_ntt_t, normalize_32) == offsetof(utils_arith_ntt_t, normalize_32), "normalize_32");
_Static_assert(offsetof(sc_hip_utils_arith_ntt_t, fwd_ntt_32_32) == offsetof(utils_arith_ntt_t, fwd_ntt_32_32), "fwd_ntt_32_32");
_Static_assert(offsetof(sc_hip_utils_arith_ntt_t, inv_ntt_32_16_large) == offsetof(utils_arith_ntt_t, inv_ntt_32_16_large), "inv_ntt_32_16_large");
_Static_assert(offsetof(sc_hip_utils_arith_ntt_t, slots_limb) == offsetof(utils_arith_ntt_t, modn_limb), "limb third");
_Static_assert(sizeof(sc_hip_utils_arith_poly_t) == sizeof(utils_arith_poly_t), "utils_arith_poly_t mirror");
_Static_assert(offsetof(sc_hip_utils_arith_poly_t, copy_32) == offsetof(utils_arith_poly_t, copy_32), "copy_32");
_Static_assert(offsetof(sc_hip_utils_arith_poly_t, mul_32_scalar) == offsetof(utils_arith_poly_t, mul_32_scalar), "mul_32_scalar");
_Static_assert(offsetof(sc_hip_utils_arith_poly_t, add_32) == offsetof(utils_arith_poly_t, add_32), "add_32");
_Static_assert(offsetof(sc_hip_utils_arith_poly_t, sub_single_32) == offsetof(utils_arith_poly_t, sub_single_32), "sub_single_32");
_Static_assert(offsetof(sc_hip_utils_arith_poly_t, mod_negate_32) == offsetof(utils_arith_poly_t, mod_negate_32), "mod_negate_32");
_Static_assert(offsetof(sc_hip_utils_arith_poly_t, cmp_not_equal_32) == offsetof(utils_arith_poly_t, cmp_not_equal_32), "cmp_not_equal_32");
_Static_assert(offsetof(sc_hip_utils_arith_poly_t, uniform_32) == offsetof(utils_arith_poly_t, uniform_32), "uniform_32");
_Static_assert(sizeof(sc_hip_utils_arith_vec_t) == sizeof(utils_arith_vec_t), "utils_arith_vec_t mirror");
_Static_assert(sizeof(sc_hip_utils_sampling_t) == sizeof(utils_sampling_t), "utils_sampling_t mirror");
_Static_assert(offsetof(sc_hip_utils_sampling_t, vector_32) == offsetof(utils_sampling_t, vector_32), "vector_32");
_Static_assert(offsetof(sc_hip_utils_sampling_t, gauss) == offsetof(utils_sampling_t, gauss), "gauss");
_Static_assert(offsetof(sc_hip_utils_sampling_t, prng_ctx) == offsetof(utils_sampling_t, prng_ctx), "prng_ctx");
_Static_assert(offsetof(sc_hip_utils_sampling_t, discard) == offsetof(utils_sampling_t, discard), "discard");
_Static_assert(sizeof(sc_hip_ntt_params_t) == sizeof(ntt_params_t), "ntt_params_t mirror");
_Static_assert(offsetof(sc_hip_ntt_params_t, u.ntt32.q) == offsetof(ntt_params_t, u.ntt32.q), "ntt32.q");

/* ---- inner boundary ---------------------------------------------------------------------------------------------------------------- */

struct sc_hip_glue {
    safecrypto_t *sc;
    const utils_arith_ntt_t *ntt0; const utils_arith_poly_t *poly0; const utils_arith_vec_t *vec0; utils_sampling_t *gauss0;
    sc_hip_utils_arith_ntt_t ntt; sc_hip_utils_arith_poly_t poly; sc_hip_utils_arith_vec_t vec;
    sc_hip_sampler_t *sampler; sc_hip_utils_sampling_t sampling;
};

/* the private struct behind a 64-bit CDF sampler's `gauss` pointer (src/utils/sampling/gaussian_cdf.c:74-81) */
typedef struct __attribute__((packed)) { UINT64 *cdf; SINT32 cdf_size; SINT32 k; SINT32 use_kl; prng_ctx_t *prng_ctx; } gauss_cdf_64_view_t;

/* ... and behind a 128- / 192-bit one (gauss_cdf_high_t, :59-68): limbs of SC_LIMB_BITS = 64, least significant first */
typedef struct __attribute__((packed)) { UINT64 *cdf_256, *cdf_192, *cdf_128; SINT32 cdf_size; SINT32 k; SINT32 use_kl; prng_ctx_t *prng_ctx; } gauss_cdf_high_view_t;

static uint64_t glue_rng64(void *ctx) { return prng_64((prng_ctx_t *)ctx); }

sc_hip_glue_t *sc_hip_glue_attach(safecrypto_t *sc, int device)
{
    if (NULL == sc || SC_HIP_SUCCESS != sc_hip_vtable_bind_device(device)) return NULL;
    sc_hip_glue_t *g = calloc(1, sizeof *g);
    if (NULL == g) return NULL;
    g->sc = sc; g->ntt0 = sc->sc_ntt; g->poly0 = sc->sc_poly; g->vec0 = sc->sc_vec; g->gauss0 = sc->sc_gauss;
    /* what the engine does not serve (scalar slots, 16-bit and limb thirds) stays the context's own backend */
    if (sc_hip_utils_arith_ntt_fill(&g->ntt, g->ntt0) || sc_hip_utils_arith_poly_fill(&g->poly, g->poly0) || sc_hip_utils_arith_vec_fill(&g->vec)) {
        free(g);
        return NULL;
    }
    if (NULL != g->ntt0) sc->sc_ntt = (const utils_arith_ntt_t *)&g->ntt;
    if (NULL != g->poly0) sc->sc_poly = (const utils_arith_poly_t *)&g->poly;
    if (NULL != g->vec0) sc->sc_vec = (const utils_arith_vec_t *)&g->vec;
    /* a 64-bit CDF sampler (SC_FLAG_0_SAMPLE_64BIT, CDF_GAUSSIAN_SAMPLING) is served with the context's OWN table, so the samples are
     * the reference's for the same prng_64 words on any host */
    if (NULL != g->gauss0 && CDF_GAUSSIAN_SAMPLING == sc->sampling && SAMPLING_64BIT == g->gauss0->precision &&
        SAMPLING_DISABLE_BOOTSTRAP == g->gauss0->bootstrapped && 0 == g->gauss0->discard) {
        const gauss_cdf_64_view_t *v = (const gauss_cdf_64_view_t *)g->gauss0->gauss;
        if (SC_HIP_SUCCESS == sc_hip_sampler_create(g->gauss0->tail, g->gauss0->sigma, v->cdf, v->cdf_size, glue_rng64, g->gauss0->prng_ctx, &g->sampler) &&
            SC_HIP_SUCCESS == sc_hip_utils_sampling_fill(&g->sampling, g->sampler)) {
            g->sampling.dimension = g->gauss0->dimension;
            sc->sc_gauss = (utils_sampling_t *)&g->sampling;
        }
    }
    /* a 128- / 192-bit CDF sampler (SC_FLAG_0_SAMPLE_128BIT / _192BIT: what the reference's BLISS-B harness asks for, test/functional/func_alg_bliss_b.c:118): the
     * context's table - MPFR arithmetic, made by the library as it always was (gauss_cdf_create_high_precision, gaussian_cdf.c:186-330) - is handed to the engine, which
     * runs the search (gaussian_cdf_sample_128 / _192) for the words the context's generator yields, two / three prng_64 a sample */
    else if (NULL != g->gauss0 && CDF_GAUSSIAN_SAMPLING == sc->sampling && (SAMPLING_128BIT == g->gauss0->precision || SAMPLING_192BIT == g->gauss0->precision) &&
             SAMPLING_DISABLE_BOOTSTRAP == g->gauss0->bootstrapped && 0 == g->gauss0->discard) {
        const gauss_cdf_high_view_t *v = (const gauss_cdf_high_view_t *)g->gauss0->gauss;
        const int prec = (int)g->gauss0->precision;
        if (SC_HIP_SUCCESS == sc_hip_sampler_create_hp(prec, g->gauss0->tail, g->gauss0->sigma, 128 == prec ? v->cdf_128 : v->cdf_192, v->cdf_size, glue_rng64,
                                                       g->gauss0->prng_ctx, &g->sampler) &&
            SC_HIP_SUCCESS == sc_hip_utils_sampling_fill(&g->sampling, g->sampler)) {
            g->sampling.dimension = g->gauss0->dimension;
            sc->sc_gauss = (utils_sampling_t *)&g->sampling;
        }
    }
    return g;
}

void sc_hip_glue_refresh_prng(sc_hip_glue_t *g)
{
    if (NULL == g || NULL == g->sampler) return;
    sc_hip_sampler_set_rng(g->sampler, glue_rng64, g->sc->prng_ctx[0]);
    g->sampling.prng_ctx = g->sc->prng_ctx[0];
}

void sc_hip_glue_detach(sc_hip_glue_t *g)
{
    if (NULL == g) return;
    g->sc->sc_ntt = g->ntt0; g->sc->sc_poly = g->poly0; g->sc->sc_vec = g->vec0; g->sc->sc_gauss = g->gauss0;
    if (NULL != g->sampler) sc_hip_sampler_destroy(g->sampler);
    free(g);
}

/* ---- outer boundary: Dilithium ---------------------------------------------------------------------------------------------------------- */

/* a maintainer would add one pointer to safecrypto_t; from outside the library a small registry does the same job */
#define GLUE_MAX 16
static struct { safecrypto_t *sc; sc_hip_ctx_t *ctx; sc_hip_dilithium_t *dil; int have_priv, have_pub;
                sc_hip_bliss_t *bliss; sc_hip_tesla_t *tesla; sc_hip_falcon_t *falcon; sc_hip_kyber_t *kyber; } g_reg[GLUE_MAX];

static int reg_find(safecrypto_t *sc, int create)
{
    int free_slot = -1;
    for (int i = 0; i < GLUE_MAX; i++) {
        if (g_reg[i].sc == sc) return i;
        if (NULL == g_reg[i].sc && free_slot < 0) free_slot = i;
    }
    if (!create || free_slot < 0) return -1;
    memset(&g_reg[free_slot], 0, sizeof g_reg[free_slot]);
    g_reg[free_slot].sc = sc;
    return free_slot;
}

void sc_hip_glue_forget(safecrypto_t *sc)
{
    const int i = reg_find(sc, 0);
    if (i < 0) return;
    if (g_reg[i].dil) sc_hip_dilithium_destroy(g_reg[i].dil);
    if (g_reg[i].bliss) sc_hip_bliss_destroy(g_reg[i].bliss);
    if (g_reg[i].tesla) sc_hip_tesla_destroy(g_reg[i].tesla);
    if (g_reg[i].falcon) sc_hip_falcon_destroy(g_reg[i].falcon);
    if (g_reg[i].kyber) sc_hip_kyber_destroy(g_reg[i].kyber);
    if (g_reg[i].ctx) sc_hip_ctx_destroy(g_reg[i].ctx);
    memset(&g_reg[i], 0, sizeof g_reg[i]);
}

/* the engine handle for this context, its keys loaded from sc->privkey / sc->pubkey (layouts: dilithium.c:810-824) */
static sc_hip_dilithium_t *dil_handle(safecrypto_t *sc, int need_priv, int need_pub)
{
    if (NULL == sc || SC_SCHEME_SIG_DILITHIUM != sc->scheme || NULL == sc->dilithium) return NULL;
    const int i = reg_find(sc, 1);
    if (i < 0) return NULL;
    const dilithium_set_t *P = sc->dilithium->params;
    const size_t n = P->n, k = P->k, l = P->l;
    if (NULL == g_reg[i].dil) {
        if (SC_HIP_SUCCESS != sc_hip_ctx_create(0, &g_reg[i].ctx)) return NULL;
        sc_hip_dilithium_params_t hp = { (int32_t)P->n, (int32_t)P->q, (int32_t)P->q_bits, (int32_t)P->k, (int32_t)P->l, (int32_t)P->d,
                                         (int32_t)P->weight_of_c, (int32_t)P->gamma_1, (int32_t)P->gamma_1_bits, (int32_t)P->gamma_2,
                                         (int32_t)P->eta, (int32_t)P->z_bits, (int32_t)P->beta, (int32_t)P->omega, (int32_t)P->omega_bits, (int32_t)P->set };
        if (SC_HIP_SUCCESS != sc_hip_dilithium_create(g_reg[i].ctx, &hp, &g_reg[i].dil)) { sc_hip_glue_forget(sc); return NULL; }
    }
    if ((need_priv && !g_reg[i].have_priv) || (need_pub && !g_reg[i].have_pub)) {
        const SINT32 *priv = sc->privkey ? (const SINT32 *)sc->privkey->key : NULL, *pub = sc->pubkey ? (const SINT32 *)sc->pubkey->key : NULL;
        const UINT8 *rho = NULL;
        if (need_priv && NULL == priv) return NULL;
        if (need_pub && NULL == pub) return NULL;
        if (NULL != priv) rho = (const UINT8 *)(priv + (NUM_DILITHIUM_PRIVKEY_K * k + l) * n); else rho = (const UINT8 *)(pub + k * n);
        /* the matrix, in the order create_rand_product_32_csprng draws it from rho (module_lwe.c:519-547, 641-660; create_csprng :914-940) */
        SINT32 *A = malloc(sizeof(SINT32) * k * l * n);
        if (NULL == A) return NULL;
        prng_ctx_t *csprng = create_csprng(sc, rho, 32);
        for (size_t j = 0; j < k * l; j++) uniform_random_ring_q_csprng(csprng, A + j * n, n, P->q, P->q_bits);
        prng_destroy(csprng);
        int rc = SC_HIP_SUCCESS;
        if (need_priv && !g_reg[i].have_priv) {
            const UINT8 *tail = (const UINT8 *)(priv + (NUM_DILITHIUM_PRIVKEY_K * k + l) * n);
            rc = sc_hip_dilithium_set_privkey(g_reg[i].dil, priv, priv + l * n, priv + (l + k) * n, tail + 32, tail + 64, A);
            if (SC_HIP_SUCCESS == rc) g_reg[i].have_priv = 1;
        }
        if (SC_HIP_SUCCESS == rc && need_pub && !g_reg[i].have_pub) {
            rc = sc_hip_dilithium_set_pubkey(g_reg[i].dil, pub, (const UINT8 *)(pub + k * n), A);
            if (SC_HIP_SUCCESS == rc) g_reg[i].have_pub = 1;
        }
        free(A);
        if (SC_HIP_SUCCESS != rc) return NULL;
    }
    return g_reg[i].dil;
}

size_t sc_hip_glue_dilithium_sig_max(safecrypto_t *sc)
{
    sc_hip_dilithium_t *d = dil_handle(sc, 0, 0);
    return d ? sc_hip_dilithium_sig_max(d) : 0;
}

SINT32 sc_hip_glue_dilithium_sign_batch(safecrypto_t *sc, size_t batch, const UINT8 *msgs, size_t msg_stride, const UINT32 *mlens,
                                        UINT8 *sigs, size_t sig_stride, UINT32 *siglens)
{
    sc_hip_dilithium_t *d = dil_handle(sc, 1, 0);
    if (NULL == d) return SC_FUNC_FAILURE;
    UINT32 *trials = malloc(sizeof(UINT32) * (batch ? batch : 1));
    if (NULL == trials) return SC_FUNC_FAILURE;
    const int rc = sc_hip_dilithium_sign_host(d, batch, msgs, msg_stride, mlens, 0, sigs, sig_stride, siglens, trials);
    if (SC_HIP_SUCCESS == rc) for (size_t i = 0; i < batch; i++) { sc->stats.sig_num++; sc->stats.sig_num_trials += trials[i]; }   /* as dilithium_sign books them */
    free(trials);
    return SC_HIP_SUCCESS == rc ? SC_FUNC_SUCCESS : SC_FUNC_FAILURE;
}

SINT32 sc_hip_glue_dilithium_verify_batch(safecrypto_t *sc, size_t batch, const UINT8 *msgs, size_t msg_stride, const UINT32 *mlens,
                                          const UINT8 *sigs, size_t sig_stride, const UINT32 *siglens, UINT32 *verdicts)
{
    sc_hip_dilithium_t *d = dil_handle(sc, 0, 1);
    if (NULL == d) return SC_FUNC_FAILURE;
    return SC_HIP_SUCCESS == sc_hip_dilithium_verify_host(d, batch, msgs, msg_stride, mlens, 0, sigs, sig_stride, siglens, verdicts) ? SC_FUNC_SUCCESS : SC_FUNC_FAILURE;
}

SINT32 sc_hip_glue_dilithium_sign(safecrypto_t *sc, const UINT8 *m, size_t m_len, UINT8 **sigret, size_t *siglen)
{
    if (NULL == sigret || NULL == siglen || (NULL == m && m_len > 0)) return SC_FUNC_FAILURE;
    const size_t cap = sc_hip_glue_dilithium_sig_max(sc);
    if (0 == cap) return SC_FUNC_FAILURE;
    UINT8 *tmp = malloc(cap);
    if (NULL == tmp) return SC_FUNC_FAILURE;
    UINT32 ml = (UINT32)m_len, sl = 0;
    UINT8 dummy = 0;
    if (SC_FUNC_SUCCESS != sc_hip_glue_dilithium_sign_batch(sc, 1, m_len ? m : &dummy, m_len ? m_len : 1, &ml, tmp, cap, &sl)) { free(tmp); return SC_FUNC_FAILURE; }
    if (NULL == *sigret || 0 == *siglen) {                               /* packer.c:142-176: the callee allocates, the caller frees */
        *sigret = realloc(tmp, sl ? sl : 1);
        if (NULL == *sigret) { free(tmp); return SC_FUNC_FAILURE; }
    } else {
        if (*siglen < sl) { free(tmp); return SC_FUNC_FAILURE; }
        memcpy(*sigret, tmp, sl);
        free(tmp);
    }
    *siglen = sl;
    return SC_FUNC_SUCCESS;
}

SINT32 sc_hip_glue_dilithium_verify(safecrypto_t *sc, const UINT8 *m, size_t m_len, const UINT8 *sigbuf, size_t siglen)
{
    if (NULL == sigbuf || (NULL == m && m_len > 0)) return SC_FUNC_FAILURE;
    const size_t cap = sc_hip_glue_dilithium_sig_max(sc);
    if (0 == cap) return SC_FUNC_FAILURE;
    /* the engine reads whole rows of sc_hip_dilithium_sig_max() bytes: give it one */
    UINT8 *row = calloc(1, cap);
    if (NULL == row) return SC_FUNC_FAILURE;
    memcpy(row, sigbuf, siglen < cap ? siglen : cap);
    UINT32 ml = (UINT32)m_len, sl = (UINT32)siglen, verdict = 1;
    UINT8 dummy = 0;
    const SINT32 rc = sc_hip_glue_dilithium_verify_batch(sc, 1, m_len ? m : &dummy, m_len ? m_len : 1, &ml, row, cap, &sl, &verdict);
    free(row);
    if (SC_FUNC_SUCCESS != rc) return SC_FUNC_FAILURE;
    return verdict == 0 ? SC_FUNC_SUCCESS : SC_FUNC_FAILURE;
}

/* ---- outer boundary: BLISS-B, ring-TESLA, Falcon, Kyber --------------------------------------------------------------------------------------- */

/* hands `len` bytes to the caller under the packer's buffer rules (packer.c:142-176): *ret == NULL or *retlen == 0 -> malloc'ed for the caller */
static SINT32 emit(const UINT8 *buf, size_t len, UINT8 **ret, size_t *retlen)
{
    if (NULL == ret || NULL == retlen) return SC_FUNC_FAILURE;
    if (NULL == *ret || 0 == *retlen) {
        UINT8 *p = malloc(len ? len : 1);
        if (NULL == p) return SC_FUNC_FAILURE;
        memcpy(p, buf, len);
        *ret = p;
    } else {
        if (*retlen < len) return SC_FUNC_FAILURE;
        memcpy(*ret, buf, len);
    }
    *retlen = len;
    return SC_FUNC_SUCCESS;
}

static int reg_ctx(safecrypto_t *sc)
{
    const int i = reg_find(sc, 1);
    if (i < 0) return -1;
    if (NULL == g_reg[i].ctx && SC_HIP_SUCCESS != sc_hip_ctx_create(0, &g_reg[i].ctx)) return -1;
    return i;
}

/* engine handles, keys loaded from the context on first use */
/* the static table entropy_huffman.c:22-34 walks for signed fields of `bits` bits */
static const huffman_table_t *glue_huff_table(int bits)
{
    const int hb = bits < 7 ? bits : 7;
    return hb == 7 ? huff_table_gaussian_6 : hb == 6 ? huff_table_gaussian_5 : hb == 5 ? huff_table_gaussian_4 : hb == 4 ? huff_table_gaussian_3 : huff_table_gaussian_2;
}
/* the context's signature coding on the engine's handle: raw fields, the static Huffman tables of the library, or the arithmetic coder with the two
 * distributions the context built (bliss_b.c:397-398) */
static int bliss_coding(safecrypto_t *sc, sc_hip_bliss_t *h)
{
    const bliss_set_t *P = sc->bliss->params;
    if (SC_ENTROPY_NONE == sc->coding_signature.type) return sc_hip_bliss_set_entropy(h, 0, NULL, NULL, 0, NULL, NULL, 0);
    if (SC_ENTROPY_BAC == sc->coding_signature.type) {
        if (NULL == sc->dist[2] || NULL == sc->dist[3]) return SC_HIP_FAILURE;
        return sc_hip_bliss_set_entropy_bac(h, (const uint64_t *)sc->dist[2], (const uint64_t *)sc->dist[3]);
    }
    if (SC_ENTROPY_HUFFMAN_STATIC == sc->coding_signature.type) {
        const huffman_table_t *t[2] = { glue_huff_table(P->z1_bits), glue_huff_table(P->z2_bits) };
        uint32_t codes[2][64]; uint16_t bits[2][64];
        for (int k = 0; k < 2; k++) {
            if (t[k]->depth > 64) return SC_HIP_FAILURE;
            for (UINT32 i = 0; i < t[k]->depth; i++) { codes[k][i] = t[k]->codes[i].code; bits[k][i] = (uint16_t)t[k]->codes[i].bits; }
        }
        return sc_hip_bliss_set_entropy(h, 1, codes[0], bits[0], t[0]->depth, codes[1], bits[1], t[1]->depth);
    }
    return SC_HIP_FAILURE;
}

static sc_hip_bliss_t *bliss_handle(safecrypto_t *sc, int need_priv)
{
    if (NULL == sc || SC_SCHEME_SIG_BLISS != sc->scheme || NULL == sc->bliss || NULL == sc->pubkey || NULL == sc->pubkey->key) return NULL;
    if (SC_ENTROPY_NONE != sc->coding_signature.type && SC_ENTROPY_BAC != sc->coding_signature.type && SC_ENTROPY_HUFFMAN_STATIC != sc->coding_signature.type) return NULL;
    /* 64-bit samplers, and the 128- / 192-bit ones the reference's own BLISS-B harness creates (func_alg_bliss_b.c:118); 32 and 256 bits stay on the reference path */
    if (need_priv && (NULL == sc->sc_gauss || CDF_GAUSSIAN_SAMPLING != sc->sampling ||
                      (SAMPLING_64BIT != sc->sc_gauss->precision && SAMPLING_128BIT != sc->sc_gauss->precision && SAMPLING_192BIT != sc->sc_gauss->precision))) return NULL;
    if (need_priv && (NULL == sc->privkey || NULL == sc->privkey->key)) return NULL;
    const int i = reg_ctx(sc);
    if (i < 0) return NULL;
    if (NULL == g_reg[i].bliss || (need_priv && !g_reg[i].have_priv)) {
        const bliss_set_t *P = sc->bliss->params;
        sc_hip_bliss_params_t hp;
        memset(&hp, 0, sizeof hp);
        hp.n = P->n; hp.q = P->q; hp.q_bits = P->q_bits; hp.n_bits = P->n_bits; hp.d = P->d; hp.p = P->p; hp.kappa = P->kappa; hp.b_inf = P->b_inf;
        hp.b_l2 = (int32_t)P->b_l2; hp.z1_bits = P->z1_bits; hp.z2_bits = P->z2_bits; hp.set = P->set; hp.sigma = P->sig; hp.m = P->m; hp.tail = 13.42f;   /* bliss_b.c:342-344 */
        if (NULL == g_reg[i].bliss) {
            if (SC_HIP_SUCCESS != sc_hip_bliss_create(g_reg[i].ctx, &hp, &g_reg[i].bliss)) return NULL;
            /* the context's OWN sampler table (long double expl is host-dependent in its last bit): the samples are the reference's on any host */
            if (NULL != sc->sc_gauss && CDF_GAUSSIAN_SAMPLING == sc->sampling && SAMPLING_64BIT == sc->sc_gauss->precision) {
                const gauss_cdf_64_view_t *v = (const gauss_cdf_64_view_t *)sc->sc_gauss->gauss;
                if (SC_HIP_SUCCESS != sc_hip_bliss_set_cdt_table(g_reg[i].bliss, v->cdf, v->cdf_size)) goto drop;
            }
            else if (NULL != sc->sc_gauss && CDF_GAUSSIAN_SAMPLING == sc->sampling && (SAMPLING_128BIT == sc->sc_gauss->precision || SAMPLING_192BIT == sc->sc_gauss->precision)) {
                /* the table stays the library's (MPFR arithmetic, gaussian_cdf.c:186-330); the engine runs the search, two / three prng_64 draws a sample */
                const gauss_cdf_high_view_t *v = (const gauss_cdf_high_view_t *)sc->sc_gauss->gauss;
                const int prec = (int)sc->sc_gauss->precision;
                if (SC_HIP_SUCCESS != sc_hip_bliss_set_cdt_table_hp(g_reg[i].bliss, prec, 128 == prec ? v->cdf_128 : v->cdf_192, v->cdf_size)) goto drop;
            }
            if (SC_HIP_SUCCESS != bliss_coding(sc, g_reg[i].bliss)) goto drop;
        }
        const SINT16 *priv = (sc->privkey && sc->privkey->key) ? (const SINT16 *)sc->privkey->key : NULL;
        if (SC_HIP_SUCCESS != sc_hip_bliss_set_key(g_reg[i].bliss, priv, priv ? priv + P->n : NULL, (const SINT16 *)sc->pubkey->key)) goto drop;
        g_reg[i].have_priv = priv != NULL; g_reg[i].have_pub = 1;
    }
    return g_reg[i].bliss;
drop:
    /* a handle that is only partly set up (host-tabulated table, raw coding) must not be found by the next call, which would skip the set-up and sign in
     * the wrong format: destroyed and forgotten, as the other schemes' handles are on a partial failure */
    sc_hip_bliss_destroy(g_reg[i].bliss);
    g_reg[i].bliss = NULL; g_reg[i].have_priv = g_reg[i].have_pub = 0;
    return NULL;
}

static sc_hip_tesla_t *tesla_handle(safecrypto_t *sc)
{
    if (NULL == sc || SC_SCHEME_SIG_RING_TESLA != sc->scheme || NULL == sc->ring_tesla) return NULL;
    if (SC_ENTROPY_NONE != sc->coding_signature.type) return NULL;
    const int i = reg_ctx(sc);
    if (i < 0) return NULL;
    if (NULL == g_reg[i].tesla) {
        const ring_tesla_set_t *P = sc->ring_tesla->params;
        sc_hip_tesla_params_t hp = { (int32_t)P->n, (int32_t)P->n_bits, (int32_t)P->q, (int32_t)P->q_bits, (int32_t)P->b, (int32_t)P->b_bits, (int32_t)P->omega,
                                     (int32_t)P->d, (int32_t)P->bound, (int32_t)P->u, (int32_t)P->set, 0 };
        if (SC_HIP_SUCCESS != sc_hip_tesla_create(g_reg[i].ctx, &hp, &g_reg[i].tesla)) return NULL;
        const SINT16 *priv = (sc->privkey && sc->privkey->key) ? (const SINT16 *)sc->privkey->key : NULL;
        const SINT32 *pub = (sc->pubkey && sc->pubkey->key) ? (const SINT32 *)sc->pubkey->key : NULL;
        const size_t n = P->n;
        if (SC_HIP_SUCCESS != sc_hip_tesla_set_key(g_reg[i].tesla, 0 == P->set ? a1_0 : a1_1, 0 == P->set ? a2_0 : a2_1, priv, priv ? priv + n : NULL,
                                                   priv ? priv + 2 * n : NULL, pub, pub ? pub + n : NULL)) { sc_hip_tesla_destroy(g_reg[i].tesla); g_reg[i].tesla = NULL; return NULL; }
        g_reg[i].have_priv = priv != NULL; g_reg[i].have_pub = pub != NULL;
    }
    return g_reg[i].tesla;
}

static sc_hip_falcon_t *falcon_handle(safecrypto_t *sc)
{
    if (NULL == sc || SC_SCHEME_SIG_FALCON != sc->scheme || NULL == sc->falcon) return NULL;
    if (SC_ENTROPY_NONE != sc->coding_signature.type) return NULL;
    const int i = reg_ctx(sc);
    if (i < 0) return NULL;
    if (NULL == g_reg[i].falcon) {
        const falcon_set_t *P = sc->falcon->params;
        sc_hip_falcon_params_t hp;
        memset(&hp, 0, sizeof hp);
        hp.n = P->n; hp.q = P->q; hp.q_bits = P->q_bits; hp.n_bits = P->n_bits; hp.set = P->set; hp.bd = P->bd;
        if (SC_HIP_SUCCESS != sc_hip_falcon_create(g_reg[i].ctx, &hp, &g_reg[i].falcon)) return NULL;
        int ok = 1;
        if (sc->privkey && sc->privkey->key && sc->falcon->sk) { ok = SC_HIP_SUCCESS == sc_hip_falcon_set_privkey(g_reg[i].falcon, (const double *)sc->falcon->sk); g_reg[i].have_priv = ok; }
        if (ok && sc->pubkey && sc->pubkey->key) { ok = SC_HIP_SUCCESS == sc_hip_falcon_set_pubkey(g_reg[i].falcon, (const SINT16 *)sc->pubkey->key); g_reg[i].have_pub = ok; }
        if (!ok) { sc_hip_falcon_destroy(g_reg[i].falcon); g_reg[i].falcon = NULL; return NULL; }
    }
    return g_reg[i].falcon;
}

static sc_hip_kyber_t *kyber_handle(safecrypto_t *sc, int need_priv)
{
    if (NULL == sc || SC_SCHEME_KEM_KYBER != sc->scheme || NULL == sc->kyber || NULL == sc->pubkey || NULL == sc->pubkey->key) return NULL;
    if (SC_ENTROPY_NONE != sc->coding_encryption.type) return NULL;
    if (need_priv && (NULL == sc->privkey || NULL == sc->privkey->key)) return NULL;
    const int i = reg_ctx(sc);
    if (i < 0) return NULL;
    const kyber_set_t *P = sc->kyber->params;
    const size_t n = P->n, k = P->k;
    if (NULL == g_reg[i].kyber) {
        sc_hip_kyber_params_t hp;
        memset(&hp, 0, sizeof hp);
        hp.n = (int32_t)P->n; hp.q = (int32_t)P->q; hp.q_bits = (int32_t)P->q_bits; hp.q_inv = (uint32_t)P->q_inv; hp.q_norm = (int32_t)P->q_norm; hp.k = (int32_t)P->k;
        hp.eta = (int32_t)P->eta; hp.d_u = (int32_t)P->d_u; hp.d_v = (int32_t)P->d_v; hp.d_t = (int32_t)P->d_t; hp.set = (int32_t)P->set;
        if (SC_HIP_SUCCESS != sc_hip_kyber_create(g_reg[i].ctx, &hp, &g_reg[i].kyber)) return NULL;
        /* public key (kyber_kem.c:636-650): t | NTT(t) | rho; the matrix in the order create_rand_product draws it from rho (module_lwe.c:519-547) */
        const SINT32 *t = (const SINT32 *)sc->pubkey->key, *t_ntt = t + k * n;
        const UINT8 *rho = (const UINT8 *)(t_ntt + k * n);
        SINT32 *A = malloc(sizeof(SINT32) * k * k * n);
        if (NULL == A) return NULL;
        prng_ctx_t *csprng = create_csprng(sc, rho, 32);
        for (size_t j = 0; j < k * k; j++) uniform_random_ring_q_csprng(csprng, A + j * n, n, P->q, P->q_bits);
        prng_destroy(csprng);
        const int ok = SC_HIP_SUCCESS == sc_hip_kyber_set_pubkey(g_reg[i].kyber, t_ntt, A) && SC_HIP_SUCCESS == sc_hip_kyber_set_kem_pubkey(g_reg[i].kyber, t, rho);
        free(A);
        if (!ok) { sc_hip_kyber_destroy(g_reg[i].kyber); g_reg[i].kyber = NULL; return NULL; }
        g_reg[i].have_pub = 1;
    }
    if (need_priv && !g_reg[i].have_priv) {
        if (SC_HIP_SUCCESS != sc_hip_kyber_set_privkey(g_reg[i].kyber, (const SINT32 *)sc->privkey->key)) return NULL;    /* NTT(s), kyber_kem.c:624-639 */
        g_reg[i].have_priv = 1;
    }
    return g_reg[i].kyber;
}

size_t sc_hip_glue_sig_bytes(safecrypto_t *sc)
{
    if (NULL == sc) return 0;
    if (SC_SCHEME_SIG_BLISS == sc->scheme) { sc_hip_bliss_t *h = bliss_handle(sc, 0); return h ? sc_hip_bliss_sig_bytes(h) : 0; }
    if (SC_SCHEME_SIG_RING_TESLA == sc->scheme) { sc_hip_tesla_t *h = tesla_handle(sc); return h ? sc_hip_tesla_sig_bytes(h) : 0; }
    if (SC_SCHEME_SIG_FALCON == sc->scheme) { sc_hip_falcon_t *h = falcon_handle(sc); return h ? sc_hip_falcon_sig_bytes(h) : 0; }
    return 0;
}

size_t sc_hip_glue_tape_words(safecrypto_t *sc, unsigned attempts)
{
    if (NULL == sc) return 0;
    if (SC_SCHEME_SIG_BLISS == sc->scheme && sc->bliss)        /* a 128- / 192-bit sampler takes two / three prng_64 draws a sample */
        return sc_hip_bliss_tape_words_hp((int)sc->bliss->params->n, attempts, (sc->sc_gauss && sc->sc_gauss->precision > 64) ? (int)sc->sc_gauss->precision : 64);
    if (SC_SCHEME_SIG_RING_TESLA == sc->scheme && sc->ring_tesla) return (size_t)sc->ring_tesla->params->n * attempts;     /* n words per attempt (ring_tesla.c:790-800) */
    if (SC_SCHEME_SIG_FALCON == sc->scheme && sc->falcon) return 4 * (size_t)sc->falcon->params->n;                         /* two prng_64 per tree leaf, no retry */
    return 0;
}

SINT32 sc_hip_glue_sign_batch(safecrypto_t *sc, size_t batch, const UINT8 *msgs, size_t msg_stride, const UINT32 *mlens, unsigned attempts,
                              UINT8 *sigs, size_t sig_stride, UINT32 *siglens, UINT32 *status)
{
    if (NULL == sc || NULL == sigs || NULL == siglens || NULL == status || 0 == attempts) return SC_FUNC_FAILURE;
    if (0 == batch) return SC_FUNC_SUCCESS;
    size_t words = sc_hip_glue_tape_words(sc, attempts);
    if (0 == words) return SC_FUNC_FAILURE;
    words = (words + 3) & ~(size_t)3;
    UINT32 *tapes = malloc(sizeof(UINT32) * words * batch), *trials = malloc(sizeof(UINT32) * batch);
    if (NULL == tapes || NULL == trials) { free(tapes); free(trials); return SC_FUNC_FAILURE; }
    for (size_t i = 0; i < words * batch; i++) tapes[i] = prng_32(sc->prng_ctx[0]);          /* message 0's words first, as successive safecrypto_sign calls would draw them */
    int rc = SC_HIP_FAILURE;
    if (SC_SCHEME_SIG_BLISS == sc->scheme) {
        sc_hip_bliss_t *h = bliss_handle(sc, 1);
        if (h) rc = sc_hip_bliss_sign_host(h, batch, msgs, msg_stride, mlens, 0, tapes, words, NULL, sigs, sig_stride, siglens, trials, status);
    } else if (SC_SCHEME_SIG_RING_TESLA == sc->scheme) {
        sc_hip_tesla_t *h = tesla_handle(sc);
        if (h) rc = sc_hip_tesla_sign_host(h, batch, msgs, msg_stride, mlens, 0, tapes, words, NULL, sigs, sig_stride, siglens, trials, status);
    } else if (SC_SCHEME_SIG_FALCON == sc->scheme) {
        sc_hip_falcon_t *h = falcon_handle(sc);
        for (size_t i = 0; i < batch; i++) trials[i] = 1;
        if (h) rc = sc_hip_falcon_sign_host(h, batch, msgs, msg_stride, mlens, 0, tapes, words, NULL, sigs, sig_stride, siglens, status);
    }
    if (SC_HIP_SUCCESS == rc) for (size_t i = 0; i < batch; i++) if (1 == status[i]) { sc->stats.sig_num++; sc->stats.sig_num_trials += trials[i]; }
    memset(tapes, 0, sizeof(UINT32) * words * batch);
    free(tapes); free(trials);
    return SC_HIP_SUCCESS == rc ? SC_FUNC_SUCCESS : SC_FUNC_FAILURE;
}

SINT32 sc_hip_glue_verify_batch(safecrypto_t *sc, size_t batch, const UINT8 *msgs, size_t msg_stride, const UINT32 *mlens,
                                const UINT8 *sigs, size_t sig_stride, const UINT32 *siglens, UINT32 *verdicts)
{
    if (NULL == sc || NULL == sigs || NULL == siglens || NULL == verdicts) return SC_FUNC_FAILURE;
    int rc = SC_HIP_FAILURE;
    if (SC_SCHEME_SIG_BLISS == sc->scheme) { sc_hip_bliss_t *h = bliss_handle(sc, 0); if (h) rc = sc_hip_bliss_verify_host(h, batch, msgs, msg_stride, mlens, 0, sigs, sig_stride, siglens, verdicts); }
    else if (SC_SCHEME_SIG_RING_TESLA == sc->scheme) { sc_hip_tesla_t *h = tesla_handle(sc); if (h) rc = sc_hip_tesla_verify_host(h, batch, msgs, msg_stride, mlens, 0, sigs, sig_stride, siglens, verdicts); }
    else if (SC_SCHEME_SIG_FALCON == sc->scheme) { sc_hip_falcon_t *h = falcon_handle(sc); if (h) rc = sc_hip_falcon_verify_host(h, batch, msgs, msg_stride, mlens, 0, sigs, sig_stride, siglens, verdicts); }
    return SC_HIP_SUCCESS == rc ? SC_FUNC_SUCCESS : SC_FUNC_FAILURE;
}

/* one message through the batch entry: the reference's loop in miniature - more tape when the attempts drawn for did not suffice */
static SINT32 sign_one(safecrypto_t *sc, const UINT8 *m, size_t m_len, UINT8 **sigret, size_t *siglen)
{
    if (NULL == sigret || NULL == siglen || (NULL == m && m_len > 0)) return SC_FUNC_FAILURE;
    const size_t cap = sc_hip_glue_sig_bytes(sc);
    if (0 == cap) return SC_FUNC_FAILURE;
    const size_t row = (cap + 15) & ~(size_t)15, mrow = (m_len + 8) & ~(size_t)7;
    UINT8 *tmp = calloc(1, row + mrow);                                   /* signature row | the message in a row of its own (the engine reads whole rows) */
    if (NULL == tmp) return SC_FUNC_FAILURE;
    if (m_len) memcpy(tmp + row, m, m_len);
    UINT32 ml = (UINT32)m_len, sl = 0, st = 0;
    /* 8 attempts of tape cover all but ~1 signature in 10^4 (BLISS-B accepts ~ 1 in 1.6, ring-TESLA ~ 1 in 3); a short tape is reported, never guessed around */
    const SINT32 rc = sc_hip_glue_sign_batch(sc, 1, tmp + row, mrow, &ml, SC_SCHEME_SIG_RING_TESLA == sc->scheme ? 24 : 8, tmp, row, &sl, &st);
    SINT32 out = SC_FUNC_FAILURE;
    if (SC_FUNC_SUCCESS == rc && 1 == st) out = emit(tmp, sl, sigret, siglen);
    free(tmp);
    return out;
}
static SINT32 verify_one(safecrypto_t *sc, const UINT8 *m, size_t m_len, const UINT8 *sigbuf, size_t siglen)
{
    if (NULL == sigbuf || (NULL == m && m_len > 0)) return SC_FUNC_FAILURE;
    const size_t cap = sc_hip_glue_sig_bytes(sc);
    if (0 == cap) return SC_FUNC_FAILURE;
    const size_t row = (cap + 15) & ~(size_t)15, mrow = (m_len + 8) & ~(size_t)7;
    UINT8 *tmp = calloc(1, row + mrow);
    if (NULL == tmp) return SC_FUNC_FAILURE;
    memcpy(tmp, sigbuf, siglen < row ? siglen : row);
    if (m_len) memcpy(tmp + row, m, m_len);
    UINT32 ml = (UINT32)m_len, sl = (UINT32)siglen, verdict = 1;
    const SINT32 rc = sc_hip_glue_verify_batch(sc, 1, tmp + row, mrow, &ml, tmp, row, &sl, &verdict);
    free(tmp);
    return (SC_FUNC_SUCCESS == rc && 0 == verdict) ? SC_FUNC_SUCCESS : SC_FUNC_FAILURE;
}

SINT32 sc_hip_glue_bliss_b_sign(safecrypto_t *sc, const UINT8 *m, size_t m_len, UINT8 **sigret, size_t *siglen)
{ return (sc && SC_SCHEME_SIG_BLISS == sc->scheme) ? sign_one(sc, m, m_len, sigret, siglen) : SC_FUNC_FAILURE; }
SINT32 sc_hip_glue_bliss_b_verify(safecrypto_t *sc, const UINT8 *m, size_t m_len, const UINT8 *sigbuf, size_t siglen)
{ return (sc && SC_SCHEME_SIG_BLISS == sc->scheme) ? verify_one(sc, m, m_len, sigbuf, siglen) : SC_FUNC_FAILURE; }
SINT32 sc_hip_glue_ring_tesla_sign(safecrypto_t *sc, const UINT8 *m, size_t m_len, UINT8 **sigret, size_t *siglen)
{ return (sc && SC_SCHEME_SIG_RING_TESLA == sc->scheme) ? sign_one(sc, m, m_len, sigret, siglen) : SC_FUNC_FAILURE; }
SINT32 sc_hip_glue_ring_tesla_verify(safecrypto_t *sc, const UINT8 *m, size_t m_len, const UINT8 *sigbuf, size_t siglen)
{ return (sc && SC_SCHEME_SIG_RING_TESLA == sc->scheme) ? verify_one(sc, m, m_len, sigbuf, siglen) : SC_FUNC_FAILURE; }
SINT32 sc_hip_glue_falcon_sign(safecrypto_t *sc, const UINT8 *m, size_t m_len, UINT8 **sigret, size_t *siglen)
{ return (sc && SC_SCHEME_SIG_FALCON == sc->scheme) ? sign_one(sc, m, m_len, sigret, siglen) : SC_FUNC_FAILURE; }
SINT32 sc_hip_glue_falcon_verify(safecrypto_t *sc, const UINT8 *m, size_t m_len, const UINT8 *sigbuf, size_t siglen)
{ return (sc && SC_SCHEME_SIG_FALCON == sc->scheme) ? verify_one(sc, m, m_len, sigbuf, siglen) : SC_FUNC_FAILURE; }

size_t sc_hip_glue_kyber_ct_bytes(safecrypto_t *sc)
{
    sc_hip_kyber_t *h = kyber_handle(sc, 0);
    return h ? sc_hip_kyber_ct_bytes(h) : 0;
}

SINT32 sc_hip_glue_kyber_encapsulation_batch(safecrypto_t *sc, size_t batch, UINT8 *cts, size_t ct_stride, UINT8 *keys)
{
    sc_hip_kyber_t *h = kyber_handle(sc, 0);
    if (NULL == h || NULL == cts || NULL == keys) return SC_FUNC_FAILURE;
    if (0 == batch) return SC_FUNC_SUCCESS;
    UINT8 *m = malloc(32 * batch);
    if (NULL == m) return SC_FUNC_FAILURE;
    for (size_t i = 0; i < batch; i++) prng_mem(sc->prng_ctx[0], m + 32 * i, 32);             /* kyber_kem.c:856: what each safecrypto_encapsulation call would draw */
    const int rc = sc_hip_kyber_kem_encaps_host(h, batch, m, cts, ct_stride, keys);
    memset(m, 0, 32 * batch);
    free(m);
    if (SC_HIP_SUCCESS == rc) sc->stats.encapsulate_num += batch;
    return SC_HIP_SUCCESS == rc ? SC_FUNC_SUCCESS : SC_FUNC_FAILURE;
}

SINT32 sc_hip_glue_kyber_decapsulation_batch(safecrypto_t *sc, size_t batch, const UINT8 *cts, size_t ct_stride, UINT8 *keys)
{
    sc_hip_kyber_t *h = kyber_handle(sc, 1);
    if (NULL == h || NULL == cts || NULL == keys) return SC_FUNC_FAILURE;
    const int rc = sc_hip_kyber_kem_decaps_host(h, batch, cts, ct_stride, keys);
    if (SC_HIP_SUCCESS == rc) sc->stats.decapsulate_num += batch;
    return SC_HIP_SUCCESS == rc ? SC_FUNC_SUCCESS : SC_FUNC_FAILURE;
}

SINT32 sc_hip_glue_kyber_encapsulation(safecrypto_t *sc, UINT8 **c, size_t *c_len, UINT8 **k, size_t *k_len)
{
    const size_t ctb = sc_hip_glue_kyber_ct_bytes(sc);
    if (0 == ctb || NULL == c || NULL == c_len || NULL == k || NULL == k_len) return SC_FUNC_FAILURE;
    const size_t row = (ctb + 15) & ~(size_t)15;
    UINT8 *ct = calloc(1, row), key[32];
    if (NULL == ct) return SC_FUNC_FAILURE;
    SINT32 rc = sc_hip_glue_kyber_encapsulation_batch(sc, 1, ct, row, key);
    if (SC_FUNC_SUCCESS == rc) rc = emit(ct, ctb, c, c_len);
    if (SC_FUNC_SUCCESS == rc) rc = emit(key, 32, k, k_len);
    memset(key, 0, sizeof key);
    free(ct);
    return rc;
}

SINT32 sc_hip_glue_kyber_decapsulation(safecrypto_t *sc, const UINT8 *c, size_t c_len, UINT8 **k, size_t *k_len)
{
    const size_t ctb = sc_hip_glue_kyber_ct_bytes(sc);
    if (0 == ctb || NULL == c || NULL == k || NULL == k_len || c_len < ctb) return SC_FUNC_FAILURE;
    const size_t row = (ctb + 15) & ~(size_t)15;
    UINT8 *ct = calloc(1, row), key[32];
    if (NULL == ct) return SC_FUNC_FAILURE;
    memcpy(ct, c, ctb);
    SINT32 rc = sc_hip_glue_kyber_decapsulation_batch(sc, 1, ct, row, key);
    if (SC_FUNC_SUCCESS == rc) rc = emit(key, 32, k, k_len);
    memset(key, 0, sizeof key);
    free(ct);
    return rc;
}
