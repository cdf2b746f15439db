/*
 * sc_hip_vtable.h - the reference's own function-pointer tables, served by the MI355X engine.
 *
 * libsafecrypto reaches its polynomial arithmetic only through tables stored in each scheme context:
 *   sc->sc_ntt   = utils_arith_ntt(type)     const utils_arith_ntt_t *   src/utils/arith/ntt.h:217-298, arith.c:360-396
 *   sc->sc_poly  = utils_arith_poly()        const utils_arith_poly_t *  src/utils/arith/arith.h:74-111, arith.c:22-53
 *   sc->sc_vec   = utils_arith_vectors()     const utils_arith_vec_t *   src/utils/arith/arith.h:123-128, arith.c:56-63
 *   sc->sc_gauss = create_sampler(...)       utils_sampling_t *          src/utils/sampling/sampling.h:68-86
 * This header declares LAYOUT-IDENTICAL mirrors of those four structs (same slot order, same packing; the slots
 * the engine serves carry the reference's exact signatures, the rest are opaque pointers) and the functions
 * that fill them, so that a maintainer writes
 *       static sc_hip_utils_arith_ntt_t t;
 *       sc_hip_utils_arith_ntt_fill(&t, utils_arith_ntt(SC_NTT_FLOATING_POINT));
 *       sc->sc_ntt = (const utils_arith_ntt_t *)&t;
 * and the reference's unmodified scheme code (dilithium_sign, bliss_b_sign, ...) runs over the GPU - which is
 * what examples/safecrypto_glue/dropin_driver.c does to prove the drop-in (sc_hip_glue.c beside it is the binding).
 *
 * What the engine serves: every ARRAY slot of the 32-bit third of utils_arith_ntt_t (transforms, pointwise /
 * scalar / sparse products, normalise / centre / flip, invert / div), the array slots of utils_arith_poly_t on
 * SINT32 data, utils_arith_vec_t, and the sampler's vector_16 / vector_32 / sample.  What it does not serve keeps
 * the pointer of the `base` table handed to the fill function: the scalar slots (modn_32, muln_32, sqrn_32,
 * pwr_32 take and return one SINT32 - there is no array to send anywhere), the 16-bit and limb thirds no scheme
 * calls (SURVEY.md 8(b) call-site census), single-coefficient updates (add_32_scalar / sub_32_scalar touch
 * poly[0] only), the slots that take the reference's prng_ctx_t (uniform_32, uniform_z2) and the Z2 helpers.
 *
 * Each call moves ONE polynomial host -> device -> host: this boundary proves equivalence and lets the
 * reference's own tests run over the engine; throughput comes from the batched ABI in sc_hip.h.
 * The slots have no context argument, so the device is bound per host thread (sc_hip_vtable_bind_device);
 * threads bound to different devices run concurrently, calls on one device are serialised.
 * There is no CPU fallback: with no usable device the fill functions return 1 and leave dst untouched.
 */
#ifndef SC_HIP_VTABLE_H
#define SC_HIP_VTABLE_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

/* mirrors ntt_params_t (src/utils/arith/ntt.h:91-103, packed) field for field; only n and u.ntt32.q are read */
typedef struct __attribute__((__packed__)) sc_hip_ntt_params {
    double q_dbl;
    double inv_q_dbl;
    float  inv_q_flt;
    size_t n;
    union {
        struct __attribute__((__packed__)) { int64_t q; uint64_t q_inv; int64_t m; int64_t k; } ntt64;   /* also the limb form */
        struct __attribute__((__packed__)) { int32_t q; uint32_t q_inv; int32_t m; int32_t k; } ntt32;
        struct __attribute__((__packed__)) { int16_t q; uint16_t q_inv; int16_t m; int16_t k; } ntt16;
    } u;
} sc_hip_ntt_params_t;

typedef void (*sc_hip_opaque_fn)(void);          /* a slot the engine does not serve: the base table's pointer, whatever its type */

/* ---- utils_arith_ntt_t (ntt.h:217-298): 18 16-bit slots | 26 32-bit slots | 33 limb slots = 77 ------------------- */
typedef struct __attribute__((__packed__)) sc_hip_utils_arith_ntt {
    sc_hip_opaque_fn slots_16[18];               /* modn_16 ... inv_ntt_16_large (ntt.h:218-235) */
    /* the 32-bit third, ntt.h:237-262, in the reference's order */
    int32_t (*modn_32)(int32_t, const sc_hip_ntt_params_t *);                                            /* base */
    int32_t (*muln_32)(int32_t, int32_t, const sc_hip_ntt_params_t *);                                   /* base */
    int32_t (*sqrn_32)(int32_t, const sc_hip_ntt_params_t *);                                            /* base */
    void (*mul_32_sparse)(int32_t *v, size_t n, uint16_t omega, const int32_t *t, const int32_t *c);
    void (*mul_32_sparse_16)(int32_t *v, size_t n, uint16_t omega, const int16_t *t, const int32_t *c);
    void (*mul_32_pointwise)(int32_t *v, const sc_hip_ntt_params_t *p, const int32_t *t, const int32_t *u);
    void (*mul_32_pointwise_16)(int32_t *v, const sc_hip_ntt_params_t *p, const int32_t *t, const int16_t *u);
    void (*mul_32_scalar)(int32_t *v, const sc_hip_ntt_params_t *p, const int32_t *t, int32_t c);
    void (*fft_32_32)(int32_t *v, const sc_hip_ntt_params_t *p, const int32_t *w);
    void (*fft_32_32_large)(int32_t *v, const sc_hip_ntt_params_t *p, const int32_t *w);
    void (*fft_32_16)(int32_t *v, const sc_hip_ntt_params_t *p, const int16_t *w);
    void (*fft_32_16_large)(int32_t *v, const sc_hip_ntt_params_t *p, const int16_t *w);
    int32_t (*pwr_32)(int32_t, int32_t, const sc_hip_ntt_params_t *);                                    /* base */
    int32_t (*invert_32)(int32_t *v, const sc_hip_ntt_params_t *p, size_t n);
    int32_t (*div_32)(int32_t *num, const int32_t *den, const sc_hip_ntt_params_t *p, size_t n);
    void (*flip_32)(int32_t *v, const sc_hip_ntt_params_t *p);
    void (*center_32)(int32_t *v, size_t n, const sc_hip_ntt_params_t *p);
    void (*normalize_32)(int32_t *v, size_t n, const sc_hip_ntt_params_t *p);
    void (*fwd_ntt_32_32)(int32_t *v, const sc_hip_ntt_params_t *p, const int32_t *t, const int32_t *w);
    void (*inv_ntt_32_32)(int32_t *v, const sc_hip_ntt_params_t *p, const int32_t *t, const int32_t *w, const int32_t *r);
    void (*fwd_ntt_32_32_large)(int32_t *v, const sc_hip_ntt_params_t *p, const int32_t *t, const int32_t *w);
    void (*inv_ntt_32_32_large)(int32_t *v, const sc_hip_ntt_params_t *p, const int32_t *t, const int32_t *w, const int32_t *r);
    void (*fwd_ntt_32_16)(int32_t *v, const sc_hip_ntt_params_t *p, const int32_t *t, const int16_t *w);
    void (*inv_ntt_32_16)(int32_t *v, const sc_hip_ntt_params_t *p, const int32_t *t, const int16_t *w, const int16_t *r);
    void (*fwd_ntt_32_16_large)(int32_t *v, const sc_hip_ntt_params_t *p, const int32_t *t, const int16_t *w);
    void (*inv_ntt_32_16_large)(int32_t *v, const sc_hip_ntt_params_t *p, const int32_t *t, const int16_t *w, const int16_t *r);
    sc_hip_opaque_fn slots_limb[33];             /* modn_limb ... inv_ntt_limb_16_large (ntt.h:264-296) */
} sc_hip_utils_arith_ntt_t;

/* dst <- base (77 pointers: e.g. utils_arith_ntt(SC_NTT_FLOATING_POINT)), then the served 32-bit slots <- the engine.
 * base may be NULL: the unserved slots are then NULL (a table for callers that use the array slots only).
 * Semantics of the served slots: outputs are canonical representatives in [0, q) (centre: [-(q-1)/2, (q-1)/2]) - a legal
 * choice, the reference's own backends return different representatives; the twiddle arguments w / r
 * are ignored, the engine derives the same tables from (q, n) with the reference's root rule and counts an error for a
 * (q, n) it cannot serve.  Returns 0, or 1 when no device is usable. */
int sc_hip_utils_arith_ntt_fill(sc_hip_utils_arith_ntt_t *dst, const void *base);

/* ---- utils_arith_poly_t (arith.h:74-111): 35 slots -------------------------------------------------------------- */
typedef struct __attribute__((__packed__)) sc_hip_utils_arith_poly {
    sc_hip_opaque_fn slots_8_16[15];             /* cmp_not_equal_8, copy_16 ... degree_16 */
    sc_hip_opaque_fn copy_32, reset_32;          /* base (memcpy / memset of host memory) */
    sc_hip_opaque_fn add_32_scalar, sub_32_scalar;                                                       /* base: poly[0] only */
    void (*mul_32_scalar)(int32_t *poly, size_t n, int32_t in);                                          /* poly_32.c:55-61 */
    void (*add_32)(int32_t *out, size_t n, const int32_t *in1, const int32_t *in2);                      /* :63-70 */
    void (*sub_32)(int32_t *out, size_t n, const int32_t *in1, const int32_t *in2);                      /* :72-79 */
    void (*add_single_32)(int32_t *out, size_t n, const int32_t *in);                                    /* :82-89 */
    void (*sub_single_32)(int32_t *out, size_t n, const int32_t *in);                                    /* :92-99 */
    sc_hip_opaque_fn mul_32;                     /* base (schoolbook product into 2n-1 words; no scheme on the hot path calls it) */
    void (*mod_negate_32)(int32_t *out, size_t n, int32_t q, const int32_t *in);                         /* :158-164 */
    int32_t (*cmp_not_equal_32)(volatile const int32_t *in1, volatile const int32_t *in2, size_t n);     /* :166-174 */
    sc_hip_opaque_fn uniform_32, degree_32;      /* base (uniform_32 takes the reference's prng_ctx_t) */
    sc_hip_opaque_fn slots_z2[6];                /* bin_inv_32, z2_mul, z2_div, z2_mul_mod2, z2_conv, uniform_z2 */
} sc_hip_utils_arith_poly_t;
int sc_hip_utils_arith_poly_fill(sc_hip_utils_arith_poly_t *dst, const void *base);

/* ---- utils_arith_vec_t (arith.h:123-128; vectors.c:25-95) -------------------------------------------------------- */
typedef struct __attribute__((__packed__)) sc_hip_utils_arith_vec {
    int32_t (*absmax_32)(const int32_t *v, size_t n);
    int32_t (*scalar_32)(const int32_t *t, const int32_t *u, size_t n);
    int32_t (*absmax_16)(const int16_t *v, size_t n);
    int32_t (*scalar_16)(const int16_t *t, const int16_t *u, size_t n);
} sc_hip_utils_arith_vec_t;
int sc_hip_utils_arith_vec_fill(sc_hip_utils_arith_vec_t *dst);

/* ---- utils_sampling_t (sampling.h:68-86) -------------------------------------------------------------------------
 * The engine owns no PRNG: the sampler pulls the reference's prng_64() words through a callback, n per vector in the
 * order sample_vector_32 draws them (sampling.c:211-228; gaussian_cdf.c:639-659), and maps them through the 64-bit CDT on the
 * device: v[i] = sample(word_i) + centre.  h_cdf: the reference's own table (gaussian_cdf_create_64's, `size` entries) when
 * bit-identity across hosts matters - the table is long-double expl() arithmetic and x87 expl differs in the last bit between
 * CPU vendors - or NULL to build it here with the same recurrence. */
typedef uint64_t (*sc_hip_rng64_fn)(void *rng_ctx);
typedef struct sc_hip_sampler sc_hip_sampler_t;
int  sc_hip_sampler_create(float tail, float sigma, const uint64_t *h_cdf, int size, sc_hip_rng64_fn rng, void *rng_ctx, sc_hip_sampler_t **out);
/* the same at 128- / 192-bit precision (create_sampler(..., SAMPLING_128BIT / SAMPLING_192BIT, ...): gaussian_cdf_sample_128 / _192, gaussian_cdf.c:468-509): h_cdf is the
 * reference's own table (gauss_cdf_high_t::cdf_128 / cdf_192, [size][precision_bits / 64] limbs - MPFR arithmetic the engine does not redo), precision_bits / 64 draws of
 * `rng` per sample; sc_hip_utils_sampling_fill then reports precision 128 / 192; the shuffle / blinding fillers stay 64-bit only */
int  sc_hip_sampler_create_hp(int precision_bits, float tail, float sigma, const uint64_t *h_cdf, int size, sc_hip_rng64_fn rng, void *rng_ctx, sc_hip_sampler_t **out);
void sc_hip_sampler_set_rng(sc_hip_sampler_t *s, sc_hip_rng64_fn rng, void *rng_ctx);
void sc_hip_sampler_destroy(sc_hip_sampler_t *s);

typedef struct __attribute__((__packed__)) sc_hip_utils_sampling {
    sc_hip_opaque_fn create;                                             /* NULL: the mirror is built around an existing handle */
    int32_t (*destroy)(void **gauss);                                    /* utils_sampling_destroy */
    sc_hip_opaque_fn get_prng;                                           /* NULL */
    int32_t (*sample)(void *gauss);                                      /* utils_sampling_sample: one draw */
    int32_t (*vector_16)(const struct sc_hip_utils_sampling *, int16_t *v, size_t n, int32_t centre);
    int32_t (*vector_32)(const struct sc_hip_utils_sampling *, int32_t *v, size_t n, int32_t centre);
    int32_t precision;                                                   /* sample_precision_e: 64 (128 / 192 behind sc_hip_sampler_create_hp) */
    int32_t dimension;
    int32_t bootstrapped;                                                /* SAMPLING_DISABLE_BOOTSTRAP */
    float   tail, sigma, sigma2;
    void   *gauss;                                                       /* the sc_hip_sampler_t */
    void   *prng_ctx;                                                    /* the reference's prng_ctx_t, as given to create */
    uint32_t discard;                                                    /* 0 (the default, src/safecrypto.c:486) or SCA_PATTERN_SAMPLE_DISCARD_LO / MD / HI = 2 / 4 / 6 with a prng_32 source set */
    void   *bootstrap;
} sc_hip_utils_sampling_t;
/* fills dst around s (dst->gauss = s; dst->prng_ctx = the rng_ctx given at creation); get_vector_32 / get_vector_16 /
 * get_sample of the reference (sampling.c:521-580) then run on the engine */
int sc_hip_utils_sampling_fill(sc_hip_utils_sampling_t *dst, sc_hip_sampler_t *s);
/* The side-channel variants of the vector fillers (sampling.c:107-228; create_sampler's `blinding`, selected by flags[2] & SC_FLAG_2_SAMPLE_SCA_SHUFFLE /
 * _BLINDING, src/safecrypto.c:492-497; sample discarding SC_FLAG_2_SAMPLE_SCA_DISCARD_*, set_discard -> dst->discard = 2 / 4 / 6).  They interleave
 * prng_32 draws (rand_range, discard_sample) with the samples' prng_64 draws on the SAME generator, so the sampler needs the reference's prng_32 as well:
 * sc_hip_sampler_set_rng32 (same rng_ctx).  blinding: 0 NORMAL_SAMPLES, 1 BLINDING_SAMPLES (the sampler's table must be the blinding one:
 * sc_hip_cdt_build_table_blinding, or the reference's own), 2 SHUFFLE_SAMPLES (safecrypto_private.h:154-158).  What runs where: the order of draws and
 * the shuffle's moves are control flow that depends on the drawn words - replayed on the host exactly as sampling.c does; every table search (the samples
 * themselves) runs on the device, all of a vector's draws in one launch. */
typedef uint32_t (*sc_hip_rng32_fn)(void *rng_ctx);
void sc_hip_sampler_set_rng32(sc_hip_sampler_t *s, sc_hip_rng32_fn rng32);
int sc_hip_utils_sampling_fill_sca(sc_hip_utils_sampling_t *dst, sc_hip_sampler_t *s, int blinding);

/* ---- device binding and error reporting ---------------------------------------------------------------------------- */
/* binds the calling host thread's table calls to `device` (default: device 0).  One host thread per GPU, each with its own
 * safecrypto_t, is the reference's threading model. */
int sc_hip_vtable_bind_device(int device);
/* the reference's slots return void: failed calls are counted here and described by sc_hip_last_error() */
unsigned long sc_hip_vtable_errors(void);

/* round-1 form: the 16 array slots of the 32-bit third as a compact table (kept for examples/c_host.c) */
typedef struct sc_hip_ntt32_vtable {
    void (*mul_32_pointwise)(int32_t *v, const sc_hip_ntt_params_t *p, const int32_t *t, const int32_t *u);
    void (*mul_32_pointwise_16)(int32_t *v, const sc_hip_ntt_params_t *p, const int32_t *t, const int16_t *u);
    void (*center_32)(int32_t *v, size_t n, const sc_hip_ntt_params_t *p);
    void (*normalize_32)(int32_t *v, size_t n, const sc_hip_ntt_params_t *p);
    void (*fwd_ntt_32_32)(int32_t *v, const sc_hip_ntt_params_t *p, const int32_t *t, const int32_t *w);
    void (*inv_ntt_32_32)(int32_t *v, const sc_hip_ntt_params_t *p, const int32_t *t, const int32_t *w, const int32_t *r);
    void (*fwd_ntt_32_32_large)(int32_t *v, const sc_hip_ntt_params_t *p, const int32_t *t, const int32_t *w);
    void (*inv_ntt_32_32_large)(int32_t *v, const sc_hip_ntt_params_t *p, const int32_t *t, const int32_t *w, const int32_t *r);
    void (*fwd_ntt_32_16)(int32_t *v, const sc_hip_ntt_params_t *p, const int32_t *t, const int16_t *w);
    void (*inv_ntt_32_16)(int32_t *v, const sc_hip_ntt_params_t *p, const int32_t *t, const int16_t *w, const int16_t *r);
    void (*fwd_ntt_32_16_large)(int32_t *v, const sc_hip_ntt_params_t *p, const int32_t *t, const int16_t *w);
    void (*inv_ntt_32_16_large)(int32_t *v, const sc_hip_ntt_params_t *p, const int32_t *t, const int16_t *w, const int16_t *r);
    void (*mul_32_sparse)(int32_t *v, size_t n, uint16_t omega, const int32_t *t, const int32_t *c);
    void (*mul_32_sparse_16)(int32_t *v, size_t n, uint16_t omega, const int16_t *t, const int32_t *c);
    int32_t (*invert_32)(int32_t *v, const sc_hip_ntt_params_t *p, size_t n);
    int32_t (*div_32)(int32_t *num, const int32_t *den, const sc_hip_ntt_params_t *p, size_t n);
} sc_hip_ntt32_vtable_t;
/* NULL (and sc_hip_last_error set) when no GPU is usable */
const sc_hip_ntt32_vtable_t *sc_hip_utils_arith_ntt32(void);

#ifdef __cplusplus
}
#endif
#endif
