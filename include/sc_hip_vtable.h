/*
 * sc_hip_vtable.h - single-polynomial entry points with the EXACT signatures of the reference's
 * utils_arith_ntt_t 32-bit slots (src/utils/arith/ntt.h:147-176 typedefs, :217-298 struct), so a
 * maintainer can point sc->sc_ntt at the GPU engine.  Each call moves one
 * polynomial host -> device -> host: it exists to prove drop-in equivalence (and to run the
 * reference's own functional tests over the engine), not for speed - throughput comes from
 * the batched ABI in sc_hip.h.
 *
 * sc_hip_ntt_params_t mirrors ntt_params_t (src/utils/arith/ntt.h:91-103, packed) field for field;
 * only n and u.ntt32.q are read.
 */
#ifndef SC_HIP_VTABLE_H
#define SC_HIP_VTABLE_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

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

/* the 32-bit array slots of utils_arith_ntt_t, same names, same order (ntt.h:241-261) */
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
    /* ntt.h:150-153,164-165,240-241,250-251: sparse products (ring-TESLA) and the fallible pointwise inverse / division
     * (0 = SC_FUNC_SUCCESS, 1 = SC_FUNC_FAILURE; an engine error also returns 1 and is counted) */
    void (*mul_32_sparse)(int32_t *v, size_t n, uint16_t omega, const int32_t *t, const int32_t *c);
    void (*mul_32_sparse_16)(int32_t *v, size_t n, uint16_t omega, const int16_t *t, const int32_t *c);
    int32_t (*invert_32)(int32_t *v, const sc_hip_ntt_params_t *p, size_t n);
    int32_t (*div_32)(int32_t *num, const int32_t *den, const sc_hip_ntt_params_t *p, size_t n);
} sc_hip_ntt32_vtable_t;

/* replaces: utils_arith_ntt(safecrypto_ntt_e) (src/utils/arith/arith.c:360-396) for a new
 * SC_NTT_HIP selector value.  Returns NULL (and sets sc_hip_last_error) when no GPU is usable -
 * there is no CPU fallback behind these pointers.  The twiddle arguments w/r are ignored: the
 * engine derives the same tables from (q, n) (and would reject a (q, n) it cannot serve). */
const sc_hip_ntt32_vtable_t *sc_hip_utils_arith_ntt32(void);

/* number of failed vtable calls so far (the reference's slots return void, so errors are
 * counted and reported through sc_hip_last_error instead of being dropped) */
unsigned long sc_hip_vtable_errors(void);

#ifdef __cplusplus
}
#endif
#endif
