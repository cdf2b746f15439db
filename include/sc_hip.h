/*
 * sc_hip.h - C ABI of the MI355X (gfx950) batched lattice-arithmetic engine that stands
 * behind libsafecrypto's polynomial hot path.
 *
 * Every entry point is plain C (pointers + sizes, no C++/torch types).  Each one names the
 * reference interface it replaces (paths relative to the libsafecrypto tree, 0.2.0-79).
 * The reference works on ONE polynomial per call; the engine's native unit is a BATCH of
 * polynomials stored row-major as int32 [batch][n] (SINT32, src/utils/arith/ntt.h), so
 * every call takes a batch count.  batch = 1 reproduces the reference call exactly.
 *
 * Conventions
 *   - return 0 (SC_FUNC_SUCCESS, include/safecrypto_types.h:71) or 1 (SC_FUNC_FAILURE, :72)
 *   - "d_" pointers are device (HBM) addresses, "h_" pointers host addresses
 *   - stream is a hipStream_t passed as void* (NULL = the null stream); device-pointer
 *     calls are asynchronous on that stream, host-pointer calls return after the copy back
 *   - outputs are canonical where the reference's are (SURVEY.md 8(a) bit-exact contract):
 *     forward NTT results are returned already normalised to [0,q), i.e. equal to the
 *     reference's fwd_ntt_32_* followed by normalize_32
 *   - there is NO CPU fallback: every compute entry fails (returns 1, message through
 *     sc_hip_last_error) when no gfx950 device / kernel image is available
 *   - threading follows the reference (one safecrypto_t per thread, src/safecrypto.c): plans and CDTs may be shared between threads
 *     and streams - their tables are immutable after creation; sc_hip_matvec keeps its scratch in the plan and orders overlapping
 *     calls itself (host mutex + device event); sc_hip_plan_set_arith is configuration, to be called before the plan is used
 *     concurrently.  A scheme handle (sc_hip_dilithium_t, _bliss_t, _tesla_t, _kyber_t, _falcon_t, _rlwe_t, _ens_t, _dlp_t) owns the
 *     device workspace its batched calls use, so calls on ONE handle must not overlap in time - give each host thread /
 *     stream its own handle (they are cheap next to a batch); sc_hip_last_error is per thread
 *   - a batched call may run parts of its work on streams of the handle's own beside the caller's (the calls with a key per item do: independent
 *     per-key chains, forked and joined by events inside the call); everything it does is ordered after what the caller queued on `stream`
 *     before the call and before what the caller queues after it, as if it ran on `stream` alone.  Rejection loops (sign calls) read one
 *     4-byte counter per round back to the host, so those calls return with most of their work done
 *   - device memory that held key material, masks, noise or CSPRNG output is zeroed before it is freed, and the per-call secrets of
 *     sign / decapsulate / encrypt calls are zeroed on the call's stream behind its kernels (the reference zeroes sc->temp the same way)
 */
#ifndef SC_HIP_H
#define SC_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SC_HIP_SUCCESS 0
#define SC_HIP_FAILURE 1

typedef struct sc_hip_ctx  sc_hip_ctx_t;   /* one per (process, device)                    */
typedef struct sc_hip_plan sc_hip_plan_t;  /* one per (q, n): twiddles resident in HBM     */
typedef struct sc_hip_cdt  sc_hip_cdt_t;   /* one per (tail, sigma): CDT resident in HBM   */

/* ---- lifetime ------------------------------------------------------------------ */

/* replaces nothing in the reference (it has no device notion); the analogue of
 * safecrypto_create's per-context vtable wiring, src/schemes/sig/dilithium/dilithium.c:207-209 */
int  sc_hip_ctx_create(int device, sc_hip_ctx_t **ctx);
void sc_hip_ctx_destroy(sc_hip_ctx_t *ctx);
const char *sc_hip_last_error(void);
int  sc_hip_device_count(void);
int  sc_hip_synchronize(sc_hip_ctx_t *ctx, void *stream);

/* replaces: init_reduce (src/utils/arith/ntt.c:132-140) + the generated tables w<q>_n<n>,
 * r<q>_n<n> (build_tools/ntt_table_gen/main.c:205-297, roots_of_unity.c:67-173).
 * g = smallest m >= 2 with m^n == -1 (mod q), exactly the reference's choice, so the
 * transform is the same function.  n in {256,512,1024}; q prime, q == 1 (mod 2n), q < 2^30. */
int  sc_hip_plan_create(sc_hip_ctx_t *ctx, int32_t q, int n, sc_hip_plan_t **plan);
void sc_hip_plan_destroy(sc_hip_plan_t *plan);
int32_t sc_hip_plan_q(const sc_hip_plan_t *plan);
int     sc_hip_plan_n(const sc_hip_plan_t *plan);
int32_t sc_hip_plan_root(const sc_hip_plan_t *plan);           /* g                         */
/* host copies of the reference-format tables (w[i]=g^i, r[i]=((q-1)/n) g^i), n entries each */
int  sc_hip_plan_tables(const sc_hip_plan_t *plan, int32_t *h_w, int32_t *h_r);
/* selects the modular-multiply code path (csrc/modarith.h); for measurement and tests - every path returns the same canonical values:
 *   1 = 24-bit Shoup products (q < 2^23)                     2 = 32-bit Shoup products, unsigned lazy residues (q < 2^30)
 *   3 = small-modulus multiply-add form (q < 2^14 and 2 q (q-1) (log2 n + 1) < 2^32)
 *   4 = the same on SIGNED lazy residues (n = 256: 36 q^2 < 2^31, i.e. q = 7681; n = 512, 1024: q^2 log2 n < 2^31 and q < 2^14, i.e. 7681, 12289)
 *   5 = signed lazy residues on the 32-bit multiplier (q < 2^25; q < 2^24 for n = 512, 1024: Dilithium's 8380417, ring-TESLA's 8399873, ...)
 *   0 = auto, the rule sc_hip_plan_create applies: 4 where it holds, else 5 where it holds, else 3 where it holds, else 2.
 * A mode whose bound the plan's (q, n) does not meet is refused (SC_HIP_FAILURE). */
int  sc_hip_plan_set_arith(sc_hip_plan_t *plan, int mode);

/* ---- batched transforms, device pointers ------------------------------------------- */

/* replaces: fwd_ntt_32_32 / fwd_ntt_32_16 (ntt_template.c.in:1541-1552,1615-1626) followed by
 * normalize_32 (:1841-1901).  in: any int32 representatives; out: V[k] = sum_i t[i] g^{i(2k+1)}
 * in [0,q), natural order.  d_out may alias d_in. */
int sc_hip_ntt_fwd(sc_hip_plan_t *plan, size_t batch, const int32_t *d_in, int32_t *d_out, void *stream);

/* replaces: inv_ntt_32_32 / inv_ntt_32_16 (ntt_template.c.in:1567-1589,1641-1663) including
 * ntt32_flip_generic (ntt.c:584-604): exact inverse of the above, canonical [0,q). */
int sc_hip_ntt_inv(sc_hip_plan_t *plan, size_t batch, const int32_t *d_in, int32_t *d_out, void *stream);

/* replaces: mul_32_pointwise / mul_32_pointwise_16 (ntt_template.c.in:956-1141) followed by
 * normalize_32: out[i] = a[i]*b[i] mod q in [0,q).  len = total coefficients. */
int sc_hip_pointwise(sc_hip_plan_t *plan, size_t len, const int32_t *d_a, const int32_t *d_b, int32_t *d_out, void *stream);

/* fused fwd_ntt -> mul_32_pointwise -> inv_ntt (the sequence at e.g. bliss_b.c:1378-1384,
 * module_lwe.c:1357-1370): out = INTT(NTT(a) o bhat), bhat given in the NTT domain in the
 * reference's natural order (e.g. a stored public key).  BASELINE.json config 2. */
int sc_hip_polymul_ntt(sc_hip_plan_t *plan, size_t batch, const int32_t *d_a, const int32_t *d_bhat, int32_t *d_out, void *stream);
/* the same for TWO stored NTT-domain polynomials and one batch of inputs (ring-TESLA's v1 = a1 y, v2 = a2 y: ring_tesla.c:842-850 / 1005-1012 call
 * fwd_ntt once and mul_32_pointwise + inv_ntt twice): out1[i] = INTT(NTT(a[i]) o b1hat), out2[i] = INTT(NTT(a[i]) o b2hat), ONE forward transform per
 * input.  b1hat, b2hat: n words each, used for every input (broadcast). */
int sc_hip_polymul_ntt_pair(sc_hip_plan_t *plan, size_t batch, const int32_t *d_a, const int32_t *d_b1hat, const int32_t *d_b2hat, int32_t *d_out1, int32_t *d_out2, void *stream);

/* fused product of two coefficient-domain polynomials: out = a*b mod (x^n+1, q), canonical
 * (fwd, fwd, mul_32_pointwise, inv: ring_tesla.c:752-910 pattern). */
int sc_hip_polymul(sc_hip_plan_t *plan, size_t batch, const int32_t *d_a, const int32_t *d_b, int32_t *d_out, void *stream);

/* replaces: normalize_32 / center_32 (ntt_template.c.in:1777-1901), in place allowed */
int sc_hip_normalize(sc_hip_plan_t *plan, size_t len, const int32_t *d_in, int32_t *d_out, void *stream);
int sc_hip_center(sc_hip_plan_t *plan, size_t len, const int32_t *d_in, int32_t *d_out, void *stream);

/* replaces: utils_arith_poly_t add_32 / sub_32 (src/utils/arith/poly_32.c:48-62); no reduction */
int sc_hip_poly_add(sc_hip_ctx_t *ctx, size_t len, const int32_t *d_a, const int32_t *d_b, int32_t *d_out, void *stream);
int sc_hip_poly_sub(sc_hip_ctx_t *ctx, size_t len, const int32_t *d_a, const int32_t *d_b, int32_t *d_out, void *stream);

/* replaces: create_rand_product_32_csprng / _16_ with the matrix already expanded
 * (src/utils/arith/module_lwe.c:588-748): t[b][i] = INTT( sum_j Ahat[i][j] o NTT(y[b][j]) ),
 * canonical.  d_Ahat: [k][l][n] NTT-domain rows in the reference's natural order (one key,
 * shared by the batch); d_y: [batch][l][n]; d_t: [batch][k][n].  transposed != 0 uses
 * Ahat[j][i] (module_lwe.c:617-633, Kyber encryption). */
int sc_hip_matvec(sc_hip_plan_t *plan, size_t batch, int k, int l, int transposed,
                  const int32_t *d_Ahat, const int32_t *d_y, int32_t *d_t, void *stream);

/* ---- CDT Gaussian sampler -------------------------------------------------------------- */

/* replaces: create_sampler(CDF_GAUSSIAN_SAMPLING, SAMPLING_64BIT, ...) ->
 * gaussian_cdf_create_64 (src/utils/sampling/sampling.c:425-469, gaussian_cdf.c:555-610).
 * The table is built on the host in long double exactly as the reference does and uploaded. */
int  sc_hip_cdt_create(sc_hip_ctx_t *ctx, float tail, float sigma, sc_hip_cdt_t **cdt);
/* host-only half of the above (no device needed): number of entries 2^ceil(log2(tail*sigma))
 * and the table itself.  NOTE the reference's table is a function of the host's x87 expl()
 * (f2xm1 differs in the last bit between CPU vendors), so tables are reproducible per host,
 * not across hosts; samples are a pure function of (table, uniform word). */
int  sc_hip_cdt_table_size(float tail, float sigma);
int  sc_hip_cdt_build_table(float tail, float sigma, uint64_t *h_cdf, int size);
/* the table gaussian_cdf_create_64 builds for a sampler created with BLINDING_SAMPLES (gaussian_cdf.c:586-589): the plain sampler's size, tabulated at
 * sigma / sqrt(2) - the blinding vector filler (sampling.c:168-191) subtracts a second sample from every coefficient */
int  sc_hip_cdt_build_table_blinding(float tail, float sigma, uint64_t *h_cdf, int size);
/* sampler over a caller-supplied monotone table (size a power of two, 2..4096), e.g. the
 * per-leaf tables Falcon's tree sampler creates on the fly (src/utils/arith/gpv.c:2671-2686)
 * precomputed at key load */
int  sc_hip_cdt_create_from_table(sc_hip_ctx_t *ctx, const uint64_t *h_cdf, int size, sc_hip_cdt_t **cdt);
void sc_hip_cdt_destroy(sc_hip_cdt_t *cdt);
int  sc_hip_cdt_size(const sc_hip_cdt_t *cdt);
int  sc_hip_cdt_table(const sc_hip_cdt_t *cdt, uint64_t *h_cdf);

/* replaces: get_vector_32 -> sample_vector_32 -> gaussian_cdf_sample_64 + binary_search_64
 * (sampling.c:211-228,556-580; gaussian_cdf.c:536-553,639-659) with the uniform 64-bit words
 * (the reference's prng_64 outputs) supplied by the caller: out[i] = sample(x[i]) + centre. */
int sc_hip_cdt_sample(sc_hip_cdt_t *cdt, size_t count, const uint64_t *d_uniform, int32_t centre, int32_t *d_out, void *stream);

/* The sampler at 32-bit precision.
 * replaces: create_sampler(CDF_GAUSSIAN_SAMPLING, SAMPLING_32BIT, ...) -> gaussian_cdf_create_32, and get_vector_32 -> gaussian_cdf_sample_32 +
 * binary_search_32 (src/utils/sampling/gaussian_cdf.c:661-774): a table of 32-bit entries built with FLOAT (float) accumulators, one prng_32 word per
 * sample, "entry < word", the sign from the word's bit 0.  A handle made here is served by sc_hip_cdt32_sample only (and the other way round); size /
 * destroy are sc_hip_cdt_size / sc_hip_cdt_destroy. */
int sc_hip_cdt32_build_table(float tail, float sigma, uint32_t *h_cdf, int size);
int sc_hip_cdt32_create(sc_hip_ctx_t *ctx, float tail, float sigma, sc_hip_cdt_t **cdt);
int sc_hip_cdt32_create_from_table(sc_hip_ctx_t *ctx, const uint32_t *h_cdf, int size, sc_hip_cdt_t **cdt);
int sc_hip_cdt32_table(const sc_hip_cdt_t *cdt, uint32_t *h_cdf);
int sc_hip_cdt32_sample(sc_hip_cdt_t *cdt, size_t count, const uint32_t *d_uniform, int32_t centre, int32_t *d_out, void *stream);

/* The sampler at 128- and 192-bit precision, over a caller-supplied table.
 * replaces: get_vector_32 -> gaussian_cdf_sample_128 / _192 + binary_search_128 / _192 + compare_ge_prec (src/utils/sampling/gaussian_cdf.c:110-184,468-509) of a sampler
 * made by create_sampler(CDF_GAUSSIAN_SAMPLING, SAMPLING_128BIT / SAMPLING_192BIT, ...) - what the reference's BLISS-B harness asks for (test/functional/func_alg_bliss_b.c:118).
 * h_cdf: the table as the reference holds it, [size][W] 64-bit limbs, least significant limb first (u128_t / u192_t with SC_LIMB_BITS = 64), W = precision_bits / 64;
 * d_uniform: [count][W] 64-bit words, the sample's prng_64 draws in the order the reference makes them (limb 0 first, :482-483,498-500).
 * What the reference's search does, and this reproduces: the sign is bit 0 of limb 0; the step to entry b is taken when compare_ge_prec says "x >= entry" - and
 * that function's loop over the limbs ends with the verdict of the MOST SIGNIFICANT limb alone (retval = !(a < b) | (equal & retval): when a limb is equal the first term is
 * already 1), so the limbs below the top one never decide anything; ties in the top limb go to the entry (">=", where the 64-bit sampler has "<").  The fixture
 * tests/golden/cdt_hp_ref.txt holds the reference's own get_vector_32 run over tables whose lower limbs are noise and whose top limbs tie with drawn words.
 * The TABLE is not built here: gauss_cdf_create_high_precision (:186-330) is MPFR arithmetic (sc_mpf_exp, sc_mpf_get_pi) - a libsafecrypto built with MPFR hands its
 * table over; the monotone check is on the top limbs, which is all the search reads.  256-bit precision is refused: the reference's
 * gaussian_cdf_sample_256 never writes the top limb of its word and stores one draw past the end of it (:519-522) - there is no defined result to reproduce.
 * A handle made here is served by sc_hip_cdt_hp_sample only; size / destroy are sc_hip_cdt_size / sc_hip_cdt_destroy. */
int sc_hip_cdt_hp_create_from_table(sc_hip_ctx_t *ctx, int precision_bits, const uint64_t *h_cdf, int size, sc_hip_cdt_t **cdt);
int sc_hip_cdt_hp_precision(const sc_hip_cdt_t *cdt);                  /* 32 / 64 / 128 / 192: what a handle was created at */
int sc_hip_cdt_hp_table(const sc_hip_cdt_t *cdt, uint64_t *h_cdf);      /* [size][W], as handed in */
int sc_hip_cdt_hp_sample(sc_hip_cdt_t *cdt, size_t count, const uint64_t *d_uniform, int32_t centre, int32_t *d_out, void *stream);
int sc_hip_cdt_hp_sample_host(sc_hip_cdt_t *cdt, size_t count, const uint64_t *h_uniform, int32_t centre, int32_t *h_out);   /* host buffers in and out */

/* ---- Dilithium: batched sign / verify under one key --------------------------------------- */

/* the fields of dilithium_set_t the engine needs (src/schemes/sig/dilithium/dilithium_params.h:60-93,
 * values dilithium_params.c:22-113; "Dilithium-III" = set 3: k=6, l=5) */
typedef struct sc_hip_dilithium_params {
    int32_t n, q, q_bits, k, l, d, weight_of_c, gamma_1, gamma_1_bits, gamma_2, eta, z_bits, beta, omega, omega_bits, set;
} sc_hip_dilithium_params_t;
typedef struct sc_hip_dilithium sc_hip_dilithium_t;

/* replaces: dilithium_create (dilithium.c:79) for the arithmetic side */
int  sc_hip_dilithium_create(sc_hip_ctx_t *ctx, const sc_hip_dilithium_params_t *params, sc_hip_dilithium_t **h);
void sc_hip_dilithium_destroy(sc_hip_dilithium_t *h);
/* key material exactly as the reference holds it in sc->privkey->key / sc->pubkey->key (dilithium.c:810-824):
 * s1 [l][n], s2 [k][n], t [k][n] (SINT32), K[32], tr[48]; t1 [k][n], rho[32].  Ahat = the matrix
 * create_rand_product_32_csprng draws from rho ([k][l][n], NTT domain, row-major; module_lwe.c:519-547,641-660):
 * the host's reference-compatible DRBG expands it once per key and hands it over. */
int  sc_hip_dilithium_set_privkey(sc_hip_dilithium_t *h, const int32_t *s1, const int32_t *s2, const int32_t *t,
                                  const uint8_t *K, const uint8_t *tr, const int32_t *Ahat);
int  sc_hip_dilithium_set_pubkey(sc_hip_dilithium_t *h, const int32_t *t1, const uint8_t *rho, const int32_t *Ahat);
size_t sc_hip_dilithium_sig_max(const sc_hip_dilithium_t *h);      /* bytes a signature can take */

/* replaces: safecrypto_sign -> dilithium_sign (src/safecrypto.c:1108, dilithium.c:1397-1958) for `batch`
 * messages.  d_msgs [batch][msg_stride] bytes; d_mlens uint32[batch] or NULL (then every message is fixed_len
 * bytes); d_sigs [batch][sig_stride], sig_stride >= sc_hip_dilithium_sig_max(); d_siglens uint32[batch];
 * d_trials (optional) uint32[batch] = attempts taken.  Signatures are byte-identical to the reference's
 * (deterministic signing).  Synchronous with respect to `stream` on return (the rejection loop reads back
 * one counter per round). */
int sc_hip_dilithium_sign(sc_hip_dilithium_t *h, size_t batch, const uint8_t *d_msgs, size_t msg_stride, const uint32_t *d_mlens,
                          uint32_t fixed_len, uint8_t *d_sigs, size_t sig_stride, uint32_t *d_siglens, uint32_t *d_trials, void *stream);
/* replaces: safecrypto_verify -> dilithium_verify (dilithium.c:2002-2240).  d_verdicts[i] = 0 (SC_FUNC_SUCCESS)
 * for a valid signature, 1 otherwise.  Asynchronous on `stream`. */
int sc_hip_dilithium_verify(sc_hip_dilithium_t *h, size_t batch, const uint8_t *d_msgs, size_t msg_stride, const uint32_t *d_mlens,
                            uint32_t fixed_len, const uint8_t *d_sigs, size_t sig_stride, const uint32_t *d_siglens,
                            uint32_t *d_verdicts, void *stream);
int sc_hip_dilithium_sign_host(sc_hip_dilithium_t *h, size_t batch, const uint8_t *h_msgs, size_t msg_stride, const uint32_t *h_mlens,
                               uint32_t fixed_len, uint8_t *h_sigs, size_t sig_stride, uint32_t *h_siglens, uint32_t *h_trials);
int sc_hip_dilithium_verify_host(sc_hip_dilithium_t *h, size_t batch, const uint8_t *h_msgs, size_t msg_stride, const uint32_t *h_mlens,
                                 uint32_t fixed_len, const uint8_t *h_sigs, size_t sig_stride, const uint32_t *h_siglens, uint32_t *h_verdicts);

/* ---- remaining sc_ntt / sc_poly / sc_vec entries, batched (csrc/ops.hip) -------------------- */

/* replaces: pwr_32 (src/utils/arith/ntt_template.c.in:1689-1720) elementwise: d_out[i] = d_x[i]^e mod q, canonical; e >= 0 */
int sc_hip_ntt_pwr(sc_hip_plan_t *plan, size_t len, const int32_t *d_x, int32_t e, int32_t *d_out, void *stream);
/* replaces: flip_32 = ntt32_flip_generic (src/utils/arith/ntt.c:584-604) for `batch` polynomials of the plan's n: out[0] = -in[0], out[i] = in[n-i],
 * then one conditional +q and one conditional -q per coefficient (inputs in (-q, 2q) come out canonical).  d_out must not alias d_in. */
int sc_hip_ntt_flip(sc_hip_plan_t *plan, size_t batch, const int32_t *d_in, int32_t *d_out, void *stream);
/* replaces: invert_32 (ntt_template.c.in:1722-1747) for `batch` arrays of n coefficients (any n >= 1; the plan supplies q), in place: v <- v^(q-2).
 * d_rc[i] = 0 (SC_FUNC_SUCCESS) or 1 (SC_FUNC_FAILURE: polynomial i has a coefficient = 0 mod q; as in the reference the
 * coefficients before the first such one are inverted and the rest left untouched).  Outputs canonical. */
int sc_hip_ntt_invert(sc_hip_plan_t *plan, size_t batch, size_t n, int32_t *d_v, int32_t *d_rc, void *stream);
/* replaces: div_32 (ntt_template.c.in:1749-1775): num <- num * den^(q-2), same failure rule */
int sc_hip_ntt_div(sc_hip_plan_t *plan, size_t batch, size_t n, int32_t *d_num, const int32_t *d_den, int32_t *d_rc, void *stream);
/* replaces: mul_32_sparse / mul_32_sparse_16 (src/utils/arith/ntt.c:381-422; ring_tesla.c:856-871,1015-1023):
 * out[j] = sum_k (j < pos_k ? t[j + n - pos_k] : -t[j - pos_k]) in exact (wrapping) integers, i.e. -t * sum_k x^pos_k in
 * Z[x]/(x^n+1) as the reference computes it; d_pos int32 [batch][omega], d_t [batch][n]; d_out must not alias d_t */
int sc_hip_mult_sparse(sc_hip_ctx_t *ctx, size_t batch, int n, int omega, const int32_t *d_t, const int32_t *d_pos, int32_t *d_out, void *stream);
int sc_hip_mult_sparse_16(sc_hip_ctx_t *ctx, size_t batch, int n, int omega, const int16_t *d_t, const int32_t *d_pos, int32_t *d_out, void *stream);
/* replaces: utils_arith_poly mul_32_scalar, mod_negate_32 (src/utils/arith/poly_32.c:46-52,149-155) on flat arrays (16-byte
 * aligned), add_32_scalar (:38-43; coefficient 0 of each of `batch` polynomials of n coefficients, per-polynomial scalars;
 * sub_32_scalar = negated scalars), cmp_not_equal_32 (:157-165; d_out[i] = OR of the XORs, 0 iff equal).
 * add_single_32 / sub_single_32 (:64-82) are sc_hip_poly_add / sc_hip_poly_sub with d_out == d_a. */
int sc_hip_poly_mul_scalar(sc_hip_ctx_t *ctx, size_t len, const int32_t *d_in, int32_t scalar, int32_t *d_out, void *stream);
int sc_hip_poly_mod_negate(sc_hip_ctx_t *ctx, size_t len, int32_t q, const int32_t *d_in, int32_t *d_out, void *stream);
int sc_hip_poly_add_scalar(sc_hip_ctx_t *ctx, size_t batch, size_t n, int32_t *d_polys, const int32_t *d_scalars, void *stream);
int sc_hip_poly_cmp_not_equal(sc_hip_ctx_t *ctx, size_t batch, size_t n, const int32_t *d_a, const int32_t *d_b, int32_t *d_out, void *stream);
/* replaces: utils_arith_poly_t uniform_32 = poly_32_uniform_rand (src/utils/arith/poly_32.c:132-156; callers bliss_b.c:1174,1187, ens_kem.c:631,642,
 * 817-818) for `batch` polynomials: h_c[j] coefficients of value +-(c_len - j), placed by one CSPRNG word per attempt (tape b = the prng_32 stream of
 * the context the reference passes).  n: a power of two <= 1024.  d_used / d_status optional (words consumed; 1 done, 2 tape exhausted -> zeros). */
int sc_hip_poly_uniform(sc_hip_ctx_t *ctx, size_t batch, int n, const uint16_t *h_c, int c_len, const uint32_t *d_tapes, size_t tape_stride,
                        const uint32_t *d_tape_lens, int32_t *d_out, uint32_t *d_used, uint32_t *d_status, void *stream);
/* replaces: utils_arith_vectors absmax_32 / scalar_32 (src/utils/arith/vectors.c:25-60) per polynomial: max |v_j| ; sum t_j u_j
 * (SINT32, wrapping) */
int sc_hip_vec_absmax(sc_hip_ctx_t *ctx, size_t batch, size_t n, const int32_t *d_v, int32_t *d_out, void *stream);
int sc_hip_vec_scalar(sc_hip_ctx_t *ctx, size_t batch, size_t n, const int32_t *d_t, const int32_t *d_u, int32_t *d_out, void *stream);

/* ---- the reference's default CSPRNG on the device (csrc/drbg.hip, drbg_dev.h) ------------------ */

/* replaces: create_csprng(sc, seed, seed_len) (src/utils/arith/module_lwe.c:914-940; AES-256 CTR-DRBG of
 * src/utils/crypto/ctr_drbg.c with the seed as circular user entropy) followed by prng_mem (src/utils/crypto/prng.c:1050-1107),
 * for `batch` independent seeds d_seeds [batch][seed_stride]: d_out row i = the first nbytes of generator i's byte stream
 * (nbytes and out_stride multiples of 16; a prng_mem(len) call consumes ceil(len/64)*64 bytes of this stream). */
int sc_hip_drbg_mem(sc_hip_ctx_t *ctx, size_t batch, const uint8_t *d_seeds, size_t seed_stride, size_t seed_len,
                    uint8_t *d_out, size_t out_stride, size_t nbytes, void *stream);
/* same generators read through prng_32 / prng_64 (prng.c:95-143,963-993: 64-bit little-endian draws, high word first):
 * d_out row i = the first nwords pool words - the "tape" format sc_hip_bliss_sign / sc_hip_falcon_sign consume
 * (nwords and out_stride_words multiples of 4) */
int sc_hip_drbg_words(sc_hip_ctx_t *ctx, size_t batch, const uint8_t *d_seeds, size_t seed_stride, size_t seed_len,
                      uint32_t *d_out, size_t out_stride_words, size_t nwords, void *stream);

/* replaces: the npoly calls of uniform_random_ring_q_csprng on create_csprng(sc, rho, rho_len) (src/utils/arith/module_lwe.c:519-547,
 * n = 256) with which Kyber and Dilithium expand their public matrices: d_out int32 [nkeys][npoly][256] in generation order -
 * the `Agen` / `Ahat` argument of sc_hip_kyber_set_pubkey / sc_hip_dilithium_set_*key, one 32-byte seed per key instead of
 * npoly KB.  (As in the reference the entries are 16-bit draws masked to q_bits, minus q when >= q.) */
int sc_hip_mlwe_expand_matrix(sc_hip_ctx_t *ctx, size_t nkeys, const uint8_t *d_rho, size_t rho_stride, size_t rho_len, int npoly, int32_t q, int q_bits,
                              int32_t *d_out, void *stream);

/* ---- the XOF-driven forms of the module-LWE helpers (src/utils/arith/module_lwe.c:330-367,449-512,550-560,751-912) ----
 * The reference compiles two forms of every module-LWE helper: *_csprng (above; what its Kyber and Dilithium are built with, kyber_kem_params.h:31 /
 * dilithium_params.h) and *_xof, driven by a SHAKE context prepared with xof_init / xof_absorb(seed) / xof_final (dilithium.c:866-870 in a build without
 * DILITHIUM_USE_CSPRNG_SAM).  The stream is SHAKE128 (shake_bits = 128; sc->xof of Kyber and Dilithium) or SHAKE256 with the reference's permutation.
 * Every entry takes the byte position of the stream to start at (`skip_bytes`: where the previous call on the same xof context stopped, 0 after xof_final).
 *
 * replaces: the npoly calls of uniform_random_ring_q_xof (module_lwe.c:550-560) with which create_rand_product_32_xof / _16_xof (:751-912) draw the matrix:
 * d_out int32 [nkeys][npoly][n] in generation order - row-major [k][l] for the plain product, [l][k] for the transposed one (the order the loops of
 * :776-800 / :806-830 visit it), i.e. the d_Ahat argument of sc_hip_matvec either way.  An entry is a 32-bit little-endian word of the stream masked to
 * q_bits, minus q when >= q (NOT the 16-bit draws of the CSPRNG form).  create_rand_product_*_xof = this call, then sc_hip_matvec.
 * NOTE on the transposed product: sc_hip_matvec computes A^T y, as create_rand_product_16_xof / _16_csprng do (module_lwe.c:875).  The _32 forms'
 * transposed branch, which no caller of the reference reaches, has add_32(t_i, n, out, c) in place of the accumulation (:629, :796) and returns
 * INTT(A[l-1][i] o (NTT(y[l-1]) + 1)) for l > 1; that is not served (the test suite's CPU checker restates it to explain the fixture). */
int sc_hip_mlwe_expand_matrix_xof(sc_hip_ctx_t *ctx, size_t nkeys, const uint8_t *d_seeds, size_t seed_stride, size_t seed_len, int shake_bits, size_t skip_bytes,
                                  int npoly, int n, int32_t q, int q_bits, int32_t *d_out, void *stream);
/* replaces: uniform_rand_sample_xof (kind SC_HIP_XOF_UNIFORM, module_lwe.c:330-344), uniform_rand_sample_small_xof (SC_HIP_XOF_UNIFORM_SMALL, :347-367) and
 * binomial_rand_sample_xof (SC_HIP_XOF_BINOMIAL, :449-485; `bits` unused) for nseeds independent streams: count = m * n values asked for per seed, row i of
 * d_out (int32, pitch out_stride values) receives EXACTLY the entries the reference's function writes and no others - the small form fills whole blocks of
 * 1024 values only (count = 1280 leaves the last 256 entries of the row as they were), the binomial form writes every block of 1024 values at the START of
 * the row (the last block's values stay, the stream still moves by a block each time).  *pos_after (optional, host) = the stream position after the call. */
#define SC_HIP_XOF_UNIFORM        0
#define SC_HIP_XOF_UNIFORM_SMALL  1
#define SC_HIP_XOF_BINOMIAL       2
int sc_hip_mlwe_sample_xof(sc_hip_ctx_t *ctx, int kind, size_t nseeds, const uint8_t *d_seeds, size_t seed_stride, size_t seed_len, int shake_bits, size_t skip_bytes,
                           int32_t eta, int bits, size_t count, int32_t *d_out, size_t out_stride, size_t *pos_after, void *stream);

/* ---- BLISS-B: batched sign / verify under one key ------------------------------------------ */

/* the fields of bliss_set_t the engine needs (src/schemes/sig/bliss_b/bliss_params.h:43-75, values
 * bliss_params.c:25-73; BLISS-B-I = set 1) + the sampler tail (13.42, bliss_b.c:342-344) */
typedef struct sc_hip_bliss_params {
    int32_t n, q, q_bits, n_bits, d, p, kappa, b_inf, b_l2, z1_bits, z2_bits, set;
    float sigma, m, tail;
    int32_t pad;
} sc_hip_bliss_params_t;
typedef struct sc_hip_bliss sc_hip_bliss_t;

int  sc_hip_bliss_create(sc_hip_ctx_t *ctx, const sc_hip_bliss_params_t *params, sc_hip_bliss_t **h);
void sc_hip_bliss_destroy(sc_hip_bliss_t *h);
/* keys as the reference stores them (bliss_b.c:1220-1245,1328-1330): f, g SINT16[n] (NULL, NULL for verify only),
 * a SINT16[n] in the NTT domain */
int  sc_hip_bliss_set_key(sc_hip_bliss_t *h, const int16_t *f, const int16_t *g, const int16_t *a);
size_t sc_hip_bliss_sig_bytes(const sc_hip_bliss_t *h);

/* replaces: safecrypto_sign -> bliss_b_sign (bliss_b.c:1266-1535) for `batch` messages.  The reference draws all of a
 * signature's randomness from its CSPRNG as 32-bit pool words (prng.c:963-1048); the engine owns no CSPRNG, so each
 * message comes with that word stream: d_tapes uint32 [batch][tape_stride], d_tape_lens (or NULL: tape_stride words
 * each).  A fresh attempt consumes 4n words, every attempt 1-2 more.  d_status[i]: 1 signed, 2 tape exhausted,
 * 3 empty message (refused by the reference's oracle), 5 undecided: the rejection draw fell within the tie margin of the threshold
 * 1 / (M exp(-|Sc|^2 / 2 sigma^2) cosh(<z, Sc> / sigma^2)) (bliss_b.c:1420-1428), which the device evaluates with its own f64 exp / cosh
 * and glibc evaluates with its own - such a message (about one attempt in 10^12 at the default margin 2^-40) is not signed here, the
 * caller signs it on the reference path.  Signatures equal the reference's for the same words. */
int sc_hip_bliss_sign(sc_hip_bliss_t *h, size_t batch, const uint8_t *d_msgs, size_t msg_stride, const uint32_t *d_mlens, uint32_t fixed_len,
                      const uint32_t *d_tapes, size_t tape_stride, const uint32_t *d_tape_lens,
                      uint8_t *d_sigs, size_t sig_stride, uint32_t *d_siglens, uint32_t *d_trials, uint32_t *d_status, void *stream);
/* sc_hip_bliss_sign with the reference's default CSPRNG inside the call (csrc/drbg.hip): message i draws from the AES-256 CTR-DRBG the reference makes of
 * d_seeds row i (prng_create / create_csprng with user entropy: src/utils/crypto/prng.c:963-1107, ctr_drbg.c:37-197), read as bliss_b_sign reads
 * sc->prng_ctx[0] (bliss_b.c:1372-1373).  Words are expanded on the device only as far as a message's attempts get and zeroed behind the call; the
 * signatures equal sc_hip_bliss_sign's over sc_hip_drbg_words(seeds) rows.  d_status 2: more than 26 attempts (p ~ 1e-12). */
int sc_hip_bliss_sign_seeded(sc_hip_bliss_t *h, size_t batch, const uint8_t *d_msgs, size_t msg_stride, const uint32_t *d_mlens, uint32_t fixed_len,
                             const uint8_t *d_seeds, size_t seed_stride, size_t seed_len,
                             uint8_t *d_sigs, size_t sig_stride, uint32_t *d_siglens, uint32_t *d_trials, uint32_t *d_status, void *stream);
/* how many CSPRNG words the reference has drawn from sc->prng_ctx[0] when the `attempts`-th attempt of a signature ends (prng.c:963-1048: 4n words per fresh
 * attempt - every other one - and 1 + 27 + 26 reservoir bits per attempt, a word whenever the reservoir runs short): the tape length a signature of that many
 * attempts needs.  Host arithmetic; no device is touched. */
size_t sc_hip_bliss_tape_words(int n, unsigned attempts);
/* the same with a key pair PER MESSAGE (distinct keys; 3 KB of key material per signature instead of a cache-resident set): d_f, d_g, d_a
 * SINT16 [batch][n] rows in the key pair's own layouts - what sc_hip_bliss_keygen writes (f, 2g+1, a in the NTT domain).  No key needs to be loaded. */
int sc_hip_bliss_sign_keys(sc_hip_bliss_t *h, size_t batch, const int16_t *d_f, const int16_t *d_g, const int16_t *d_a,
                           const uint8_t *d_msgs, size_t msg_stride, const uint32_t *d_mlens, uint32_t fixed_len,
                           const uint32_t *d_tapes, size_t tape_stride, const uint32_t *d_tape_lens,
                           uint8_t *d_sigs, size_t sig_stride, uint32_t *d_siglens, uint32_t *d_trials, uint32_t *d_status, void *stream);
int sc_hip_bliss_verify_keys(sc_hip_bliss_t *h, size_t batch, const int16_t *d_a, const uint8_t *d_msgs, size_t msg_stride, const uint32_t *d_mlens, uint32_t fixed_len,
                             const uint8_t *d_sigs, size_t sig_stride, const uint32_t *d_siglens, uint32_t *d_verdicts, void *stream);
/* replaces the sampler table sc_hip_bliss_create built on this host (long double expl: host-dependent in its last bit) by the caller's - the reference
 * context's own (the private struct behind sc->sc_gauss->gauss, gaussian_cdf.c:74-81: cdf, cdf_size): the samples are then the reference's on any host */
int sc_hip_bliss_set_cdt_table(sc_hip_bliss_t *h, const uint64_t *h_cdf, int size);
/* the same for a context created with SC_FLAG_0_SAMPLE_128BIT / _192BIT (the reference's BLISS-B harness: test/functional/func_alg_bliss_b.c:118): the context's table
 * (gauss_cdf_high_t::cdf_128 / cdf_192, [size][precision_bits / 64] limbs: see sc_hip_cdt_hp_create_from_table), after which a Gaussian sample of sc_hip_bliss_sign / _sign_seeded /
 * _sign_keys takes precision_bits / 64 prng_64 draws of the tape - 4 n precision_bits / 64 words per fresh attempt; sc_hip_bliss_tape_words_hp is the matching tape length.
 * Signatures byte-identical to the reference's from such a context (tests/golden/bliss_hp_ref.txt).  sc_hip_bliss_set_cdt_table returns the handle to 64 bits. */
int sc_hip_bliss_set_cdt_table_hp(sc_hip_bliss_t *h, int precision_bits, const uint64_t *h_cdf, int size);
size_t sc_hip_bliss_tape_words_hp(int n, unsigned attempts, int precision_bits);
/* the relative margin of the status-5 report above (default 2^-40; 0 decides everything on the device; 1 reports every attempt - self-test) */
int sc_hip_bliss_set_tie_margin(sc_hip_bliss_t *h, double relative);
/* replaces: sc->coding_signature.type (bliss_b.c:197-200, set from SC_FLAG_0_ENTROPY_HUFFMAN by safecrypto_create).  type 0 = SC_ENTROPY_NONE: the
 * signature's t, z and indices as raw z1_bits / z2_bits / n_bits fields (the default).  type 1 = SC_ENTROPY_HUFFMAN_STATIC: t and z coded as
 * entropy_poly_encode_32 / _16 code them with SIGNED_COEFF (src/utils/entropy/entropy.c:70-112,147-190; entropy_huffman.c:22-57,155-190): the low
 * max(bits - 7, 0) bits of |v| raw, |v| >> beta as a codeword of a static prefix code, a sign bit unless v = 0; the indices stay raw; one MSB-first stream
 * (packer.c:299-351).  The code tables are data of the reference (src/utils/entropy/huffman.h: huff_code_gaussian_<k>[].code / .bits), so the caller hands
 * over the two its coder would walk - for t: k = min(z1_bits, 7) - 1, for z: k = min(z2_bits, 7) - 1, k = 2 when that is below 3 (entropy_huffman.c:27-31) -
 * as codes[depth], bits[depth]; they must form a complete prefix code.  From then on the sign calls write coded signatures (d_siglens[i] varies, below
 * sc_hip_bliss_sig_bytes(), which stays the row size to provide; d_status[i] = 6: a coefficient outside the code's alphabet - the reference's
 * encode_huffman fails on it too - or a coded signature longer than the row) and the verify calls read them (a stream that ends early is invalid).
 * The arithmetic coder (SC_ENTROPY_BAC): sc_hip_bliss_set_entropy_bac below. */
int sc_hip_bliss_set_entropy(sc_hip_bliss_t *h, int type, const uint32_t *codes_t, const uint16_t *bits_t, uint32_t depth_t,
                             const uint32_t *codes_z, const uint16_t *bits_z, uint32_t depth_z);
/* replaces: the signature coding of SC_FLAG_0_ENTROPY_BAC (bliss_b.c:198, 397-398, 1453-1463, 1650-1657): t and z each through bac_encode_64_32 / _16
 * (src/utils/entropy/bac.c:162-258: a byte-aligned block - a 2-byte big-endian length, then that many bytes of a 64-bit binary range coder's output, a
 * coefficient being `bits` binary decisions whose probabilities are dist[prefix | 1 << k]), the indices raw behind them; decoding by bac_decode_64_32 / _16
 * (:260-334).  dist_t (2^z1_bits entries) and dist_z (2^z2_bits): the two distributions as entropy_dist_create -> gauss_freq_bac_64 leaves them in
 * sc->dist[2] and sc->dist[3] (sigma and 0.7; long double arithmetic of the caller's host, so handed over as data like the sampler's table).  From then on
 * the sign calls write coded signatures (d_status[i] = 6: a coded signature that does not fit its row) and the verify calls read them (anything read past a
 * signature's end makes it invalid).  sc_hip_bliss_set_entropy(h, 0, ...) switches back to raw fields.  A range coder is serial from a signature's first
 * bit to its last: one lane per signature. */
int sc_hip_bliss_set_entropy_bac(sc_hip_bliss_t *h, const uint64_t *dist_t, const uint64_t *dist_z);
/* replaces: safecrypto_verify -> bliss_b_verify (bliss_b.c:1583-1740); d_verdicts[i] = 0 valid / 1 invalid */
int sc_hip_bliss_verify(sc_hip_bliss_t *h, size_t batch, const uint8_t *d_msgs, size_t msg_stride, const uint32_t *d_mlens, uint32_t fixed_len,
                        const uint8_t *d_sigs, size_t sig_stride, const uint32_t *d_siglens, uint32_t *d_verdicts, void *stream);

/* replaces: safecrypto_keygen -> bliss_b_keygen (src/schemes/sig/bliss_b/bliss_b.c:1124-1264) for `batch` independent key pairs.  Key
 * pair i consumes tape i (the prng_32 word stream of sc->prng_ctx[0]: poly_32_uniform_rand, src/utils/arith/poly_32.c:132-156, one word
 * per placement attempt; a non-invertible NTT(f) costs another polynomial's worth - allow ~6 (nz0 + nz1) words).  nz0, nz1 =
 * bliss_set_t::nz (bliss_params.c).  Outputs [batch][n] SINT16 as the reference stores them: f, g (after the in-place 2g + 1), a (NTT
 * domain, the first n entries of the public key).  d_attempts (optional): candidates f drawn; d_status[i]: 1 generated, 2 tape exhausted. */
int sc_hip_bliss_keygen(sc_hip_bliss_t *h, size_t batch, int nz0, int nz1, const uint32_t *d_tapes, size_t tape_stride, const uint32_t *d_tape_lens,
                        int16_t *d_f, int16_t *d_g, int16_t *d_a, uint32_t *d_attempts, uint32_t *d_status, void *stream);

/* ---- ring-TESLA: batched sign / verify under one key ---------------------------------------- */

/* ring_tesla_set_t fields (src/schemes/sig/ring_tesla/ring_tesla_params.h:43-67, values ring_tesla_params.c:22-40; sets 0 and 1) */
typedef struct sc_hip_tesla_params {
    int32_t n, n_bits, q, q_bits, b, b_bits, omega, d, bound, u, set, pad;
} sc_hip_tesla_params_t;
typedef struct sc_hip_tesla sc_hip_tesla_t;

int  sc_hip_tesla_create(sc_hip_ctx_t *ctx, const sc_hip_tesla_params_t *params, sc_hip_tesla_t **h);
void sc_hip_tesla_destroy(sc_hip_tesla_t *h);
/* a1, a2: the reference's public polynomials in the NTT domain (a1_0 / a2_0, ring_tesla_params.h:85-88); s, e1, e2: the private key
 * as stored (SINT16[n] each, ring_tesla.c:703-730; NULLs for a verify-only handle); t1, t2: the public key (SINT32[n] each; NULLs
 * for a sign-only handle) */
int  sc_hip_tesla_set_key(sc_hip_tesla_t *h, const int32_t *a1, const int32_t *a2, const int16_t *s, const int16_t *e1, const int16_t *e2,
                          const int32_t *t1, const int32_t *t2);
size_t sc_hip_tesla_sig_bytes(const sc_hip_tesla_t *h);        /* n*q_bits/8 + 64 */
/* replaces: safecrypto_sign -> ring_tesla_sign (src/schemes/sig/ring_tesla/ring_tesla.c:752-910) for `batch` messages; tapes as for
 * BLISS (n CSPRNG pool words per attempt).  d_status[i]: 1 signed, 2 tape exhausted.  Signatures equal the reference's for the
 * same words - including the behaviours of its oracle documented in csrc/tesla.hip. */
int sc_hip_tesla_sign(sc_hip_tesla_t *h, size_t batch, const uint8_t *d_msgs, size_t msg_stride, const uint32_t *d_mlens, uint32_t fixed_len,
                      const uint32_t *d_tapes, size_t tape_stride, const uint32_t *d_tape_lens,
                      uint8_t *d_sigs, size_t sig_stride, uint32_t *d_siglens, uint32_t *d_trials, uint32_t *d_status, void *stream);
/* the same with the CSPRNG inside the call: the masks of message i come straight out of the AES-256 CTR-DRBG seeded with d_seeds row i
 * (ring_tesla.c:818-822 reads sc->prng_ctx[0] through prng_32); no tape exists and none can run out */
int sc_hip_tesla_sign_seeded(sc_hip_tesla_t *h, size_t batch, const uint8_t *d_msgs, size_t msg_stride, const uint32_t *d_mlens, uint32_t fixed_len,
                             const uint8_t *d_seeds, size_t seed_stride, size_t seed_len,
                             uint8_t *d_sigs, size_t sig_stride, uint32_t *d_siglens, uint32_t *d_trials, uint32_t *d_status, void *stream);
/* replaces: safecrypto_verify -> ring_tesla_verify (ring_tesla.c:922-1040); d_verdicts[i] = 0 valid / 1 invalid */
int sc_hip_tesla_verify(sc_hip_tesla_t *h, size_t batch, const uint8_t *d_msgs, size_t msg_stride, const uint32_t *d_mlens, uint32_t fixed_len,
                        const uint8_t *d_sigs, size_t sig_stride, const uint32_t *d_siglens, uint32_t *d_verdicts, void *stream);

/* ---- RLWE-Enc: batched Ring-LWE encryption / decryption under one key pair, batched key generation ---- */

/* rlwe_set_t fields (src/schemes/enc/rlwe_enc/rlwe_enc_params.h:37-61, values rlwe_enc_params.c:23-41); tail = 13.0, the cut
 * rlwe_enc_create hands to create_sampler (rlwe_enc.c:134-136).  Sets: {0, 7681, 13, 256, 8, 3840, 1920, 5760, 4.51f} and
 * {1, 12289, 14, 512, 9, 6144, 3072, 9216, 4.86f}. */
typedef struct sc_hip_rlwe_params {
    int32_t set, q, q_bits, n, n_bits, m_scale, o_scale_0, o_scale_1;
    float sigma, tail;
    int32_t pad[2];
} sc_hip_rlwe_params_t;
typedef struct sc_hip_rlwe sc_hip_rlwe_t;

int  sc_hip_rlwe_create(sc_hip_ctx_t *ctx, const sc_hip_rlwe_params_t *params, sc_hip_rlwe_t **h);
void sc_hip_rlwe_destroy(sc_hip_rlwe_t *h);
/* a, p: the public key as the reference stores it (SINT16[n] each, NTT domain: sc->pubkey->key, rlwe_enc.c:560-561); r2: the
 * private key (SINT16[n], NTT domain, :673).  Host pointers. */
int  sc_hip_rlwe_set_pubkey(sc_hip_rlwe_t *h, const int16_t *a, const int16_t *p);
int  sc_hip_rlwe_set_privkey(sc_hip_rlwe_t *h, const int16_t *r2);
size_t sc_hip_rlwe_msg_bytes(const sc_hip_rlwe_t *h);          /* n/8: one plaintext block (rlwe_enc.c:546-549) */
size_t sc_hip_rlwe_ct_bytes(const sc_hip_rlwe_t *h);           /* 2 n q_bits / 8 */
size_t sc_hip_rlwe_enc_words(const sc_hip_rlwe_t *h);          /* 6n CSPRNG pool words per encryption (3n prng_64 draws) */
size_t sc_hip_rlwe_keygen_words(const sc_hip_rlwe_t *h);       /* n/2 + 2n + n/16 words per key pair */
/* replaces: safecrypto_public_encrypt -> rlwe_enc_encrypt (src/schemes/enc/rlwe_enc/rlwe_enc.c:511-631) for `batch` n-bit
 * messages.  d_tapes[i * tape_stride ...]: the prng_32 word stream of sc->prng_ctx[0] for message i (8-byte aligned, even stride;
 * sc_hip_drbg_words produces it from seeds).  Ciphertext i = c1 (NTT domain, the reference's RLWE_ENC_ENABLE_NTT_TRANSMISSION
 * build, rlwe_enc.h:24) | c2, q_bits per coefficient MSB first - byte for byte the reference's output for the same words.
 * msg_stride, ct_stride: multiples of 4 (4-byte aligned rows). */
int sc_hip_rlwe_encrypt(sc_hip_rlwe_t *h, size_t batch, const uint8_t *d_msgs, size_t msg_stride, const uint32_t *d_tapes, size_t tape_stride,
                        uint8_t *d_ct, size_t ct_stride, void *stream);
/* replaces: safecrypto_private_decrypt -> rlwe_enc_decrypt (rlwe_enc.c:633-707); any 2 n q_bits / 8 bytes are accepted, as there */
int sc_hip_rlwe_decrypt(sc_hip_rlwe_t *h, size_t batch, const uint8_t *d_ct, size_t ct_stride, uint8_t *d_msgs, size_t msg_stride, void *stream);
/* replaces: safecrypto_keygen -> rlwe_enc_keygen with a_gen / r1_gen / r2_gen (rlwe_enc.c:205-290, 425-493) for `batch`
 * independent key pairs, one tape each; outputs [batch][n] SINT16 in the reference's storage format (a, p = r1 - a r2, r2) */
int sc_hip_rlwe_keygen(sc_hip_rlwe_t *h, size_t batch, const uint32_t *d_tapes, size_t tape_stride, int16_t *d_a, int16_t *d_p, int16_t *d_r2, void *stream);

/* ---- DLP-IBE: batched encrypt (per-message identity) / decrypt (one user key); extract is sc_hip_dlp_ibe_extract above ---- */

/* dlp_ibe_set_t fields (src/schemes/ibe/dlp/dlp_ibe_params.h:106-125, values dlp_ibe_params.c:22-40: q = 4206593, q_bits = 23, m_scale =
 * 0x201800, l = 18 / 17) and ubits, the width of a packed user-key coefficient (dlp_ibe.c:916-917) */
typedef struct sc_hip_dlp_params {
    int32_t n, q, q_bits, n_bits, set, ubits, m_scale, l;
} sc_hip_dlp_params_t;
typedef struct sc_hip_dlp sc_hip_dlp_t;

int  sc_hip_dlp_create(sc_hip_ctx_t *ctx, const sc_hip_dlp_params_t *params, sc_hip_dlp_t **h);
void sc_hip_dlp_destroy(sc_hip_dlp_t *h);
/* h_ntt: the second half of sc->pubkey->key (NTT(h), SINT32[n], dlp_ibe.c:1190-1191); uk_ntt: sc->dlp_ibe->user_key + n, the NTT of the user
 * secret key as dlp_ibe_secret_key leaves it (:1100-1104).  Host pointers. */
int  sc_hip_dlp_set_pubkey(sc_hip_dlp_t *h, const int32_t *h_ntt);
int  sc_hip_dlp_set_user_key(sc_hip_dlp_t *h, const int32_t *uk_ntt);
size_t sc_hip_dlp_msg_bytes(const sc_hip_dlp_t *h);            /* n / 8 (SC_IBE_MESSAGE_LENGTH_N) */
size_t sc_hip_dlp_ct_bytes(const sc_hip_dlp_t *h);             /* n (2 q_bits - l) / 8 + n / 8 */
size_t sc_hip_dlp_enc_words(const sc_hip_dlp_t *h);            /* CSPRNG words one encryption can touch */
/* replaces: safecrypto_ibe_public_encrypt -> dlp_ibe_encrypt (src/schemes/ibe/dlp/dlp_ibe.c:1140-1370) for `batch` (identity, message) pairs;
 * tape i = the prng_32 word stream of sc->prng_ctx[0] for message i (key bits, then the prng_var(6) reservoir draws).  Ciphertexts equal the
 * reference's for the same words.  d_status[i]: 1 done, 2 tape exhausted. */
int sc_hip_dlp_encrypt(sc_hip_dlp_t *h, size_t batch, const uint8_t *d_ids, size_t id_stride, const uint32_t *d_idlens, uint32_t fixed_len,
                       const uint8_t *d_msgs, size_t msg_stride, const uint32_t *d_tapes, size_t tape_stride, const uint32_t *d_tape_lens,
                       uint8_t *d_ct, size_t ct_stride, uint32_t *d_status, void *stream);
/* replaces: safecrypto_private_decrypt -> dlp_ibe_decrypt (dlp_ibe.c:1372-1520) under the loaded user key */
int sc_hip_dlp_decrypt(sc_hip_dlp_t *h, size_t batch, const uint8_t *d_ct, size_t ct_stride, uint8_t *d_msgs, size_t msg_stride, void *stream);

/* ---- ENS-KEM: batched encapsulation / decapsulation under one key pair ------------------------------ */

/* ens_kem_set_t fields (src/schemes/kem/ens/ens_kem_params.h:30-47, values ens_kem_params.c:23-61): set, q, q_bits, n, n_bits and the
 * 13 coefficient counts coeff_rnd (the first 12 are used: coeff_rnd[j] coefficients of value +-(12 - j), ens_kem.c:817-818) */
typedef struct sc_hip_ens_params {
    int32_t set, q, q_bits, n, n_bits, pad0, pad1, coeff_rnd[13];
} sc_hip_ens_params_t;
typedef struct sc_hip_ens sc_hip_ens_t;

int  sc_hip_ens_create(sc_hip_ctx_t *ctx, const sc_hip_ens_params_t *params, sc_hip_ens_t **h);
void sc_hip_ens_destroy(sc_hip_ens_t *h);
/* the keys as the reference holds them (ens_kem.c:609-618, 757-762): h = sc->pubkey->key (SINT16[n], NTT domain, centred); g = the first n
 * SINT16 of sc->privkey->key, inv_g = the n/32 words behind them (1/g mod 2, bit-reversed and packed, :191-211).  Host pointers. */
int  sc_hip_ens_set_pubkey(sc_hip_ens_t *h, const int16_t *hh);
int  sc_hip_ens_set_privkey(sc_hip_ens_t *h, const int16_t *g, const uint32_t *inv_g);
size_t sc_hip_ens_ct_bytes(const sc_hip_ens_t *h);             /* n q_bits / 8 */
size_t sc_hip_ens_key_bytes(const sc_hip_ens_t *h);            /* n / 8 */
/* replaces: safecrypto_encapsulation -> ens_kem_encapsulation (src/schemes/kem/ens/ens_kem.c:785-865) `batch` times.  Operation i draws
 * r and e from tape i (the prng_32 word stream of sc->prng_ctx[0] through poly_32_uniform_rand: one word per placement attempt, ~1.2 x
 * the number of non-zero coefficients per polynomial); d_used (optional): words consumed; d_status: 1 done, 2 tape exhausted.
 * Ciphertexts and keys equal the reference's for the same words. */
int sc_hip_ens_encaps(sc_hip_ens_t *h, size_t batch, const uint32_t *d_tapes, size_t tape_stride, const uint32_t *d_tape_lens,
                      uint8_t *d_ct, size_t ct_stride, uint8_t *d_keys, size_t key_stride, uint32_t *d_used, uint32_t *d_status, void *stream);
/* replaces: safecrypto_decapsulation -> ens_kem_decapsulation (ens_kem.c:867-960); any n q_bits / 8 bytes are accepted, as there */
int sc_hip_ens_decaps(sc_hip_ens_t *h, size_t batch, const uint8_t *d_ct, size_t ct_stride, uint8_t *d_keys, size_t key_stride, void *stream);

/* ---- Kyber (round 1, q = 7681): batched CPA encryption / decryption under one key ------------ */

/* kyber_set_t fields (src/utils/arith/module_lwe.h:70-93, values kyber_kem_params.c:24-49; set 1: k = 3) */
typedef struct sc_hip_kyber_params {
    int32_t n, q, q_bits; uint32_t q_inv; int32_t q_norm, k, eta, d_u, d_v, d_t, set, pad[5];
} sc_hip_kyber_params_t;
typedef struct sc_hip_kyber sc_hip_kyber_t;

int  sc_hip_kyber_create(sc_hip_ctx_t *ctx, const sc_hip_kyber_params_t *params, sc_hip_kyber_t **h);
void sc_hip_kyber_destroy(sc_hip_kyber_t *h);
/* t_ntt [k][n]: NTT(t) as the public key stores it (kyber_kem.c:636-650); Agen [k*k][n]: the matrix polynomials in the order
 * the CSPRNG seeded with rho generates them (module_lwe.c:519-547) - the host expands them once per key */
int  sc_hip_kyber_set_pubkey(sc_hip_kyber_t *h, const int32_t *t_ntt, const int32_t *Agen);
int  sc_hip_kyber_set_privkey(sc_hip_kyber_t *h, const int32_t *s_ntt);      /* NTT(s) [k][n], kyber_kem.c:624-639 */
size_t sc_hip_kyber_noise_bytes(const sc_hip_kyber_t *h);                   /* (2k+1)*n*eta/4 */

/* replaces: kyber_cpa_enc (module_lwe.c:1268-1401) for `batch` 32-byte messages d_m [batch][32]; d_noise [batch][noise_stride]
 * holds, per message, the bytes binomial_rand_sample_csprng draws from the CSPRNG seeded with r (r, e1, e2 in that order);
 * outputs the compressed ciphertext polynomials u int32 [batch][k][n] (d_u bits) and v int32 [batch][n] (d_v bits), both 16-byte aligned */
int sc_hip_kyber_cpa_enc(sc_hip_kyber_t *h, size_t batch, const uint8_t *d_m, const uint8_t *d_noise, size_t noise_stride,
                         int32_t *d_u, int32_t *d_v, void *stream);
/* replaces: kyber_cpa_dec (module_lwe.c:1415-1481); u, v are not modified; d_m [batch][32] */
int sc_hip_kyber_cpa_dec(sc_hip_kyber_t *h, size_t batch, const int32_t *d_u, const int32_t *d_v, uint8_t *d_m, void *stream);

/* ---- Falcon (binary rings, sets 0 and 2): batched tree sampler, sign / verify under one key -- */

/* falcon_set_t fields (src/schemes/sig/falcon/falcon_params.h:38-55, values falcon_params.c:24-55) */
typedef struct sc_hip_falcon_params {
    int32_t n, q, q_bits, n_bits, set;
    float bd;
    int32_t pad[2];
} sc_hip_falcon_params_t;
typedef struct sc_hip_falcon sc_hip_falcon_t;

int  sc_hip_falcon_create(sc_hip_ctx_t *ctx, const sc_hip_falcon_params_t *params, sc_hip_falcon_t **h);
void sc_hip_falcon_destroy(sc_hip_falcon_t *h);
/* sk: the expanded private key sc->falcon->sk that load_skey builds (src/schemes/sig/falcon/falcon.c:786-790,
 * src/utils/arith/falcon_keygen.c:6107-6170; offsets falcon_keygen.h:107-136): b00 | b01 | b10 | b11 in FFT form and the
 * normalised LDL tree, (n_bits + 5) * n doubles.  The n per-leaf CDF tables the reference builds on the fly
 * (src/utils/arith/gpv.c:2671-2686) are tabulated here, once, on the host. */
int  sc_hip_falcon_set_privkey(sc_hip_falcon_t *h, const double *sk);
/* pk: h in the coefficient domain, SINT16[n] (first half of sc->pubkey->key, falcon.c:772-780) */
int  sc_hip_falcon_set_pubkey(sc_hip_falcon_t *h, const int16_t *pk);
size_t sc_hip_falcon_sig_bytes(const sc_hip_falcon_t *h);      /* n * (q_bits - 2) / 8 */
/* key sets (distinct keys): nkeys expanded private keys / public keys back to back in the layouts above, resident on the device
 * ((n_bits + 5) n doubles + n leaf tables per private key: 0.4 MB for Falcon-1024).  set_privkey / set_pubkey load a set of one; the one-key calls
 * use key 0.  sc_hip_falcon_sign_keys / _verify_keys take the key index of every message (d_key_of [batch]); an index outside the set gives
 * d_status 6 / verdict 1.  Submit messages GROUPED BY KEY where possible: a run of 8 consecutive messages (aligned to 8) under one key is signed by the
 * same kernel as a one-key batch (7.0 M sign/s at batch 8192 over 64 keys); a group of 8 that mixes keys runs 8 lanes per signature with every key read
 * where it lies (2.6 M sign/s when no two neighbours share a key).  The bytes are the same either way. */
int  sc_hip_falcon_set_privkeys(sc_hip_falcon_t *h, size_t nkeys, const double *sk);
int  sc_hip_falcon_set_pubkeys(sc_hip_falcon_t *h, size_t nkeys, const int16_t *pk);
/* how the tree sampler is laid over the machine; results are identical (the same double operations in the same order).  0 (default) = 64:
 * a wavefront owns 8 signatures, tree levels below 128 coefficients in LDS; 1: a lane per signature, 8: 8 lanes per signature, working
 * polynomials interleaved across the batch in HBM (the first generation, kept for comparison and for the parity tests) */
int  sc_hip_falcon_set_mapping(sc_hip_falcon_t *h, int lanes);

/* replaces: gaussian_sample_with_tree (src/utils/arith/gpv.c:2961-3090, ter == 0) for `batch` targets d_c int32 [batch][n].
 * Randomness as for BLISS: d_tapes uint32 [batch][tape_stride] holds each target's CSPRNG pool words (4n are consumed:
 * two prng_64 per tree leaf, high word first); d_tape_lens NULL = tape_stride words each.  Outputs d_s1 (may be NULL) and
 * d_s2 int32 [batch][n], bit for bit the reference's; d_status[i] (may be NULL) = 2 when tape i is too short, else 0. */
int sc_hip_falcon_sample(sc_hip_falcon_t *h, size_t batch, const int32_t *d_c, const uint32_t *d_tapes, size_t tape_stride, const uint32_t *d_tape_lens,
                         int32_t *d_s1, int32_t *d_s2, uint32_t *d_status, void *stream);
/* replaces: safecrypto_sign -> falcon_sig_sign (src/schemes/sig/falcon/falcon.c:808-935): SHAKE128 hash to the ring, the
 * tree sampler, centring, the norm bound, raw (q_bits-2)-bit packing of s2.  d_status[i]: 1 signed, 2 tape too short,
 * 4 norm bound exceeded (the reference returns SC_FUNC_FAILURE there, it does not retry); d_siglens[i] = 0 unless signed. */
int sc_hip_falcon_sign(sc_hip_falcon_t *h, size_t batch, const uint8_t *d_msgs, size_t msg_stride, const uint32_t *d_mlens, uint32_t fixed_len,
                       const uint32_t *d_tapes, size_t tape_stride, const uint32_t *d_tape_lens,
                       uint8_t *d_sigs, size_t sig_stride, uint32_t *d_siglens, uint32_t *d_status, void *stream);
/* the same with the CSPRNG inside the call: the 4n pool words of signature i are expanded on the device from d_seeds row i (AES-256 CTR-DRBG as the
 * reference seeds it, see sc_hip_bliss_sign_seeded; falcon.c:808-935 draws through gpv.c:2671-2686) and zeroed behind the call */
int sc_hip_falcon_sign_seeded(sc_hip_falcon_t *h, size_t batch, const uint8_t *d_msgs, size_t msg_stride, const uint32_t *d_mlens, uint32_t fixed_len,
                              const uint8_t *d_seeds, size_t seed_stride, size_t seed_len,
                              uint8_t *d_sigs, size_t sig_stride, uint32_t *d_siglens, uint32_t *d_status, void *stream);
int sc_hip_falcon_sign_keys(sc_hip_falcon_t *h, size_t batch, const uint32_t *d_key_of, const uint8_t *d_msgs, size_t msg_stride, const uint32_t *d_mlens, uint32_t fixed_len,
                            const uint32_t *d_tapes, size_t tape_stride, const uint32_t *d_tape_lens,
                            uint8_t *d_sigs, size_t sig_stride, uint32_t *d_siglens, uint32_t *d_status, void *stream);
int sc_hip_falcon_verify_keys(sc_hip_falcon_t *h, size_t batch, const uint32_t *d_key_of, const uint8_t *d_msgs, size_t msg_stride, const uint32_t *d_mlens, uint32_t fixed_len,
                              const uint8_t *d_sigs, size_t sig_stride, const uint32_t *d_siglens, uint32_t *d_verdicts, void *stream);
/* replaces: safecrypto_verify -> falcon_sig_verify (falcon.c:955-1040); d_verdicts[i] = 0 valid / 1 invalid */
int sc_hip_falcon_verify(sc_hip_falcon_t *h, size_t batch, const uint8_t *d_msgs, size_t msg_stride, const uint32_t *d_mlens, uint32_t fixed_len,
                         const uint8_t *d_sigs, size_t sig_stride, const uint32_t *d_siglens, uint32_t *d_verdicts, void *stream);

/* replaces: safecrypto_ibe_extract -> dlp_ibe_extract (src/schemes/ibe/dlp/dlp_ibe.c:874-1043: enhanced extract, XOF id oracle :805-816, H(id) in
 * the NTT domain :935-940) for `batch` identities under one master key.  h: a handle created with the DLP-IBE parameters {n, q = 4206593,
 * q_bits = 23, n_bits, set, bd unused} and loaded (sc_hip_falcon_set_privkey) with sc->dlp_ibe->master_tree - the master key expanded by the same
 * load_skey as Falcon's (dlp_ibe.c:394-425).  ubits = 8 + ceil_log2(floor(6 * 1.17 * sqrt(q / 2n))) (:916-917; 17 for both sets).  Tapes as for
 * Falcon signing (4n CSPRNG words per identity).  d_sk: n * ubits / 8 bytes each - the reference's user secret key for the same words.
 * d_status[i]: 1 extracted, 2 tape exhausted. */
int sc_hip_dlp_ibe_extract(sc_hip_falcon_t *h, size_t batch, int ubits, const uint8_t *d_ids, size_t id_stride, const uint32_t *d_idlens, uint32_t fixed_len,
                           const uint32_t *d_tapes, size_t tape_stride, const uint32_t *d_tape_lens, uint8_t *d_sk, size_t sk_stride, uint32_t *d_status, void *stream);

/* replaces: safecrypto_verify -> ens_dlp_sig_verify (src/schemes/sig/ens_dlp/ens_dlp_sig.c:1564-1665; SC_SCHEME_SIG_ENS / _DLP without message
 * recovery) for `batch` messages under one public key.  h: a handle created with the scheme's parameters {n, q, q_bits, n_bits, set, bd}
 * (ens_dlp_sig_params.c:23-62, n = 256 / 512 / 1024) and loaded (sc_hip_falcon_set_pubkey) with the first n SINT16 of sc->pubkey->key.  Signatures are
 * the reference's wire format (n fields of q_bits bits, signed, MSB first); the signer is sc_hip_ens_dlp_sig_sign below.
 * d_verdicts[i] = 0 valid / 1 invalid. */
int sc_hip_ens_dlp_sig_verify(sc_hip_falcon_t *h, size_t batch, const uint8_t *d_msgs, size_t msg_stride, const uint32_t *d_mlens, uint32_t fixed_len,
                              const uint8_t *d_sigs, size_t sig_stride, const uint32_t *d_siglens, uint32_t *d_verdicts, void *stream);

/* ---- ENS / DLP signing -------------------------------------------------------------------------------------------------------------------
 * replaces: safecrypto_sign -> ens_dlp_sig_sign (src/schemes/sig/ens_dlp/ens_dlp_sig.c:1171-1315; ENS and DLP signatures without message recovery)
 * for a batch of messages under one private key.  The signer is Klein's sampler (gaussian_lattice_sample_flt, src/utils/arith/gpv.c:2409-2492) over
 * the expanded basis and its single-precision Gram-Schmidt orthogonalisation, which the reference context already holds
 * (sc->ens_dlp_sig->b / b_gs / b_gs_inv_norm, built by create_gpv_matrices, ens_dlp_sig.c:364-421) and hands over at key load.
 * params = the scheme's {n, q, q_bits, n_bits, set, bd} (ens_dlp_sig_params.c:23-62; n = 256 / 512 / 1024; bd unused by the signer). */
typedef struct sc_hip_ens_dlp_signer sc_hip_ens_dlp_signer_t;
int  sc_hip_ens_dlp_signer_create(sc_hip_ctx_t *ctx, const sc_hip_falcon_params_t *params, sc_hip_ens_dlp_signer_t **h);
void sc_hip_ens_dlp_signer_destroy(sc_hip_ens_dlp_signer_t *h);
/* host pointers: b SINT32 [2n][2n], b_gs FLOAT [2n][2n], b_gs_inv_norm FLOAT [2n] */
int  sc_hip_ens_dlp_signer_set_key(sc_hip_ens_dlp_signer_t *h, const int32_t *b, const float *b_gs, const float *b_gs_inv_norm);
/* the same from the private key alone: f | g | F | G (SINT32, 4n words = sc->privkey->key); derives the three matrices as create_gpv_matrices does
 * (gpv_expand_basis, modified_gram_schmidt_fast_flt, gpv_precompute_inv_flt: src/utils/arith/gpv.c:709-760, 1021-1117, 977-984), on the host at key load */
int  sc_hip_ens_dlp_signer_set_privkey(sc_hip_ens_dlp_signer_t *h, const int32_t *fgFG);
/* lanes per signature of Klein's sampler: 0 (default) chosen from the batch size; 1 = a lane per signature with the state in HBM, 4 / 8 / 16 = that
 * many lanes with the state in registers (>= n / 64); + 0x100 / + 0x200 pin the register form's table search (entries of the next row held across
 * the sweep / fetched in their own step; default by batch size).  Results are identical. */
int  sc_hip_ens_dlp_signer_set_mapping(sc_hip_ens_dlp_signer_t *h, int lanes);
size_t sc_hip_ens_dlp_sig_bytes(const sc_hip_ens_dlp_signer_t *h);        /* n * q_bits / 8 */
size_t sc_hip_ens_dlp_sig_tape_words(const sc_hip_ens_dlp_signer_t *h);   /* 4n: one prng_64 per basis row */
/* d_tapes row i: the prng_32 word stream of sc->prng_ctx[0] during signature i (8-byte aligned rows, tape_stride even and >= 4n);
 * d_status[i]: 1 signed, 2 tape shorter than 4n words, 3 the sampler state left the 32-bit range (not reachable with a valid key) */
int sc_hip_ens_dlp_sig_sign(sc_hip_ens_dlp_signer_t *h, size_t batch, const uint8_t *d_msgs, size_t msg_stride, const uint32_t *d_mlens, uint32_t fixed_len,
                            const uint32_t *d_tapes, size_t tape_stride, const uint32_t *d_tape_lens, uint8_t *d_sigs, size_t sig_stride, uint32_t *d_status, void *stream);

/* ---- ENS / DLP signatures WITH MESSAGE RECOVERY (SC_SCHEME_SIG_ENS_WITH_RECOVERY / SC_SCHEME_SIG_DLP_WITH_RECOVERY) -------------------------------
 * replaces: safecrypto_sign_with_recovery -> ens_dlp_sig_sign_recovery (src/schemes/sig/ens_dlp/ens_dlp_sig.c:1319-1501) and
 * safecrypto_verify_with_recovery -> ens_dlp_sig_verify_recovery (:1663-1826), with the XOF random oracles the default build compiles (ens_dlp_sig.h:31).
 * Handles as for the plain signatures, created with k_recovery (ens_dlp_sig_params.c:23-62: 19 / 20) in params.pad[0].  The first
 * (n - k)(q_bits - 1) bits of a message (m1) travel inside the signature s1 | s2 (2 n q_bits / 8 bytes); the signer hands back the rest (m2).
 * Out of place where the reference rewrites the caller's message buffer: row i of d_m2 / d_m2lens[i] = what the reference leaves in *m / *m_len
 * (m2_stride >= the longest message less the whole m1 bytes).  d_status as sc_hip_ens_dlp_sig_sign. */
size_t sc_hip_ens_dlp_sig_rec_bytes(const sc_hip_ens_dlp_signer_t *h);      /* 2 n q_bits / 8 */
size_t sc_hip_ens_dlp_sig_rec_m1_bytes(const sc_hip_ens_dlp_signer_t *h);   /* ceil((n - k)(q_bits - 1) / 8); 0 when the handle carries no k_recovery */
int sc_hip_ens_dlp_sig_sign_recovery(sc_hip_ens_dlp_signer_t *h, size_t batch, const uint8_t *d_msgs, size_t msg_stride, const uint32_t *d_mlens, uint32_t fixed_len,
                                     const uint32_t *d_tapes, size_t tape_stride, const uint32_t *d_tape_lens, uint8_t *d_sigs, size_t sig_stride,
                                     uint8_t *d_m2, size_t m2_stride, uint32_t *d_m2lens, uint32_t *d_status, void *stream);
/* the verifier (a Falcon-shaped handle with h loaded, as sc_hip_ens_dlp_sig_verify): d_verdicts[i] = 0 valid / 1 invalid; a valid signature's message
 * is rebuilt in row i of d_msgs_out (d_mlens_out[i] bytes: the m1 bytes, then what m2 carried; out_stride >= that) - what the reference leaves in *m
 * when it is handed m2 in a zeroed buffer.  d_m2lens null: no remainders.  A refused signature reports length 0. */
int sc_hip_ens_dlp_sig_verify_recovery(sc_hip_falcon_t *h, size_t batch, const uint8_t *d_sigs, size_t sig_stride, const uint32_t *d_siglens,
                                       const uint8_t *d_m2, size_t m2_stride, const uint32_t *d_m2lens,
                                       uint8_t *d_msgs_out, size_t out_stride, uint32_t *d_mlens_out, uint32_t *d_verdicts, void *stream);

/* KEM level, with the reference's build options (SHA3-512 with its own permutation, AES CTR-DRBG for oracle bytes and noise):
 * extra public-key material that only enters the G hash - t: the coefficient-domain vector as stored (first k*n words of
 * sc->pubkey->key, kyber_kem.c:636-650), rho: the 32-byte matrix seed */
int  sc_hip_kyber_set_kem_pubkey(sc_hip_kyber_t *h, const int32_t *t, const uint8_t *rho);
size_t sc_hip_kyber_ct_bytes(const sc_hip_kyber_t *h);                       /* (k*n*d_u + n*d_v)/8 + 32 */
/* replaces: safecrypto_encapsulation -> kyber_kem_encapsulation (src/schemes/kem/kyber/kyber_kem.c:855-928) for `batch` values
 * d_m [batch][32] (what prng_mem(sc->prng_ctx[0], m, 32) returned): ciphertexts d_ct [batch][ct_stride] in the reference's wire
 * format (u | v | d) and 32-byte shared keys d_keys [batch][32] */
int sc_hip_kyber_kem_encaps(sc_hip_kyber_t *h, size_t batch, const uint8_t *d_m, uint8_t *d_ct, size_t ct_stride, uint8_t *d_keys, void *stream);
/* replaces: safecrypto_decapsulation -> kyber_kem_decapsulation (kyber_kem.c:930-1040); needs set_privkey too */
int sc_hip_kyber_kem_decaps(sc_hip_kyber_t *h, size_t batch, const uint8_t *d_ct, size_t ct_stride, uint8_t *d_keys, void *stream);

/* the same two with a key PER VALUE (distinct keys), in the layouts sc_hip_kyber_keygen writes: d_t, d_t_ntt, d_s_ntt int32
 * [batch][k][n], d_rho [batch][32].  Every value's matrix is expanded from its rho and its G prefix absorbed from its own (rho, t); no key
 * needs to be loaded into the handle.  Staging costs k*k*n*8 bytes of device memory per value. */
int sc_hip_kyber_kem_encaps_keys(sc_hip_kyber_t *h, size_t batch, const int32_t *d_t, const int32_t *d_t_ntt, const uint8_t *d_rho,
                                 const uint8_t *d_m, uint8_t *d_ct, size_t ct_stride, uint8_t *d_keys, void *stream);
int sc_hip_kyber_kem_decaps_keys(sc_hip_kyber_t *h, size_t batch, const int32_t *d_s_ntt, const int32_t *d_t, const int32_t *d_t_ntt, const uint8_t *d_rho,
                                 const uint8_t *d_ct, size_t ct_stride, uint8_t *d_keys, void *stream);

/* replaces: safecrypto_keygen -> kyber_kem_keygen (kyber_kem.c:575-670; kyber_cpa_keygen, module_lwe.c:1189-1266) for `batch` key pairs.
 * Key i draws what the reference draws from sc->prng_ctx[0] - rho, the noise of s and e, z, in that order - from a CTR-DRBG seeded
 * with d_seeds[i]; its matrix comes from a second generator seeded with rho.  Outputs in the key pair's layouts: d_s_ntt (centred
 * NTT(s)), d_t, d_t_ntt (centred NTT(t)) int32 [batch][k][n]; d_rho, d_z uint8 [batch][32].  The handle needs no key loaded. */
int sc_hip_kyber_keygen(sc_hip_kyber_t *h, size_t batch, const uint8_t *d_seeds, size_t seed_stride, size_t seed_len,
                        int32_t *d_s_ntt, int32_t *d_t, int32_t *d_t_ntt, uint8_t *d_rho, uint8_t *d_z, void *stream);

/* replaces: safecrypto_keygen -> dilithium_keygen (src/schemes/sig/dilithium/dilithium.c:752-935) for `batch` key pairs.  Key i draws
 * what the reference draws from sc->prng_ctx[0] - rho, K, the sample blocks of s1 and of s2 (uniform_rand_sample_small_csprng,
 * module_lwe.c:280-310) - from a CTR-DRBG seeded with d_seeds[i]; its matrix comes from a second generator seeded with rho.
 * eta_bits = dilithium_set_t::eta_bits.  Outputs in the key pair's layouts: d_s1 int32 [batch][l][n]; d_s2, d_t, d_t1 [batch][k][n];
 * d_rho, d_K uint8 [batch][32]; d_tr [batch][48].  The handle needs no key loaded. */
int sc_hip_dilithium_keygen(sc_hip_dilithium_t *h, size_t batch, int eta_bits, const uint8_t *d_seeds, size_t seed_stride, size_t seed_len,
                            int32_t *d_s1, int32_t *d_s2, int32_t *d_t, uint8_t *d_rho, uint8_t *d_K, uint8_t *d_tr, int32_t *d_t1, void *stream);

/* replaces: safecrypto_verify -> dilithium_verify (src/schemes/sig/dilithium/dilithium.c:2073-2260) for a batch of signatures EACH under its own public
 * key ("16384 distinct keys").  d_t1 [batch][k][n] int32 and d_rho [batch][32]: the keys as dilithium_pubkey_load leaves them
 * (t1 | rho, dilithium.c:824,904).  The matrix of every key is regenerated from rho on the device; no handle key needs to be loaded. */
int sc_hip_dilithium_verify_keys(sc_hip_dilithium_t *h, size_t batch, const int32_t *d_t1, const uint8_t *d_rho,
                                 const uint8_t *d_msgs, size_t msg_stride, const uint32_t *d_mlens, uint32_t fixed_len,
                                 const uint8_t *d_sigs, size_t sig_stride, const uint32_t *d_siglens, uint32_t *d_verdicts, void *stream);

/* replaces: safecrypto_sign -> dilithium_sign (dilithium.c:1397-1698) for a batch of messages EACH under its own private key.  Keys in the key pair's
 * layouts (dilithium.c:810-823; what sc_hip_dilithium_keygen writes): d_s1 [batch][l][n], d_s2, d_t [batch][k][n] int32; d_rho, d_K [batch][32];
 * d_tr [batch][48].  Outputs as sc_hip_dilithium_sign. */
int sc_hip_dilithium_sign_keys(sc_hip_dilithium_t *h, size_t batch, const int32_t *d_s1, const int32_t *d_s2, const int32_t *d_t,
                               const uint8_t *d_rho, const uint8_t *d_K, const uint8_t *d_tr,
                               const uint8_t *d_msgs, size_t msg_stride, const uint32_t *d_mlens, uint32_t fixed_len,
                               uint8_t *d_sigs, size_t sig_stride, uint32_t *d_siglens, uint32_t *d_trials, void *stream);

/* ---- static-Huffman coding of signatures that are one polynomial (+ raw bytes) ---------------------------------------------------------------
 * replaces: entropy_poly_encode_32 / _decode_32(..., SIGNED_COEFF, SC_ENTROPY_HUFFMAN_STATIC, ...) followed by raw entropy_poly_encode_8 fields in the same
 * MSB-first stream (src/utils/entropy/entropy.c:70-112,114-145; entropy_huffman.c:22-57,155-190; packer.c:299-351) as the signers use them with
 * SC_FLAG_0_ENTROPY_HUFFMAN: Falcon's s2 (n fields of q_bits - 2 bits, falcon.c:892-903 / 984-993), the ENS / DLP signatures' s1 (n fields of q_bits bits,
 * ens_dlp_sig.c:1275-1287), ring-TESLA's z and digest (n fields of q_bits bits + 64 bytes, ring_tesla.c:880-890).  A format conversion of rows in device
 * memory, run behind sc_hip_<scheme>_sign (raw rows -> coded rows, in place) and in front of sc_hip_<scheme>_verify (coded -> raw rows in a second
 * buffer).  codes / bits: the reference's static table for the field width (src/utils/entropy/huffman.h, huff_table_gaussian_<min(field_bits, 7) - 1>,
 * _2 below 4 bits: data of the reference, handed over as for sc_hip_bliss_set_entropy); it must be a complete prefix code. */
typedef struct sc_hip_huffman sc_hip_huffman_t;
int  sc_hip_huffman_create(sc_hip_ctx_t *ctx, const uint32_t *codes, const uint16_t *bits, uint32_t depth, int field_bits, sc_hip_huffman_t **h);
void sc_hip_huffman_destroy(sc_hip_huffman_t *h);
/* d_rows [batch][row_stride]: raw signatures (n fields, then tail_bytes bytes) -> coded, in place; d_lens[i] = coded length.  d_status (may be NULL):
 * only rows with d_status[i] == 1 are coded; 6 is written for a coefficient outside the table's alphabet or a coded row longer than row_stride */
int sc_hip_huffman_encode_rows(sc_hip_huffman_t *h, size_t batch, uint8_t *d_rows, size_t row_stride, uint32_t *d_lens, uint32_t *d_status,
                               int n, int tail_bytes, void *stream);
/* coded rows -> raw rows in d_out (a buffer of its own, out_stride >= the raw size), d_outlens[i] = raw length, d_bad[i] (may be NULL) = 1 when the
 * stream ends before the last field (the reference's decoder fails there) */
int sc_hip_huffman_decode_rows(sc_hip_huffman_t *h, size_t batch, const uint8_t *d_rows, size_t row_stride, const uint32_t *d_lens,
                               uint8_t *d_out, size_t out_stride, uint32_t *d_outlens, uint32_t *d_bad, int n, int tail_bytes, void *stream);

/* ---- multi-device driver --------------------------------------------------------------------------------
 * Independent units shard with no collective: a host batch is cut into contiguous slices (sc_hip_shard_range), one host thread per
 * worker drives its device (>= 2 streams per device inside the per-device host entry points), results are written at the slice's
 * offset of the caller's buffers.  devices: HIP device ordinals, one worker each (NULL: 0 .. nworkers-1); an ordinal may repeat. */
typedef struct sc_hip_multi sc_hip_multi_t;
typedef struct sc_hip_multi_plan sc_hip_multi_plan_t;
typedef struct sc_hip_multi_dilithium sc_hip_multi_dilithium_t;
void sc_hip_shard_range(size_t total, int parts, int index, size_t *begin, size_t *count);   /* slice `index` of `parts`: sizes differ by at most one */
int  sc_hip_multi_create(const int *devices, int nworkers, sc_hip_multi_t **m);
void sc_hip_multi_destroy(sc_hip_multi_t *m);
int  sc_hip_multi_workers(const sc_hip_multi_t *m);
sc_hip_ctx_t *sc_hip_multi_ctx(const sc_hip_multi_t *m, int worker);                         /* a worker's context, for the batched per-device ABI */
/* the plan replicated on every worker; the *_host transforms of this header over the whole host batch */
int  sc_hip_multi_plan_create(sc_hip_multi_t *m, int32_t q, int n, sc_hip_multi_plan_t **plan);
void sc_hip_multi_plan_destroy(sc_hip_multi_plan_t *plan);
int  sc_hip_multi_ntt_fwd_host(sc_hip_multi_plan_t *plan, size_t batch, const int32_t *h_in, int32_t *h_out);
int  sc_hip_multi_ntt_inv_host(sc_hip_multi_plan_t *plan, size_t batch, const int32_t *h_in, int32_t *h_out);
int  sc_hip_multi_polymul_ntt_host(sc_hip_multi_plan_t *plan, size_t batch, const int32_t *h_a, const int32_t *h_bhat, int32_t *h_out);
/* Dilithium sign / verify of a host batch under one key pair (replicated to every worker at set_*key); arguments as the one-device *_host entries */
int  sc_hip_multi_dilithium_create(sc_hip_multi_t *m, const sc_hip_dilithium_params_t *params, sc_hip_multi_dilithium_t **h);
void sc_hip_multi_dilithium_destroy(sc_hip_multi_dilithium_t *h);
int  sc_hip_multi_dilithium_set_privkey(sc_hip_multi_dilithium_t *h, const int32_t *s1, const int32_t *s2, const int32_t *t, const uint8_t *K, const uint8_t *tr, const int32_t *Ahat);
int  sc_hip_multi_dilithium_set_pubkey(sc_hip_multi_dilithium_t *h, const int32_t *t1, const uint8_t *rho, const int32_t *Ahat);
size_t sc_hip_multi_dilithium_sig_max(const sc_hip_multi_dilithium_t *h);
int  sc_hip_multi_dilithium_sign_host(sc_hip_multi_dilithium_t *h, size_t batch, const uint8_t *h_msgs, size_t msg_stride, const uint32_t *h_mlens, uint32_t fixed_len,
                                      uint8_t *h_sigs, size_t sig_stride, uint32_t *h_siglens, uint32_t *h_trials);
int  sc_hip_multi_dilithium_verify_host(sc_hip_multi_dilithium_t *h, size_t batch, const uint8_t *h_msgs, size_t msg_stride, const uint32_t *h_mlens, uint32_t fixed_len,
                                        const uint8_t *h_sigs, size_t sig_stride, const uint32_t *h_siglens, uint32_t *h_verdicts);

/* BLISS-B, Falcon, ring-TESLA sign / verify and Kyber encapsulation / decapsulation of a host batch under one key (replicated to every worker by the
 * set_* call); arguments as the one-device *_host entries below.  BASELINE configs[3] ("BLISS-B-I ... 1 -> 8 MI355X shard") and configs[4]
 * ("Falcon-1024 ... 8 x MI355X") in one process. */
typedef struct sc_hip_multi_bliss sc_hip_multi_bliss_t;
typedef struct sc_hip_multi_falcon sc_hip_multi_falcon_t;
typedef struct sc_hip_multi_tesla sc_hip_multi_tesla_t;
typedef struct sc_hip_multi_kyber sc_hip_multi_kyber_t;
int  sc_hip_multi_bliss_create(sc_hip_multi_t *m, const sc_hip_bliss_params_t *params, sc_hip_multi_bliss_t **h);
void sc_hip_multi_bliss_destroy(sc_hip_multi_bliss_t *h);
int  sc_hip_multi_bliss_set_key(sc_hip_multi_bliss_t *h, const int16_t *f, const int16_t *g, const int16_t *a);
/* the sampler table on every worker (precision_bits 64: sc_hip_bliss_set_cdt_table; 128 / 192: sc_hip_bliss_set_cdt_table_hp) */
int  sc_hip_multi_bliss_set_cdt_table(sc_hip_multi_bliss_t *h, int precision_bits, const uint64_t *h_cdf, int size);
size_t sc_hip_multi_bliss_sig_bytes(const sc_hip_multi_bliss_t *h);
int  sc_hip_multi_bliss_sign_host(sc_hip_multi_bliss_t *h, size_t batch, const uint8_t *h_msgs, size_t msg_stride, const uint32_t *h_mlens, uint32_t fixed_len,
                                  const uint32_t *h_tapes, size_t tape_stride, const uint32_t *h_tape_lens,
                                  uint8_t *h_sigs, size_t sig_stride, uint32_t *h_siglens, uint32_t *h_trials, uint32_t *h_status);
int  sc_hip_multi_bliss_verify_host(sc_hip_multi_bliss_t *h, size_t batch, const uint8_t *h_msgs, size_t msg_stride, const uint32_t *h_mlens, uint32_t fixed_len,
                                    const uint8_t *h_sigs, size_t sig_stride, const uint32_t *h_siglens, uint32_t *h_verdicts);
int  sc_hip_multi_falcon_create(sc_hip_multi_t *m, const sc_hip_falcon_params_t *params, sc_hip_multi_falcon_t **h);
void sc_hip_multi_falcon_destroy(sc_hip_multi_falcon_t *h);
int  sc_hip_multi_falcon_set_privkey(sc_hip_multi_falcon_t *h, const double *sk);
int  sc_hip_multi_falcon_set_pubkey(sc_hip_multi_falcon_t *h, const int16_t *pk);
size_t sc_hip_multi_falcon_sig_bytes(const sc_hip_multi_falcon_t *h);
int  sc_hip_multi_falcon_sign_host(sc_hip_multi_falcon_t *h, size_t batch, const uint8_t *h_msgs, size_t msg_stride, const uint32_t *h_mlens, uint32_t fixed_len,
                                   const uint32_t *h_tapes, size_t tape_stride, const uint32_t *h_tape_lens,
                                   uint8_t *h_sigs, size_t sig_stride, uint32_t *h_siglens, uint32_t *h_status);
int  sc_hip_multi_falcon_verify_host(sc_hip_multi_falcon_t *h, size_t batch, const uint8_t *h_msgs, size_t msg_stride, const uint32_t *h_mlens, uint32_t fixed_len,
                                     const uint8_t *h_sigs, size_t sig_stride, const uint32_t *h_siglens, uint32_t *h_verdicts);
int  sc_hip_multi_tesla_create(sc_hip_multi_t *m, const sc_hip_tesla_params_t *params, sc_hip_multi_tesla_t **h);
void sc_hip_multi_tesla_destroy(sc_hip_multi_tesla_t *h);
int  sc_hip_multi_tesla_set_key(sc_hip_multi_tesla_t *h, const int32_t *a1, const int32_t *a2, const int16_t *s, const int16_t *e1, const int16_t *e2,
                                const int32_t *t1, const int32_t *t2);
size_t sc_hip_multi_tesla_sig_bytes(const sc_hip_multi_tesla_t *h);
int  sc_hip_multi_tesla_sign_host(sc_hip_multi_tesla_t *h, size_t batch, const uint8_t *h_msgs, size_t msg_stride, const uint32_t *h_mlens, uint32_t fixed_len,
                                  const uint32_t *h_tapes, size_t tape_stride, const uint32_t *h_tape_lens,
                                  uint8_t *h_sigs, size_t sig_stride, uint32_t *h_siglens, uint32_t *h_trials, uint32_t *h_status);
int  sc_hip_multi_tesla_verify_host(sc_hip_multi_tesla_t *h, size_t batch, const uint8_t *h_msgs, size_t msg_stride, const uint32_t *h_mlens, uint32_t fixed_len,
                                    const uint8_t *h_sigs, size_t sig_stride, const uint32_t *h_siglens, uint32_t *h_verdicts);
int  sc_hip_multi_kyber_create(sc_hip_multi_t *m, const sc_hip_kyber_params_t *params, sc_hip_multi_kyber_t **h);
void sc_hip_multi_kyber_destroy(sc_hip_multi_kyber_t *h);
int  sc_hip_multi_kyber_set_pubkey(sc_hip_multi_kyber_t *h, const int32_t *t_ntt, const int32_t *Agen, const int32_t *t, const uint8_t *rho);   /* set_pubkey + set_kem_pubkey */
int  sc_hip_multi_kyber_set_privkey(sc_hip_multi_kyber_t *h, const int32_t *s_ntt);
size_t sc_hip_multi_kyber_ct_bytes(const sc_hip_multi_kyber_t *h);
int  sc_hip_multi_kyber_kem_encaps_host(sc_hip_multi_kyber_t *h, size_t batch, const uint8_t *h_m, uint8_t *h_ct, size_t ct_stride, uint8_t *h_keys);
int  sc_hip_multi_kyber_kem_decaps_host(sc_hip_multi_kyber_t *h, size_t batch, const uint8_t *h_ct, size_t ct_stride, uint8_t *h_keys);

/* ---- host-buffer convenience (H2D, kernel, D2H; synchronous) ---------------------------- */
/* the shape safecrypto_sign / safecrypto_verify (src/safecrypto.c:1108-1139 -> bliss_b.c:1266,1583; falcon.c:808,955; ring_tesla.c:752,922) and
 * safecrypto_encapsulation / safecrypto_decapsulation (-> kyber_kem.c:855,930) have, for a batch in the caller's memory; arguments as the device
 * entries above with host pointers, h_status / h_trials as d_status / d_trials there */
int sc_hip_bliss_sign_host(sc_hip_bliss_t *h, size_t batch, const uint8_t *h_msgs, size_t msg_stride, const uint32_t *h_mlens, uint32_t fixed_len,
                           const uint32_t *h_tapes, size_t tape_stride, const uint32_t *h_tape_lens,
                           uint8_t *h_sigs, size_t sig_stride, uint32_t *h_siglens, uint32_t *h_trials, uint32_t *h_status);
int sc_hip_bliss_verify_host(sc_hip_bliss_t *h, size_t batch, const uint8_t *h_msgs, size_t msg_stride, const uint32_t *h_mlens, uint32_t fixed_len,
                             const uint8_t *h_sigs, size_t sig_stride, const uint32_t *h_siglens, uint32_t *h_verdicts);
int sc_hip_tesla_sign_host(sc_hip_tesla_t *h, size_t batch, const uint8_t *h_msgs, size_t msg_stride, const uint32_t *h_mlens, uint32_t fixed_len,
                           const uint32_t *h_tapes, size_t tape_stride, const uint32_t *h_tape_lens,
                           uint8_t *h_sigs, size_t sig_stride, uint32_t *h_siglens, uint32_t *h_trials, uint32_t *h_status);
int sc_hip_tesla_verify_host(sc_hip_tesla_t *h, size_t batch, const uint8_t *h_msgs, size_t msg_stride, const uint32_t *h_mlens, uint32_t fixed_len,
                             const uint8_t *h_sigs, size_t sig_stride, const uint32_t *h_siglens, uint32_t *h_verdicts);
int sc_hip_falcon_sign_host(sc_hip_falcon_t *h, size_t batch, const uint8_t *h_msgs, size_t msg_stride, const uint32_t *h_mlens, uint32_t fixed_len,
                            const uint32_t *h_tapes, size_t tape_stride, const uint32_t *h_tape_lens,
                            uint8_t *h_sigs, size_t sig_stride, uint32_t *h_siglens, uint32_t *h_status);
int sc_hip_falcon_verify_host(sc_hip_falcon_t *h, size_t batch, const uint8_t *h_msgs, size_t msg_stride, const uint32_t *h_mlens, uint32_t fixed_len,
                              const uint8_t *h_sigs, size_t sig_stride, const uint32_t *h_siglens, uint32_t *h_verdicts);
int sc_hip_kyber_kem_encaps_host(sc_hip_kyber_t *h, size_t batch, const uint8_t *h_m, uint8_t *h_ct, size_t ct_stride, uint8_t *h_keys);
int sc_hip_kyber_kem_decaps_host(sc_hip_kyber_t *h, size_t batch, const uint8_t *h_ct, size_t ct_stride, uint8_t *h_keys);
int sc_hip_ntt_fwd_host(sc_hip_plan_t *plan, size_t batch, const int32_t *h_in, int32_t *h_out);
int sc_hip_ntt_inv_host(sc_hip_plan_t *plan, size_t batch, const int32_t *h_in, int32_t *h_out);
int sc_hip_polymul_ntt_host(sc_hip_plan_t *plan, size_t batch, const int32_t *h_a, const int32_t *h_bhat, int32_t *h_out);
int sc_hip_cdt_sample_host(sc_hip_cdt_t *cdt, size_t count, const uint64_t *h_uniform, int32_t centre, int32_t *h_out);
int sc_hip_pointwise_host(sc_hip_plan_t *plan, size_t len, const int32_t *h_a, const int32_t *h_b, int32_t *h_out);
int sc_hip_normalize_host(sc_hip_plan_t *plan, size_t len, const int32_t *h_in, int32_t *h_out);
int sc_hip_center_host(sc_hip_plan_t *plan, size_t len, const int32_t *h_in, int32_t *h_out);
int sc_hip_ntt_invert_host(sc_hip_plan_t *plan, size_t batch, size_t n, int32_t *h_v, int32_t *h_rc);
int sc_hip_ntt_div_host(sc_hip_plan_t *plan, size_t batch, size_t n, int32_t *h_num, const int32_t *h_den, int32_t *h_rc);
int sc_hip_mult_sparse_host(sc_hip_ctx_t *ctx, size_t batch, int n, int omega, const void *h_t, int t_is_16, const int32_t *h_pos, int32_t *h_out);

#ifdef __cplusplus
}
#endif
#endif /* SC_HIP_H */
