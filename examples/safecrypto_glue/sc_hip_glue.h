/*
 * sc_hip_glue.h - the libsafecrypto-side binding to the MI355X engine: the file a libsafecrypto maintainer adds to the library
 * (it is compiled AGAINST libsafecrypto's private headers, and linked with -lsc_hip).  Not part of the engine.
 *
 *   inner boundary  sc_hip_glue_attach():   points sc->sc_ntt / sc->sc_poly / sc->sc_vec / sc->sc_gauss of an existing context at the
 *                                           engine's tables (include/sc_hip_vtable.h); the scheme code is untouched and every
 *                                           polynomial operation it performs runs on the GPU, one polynomial per call.
 *   outer boundary  sc_hip_glue_*_sign ...: entries with the signatures of the scheme table safecrypto_alg_t
 *                                           (src/safecrypto_private.h:392-419) that hand the whole operation - hashing, sampling,
 *                                           rejection loop, packing - to the engine's scheme-level kernels, for one message or a batch.
 */
#ifndef SC_HIP_GLUE_H
#define SC_HIP_GLUE_H
#include "safecrypto.h"
#include "safecrypto_private.h"

typedef struct sc_hip_glue sc_hip_glue_t;

/* inner boundary.  device: the HIP device this host thread works on.  Returns NULL when no device is usable (nothing is changed). */
sc_hip_glue_t *sc_hip_glue_attach(safecrypto_t *sc, int device);
/* puts the context's own tables back (call before safecrypto_destroy) */
void sc_hip_glue_detach(sc_hip_glue_t *g);
/* the sampler's PRNG follows sc->prng_ctx[0]; call after swapping that pointer */
void sc_hip_glue_refresh_prng(sc_hip_glue_t *g);

/* outer boundary, Dilithium: same signatures as the `signing` / `verification` rows (safecrypto_private.h:362-366), same
 * buffer rules as the reference's packer (packer.c:142-176): *sigret == NULL or *siglen == 0 -> a buffer is malloc'ed for the caller */
SINT32 sc_hip_glue_dilithium_sign(safecrypto_t *sc, const UINT8 *m, size_t m_len, UINT8 **sigret, size_t *siglen);
SINT32 sc_hip_glue_dilithium_verify(safecrypto_t *sc, const UINT8 *m, size_t m_len, const UINT8 *sigbuf, size_t siglen);
/* the batched form that sits beside safecrypto_sign: `batch` messages of msg_stride bytes (lengths mlens[i]) under sc's key,
 * signatures into sigs[i * sig_stride], lengths into siglens[i] */
SINT32 sc_hip_glue_dilithium_sign_batch(safecrypto_t *sc, size_t batch, const UINT8 *msgs, size_t msg_stride, const UINT32 *mlens,
                                        UINT8 *sigs, size_t sig_stride, UINT32 *siglens);
SINT32 sc_hip_glue_dilithium_verify_batch(safecrypto_t *sc, size_t batch, const UINT8 *msgs, size_t msg_stride, const UINT32 *mlens,
                                          const UINT8 *sigs, size_t sig_stride, const UINT32 *siglens, UINT32 *verdicts);
size_t sc_hip_glue_dilithium_sig_max(safecrypto_t *sc);
/* outer boundary, the other schemes the north star names - the `signing` / `verification` / `encapsulation` / `decapsulation` rows of safecrypto_alg_t
 * (src/safecrypto_private.h:392-419; table src/safecrypto.c:188-215, dispatch :1108-1139) with exactly those signatures, the whole operation on the
 * device.  Keys are read where the reference keeps them (sc->privkey->key / sc->pubkey->key: bliss_b.c:1220-1245, ring_tesla.c:703-730, falcon.c:772-790 and
 * sc->falcon->sk, kyber_kem.c:624-650); randomness is drawn from sc->prng_ctx[0] - the CSPRNG words a signature consumes as prng_32 words (the engine's
 * "tape"), Kyber's m with the reference's own prng_mem call.  A signature equals safecrypto_sign's for the same CSPRNG state whenever that state has an
 * empty bit reservoir (a freshly seeded generator - prng_var's left-over bits of an earlier call are invisible from outside prng.c); the generator is
 * advanced by the whole tape drawn, which for the rejection-sampling signers is more than the reference would have consumed.
 * Output buffers follow packer.c:142-176 like the Dilithium entries.  BLISS-B: the CDF sampler at 64 bits (SC_FLAG_0_SAMPLE_64BIT) or at 128 / 192 bits
 * (SC_FLAG_0_SAMPLE_128BIT / _192BIT - what the reference's own harness creates, test/functional/func_alg_bliss_b.c:118: the context's table is handed to the
 * engine, sc_hip_bliss_set_cdt_table_hp), signatures raw, static-Huffman or arithmetic coded; anything else returns SC_FUNC_FAILURE and the caller stays on
 * the reference path. */
SINT32 sc_hip_glue_bliss_b_sign(safecrypto_t *sc, const UINT8 *m, size_t m_len, UINT8 **sigret, size_t *siglen);
SINT32 sc_hip_glue_bliss_b_verify(safecrypto_t *sc, const UINT8 *m, size_t m_len, const UINT8 *sigbuf, size_t siglen);
SINT32 sc_hip_glue_ring_tesla_sign(safecrypto_t *sc, const UINT8 *m, size_t m_len, UINT8 **sigret, size_t *siglen);
SINT32 sc_hip_glue_ring_tesla_verify(safecrypto_t *sc, const UINT8 *m, size_t m_len, const UINT8 *sigbuf, size_t siglen);
SINT32 sc_hip_glue_falcon_sign(safecrypto_t *sc, const UINT8 *m, size_t m_len, UINT8 **sigret, size_t *siglen);
SINT32 sc_hip_glue_falcon_verify(safecrypto_t *sc, const UINT8 *m, size_t m_len, const UINT8 *sigbuf, size_t siglen);
SINT32 sc_hip_glue_kyber_encapsulation(safecrypto_t *sc, UINT8 **c, size_t *c_len, UINT8 **k, size_t *k_len);
SINT32 sc_hip_glue_kyber_decapsulation(safecrypto_t *sc, const UINT8 *c, size_t c_len, UINT8 **k, size_t *k_len);
/* the batched forms beside them: `batch` messages of msg_stride bytes under sc's key, a CSPRNG tape per message drawn from sc->prng_ctx[0] in message
 * order (tape_words each: sc_hip_glue_tape_words()); status[i] as the engine reports it (1 signed; 2 tape too short - ask for more attempts) */
size_t sc_hip_glue_sig_bytes(safecrypto_t *sc);                                 /* row size of the raw-coded signature (BLISS-B, ring-TESLA, Falcon) */
size_t sc_hip_glue_tape_words(safecrypto_t *sc, unsigned attempts);             /* CSPRNG words a signature of `attempts` attempts consumes */
SINT32 sc_hip_glue_sign_batch(safecrypto_t *sc, size_t batch, const UINT8 *msgs, size_t msg_stride, const UINT32 *mlens, unsigned attempts,
                              UINT8 *sigs, size_t sig_stride, UINT32 *siglens, UINT32 *status);
SINT32 sc_hip_glue_verify_batch(safecrypto_t *sc, size_t batch, const UINT8 *msgs, size_t msg_stride, const UINT32 *mlens,
                                const UINT8 *sigs, size_t sig_stride, const UINT32 *siglens, UINT32 *verdicts);
SINT32 sc_hip_glue_kyber_encapsulation_batch(safecrypto_t *sc, size_t batch, UINT8 *cts, size_t ct_stride, UINT8 *keys);   /* keys [batch][32]; m drawn per value */
SINT32 sc_hip_glue_kyber_decapsulation_batch(safecrypto_t *sc, size_t batch, const UINT8 *cts, size_t ct_stride, UINT8 *keys);
size_t sc_hip_glue_kyber_ct_bytes(safecrypto_t *sc);
/* drops the engine handle cached for this context (keys changed, or before safecrypto_destroy) */
void sc_hip_glue_forget(safecrypto_t *sc);

#endif
