/*
 * sc_hip_bench.h - measurement and self-test hooks of libsc_hip.so.  NOT part of the drop-in ABI (include/sc_hip.h,
 * include/sc_hip_vtable.h): nothing here replaces a reference interface, a libsafecrypto integration never calls these.
 * They exist for bench.py, tools/ and tests/ and may change between rounds.
 */
#ifndef SC_HIP_BENCH_H
#define SC_HIP_BENCH_H
#include "sc_hip.h"
#ifdef __cplusplus
extern "C" {
#endif

/* `steps` back-to-back sc_hip_polymul_ntt calls on the same buffers as ONE HIP graph launch (captured on first use for these arguments,
 * cached in the plan, replayed afterwards); `stream` must not be the null stream and d_out must not alias an input.  Falls back to the
 * plain loop if the capture is refused.  For launch-bound loops: at ~37 us per kernel the host cost of a plain call is ~5 % of the wall
 * time.  Calls on one plan must not overlap (the cached graph belongs to the plan). */
int sc_hip_polymul_ntt_steps(sc_hip_plan_t *plan, size_t batch, const int32_t *d_a, const int32_t *d_bhat, int32_t *d_out, int steps, void *stream);

/* Runs `iters` back-to-back launches of one op (0 fwd, 1 inv, 2 polymul_ntt, 3 polymul) on `stream`, bracketed by hipEvents recorded on
 * that same stream; returns the average per-launch milliseconds in *ms.  bench.py's roofline figure. */
int sc_hip_time_op(sc_hip_plan_t *plan, int op, size_t batch, const int32_t *d_a, const int32_t *d_b,
                   int32_t *d_out, void *stream, int iters, float *ms);

/* the reference's XOF on the device (src/utils/crypto/xof.c: SHAKE128/256 framing around the permutation tinysha3.c runs under
 * -DSHA3_UNROLLED, which is not Keccak-f[1600] - see csrc/keccak_dev.h); standard != 0 selects the FIPS 202 permutation instead
 * (self-test of the sponge framing against hashlib) */
int sc_hip_shake_host(sc_hip_ctx_t *ctx, int bits, int standard, const uint8_t *h_in, size_t len, uint8_t *h_out, size_t outlen);

/* the default mapping of the Falcon tree sampler with per-phase cycle counters: h_cycles[6] = shader cycles summed over the wavefronts for
 * transform + targets | tree levels in HBM | the level-128 node | tree levels in LDS | register sub-trees | lattice point + inverse transforms */
int sc_hip_falcon_sample_profile(sc_hip_falcon_t *h, size_t batch, const int32_t *d_c, const uint32_t *d_tapes, size_t tape_stride,
                                 int32_t *d_s1, int32_t *d_s2, unsigned long long *h_cycles, void *stream);

/* pin the variant of the Falcon sampler's default mapping: signatures per wavefront (8 or 4) and the sub-tree level kept in LDS (7 or 6);
 * spw may carry the wavefronts per workgroup in bits 8-9 (1 or 2; 2 needs n = 1024 and the (8, 7) variant).  0, 0: chosen per batch (the production
 * setting).  Results do not depend on it. */
int sc_hip_falcon_set_variant(sc_hip_falcon_t *h, int spw, int lm);

/* Dilithium signer: (slots_per_message & 0xFF) > 0 fixes the number of attempts evaluated side by side per message in the first round (0: the
 * built-in rule, about 48 K slots per round); bit 8 set: the host looks at the count of unsigned messages after EVERY round (round 2's loop) instead
 * of queueing the planned rounds back to back; bit 9 set: ExpandMask with one lane per attempt at every batch size (default: two lanes while SIMDs
 * would idle); bit 10 set (one-key calls): w = A y between the round's kernels in 24 bits per coefficient instead of 32-bit words (-9 % HBM traffic, +1 % time); bit 11 set
 * (one-key calls): the finishing kernel WITHOUT transforms (c s as sums of shifted table rows; NOT YET RUN ON HARDWARE - see DESIGN.md 4.4) - honoured at key load
 * only: its tables are made from the raw key inside sc_hip_dilithium_set_privkey and no host copy of them is kept, so setting the bit on a handle whose key was loaded
 * without it FAILS (load the key again afterwards); unfused != 0 runs
 * NTT(y) and the matrix product as two kernels instead of the fused one */
int sc_hip_dilithium_set_tuning(sc_hip_dilithium_t *h, int slots_per_message, int unfused);

/* test hook, host only (no device call): the coefficient-domain shift tables sc_hip_dilithium_set_privkey builds for the finishing kernel without
 * transforms - us2 [ceil(k/3)][3n] | us1 [ceil(l/3)][3n] words of three 10-bit fields (row 3r + f in bits 10f.. of word row r, each entry U + *bias) |
 * ut0 [ceil(k/2)][3n] words of two signed 16-bit values; U[m] = -s[m] (m < n), s[m - n] (n <= m < 2n), -s[m - 2n] (2n <= m), s centred, *bias = max |s|
 * over s1 and s2; *fits = 0 when the key cannot use them (biased sums of `weight` entries outside 10 bits, t0 outside 16 bits, k or l > 6).
 * s1, s2, t as the private key holds them (dilithium.c:810-823); out may be null to ask for *fits only. */
int sc_hip_dilithium_fin_tables_host(const sc_hip_dilithium_params_t *params, const int32_t *s1, const int32_t *s2, const int32_t *t,
                                     uint32_t *out, size_t out_words, int *fits, uint32_t *bias);

/* test hook, host only (no device call): what sc_hip_cdt_create_from_table puts on the device for a table - guide_params: the ten words of the guide
 * (csrc/cdt_guide.h CdtGuide: lb_shift, tail_thr, K, MB, lin, total, S, sat_hi, sat_lo, jump; S = 0: the kernels fall back to the halving search), bins:
 * the guide's bin table (*bins_len entries), probe: the table the kernels search (size + 8 entries) */
int sc_hip_cdt_guide_host(const uint64_t *h_cdf, int size, uint32_t *guide_params, uint16_t *bins, size_t bins_cap, uint64_t *probe, size_t probe_cap, size_t *bins_len);

/* test hook: how many 32-bit words of the signer's per-call secrets (masks y, w = A y, NTT(y) where it is kept in HBM) are non-zero in the handle's
 * workspace - 0 behind every signing call (the reference zeroes sc->temp, dilithium.c:1964-1975) */
int sc_hip_dilithium_workspace_residue(sc_hip_dilithium_t *h, unsigned long long *nonzero_words);

/* the same for a Falcon handle: the sampler's working polynomials (6n doubles per signature) and the lattice point s1 | s2 before packing */
int sc_hip_falcon_workspace_residue(sc_hip_falcon_t *h, unsigned long long *nonzero_words);

/* test hook: arithmetic of the RLWE-Enc encrypt kernel - 0 automatic (the signed small-modulus mode where the plan qualifies, as for both of the
 * reference's parameter sets), 3 the unsigned multiply-add mode every q below 2^14 takes.  Results are identical; the tests run both. */
int sc_hip_rlwe_set_arith(sc_hip_rlwe_t *h, int mode);

/* process-wide: on = 0 stops the library from registering (pinning) caller memory in its host-buffer entries - every transfer then takes the
 * runtime's staged path; 1 (default) registers the whole pages inside the caller's buffers; 2 registers the caller's ranges as they are, edge pages
 * included (what round 2 did).  For bisecting transfer problems (profiles/r03_host_pin_bisect.txt). */
void sc_hip_bench_set_host_pin(int on);

/* BLISS-B signer: 1 (default) = after a message's first pair of attempts its later pairs are evaluated side by side (the tape position of an attempt does
 * not depend on any value, so pair p starts from a state the host computes; the lowest successful attempt is the signature); 0 = the reference's loop
 * round by round; 2 = as 1, but every third message that goes on to its later pairs is sent through the redo path (the one a failed oracle call
 * would take: state reset, the loop in order) - a self-test; 3 = as 1, but the first later pass (queued without a look at how many messages go on, for
 * batches of 1024 and more) is sized for ONE message, so that it is refused and redone with the count in hand - a self-test of that path.
 * Signatures, attempt counts and statuses do not depend on it. */
int sc_hip_bliss_set_speculation(sc_hip_bliss_t *h, int on);

/* process-wide: how the batched NTT kernels touch memory.  1 (default): operations whose operands exceed the 256 MiB Infinity Cache move their
 * coefficient rows 16 bytes per lane with the non-temporal policy (transposed through LDS), smaller ones 4 bytes per lane with the default policy -
 * the faster form on either side of that size (profiles/r04_ntt_stream_variants.txt); 0: never; 2: always.  Results are identical. */
void sc_hip_bench_set_ntt_stream(int mode);

#ifdef __cplusplus
}
#endif
#endif /* SC_HIP_BENCH_H */
