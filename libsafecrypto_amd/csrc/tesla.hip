// tesla.hip - batched ring-TESLA sign / verify on MI355X behind the C ABI of include/sc_hip.h.
//
// Replaces, for a batch of messages under one key, ring_tesla_sign / ring_tesla_verify
// (src/schemes/sig/ring_tesla/ring_tesla.c:752-910, 922-1040; raw signature coding) with test_w / test_rejection (:542-595) and
// the random oracle of src/schemes/sig/ring_tesla/random_oracle.c: poly_rounding + SHA3-512 (the reference's permutation,
// keccak_dev.h) over the big-endian words, the Salsa20-keyed stream (random_stream) and populate_c.
//
// Two behaviours of the reference are reproduced, not repaired: salsa20_core stores its result with SC_BIG_ENDIAN_32_COPY
// (src/utils/crypto/salsa/salsa20.c:93), so every 32-bit word of the stream is byte-swapped relative to Salsa20; and
// poly_rounding computes (p - modn(p)) >> d with modn over the ring modulus q, which is 0 for every centred or normalised
// coefficient.  The rounding is still evaluated here as written; when it is 0 for a whole message - always, with the reference's
// arithmetic - the hash resumes from the sponge state after the 56 whole all-zero blocks, kept per handle.
//
// Randomness: n prng_32 words of sc->prng_ctx[0] per attempt (ring_tesla.c:818-822); each message travels with its word tape
// as for BLISS (or the tape is expanded on the device with sc_hip_drbg_words).  Mapping: y / products on the batched NTT kernels
// (the public a1, a2 are NTT-domain constants of the reference, passed in as data), one lane per message for the hash and the
// Salsa20 stream, one wavefront per message for the sparse products, the three tests and the packing.
#include "engine_internal.h"
#include "keccak_dev.h"
#include "drbg_dev.h"
namespace sc { const DrbgConst &drbg_const(); }

#include <string.h>

using namespace sc;

namespace {

#include "tesla_sign_dev.h"

#include "tesla_verify_dev.h"

}  // namespace

struct sc_hip_tesla {
    sc_hip_ctx *ctx;
    sc_hip_plan *plan;
    TsP P;
    bool have_priv, have_pub;
    int32_t *d_a1, *d_a2, *d_t1, *d_t2;
    int32_t *d_ttab = nullptr;           // t1 | t2 as the verifier's shift tables (t | -t, 2n entries each)
    int16_t *d_key;                      // s | e1 | e2, each as a shift table of 2n entries (t | -t)
    uint64_t *d_zstate; uint32_t zblocks;
    size_t cap, slot_cap;                // messages / slots (attempts evaluated side by side) the workspace holds
    St st;
    uint32_t *act[2], *cnt, *rs, *need, *sbase, *nz, *trials, *acc;
    int32_t *y, *v1, *v2, *r1, *r2, *cidx, *zs;
    uint8_t *md;
    uint32_t *h_pin = nullptr;           // pinned host word the call's look at the device's count lands in (a copy to pageable memory blocks the host before the closing work is queued)
};

static void ts_free_ws(sc_hip_tesla *t)
{
    void *ptrs[] = { t->st.pos, t->st.status, t->act[0], t->act[1], t->cnt, t->rs, t->need, t->sbase, t->nz, t->trials, t->acc, t->y, t->v1, t->v2, t->r1, t->r2, t->cidx, t->zs, t->md };
    for (void *p : ptrs) if (p) (void)sc::wipe_free(p);
    t->st.pos = t->st.status = nullptr; t->act[0] = t->act[1] = t->cnt = t->rs = t->need = t->sbase = t->nz = t->trials = t->acc = nullptr;
    t->y = t->v1 = t->v2 = t->r1 = t->r2 = t->cidx = t->zs = nullptr; t->md = nullptr; t->cap = t->slot_cap = 0;
}

// per-message state for `batch` messages, per-attempt arrays for `slots` attempts (slots >= batch)
static int ts_ensure_ws(sc_hip_tesla *t, size_t batch, size_t slots)
{
    if (batch <= t->cap && slots <= t->slot_cap) return SC_HIP_SUCCESS;
    if (batch < t->cap) batch = t->cap;
    if (slots < t->slot_cap) slots = t->slot_cap;
    ts_free_ws(t);
    const size_t n = t->P.n;
#define WS(ptr, bytes) HIPCHK(hipMalloc((void **)&(ptr), (bytes)))
    WS(t->st.pos, batch * 4); WS(t->st.status, batch * 4); WS(t->act[0], batch * 4); WS(t->act[1], batch * 4); WS(t->cnt, 64); WS(t->rs, 64); WS(t->trials, batch * 4);
    WS(t->need, slots * 4); WS(t->sbase, slots * 4); WS(t->nz, slots * 4); WS(t->acc, slots * 4);
    WS(t->y, slots * n * 4); WS(t->v1, slots * n * 4); WS(t->v2, slots * n * 4); WS(t->r1, slots * n * 4); WS(t->r2, slots * n * 4); WS(t->zs, slots * n * 4);
    WS(t->cidx, slots * (size_t)t->P.omega * 4); WS(t->md, slots * 64);
#undef WS
    t->cap = batch; t->slot_cap = slots;
    return SC_HIP_SUCCESS;
}

extern "C" int sc_hip_tesla_create(sc_hip_ctx_t *ctx, const sc_hip_tesla_params_t *params, sc_hip_tesla_t **out)
{
    if (!ctx || !params || !out) return fail("tesla_create: null argument");
    static_assert(sizeof(TsP) == sizeof(sc_hip_tesla_params_t), "parameter struct layout");
    TsP P; memcpy(&P, params, sizeof P);
    if (P.n != 512 || (1 << P.n_bits) != P.n) return fail("tesla_create: n must be 512");
    if (P.omega > MAXW || (long long)P.omega * 32768 >= P.q / 2 || P.b >= P.q / 2) return fail("tesla_create: omega above 32, or omega * 2^15 or b not below q / 2 (the decide kernel's division-free normalisation)");
    if (P.omega < 1 || P.omega > 64 || P.d < 2 || P.d > 30 || P.q_bits < 8 || P.q_bits > 30 || P.b_bits < 1 || P.b_bits > 30 || P.q < 3) return fail("tesla_create: parameters out of range");
    sc_hip_plan *plan = nullptr;
    if (sc_hip_plan_create(ctx, P.q, P.n, &plan) != SC_HIP_SUCCESS) return SC_HIP_FAILURE;
    HIPCHK(hipSetDevice(ctx->device));
    sc_hip_tesla *t = new sc_hip_tesla;
    memset(t, 0, sizeof *t);
    t->ctx = ctx; t->plan = plan; t->P = P;
    t->zblocks = (uint32_t)(8 * P.n / 72);
    if (hipMalloc((void **)&t->d_zstate, 25 * 8) != hipSuccess) { sc_hip_tesla_destroy(t); return fail("tesla_create: device allocation failed"); }
    hipLaunchKernelGGL(ts_zero_prefix_kernel, dim3(1), dim3(HT), 0, nullptr, t->zblocks, t->d_zstate);
    if (hipDeviceSynchronize() != hipSuccess) { sc_hip_tesla_destroy(t); return fail("tesla_create: kernel failed"); }
    *out = t;
    return SC_HIP_SUCCESS;
}

extern "C" void sc_hip_tesla_destroy(sc_hip_tesla_t *t)
{
    if (!t) return;
    ts_free_ws(t);
    if (t->h_pin) (void)hipHostFree(t->h_pin);
    void *ptrs[] = { t->d_a1, t->d_a2, t->d_t1, t->d_t2, t->d_key, t->d_zstate, t->d_ttab };
    for (void *p : ptrs) if (p) (void)sc::wipe_free(p);
    sc_hip_plan_destroy(t->plan);
    delete t;
}

static int ts_upload32(int32_t **dst, const int32_t *src, size_t n)
{
    if (!*dst) HIPCHK(hipMalloc((void **)dst, n * 4));
    HIPCHK(sc::copy_in(*dst, src, n * 4));
    return SC_HIP_SUCCESS;
}

// a1, a2: the reference's public polynomials, NTT domain (a1_0 / a2_0, ring_tesla_params.h:85-88); s, e1, e2: the private key as
// stored (SINT16 [3][n], ring_tesla.c:703-730; NULL for a verify-only handle); t1, t2: the public key (SINT32 [2][n]; NULL for sign-only)
extern "C" int sc_hip_tesla_set_key(sc_hip_tesla_t *t, const int32_t *a1, const int32_t *a2, const int16_t *s, const int16_t *e1, const int16_t *e2,
                                    const int32_t *t1, const int32_t *t2)
{
    if (!t || !a1 || !a2) return fail("tesla_set_key: null argument");
    if ((s == nullptr) != (e1 == nullptr) || (s == nullptr) != (e2 == nullptr) || (t1 == nullptr) != (t2 == nullptr)) return fail("tesla_set_key: s, e1, e2 go together; t1, t2 go together");
    HIPCHK(hipSetDevice(t->ctx->device));
    const size_t n = t->P.n;
    if (ts_upload32(&t->d_a1, a1, n) || ts_upload32(&t->d_a2, a2, n)) return SC_HIP_FAILURE;
    if (s) {
        // the three small polynomials as the deciding kernel's shift tables (sparse_shift): T[m] = t[m], T[n + m] = -t[m]
        if (!t->d_key) HIPCHK(hipMalloc((void **)&t->d_key, 6 * n * 2));
        std::vector<int16_t> tab(6 * n);
        const int16_t *src[3] = { s, e1, e2 };
        for (int p = 0; p < 3; p++)
            for (size_t m = 0; m < n; m++) { tab[2 * n * p + m] = src[p][m]; tab[2 * n * p + n + m] = (int16_t)-src[p][m]; }
        HIPCHK(sc::copy_in(t->d_key, tab.data(), 6 * n * 2));
        for (auto &v : tab) v = 0;
        t->have_priv = true;
    }
    if (t1) {
        if (ts_upload32(&t->d_t1, t1, n) || ts_upload32(&t->d_t2, t2, n)) return SC_HIP_FAILURE;
        std::vector<int32_t> tt(4 * n);
        for (size_t m = 0; m < n; m++) { tt[m] = t1[m]; tt[n + m] = -t1[m]; tt[2 * n + m] = t2[m]; tt[3 * n + m] = -t2[m]; }
        if (ts_upload32(&t->d_ttab, tt.data(), 4 * n)) return SC_HIP_FAILURE;
        t->have_pub = true;
    }
    return SC_HIP_SUCCESS;
}

extern "C" size_t sc_hip_tesla_sig_bytes(const sc_hip_tesla_t *t)
{ return t ? ((size_t)t->P.n * t->P.q_bits + 7) / 8 + 64 : 0; }

static int ts_sign_core(sc_hip_tesla_t *t, size_t batch, const uint8_t *d_msgs, size_t msg_stride, const uint32_t *d_mlens, uint32_t fixed_len,
                        const uint32_t *d_tapes, size_t tape_stride, const uint32_t *d_tape_lens,
                        uint8_t *d_sigs, size_t sig_stride, uint32_t *d_siglens, uint32_t *d_trials, uint32_t *d_status, void *stream,
                        const uint8_t *d_seeds, size_t seed_stride, size_t seed_len)
{
    if (!t) return fail("tesla_sign: null handle");
    if (!t->have_priv) return fail("tesla_sign: no private key loaded");
    if (batch == 0) return SC_HIP_SUCCESS;
    if (batch > 0x7fffffffu / 2048) return fail("tesla_sign: batch too large");
    const bool seeded = d_seeds != nullptr;
    if (seeded) {
        if (seed_len == 0 || seed_len > seed_stride || seed_len > 0xffffffffu) return fail("tesla_sign_seeded: bad seed length / stride");
        d_tape_lens = nullptr; tape_stride = 0xfffffff0u;                 // a generator does not run out (the 32-bit word position does, after 2^23 attempts)
    }
    if (!d_msgs || (!d_tapes && !seeded) || !d_sigs || !d_siglens) return fail("tesla_sign: null buffer");
    if (sig_stride < sc_hip_tesla_sig_bytes(t)) return fail("tesla_sign: sig_stride smaller than sc_hip_tesla_sig_bytes()");
    if (tape_stride > 0xffffffffu) return fail("tesla_sign: tape too long");
    HIPCHK(hipSetDevice(t->ctx->device));
    // Attempts evaluated side by side per round: two per message while that is at most 16 K slots, one per message beyond.  Measured (tools/bench_seeded.py,
    // ms per call tape-resident / seeded, one against two attempts per message): batch 4096 0.59 / 0.67 against 0.41 / 0.50; 8192 0.62 / 0.76 against 0.59 / 0.72;
    // 16384 0.89 / 1.20 against 0.89 / 1.27; 65536 2.47 / 3.13 against 3.00 / 4.24 - a round's kernels are throughput-bound from ~16 K slots up, and every
    // speculative attempt costs its 2 KB of mask (in a seeded call: of cipher output) whether it is needed or not.
    size_t slot_target = batch * 2 < 16384 ? batch * 2 : (batch > 16384 ? batch : 16384);
    if (slot_target < 4096) slot_target = 4096;
    if (ts_ensure_ws(t, batch, slot_target)) return SC_HIP_FAILURE;
    hipStream_t s = (hipStream_t)stream;
    const TsP &P = t->P;
    const int n = P.n;
    St st = t->st;
    HIPCHK(hipMemsetAsync(st.pos, 0, batch * 4, s)); HIPCHK(hipMemsetAsync(st.status, 0, batch * 4, s));
    uint32_t *trials = d_trials ? d_trials : t->trials;
    HIPCHK(hipMemsetAsync(trials, 0, batch * 4, s));
    // Rounds are queued as the schedule predicts them (the slot rule run forward with the scheme's rejection probability, 0.68 per attempt, until fewer than four
    // messages are expected to be left: three rounds at 16384 messages), then the count is looked at ONCE and single rounds follow while messages are left.
    const uint32_t cap = (uint32_t)t->slot_cap, B = (uint32_t)batch;
    uint32_t todo = 0;
    { double left = (double)batch; while (left >= 4.0 && todo < 64) {     /* (a queued round that finds nobody left still costs two transforms over the slot budget) */ double S = cap / (left < 1.0 ? 1.0 : left); if (S > 32) S = 32; if (S < 1) S = 1; left *= pow(0.68, floor(S)); todo++; } }
    if (!todo) todo = 1;
    uint32_t round = 0, left = B;
    bool planned = true;                                                  // the queued rounds take their sizes from the device; a round behind the look has the host's
    while (left > 0) {
        for (uint32_t i = 0; i < todo; i++, round++) {
            if (round >= 9999) { left = 0; break; }                       // the reference's own cap (ring_tesla.c:814); leftovers keep status 0
            const uint32_t *active = round ? t->act[(round - 1) & 1] : nullptr;
            uint32_t *next = t->act[round & 1];
            const uint32_t *rs = planned ? t->rs : nullptr;
            uint32_t nact = B, S = 1, nslots = cap;                       // planned: upper bounds (they size the grids)
            if (planned) hipLaunchKernelGGL(ts_rs_kernel, dim3(1), dim3(1), 0, s, t->rs, t->cnt, round ? 0u : B, cap);
            else {
                nact = left; S = cap / nact; if (S > 32) S = 32; if (S < 1) S = 1; nslots = nact * S;
                HIPCHK(hipMemsetAsync(t->cnt, 0, 4, s));
            }
            const size_t ne = (size_t)nslots * n;
            size_t yb = (ne / 4 + 255) / 256;                             // a lane per four coefficients; persistent beyond 16 workgroups per CU
            if (yb > (size_t)t->ctx->cus * 16) yb = (size_t)t->ctx->cus * 16;
            if (seeded) hipLaunchKernelGGL(ts_y_seeded_kernel, dim3((unsigned)(((size_t)nslots * 32 + 255) / 256)), dim3(256), 0, s, active, nslots, S, t->need, t->sbase,
                                           d_seeds, seed_stride, (uint32_t)seed_len, t->y, P, rs, sc::drbg_const(), st);
            else hipLaunchKernelGGL(ts_y_kernel, dim3((unsigned)yb), dim3(256), 0, s, active, nslots, S, t->need, t->sbase, d_tapes, tape_stride, t->y, P, rs, st, d_tape_lens);
            HIPCHK(hipGetLastError());
            // (a planned round's products run over the whole slot budget: rows past the round's slots hold what an earlier round left there and are read by nobody)
            if (sc::polymul_ntt_bcast2(t->plan, nslots, t->y, t->d_a1, t->d_a2, t->v1, t->v2, s)) return SC_HIP_FAILURE;      // v1 = a1 y, v2 = a2 y: one forward transform of y for both
            hipLaunchKernelGGL(ts_round_kernel, dim3((nslots + 3) / 4), dim3(256), 0, s, nslots, t->v1, t->v2, t->r1, t->r2, t->nz, 1, P, rs);
            hipLaunchKernelGGL(ts_oracle_kernel, dim3((nslots + HT - 1) / HT), dim3(HT), 0, s, active, nslots, S, t->need, d_msgs, msg_stride, d_mlens, fixed_len, t->r1, t->r2, t->nz,
                               t->d_zstate, t->zblocks, t->md, 0, (uint32_t *)nullptr, (const uint8_t *)nullptr, (size_t)0, 0u, P, rs);
            hipLaunchKernelGGL(ts_f_kernel, dim3((nslots + HT - 1) / HT), dim3(HT), 0, s, (const uint32_t *)nullptr, nslots, t->need, t->md, (size_t)64, 0u, 0, t->cidx, P, rs);
            hipLaunchKernelGGL(ts_decide_kernel, dim3((nslots + 3) / 4), dim3(256), sizeof(int16_t) * 6 * (size_t)n, s, nslots, t->need, t->d_key, t->v1, t->v2, t->y, t->cidx, t->zs, t->acc, P, rs);
            hipLaunchKernelGGL(ts_select_kernel, dim3((nact + 4 * SEL_PER - 1) / (4 * SEL_PER)), dim3(256), 0, s, active, nact, S, t->need, t->acc, st, t->zs, t->md, d_sigs, sig_stride, d_siglens, trials, next, t->cnt, P, rs);
            HIPCHK(hipGetLastError());
        }
        if (!left) {                                                          // the reference's round cap: no look follows, the closing work does
            if (d_status) HIPCHK(hipMemcpyAsync(d_status, st.status, batch * 4, hipMemcpyDeviceToDevice, s));
            const size_t used = t->slot_cap * (size_t)n * 4;
            ZeroSix z6{ { t->y, t->v1, t->v2, t->r1, t->r2, t->zs }, used / 16 };
            size_t zb = (used / 16 + 255) / 256;
            if (zb > (size_t)t->ctx->cus * 8) zb = (size_t)t->ctx->cus * 8;
            hipLaunchKernelGGL(ts_zero6_kernel, dim3((unsigned)(zb ? zb : 1)), dim3(256), 0, s, z6);
            HIPCHK(hipGetLastError());
            break;
        }
        // The look at the count - and in front of it, not behind it, the call's closing work: the call usually ends here, so the zeroing of the masks y, their
        // products and the rejected candidates (one launch over the six arrays; every round's slots are cut from the handle's slot budget, so that is how far they
        // were written) and the copy of the statuses run while the host waits for the count instead of after it has woken up.  A round that follows rewrites
        // what it reads and is zeroed again.
        if (!t->h_pin) HIPCHK(hipHostMalloc((void **)&t->h_pin, 64, hipHostMallocDefault));
        HIPCHK(hipMemcpyAsync(t->h_pin, t->cnt, 4, hipMemcpyDeviceToHost, s));
        if (d_status) HIPCHK(hipMemcpyAsync(d_status, st.status, batch * 4, hipMemcpyDeviceToDevice, s));
        {
            const size_t used = t->slot_cap * (size_t)n * 4;
            ZeroSix z6{ { t->y, t->v1, t->v2, t->r1, t->r2, t->zs }, used / 16 };
            size_t zb = (used / 16 + 255) / 256;
            if (zb > (size_t)t->ctx->cus * 8) zb = (size_t)t->ctx->cus * 8;
            hipLaunchKernelGGL(ts_zero6_kernel, dim3((unsigned)(zb ? zb : 1)), dim3(256), 0, s, z6);
            HIPCHK(hipGetLastError());
        }
        HIPCHK(hipStreamSynchronize(s));
        left = *(volatile uint32_t *)t->h_pin;
        todo = 1; planned = false;
    }
    return SC_HIP_SUCCESS;
}

extern "C" int sc_hip_tesla_sign(sc_hip_tesla_t *t, size_t batch, const uint8_t *d_msgs, size_t msg_stride, const uint32_t *d_mlens, uint32_t fixed_len,
                                 const uint32_t *d_tapes, size_t tape_stride, const uint32_t *d_tape_lens,
                                 uint8_t *d_sigs, size_t sig_stride, uint32_t *d_siglens, uint32_t *d_trials, uint32_t *d_status, void *stream)
{
    if (!d_tapes && batch) return fail("tesla_sign: null buffer");
    return ts_sign_core(t, batch, d_msgs, msg_stride, d_mlens, fixed_len, d_tapes, tape_stride, d_tape_lens, d_sigs, sig_stride, d_siglens, d_trials, d_status, stream, nullptr, 0, 0);
}

// sc_hip_tesla_sign with the CSPRNG inside the call: message i draws its masks from the generator the reference makes of seed i (AES-256 CTR-DRBG,
// src/utils/crypto/prng.c:963-1107, ctr_drbg.c:37-197; ring_tesla.c:818-822 reads it through prng_32).  The words of an attempt are produced by the kernel
// that turns them into the mask (ts_y_seeded_kernel): nothing is expanded ahead and no tape exists.  Byte-identical to sc_hip_tesla_sign over
// sc_hip_drbg_words(seeds, ...) rows that are long enough.
extern "C" int sc_hip_tesla_sign_seeded(sc_hip_tesla_t *t, size_t batch, const uint8_t *d_msgs, size_t msg_stride, const uint32_t *d_mlens, uint32_t fixed_len,
                                        const uint8_t *d_seeds, size_t seed_stride, size_t seed_len,
                                        uint8_t *d_sigs, size_t sig_stride, uint32_t *d_siglens, uint32_t *d_trials, uint32_t *d_status, void *stream)
{
    if (!d_seeds && batch) return fail("tesla_sign_seeded: null seed buffer");
    return ts_sign_core(t, batch, d_msgs, msg_stride, d_mlens, fixed_len, nullptr, 0, nullptr, d_sigs, sig_stride, d_siglens, d_trials, d_status, stream, d_seeds, seed_stride, seed_len);
}

extern "C" int sc_hip_tesla_verify(sc_hip_tesla_t *t, size_t batch, const uint8_t *d_msgs, size_t msg_stride, const uint32_t *d_mlens, uint32_t fixed_len,
                                   const uint8_t *d_sigs, size_t sig_stride, const uint32_t *d_siglens, uint32_t *d_verdicts, void *stream)
{
    if (!t) return fail("tesla_verify: null handle");
    if (!t->have_pub) return fail("tesla_verify: no public key loaded");
    if (batch == 0) return SC_HIP_SUCCESS;
    if (batch > 0x7fffffffu / 2048) return fail("tesla_verify: batch too large");
    if (!d_msgs || !d_sigs || !d_siglens || !d_verdicts) return fail("tesla_verify: null buffer");
    if (sig_stride < sc_hip_tesla_sig_bytes(t)) return fail("tesla_verify: sig_stride smaller than sc_hip_tesla_sig_bytes()");
    HIPCHK(hipSetDevice(t->ctx->device));
    if (ts_ensure_ws(t, batch, batch)) return SC_HIP_FAILURE;
    hipStream_t s = (hipStream_t)stream;
    const TsP &P = t->P;
    const uint32_t B = (uint32_t)batch;
    const uint32_t md_off = ((uint32_t)P.n * P.q_bits + 7) / 8;
    HIPCHK(hipMemsetAsync(d_verdicts, 0, batch * 4, s));
    hipLaunchKernelGGL(ts_unpack_f_kernel, dim3((B + HT - 1) / HT + (B + 3) / 4), dim3(256), 0, s, B, d_sigs, sig_stride, d_siglens, t->y, d_verdicts, P, md_off, t->cidx);
    HIPCHK(hipGetLastError());
    if (sc::polymul_ntt_bcast2(t->plan, batch, t->y, t->d_a1, t->d_a2, t->v1, t->v2, s)) return SC_HIP_FAILURE;
    {
        hipLaunchKernelGGL(ts_vw_kernel, dim3((B + 3) / 4), dim3(256), sizeof(int32_t) * 4 * (size_t)P.n, s, B, t->v1, t->v2, t->d_ttab, t->cidx, P, t->plan->k[1]);
    }
    hipLaunchKernelGGL(ts_round_kernel, dim3((B + 3) / 4), dim3(256), 0, s, B, t->v1, t->v2, t->r1, t->r2, t->nz, 0, P);
    hipLaunchKernelGGL(ts_oracle_kernel, dim3((B + HT - 1) / HT), dim3(HT), 0, s, (const uint32_t *)nullptr, B, 1u, (const uint32_t *)nullptr, d_msgs, msg_stride, d_mlens, fixed_len, t->r1, t->r2, t->nz,
                       t->d_zstate, t->zblocks, (uint8_t *)nullptr, 1, d_verdicts, d_sigs, sig_stride, md_off, P);
    HIPCHK(hipGetLastError());
    return SC_HIP_SUCCESS;
}

// ---- host-buffer convenience (sc::sign_host / sc::verify_host in engine_internal.h) ----

extern "C" int sc_hip_tesla_sign_host(sc_hip_tesla_t *t, size_t batch, const uint8_t *h_msgs, size_t msg_stride, const uint32_t *h_mlens, uint32_t fixed_len,
                                      const uint32_t *h_tapes, size_t tape_stride, const uint32_t *h_tape_lens,
                                      uint8_t *h_sigs, size_t sig_stride, uint32_t *h_siglens, uint32_t *h_trials, uint32_t *h_status)
{
    if (!t) return fail("tesla_sign_host: null handle");
    return sc::sign_host(t->ctx->device, batch, h_msgs, msg_stride, h_mlens, h_tapes, tape_stride, h_tape_lens, h_sigs, sig_stride, h_siglens, h_trials, h_status,
        [&](const uint8_t *m, const uint32_t *ml, const uint32_t *tp, const uint32_t *tl, uint8_t *sg, uint32_t *sl, uint32_t *tr, uint32_t *st) {
            return sc_hip_tesla_sign(t, batch, m, msg_stride, ml, fixed_len, tp, tape_stride, tl, sg, sig_stride, sl, tr, st, nullptr);
        });
}

extern "C" int sc_hip_tesla_verify_host(sc_hip_tesla_t *t, size_t batch, const uint8_t *h_msgs, size_t msg_stride, const uint32_t *h_mlens, uint32_t fixed_len,
                                        const uint8_t *h_sigs, size_t sig_stride, const uint32_t *h_siglens, uint32_t *h_verdicts)
{
    if (!t) return fail("tesla_verify_host: null handle");
    return sc::verify_host(t->ctx->device, batch, h_msgs, msg_stride, h_mlens, h_sigs, sig_stride, h_siglens, h_verdicts,
        [&](const uint8_t *m, const uint32_t *ml, const uint8_t *sg, const uint32_t *sl, uint32_t *v) {
            return sc_hip_tesla_verify(t, batch, m, msg_stride, ml, fixed_len, sg, sig_stride, sl, v, nullptr);
        });
}
