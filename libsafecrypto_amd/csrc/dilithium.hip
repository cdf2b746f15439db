// dilithium.hip - batched Dilithium sign / verify on MI355X behind the C ABI of include/sc_hip.h.
//
// Replaces, for a batch of messages under one key, dilithium_sign / dilithium_verify
// (src/schemes/sig/dilithium/dilithium.c:1397-1958, 2002-2240; build switches of dilithium_params.h:35-56:
// DILITHIUM_USE_CSPRNG_SAM, DILITHIUM_USE_H_FUNC_XOF (SHAKE128), DILITHIUM_USE_SPARSE_MULTIPLIER,
// USE_DETERMINISTIC_DILITHIUM) and the helpers they call in src/utils/arith/module_lwe.c
// (expand_mask :201-244, create_rand_product_32_csprng :588-667 with the matrix already expanded,
// high/low_order_bits :1075-1113, kyber_oracle_core :1115-1141, round_alpha module_lwe.h:110-136,
// make_hint / use_hint / check_norm_inf dilithium.c:1185-1312).
//
// Everything of an attempt runs on the device: SHAKE (the reference's permutation, keccak_dev.h), mask
// expansion, A*y through the NTT machinery of ntt_core.h, rounding, challenge, the three c*key products
// (as NTT-domain products - equal mod q to the reference's sparse schoolbook multiplier), norm checks,
// hints and the MSB-first bit packing of the signature (src/utils/entropy/packer.c:299-351).  The
// rejection loop is batched and speculative: deterministic signing makes an attempt a pure function of
// (key, message, kappa), so each round evaluates S consecutive kappa values of every still-active message
// side by side ("slots") and accepts the LOWEST successful kappa - exactly the attempt the reference's
// sequential loop stops at, hence byte-identical signatures - while the round count (the latency of the
// one-lane-per-hash SHAKE kernels) drops from ~30 to a handful; S grows as messages finish.
#include "engine_internal.h"
#include "keccak_dev.h"
#include "mlwe_kernels.h"

#include <stdlib.h>
#include <type_traits>
#include <string.h>

using namespace sc;
using namespace sc_mlwe;

namespace {

#include "dil_sign_dev.h"

// ------------------------------------------------------------------------------------------------
// one-thread-per-message kernels (hashing, challenge, decisions, packing)
// ------------------------------------------------------------------------------------------------

// ---- key generation (dilithium.c:752-935) -----------------------------------------------------------------------------------

// rho and K of every key out of its CSPRNG stream: the first 32 bytes of the first two 64-byte prng_mem blocks
__global__ void dil_kg_slices_kernel(const uint8_t *stream, size_t stride, uint32_t count, uint8_t *rho, uint8_t *K)
{
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t item = (uint32_t)(t / 64); const int b = (int)(t % 64);
    if (item >= count) return;
    const uint8_t *s = stream + (size_t)item * stride;
    if (b < 32) rho[(size_t)item * 32 + b] = s[b]; else K[(size_t)item * 32 + (b - 32)] = s[64 + (b - 32)];
}

// uniform_rand_sample_small_csprng (module_lwe.c:280-310): sample o of a vector is half (o & 1) of byte o >> 1 of its blocks:
// (b >> 4*half) & mask, halved when > 2 eta, stored as eta - value
__global__ void dil_kg_small_kernel(const uint8_t *stream, size_t stride, size_t off, uint32_t count, int per, int eta, uint32_t mask, int32_t *out)
{
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t item = (uint32_t)(t / per); const int o = (int)(t % per);
    if (item >= count) return;
    uint32_t v = ((uint32_t)stream[(size_t)item * stride + off + (o >> 1)] >> (4 * (o & 1))) & mask;
    if (v > 2u * (uint32_t)eta) v >>= 1;
    out[t] = eta - (int32_t)v;
}

// t = normalise(A s1 + s2) ; t1 = its high part (decompose, module_lwe.c:73-104)
__global__ void dil_kg_t_kernel(const int32_t *as1, const int32_t *s2, int32_t *t, int32_t *t1, size_t len, int32_t q, int d)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= len) return;
    int32_t v = (as1[i] + s2[i]) % q;
    if (v < 0) v += q;
    t[i] = v;
    int32_t x = v & ((1 << d) - 1);
    x -= (x > (1 << (d - 1))) * (1 << d);
    t1[i] = (v - x) >> d;
}


#include "dil_verify_dev.h"

}  // namespace

// ------------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------------

struct sc_hip_dilithium {
    sc_hip_ctx *ctx;
    sc_hip_plan *plan;
    DilP P;
    bool have_priv, have_pub;
    uint32_t *d_A, *d_s1, *d_s2, *d_t0, *d_t1s;      // internal order, Montgomery form
    uint32_t *d_As;                                   // the matrix again in plain form | its Shoup companions (dil_yaw_kernel), l <= 5 only
    uint32_t *d_tabs; uint32_t fin_bias;              // shift tables of s2, s1, t0 in the coefficient domain (dil_cfins_kernel) and their fields' bias; null: the key does not fit them, or tuning bit 11 never asked for them
    bool tabs_tried;                                  // the loaded key went through the table builder (bit 11 was set at key load): d_tabs is either there or the key does not fit.  No host copy of the tables outlives set_privkey
    uint8_t *d_K, *d_tr_priv, *d_tr_pub, *d_rho;
    // workspace
    size_t cap;                                       // per message / per signature (verify): mu, kappa, trials, act, z, h, c, ch
    uint8_t *mu, *w1, *h;
    size_t slot_cap;                                  // per attempt of a signing round: y, w, w1, c8, yh, slot_item
    uint32_t *kappa, *trials, *act[2], *cnt, *rs, *slot_item;
    int8_t *c8;
    int32_t *y, *w, *c, *z;
    uint32_t *yh, *ch;
    uint8_t *msg_dev, *sig_dev;
    size_t msg_cap, sig_cap;
    sc::Workspace kws;                                // sc_hip_dilithium_verify_keys: per-item key material
    hipStream_t side, side2;                          // verify_keys / sign_keys: per-key chains that depend on nothing of each other run beside each other
    hipEvent_t ev_fork, ev_join, ev_join2;
    int tune_slots, tune_unfused, tune_sync, tune_exp1 = 0, tune_w24 = 0, tune_sparse_fin = 0;        // measurement hooks (sc_hip_dilithium_set_tuning): slots per message (0 = default), unfused NTT(y) / matrix product, a read-back after every round
};

static void free_ws(sc_hip_dilithium *d)
{
    void *ptrs[] = { d->mu, d->w1, d->h, d->kappa, d->trials, d->act[0], d->act[1], d->cnt, d->rs, d->slot_item, d->c8, d->y, d->w, d->c, d->z, d->yh, d->ch };
    for (void *p : ptrs) if (p) (void)sc::wipe_free(p);
    d->mu = d->w1 = d->h = nullptr; d->kappa = d->trials = d->act[0] = d->act[1] = d->cnt = d->rs = d->slot_item = nullptr;
    d->c8 = nullptr; d->y = d->w = d->c = d->z = nullptr; d->yh = d->ch = nullptr;
    d->cap = 0; d->slot_cap = 0;
}

// per-message state for `batch` messages, per-attempt arrays for `slots` attempts (slots >= batch)
static int ensure_ws(sc_hip_dilithium *d, size_t batch, size_t slots)
{
    if (batch <= d->cap && slots <= d->slot_cap) return SC_HIP_SUCCESS;
    if (batch < d->cap) batch = d->cap;
    if (slots < d->slot_cap) slots = d->slot_cap;
    free_ws(d);
    const size_t n = d->P.n, k = d->P.k, l = d->P.l;
#define WS(ptr, bytes) HIPCHK(hipMalloc((void **)&(ptr), (bytes)))
    WS(d->mu, batch * 48); WS(d->kappa, batch * 4); WS(d->trials, batch * 4); WS(d->act[0], batch * 4); WS(d->act[1], batch * 4); WS(d->cnt, 64); WS(d->rs, 64);
    WS(d->h, batch * k * n); WS(d->c, batch * n * 4); WS(d->z, batch * l * n * 4); WS(d->ch, batch * n * 4);          // the verifier's: one per signature
    WS(d->slot_item, slots * 4); WS(d->c8, slots * n);
    WS(d->w1, slots * k * n);
    WS(d->y, slots * l * n * 4); WS(d->w, slots * k * n * 4); WS(d->yh, slots * l * n * 4);
#undef WS
    d->cap = batch; d->slot_cap = slots;
    return SC_HIP_SUCCESS;
}

static DilNtt base_args(sc_hip_dilithium *d)
{
    DilNtt a = mlwe_base_args(d->plan, d->P.k, d->P.l, d->P.gamma2);
    a.gamma1_beta = d->P.gamma1 - d->P.beta; a.gamma2_beta = d->P.gamma2 - d->P.beta; a.d = d->P.d;
    a.yh = d->yh; a.ch = d->ch; a.y = d->y; a.w = d->w; a.w1 = d->w1; a.z = d->z; a.h = d->h;
    return a;
}

template <int OP>
static int launch_dil(sc_hip_dilithium *d, DilNtt a, uint32_t nitems, int per, hipStream_t s) { return mlwe_launch<OP>(d->ctx, a, nitems, per, s); }

extern "C" int sc_hip_dilithium_create(sc_hip_ctx_t *ctx, const sc_hip_dilithium_params_t *params, sc_hip_dilithium_t **out)
{
    if (!ctx || !params || !out) return fail("dilithium_create: null argument");
    DilP P;
    static_assert(sizeof(DilP) == sizeof(sc_hip_dilithium_params_t), "parameter struct layout");
    memcpy(&P, params, sizeof P);
    if (P.n != 256) return fail("dilithium_create: n must be 256");
    if (P.k < 1 || P.k > 8 || P.l < 1 || P.l > 8) return fail("dilithium_create: k, l out of range");
    if (P.weight < 1 || P.weight > 64 || (P.l * P.n * P.z_bits) % 8 || (2 * P.z_bits) % 8 || P.z_bits > 24) return fail("dilithium_create: unsupported packing parameters");
    if (((P.q - 1) / (2 * P.gamma2)) & (((P.q - 1) / (2 * P.gamma2)) - 1)) return fail("dilithium_create: (q-1)/(2 gamma_2) must be a power of two");
    sc_hip_plan *plan = nullptr;
    if (sc_hip_plan_create(ctx, P.q, P.n, &plan) != SC_HIP_SUCCESS) return SC_HIP_FAILURE;
    sc_hip_dilithium *d = new sc_hip_dilithium;
    memset(d, 0, sizeof *d);
    d->ctx = ctx; d->plan = plan; d->P = P;
    *out = d;
    return SC_HIP_SUCCESS;
}

extern "C" void sc_hip_dilithium_destroy(sc_hip_dilithium_t *d)
{
    if (!d) return;
    free_ws(d);
    d->kws.release();
    if (d->side) (void)hipStreamDestroy(d->side);
    if (d->side2) (void)hipStreamDestroy(d->side2);
    if (d->ev_fork) (void)hipEventDestroy(d->ev_fork);
    if (d->ev_join) (void)hipEventDestroy(d->ev_join);
    if (d->ev_join2) (void)hipEventDestroy(d->ev_join2);
    void *ptrs[] = { d->d_A, d->d_As, d->d_tabs, d->d_s1, d->d_s2, d->d_t0, d->d_t1s, d->d_K, d->d_tr_priv, d->d_tr_pub, d->d_rho, d->msg_dev, d->sig_dev };
    for (void *p : ptrs) if (p) (void)sc::wipe_free(p);
    sc_hip_plan_destroy(d->plan);
    delete d;
}

// upload natural-order coefficient-domain polynomials and leave their NTT (internal order, Montgomery form) in *dst
static int key_ntt(sc_hip_dilithium *d, const int32_t *h_polys, size_t npoly, uint32_t **dst)
{
    const size_t bytes = npoly * N * 4;
    DevBuf tb;
    if (tb.get(bytes)) return SC_HIP_FAILURE;
    int32_t *tmp = tb.as<int32_t>();
    if (!*dst) HIPCHK(hipMalloc((void **)dst, bytes));
    HIPCHK(sc::copy_in(tmp, h_polys, bytes));
    DilNtt a = base_args(d);
    a.in = tmp; a.out_int = *dst;
    int rc = launch_dil<DOP_KEYPREP>(d, a, (uint32_t)npoly, 1, nullptr);
    if (rc == SC_HIP_SUCCESS && hipDeviceSynchronize() != hipSuccess) rc = fail("dilithium key load: kernel failed");
    return rc;
}

static int load_matrix(sc_hip_dilithium *d, const int32_t *h_Ahat)
{
    const size_t np = (size_t)d->P.k * d->P.l, bytes = np * N * 4;
    DevBuf tb;
    if (tb.get(bytes)) return SC_HIP_FAILURE;
    int32_t *tmp = tb.as<int32_t>();
    if (!d->d_A) HIPCHK(hipMalloc((void **)&d->d_A, bytes));
    HIPCHK(sc::copy_in(tmp, h_Ahat, bytes));
    hipLaunchKernelGGL(dil_to_internal_mont, dim3((unsigned)((np * N + 255) / 256)), dim3(256), 0, nullptr, tmp, d->d_A, np, d->plan->k[1], d->plan->r2, d->plan->r2p);
    if (d->P.l >= 2 && d->P.l <= 5) {
        if (!d->d_As) HIPCHK(hipMalloc((void **)&d->d_As, 2 * bytes));
        hipLaunchKernelGGL(dil_to_internal_shoup, dim3((unsigned)((np * N + 255) / 256)), dim3(256), 0, nullptr, tmp, d->d_As, d->d_As + np * N, np, d->plan->k[1]);
    }
    return (hipDeviceSynchronize() == hipSuccess) ? SC_HIP_SUCCESS : fail("dilithium matrix load: kernel failed");
}

// dil_cfins_kernel's shift tables: us2 [ceil(k/3)][3n] | us1 [ceil(l/3)][3n], three rows a word as 10-bit fields holding U + bias, and ut0 [ceil(k/2)][3n], two
// rows a word as signed 16-bit values; U[m] = -s[m] (m < n), s[m - n] (n <= m < 2n), -s[m - 2n] (2n <= m), s centred; bias = max |s| over s1 and s2.
// Returns false for a key whose biased sums of `weight` entries could leave 10 bits or whose t0 does not fit 16 bits - none the reference generates;
// such a key keeps the transform kernel.  t0 canonical in [0, q) or centred.
static bool fin_tables(const DilP &P, const int32_t *s1, const int32_t *s2, const int32_t *t0, std::vector<uint32_t> &tabs, uint32_t &bias)
{
    const int k = P.k, l = P.l, q = P.q;
    const int kw = (k + 2) / 3, lw = (l + 2) / 3, kp = (k + 1) / 2;
    tabs.assign(cfins_table_words(P), 0u);
    bool fits = k <= 3 * CSW && l <= 3 * CSW && k <= 2 * CSP && P.weight >= 1 && P.weight <= 64;
    auto centred = [&](int32_t x) { int32_t r = x % q; if (r < 0) r += q; return r > (q - 1) / 2 ? r - q : r; };
    int32_t b = 0;
    for (size_t i = 0; i < (size_t)l * N; i++) { const int32_t v = centred(s1[i]); b = std::max(b, v < 0 ? -v : v); }
    for (size_t i = 0; i < (size_t)k * N; i++) { const int32_t v = centred(s2[i]); b = std::max(b, v < 0 ? -v : v); }
    if ((int64_t)2 * b * (P.weight > 0 ? P.weight : 1) > 1023) { fits = false; b = 0; }
    bias = (uint32_t)b;
    auto fill10 = [&](uint32_t *T, const int32_t *src, int rows) {
        for (int i = 0; i < rows; i++)
            for (int j = 0; j < N; j++) {
                const int32_t v = fits ? centred(src[(size_t)i * N + j]) : 0;
                uint32_t *row = T + (size_t)(i / 3) * 3 * N;
                const int sh = 10 * (i % 3);
                row[j] |= (uint32_t)(b - v) << sh; row[N + j] |= (uint32_t)(b + v) << sh; row[2 * N + j] |= (uint32_t)(b - v) << sh;
            }
    };
    fill10(tabs.data(), s2, k);
    fill10(tabs.data() + (size_t)kw * 3 * N, s1, l);
    uint32_t *T0 = tabs.data() + (size_t)(kw + lw) * 3 * N;
    for (int i = 0; i < k; i++)
        for (int j = 0; j < N; j++) {
            const int32_t v = centred(t0[(size_t)i * N + j]);
            if (v > 32767 || v < -32767) fits = false;
            const uint32_t neg = (uint32_t)(uint16_t)(int16_t)(-v), pos = (uint32_t)(uint16_t)(int16_t)v;
            uint32_t *row = T0 + (size_t)(i >> 1) * 3 * N;
            const int sh = 16 * (i & 1);
            row[j] |= neg << sh; row[N + j] |= pos << sh; row[2 * N + j] |= neg << sh;
        }
    (void)kp;
    return fits;
}

// test hook (include/sc_hip_bench.h): the tables of a key, built on the host exactly as sc_hip_dilithium_set_privkey builds them - no device involved
extern "C" int sc_hip_dilithium_fin_tables_host(const sc_hip_dilithium_params_t *params, const int32_t *s1, const int32_t *s2, const int32_t *t,
                                                uint32_t *out, size_t out_words, int *fits, uint32_t *bias)
{
    if (!params || !s1 || !s2 || !t || !fits || !bias) return fail("dilithium_fin_tables_host: null argument");
    DilP P;
    memcpy(&P, params, sizeof P);
    if (P.n != N || P.k < 1 || P.k > 8 || P.l < 1 || P.l > 8 || P.q < 2 || P.d < 1 || P.d > 30) return fail("dilithium_fin_tables_host: unsupported parameters");
    std::vector<int32_t> t0((size_t)P.k * N);
    const int32_t mask = (1 << P.d) - 1, sign = 1 << P.d, thresh = 1 << (P.d - 1);
    for (size_t i = 0; i < t0.size(); i++) { int32_t x = t[i] & mask; x -= (x > thresh) * sign; t0[i] = x; }
    std::vector<uint32_t> tabs;
    *fits = fin_tables(P, s1, s2, t0.data(), tabs, *bias) ? 1 : 0;
    if (out) {
        if (out_words < tabs.size()) return fail("dilithium_fin_tables_host: output too small");
        memcpy(out, tabs.data(), tabs.size() * 4);
    }
    return SC_HIP_SUCCESS;
}

// The opt-in finishing kernel's tables (tuning bit 11) for the key being loaded: built on the host, uploaded, and the host copy zeroed before returning - the handle
// keeps no host-side secret.  A failed upload leaves no half-filled device table behind (d_tabs stays null: sign_core then takes the transform kernel).
static int build_fin_tables(sc_hip_dilithium *d, const int32_t *s1, const int32_t *s2, const int32_t *t0)
{
    if (d->d_tabs) { (void)sc::wipe_free(d->d_tabs); d->d_tabs = nullptr; }
    d->tabs_tried = false;
    std::vector<uint32_t> tabs;
    auto wipe = [&] { for (auto &w : tabs) *(volatile uint32_t *)&w = 0; };
    const bool fits = fin_tables(d->P, s1, s2, t0, tabs, d->fin_bias);
    int rc = SC_HIP_SUCCESS;
    if (fits) {
        if (hipMalloc((void **)&d->d_tabs, tabs.size() * 4) != hipSuccess) { d->d_tabs = nullptr; rc = fail("dilithium_set_privkey: hipMalloc of the finishing tables failed"); }
        else if (sc::copy_in(d->d_tabs, tabs.data(), tabs.size() * 4)) { (void)sc::wipe_free(d->d_tabs); d->d_tabs = nullptr; rc = SC_HIP_FAILURE; }
    }
    wipe();
    d->tabs_tried = (rc == SC_HIP_SUCCESS);
    return rc;
}

extern "C" int sc_hip_dilithium_set_privkey(sc_hip_dilithium_t *d, const int32_t *s1, const int32_t *s2, const int32_t *t,
                                            const uint8_t *K, const uint8_t *tr, const int32_t *Ahat)
{
    if (!d || !s1 || !s2 || !t || !K || !tr || !Ahat) return fail("dilithium_set_privkey: null argument");
    HIPCHK(hipSetDevice(d->ctx->device));
    const int k = d->P.k, l = d->P.l, q = d->P.q, dd = d->P.d;
    // decompose(t) -> t0 in [0,q) (module_lwe.c:89-104); only t0 is needed by the signer
    std::vector<int32_t> t0((size_t)k * N);
    struct Wipe { std::vector<int32_t> &v; ~Wipe() { for (auto &w : v) *(volatile int32_t *)&w = 0; } } wipe_t0{ t0 };      // t0 is key material: the host copy does not outlive the call, on any path out
    const int32_t mask = (1 << dd) - 1, sign = 1 << dd, thresh = 1 << (dd - 1);
    for (size_t i = 0; i < t0.size(); i++) { int32_t x = t[i] & mask; x -= (x > thresh) * sign; t0[i] = (x >= 0) ? x : x + q; }
    if (key_ntt(d, s1, (size_t)l, &d->d_s1) || key_ntt(d, s2, (size_t)k, &d->d_s2) || key_ntt(d, t0.data(), (size_t)k, &d->d_t0)) return SC_HIP_FAILURE;
    if (load_matrix(d, Ahat)) return SC_HIP_FAILURE;
    // the opt-in finishing kernel's tables: only when tuning bit 11 is ALREADY set (the default path does no work for them, on the host or the device)
    if (d->d_tabs) { (void)sc::wipe_free(d->d_tabs); d->d_tabs = nullptr; }
    d->tabs_tried = false;
    if (d->tune_sparse_fin && build_fin_tables(d, s1, s2, t0.data())) return SC_HIP_FAILURE;
    if (!d->d_K) HIPCHK(hipMalloc((void **)&d->d_K, 32));
    if (!d->d_tr_priv) HIPCHK(hipMalloc((void **)&d->d_tr_priv, 48));
    HIPCHK(sc::copy_in(d->d_K, K, 32));
    HIPCHK(sc::copy_in(d->d_tr_priv, tr, 48));
    d->have_priv = true;
    return SC_HIP_SUCCESS;
}

extern "C" int sc_hip_dilithium_set_pubkey(sc_hip_dilithium_t *d, const int32_t *t1, const uint8_t *rho, const int32_t *Ahat)
{
    if (!d || !t1 || !rho || !Ahat) return fail("dilithium_set_pubkey: null argument");
    HIPCHK(hipSetDevice(d->ctx->device));
    const int k = d->P.k;
    std::vector<int32_t> t1s((size_t)k * N);
    for (size_t i = 0; i < t1s.size(); i++) t1s[i] = t1[i] << d->P.d;              // t1 . 2^d (dilithium.c:2163-2166)
    if (key_ntt(d, t1s.data(), (size_t)k, &d->d_t1s)) return SC_HIP_FAILURE;
    if (load_matrix(d, Ahat)) return SC_HIP_FAILURE;
    if (!d->d_rho) HIPCHK(hipMalloc((void **)&d->d_rho, 32));
    if (!d->d_tr_pub) HIPCHK(hipMalloc((void **)&d->d_tr_pub, 48));
    HIPCHK(sc::copy_in(d->d_rho, rho, 32));
    DevBuf t1b;
    if (t1b.get(t1s.size() * 4)) return SC_HIP_FAILURE;
    int32_t *dt1 = t1b.as<int32_t>();
    HIPCHK(sc::copy_in(dt1, t1, t1s.size() * 4));
    hipLaunchKernelGGL(dil_crh_t1_kernel, dim3(1), dim3(HT), 0, nullptr, d->d_rho, dt1, (uint32_t)t1s.size(), d->P.q_bits - d->P.d, d->d_tr_pub, 1u);
    int rc = (hipDeviceSynchronize() == hipSuccess) ? SC_HIP_SUCCESS : fail("dilithium_set_pubkey: CRH kernel failed");
    d->have_pub = (rc == SC_HIP_SUCCESS);
    return rc;
}

// generic module-LWE product (create_rand_product_32/16_csprng with the matrix expanded, module_lwe.c:588-748):
// t[b][i] = INTT( sum_j A[i][j] o NTT(y[b][j]) ), canonical.  n = 256 (every user of the reference is).
extern "C" int sc_hip_matvec(sc_hip_plan_t *p, size_t batch, int kk, int l, int transposed,
                             const int32_t *d_Ahat, const int32_t *d_y, int32_t *d_t, void *stream)
{
    if (!p) return fail("matvec: null plan");
    if (p->n != 256) return fail("matvec: n must be 256");
    if (kk < 1 || kk > 16 || l < 1 || l > 16) return fail("matvec: k, l must be in 1..16");
    if (batch == 0) return SC_HIP_SUCCESS;
    if (batch > 0x7fffffffu / 32) return fail("matvec: batch too large");
    if (!d_Ahat || !d_y || !d_t) return fail("matvec: null device pointer");
    HIPCHK(hipSetDevice(p->ctx->device));
    // The scratch (NTT(y) and the matrix in internal order) belongs to the plan, and plans are shared between threads and streams: calls are
    // serialised on the host by the plan's mutex and on the device by an event - a call on another stream waits for the previous product to
    // finish with the scratch, and the scratch is only freed once nothing uses it.
    std::lock_guard<std::mutex> mv_lock(p->mv_mu);
    hipStream_t s = (hipStream_t)stream;
    const size_t need = ((size_t)kk * l + batch * (size_t)l) * N * 4;
    if (need > p->mv_cap) {
        if (p->mv_event) HIPCHK(hipEventSynchronize((hipEvent_t)p->mv_event));
        if (p->mv_ws) (void)sc::wipe_free(p->mv_ws);
        p->mv_ws = nullptr; p->mv_cap = 0;
        HIPCHK(hipMalloc((void **)&p->mv_ws, need));
        p->mv_cap = need;
    }
    if (!p->mv_event) { hipEvent_t e; HIPCHK(hipEventCreateWithFlags(&e, hipEventDisableTiming)); p->mv_event = e; }
    else HIPCHK(hipStreamWaitEvent(s, (hipEvent_t)p->mv_event, 0));
    uint32_t *A_int = p->mv_ws, *yh = p->mv_ws + (size_t)kk * l * N;
    const size_t np = (size_t)kk * l;
    hipLaunchKernelGGL(dil_to_internal_mont, dim3((unsigned)((np * N + 255) / 256)), dim3(256), 0, s, d_Ahat, A_int, np, p->k[1], p->r2, p->r2p,
                       transposed ? kk : 0, transposed ? l : 0);
    if (hipGetLastError() != hipSuccess) { (void)hipEventRecord((hipEvent_t)p->mv_event, s); return fail("matvec: kernel launch failed"); }
    sc_hip_dilithium tmp;
    memset(&tmp, 0, sizeof tmp);
    tmp.ctx = p->ctx; tmp.plan = p; tmp.P.n = 256; tmp.P.q = p->q; tmp.P.k = kk; tmp.P.l = l; tmp.P.gamma2 = 1 << 20;
    DilNtt a = base_args(&tmp);
    a.in = d_y; a.out_int = yh; a.yh = yh; a.key = A_int; a.w = d_t; a.w1 = nullptr;
    // the event marks the END of this call's use of the scratch on every path out of here once work has been queued: a later call on another stream must
    // never wait on a stale event and overwrite the scratch under a partly queued product
    int rc = launch_dil<DOP_FWD_INT>(&tmp, a, (uint32_t)batch, l, s);
    if (rc == SC_HIP_SUCCESS) rc = launch_dil<DOP_MATVEC>(&tmp, a, (uint32_t)batch, kk, s);
    if (hipEventRecord((hipEvent_t)p->mv_event, s) != hipSuccess && rc == SC_HIP_SUCCESS) rc = fail("matvec: hipEventRecord failed");
    return rc;
}

// test hook (include/sc_hip_bench.h): words of the per-call secrets (masks y, w = A y, NTT(y)) that are not zero in the signer's workspace
namespace {
__global__ void dil_residue_kernel(const uint32_t *p, size_t words, unsigned long long *count)
{
    unsigned long long c = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < words; i += (size_t)gridDim.x * blockDim.x) c += p[i] != 0;
    if (c) atomicAdd(count, c);
}
}
extern "C" int sc_hip_dilithium_workspace_residue(sc_hip_dilithium_t *d, unsigned long long *nonzero_words)
{
    if (!d || !nonzero_words) return fail("dilithium_workspace_residue: null argument");
    *nonzero_words = 0;
    if (!d->slot_cap) return SC_HIP_SUCCESS;
    HIPCHK(hipSetDevice(d->ctx->device));
    DevBuf cnt;
    if (cnt.get(8)) return SC_HIP_FAILURE;
    HIPCHK(hipMemset(cnt.p, 0, 8));
    const size_t n = d->P.n;
    const struct { const void *p; size_t words; } parts[3] = { { d->y, d->slot_cap * d->P.l * n }, { d->w, d->slot_cap * d->P.k * n }, { d->yh, d->slot_cap * d->P.l * n } };
    for (const auto &pt : parts) hipLaunchKernelGGL(dil_residue_kernel, dim3(1024), dim3(256), 0, nullptr, (const uint32_t *)pt.p, pt.words, cnt.as<unsigned long long>());
    HIPCHK(hipDeviceSynchronize());
    HIPCHK(sc::copy_out(nonzero_words, cnt.p, 8));
    return SC_HIP_SUCCESS;
}

// measurement hook (include/sc_hip_bench.h)
extern "C" int sc_hip_dilithium_set_tuning(sc_hip_dilithium_t *d, int slots_per_message, int unfused)
{
    if (!d) return fail("dilithium_set_tuning: null handle");
    d->tune_slots = slots_per_message & 0xFF; d->tune_sync = (slots_per_message >> 8) & 1; d->tune_exp1 = (slots_per_message >> 9) & 1; d->tune_w24 = (slots_per_message >> 10) & 1; d->tune_sparse_fin = (slots_per_message >> 11) & 1; d->tune_unfused = unfused ? 1 : 0;
    // bit 11's tables are made from the raw key at key load and nothing of the key stays on the host: a key loaded without the bit has to be loaded again
    if (d->tune_sparse_fin && d->have_priv && !d->tabs_tried) {
        d->tune_sparse_fin = 0;
        return fail("dilithium_set_tuning: bit 11 (finishing kernel without transforms) must be set before sc_hip_dilithium_set_privkey - load the key again after setting it");
    }
    return SC_HIP_SUCCESS;
}

extern "C" size_t sc_hip_dilithium_sig_max(const sc_hip_dilithium_t *d)
{
    if (!d) return 0;
    const DilP &P = d->P;
    const size_t bits = (size_t)P.l * P.n * P.z_bits + P.omega_bits + (size_t)P.omega * (8 + ((P.k + 1) >> 1)) + 2 * (size_t)P.n;
    return (bits + 7) / 8;
}

// d_msgs: [batch][msg_stride] bytes, d_mlens: per-message lengths (NULL: every message is fixed_len bytes);
// d_sigs: [batch][sig_stride] bytes (sig_stride >= sc_hip_dilithium_sig_max), d_siglens / d_trials: uint32 [batch]
// the signer's key material on the device: one key for the whole batch (strides 0) or one per message
struct DilSignKeys {
    const uint32_t *A, *As;               // matrix: Montgomery internal form; plain | Shoup form for the fused kernel (null: unfused pair)
    const uint32_t *s1, *s2, *t0;         // NTT, internal order, Montgomery form
    const uint8_t *K, *tr;
    bool per_item;
    bool A_nat;                           // A is in natural order and plain form (as the expansion writes it): NTT(y) then carries the Montgomery factor (DilNtt::key_nat)
    hipEvent_t matrix_ready;              // non-null: A is being written on another stream; the first kernel that reads it waits for this
    // per_item, optional (dil_cfins_kernel<K, L, true>, tuning bit 11): the keys in the coefficient domain - s1 [item][l][n], s2 [item][k][n] as the caller
    // holds them, t0 [item][k][n] canonical; *sparse_refused = true when a key was outside the range that kernel serves (the caller signs again without it)
    const int32_t *s1n = nullptr, *s2n = nullptr, *t0n = nullptr;
    bool *sparse_refused = nullptr;
};

static int sign_core(sc_hip_dilithium_t *d, const DilSignKeys &ks, size_t batch, const uint8_t *d_msgs, size_t msg_stride, const uint32_t *d_mlens,
                     uint32_t fixed_len, uint8_t *d_sigs, size_t sig_stride, uint32_t *d_siglens, uint32_t *d_trials, void *stream);

extern "C" int sc_hip_dilithium_sign(sc_hip_dilithium_t *d, size_t batch, const uint8_t *d_msgs, size_t msg_stride, const uint32_t *d_mlens,
                                     uint32_t fixed_len, uint8_t *d_sigs, size_t sig_stride, uint32_t *d_siglens, uint32_t *d_trials, void *stream)
{
    if (!d) return fail("dilithium_sign: null handle");
    if (!d->have_priv) return fail("dilithium_sign: no private key loaded");
    const DilSignKeys ks = { d->d_A, d->tune_unfused ? nullptr : d->d_As, d->d_s1, d->d_s2, d->d_t0, d->d_K, d->d_tr_priv, false, false, nullptr };
    return sign_core(d, ks, batch, d_msgs, msg_stride, d_mlens, fixed_len, d_sigs, sig_stride, d_siglens, d_trials, stream);
}

// rounds to queue before the first look at the count of unsigned messages: the schedule run forward with a (pessimistic) rejection probability of
// 3/4 per attempt until less than half a message is expected to be left
static uint32_t plan_rounds(size_t batch, size_t slot_cap, uint32_t s_max)
{
    double n = (double)batch;
    uint32_t r = 0;
    while (n >= 0.5 && r < 64) {
        const double live = n < 1.0 ? 1.0 : n;
        double S = (double)slot_cap / live;
        if (S > s_max) S = s_max;
        if (S < 1) S = 1;
        double f = 1.0;
        for (int i = 0; i < (int)S; i++) f *= 0.75;
        n *= f; r++;
    }
    return r ? r : 1;
}

static int sign_core(sc_hip_dilithium_t *d, const DilSignKeys &ks, size_t batch, const uint8_t *d_msgs, size_t msg_stride, const uint32_t *d_mlens,
                     uint32_t fixed_len, uint8_t *d_sigs, size_t sig_stride, uint32_t *d_siglens, uint32_t *d_trials, void *stream)
{
    if (batch == 0) return SC_HIP_SUCCESS;
    if (batch > 0x7fffffffu / 16) return fail("dilithium_sign: batch too large");
    if (!d_msgs && (d_mlens || fixed_len)) return fail("dilithium_sign: null message buffer");
    if (!d_sigs || !d_siglens) return fail("dilithium_sign: null output buffer");
    if (sig_stride < sc_hip_dilithium_sig_max(d)) return fail("dilithium_sign: sig_stride smaller than sc_hip_dilithium_sig_max()");
    HIPCHK(hipSetDevice(d->ctx->device));
    // Attempts evaluated side by side per round ("slots").  More slots per message mean fewer rounds (each round costs the latency
    // of the lane-per-attempt hash kernels) but more attempts thrown away after the first success.
    // Measured on MI355X (tools/bench_dilithium.py, sc_hip_dilithium_set_tuning slots = 1..8): about 48 K slots per round is the sweet spot - 8 per
    // message for small batches, 3 at 16 K messages, 1 from 48 K messages up; as the batch thins out the same slots go deeper.
    size_t slot_target = batch * 8 < 49152 ? batch * 8 : 49152;
    if (slot_target < batch) slot_target = batch;
    if (d->tune_slots >= 1 && d->tune_slots <= 32) slot_target = batch * (size_t)d->tune_slots;
    if (slot_target < 4096) slot_target = 4096;
    if (ensure_ws(d, batch, slot_target)) return SC_HIP_FAILURE;
    hipStream_t s = (hipStream_t)stream;
    const DilP &P = d->P;
    const uint32_t B = (uint32_t)batch, cap = (uint32_t)d->slot_cap, s_max = 32;
    const int k = P.k, l = P.l;
    HIPCHK(hipMemsetAsync(d->kappa, 0, batch * 4, s));
    uint32_t *trials = d_trials ? d_trials : d->trials;
    hipLaunchKernelGGL(dil_mu_kernel, dim3((B + HT - 1) / HT), dim3(HT), 0, s, ks.tr, d_msgs, msg_stride, d_mlens, fixed_len, d->mu, B, ks.per_item ? 48u : 0u);
    HIPCHK(hipGetLastError());

    // The round loop runs on the device: every kernel of a round takes its sizes from the round state (rs), which the round's first kernel derives
    // from the count its predecessor's last kernel left behind; grids are sized for the slot budget and the surplus workgroups leave at once.  The
    // host queues as many rounds as the schedule is expected to need, looks at the count once, and adds single rounds while messages are left.
    const size_t fin_lds = (4 + 4 * cfin_wave_words(P)) * sizeof(uint32_t);
    // one key for the batch whose coefficient-domain tables exist: the finishing kernel without transforms - ONLY behind tuning bit 11: the kernel was written
    // while the GPU pool was closed to this build and has not run on hardware yet; the transform kernel stays the product path until it has
    const bool sparse_fin = d->tune_sparse_fin && !ks.per_item && ks.s1 == d->d_s1 && d->d_tabs && !d->tune_w24 && l == k - 1 && k >= 3 && k <= 6;     // (the four parameter sets, dilithium_params.c)
    // a key per message: the same kernel building every message's tables in LDS from the raw key rows, the fields sized for |s| <= eta (what key generation
    // gives; anything else is refused on the device and signed again by the caller through the transform kernel)
    const bool sparse_keys = d->tune_sparse_fin && ks.per_item && ks.s1n && ks.s2n && ks.t0n && ks.sparse_refused && l == k - 1 && k >= 3 && k <= 6
                             && P.eta >= 1 && 2 * P.eta * P.weight <= 1023 && P.d <= 16;
    const size_t fink_lds = (4 + 2 * (cfins_table_words(P) + cfins_wave_words(P))) * sizeof(uint32_t);
    uint32_t *refused_dev = d->cnt + 8;                                 // (d->cnt is 16 words; the rounds use the first)
    if (sparse_keys) HIPCHK(hipMemsetAsync(refused_dev, 0, 4, s));
    const size_t fins_lds = (4 + cfins_table_words(P) + 4 * cfins_wave_words(P)) * sizeof(uint32_t);
    DilFin f;
    f.c8 = d->c8; f.s1 = ks.s1; f.s2 = ks.s2; f.t0 = ks.t0;
    f.s1_stride = ks.per_item ? (size_t)l * N : 0; f.k_stride = ks.per_item ? (size_t)k * N : 0;
    f.rs = d->rs; f.kappa = d->kappa; f.sigs = d_sigs; f.sig_stride = sig_stride; f.siglens = d_siglens; f.trials = trials; f.next_count = d->cnt; f.P = P;
    // ExpandMask with two lanes per attempt while the one-lane form would leave SIMDs without a second wavefront; one workgroup more than the slot
    // budget needs, so that there is always one to zero the slots a shrinking round leaves behind
    const bool exp2 = d->tune_exp1 ? false : (size_t)cap <= (size_t)d->ctx->cus * 4 * 64;
    const unsigned eb = (unsigned)(((size_t)cap * (exp2 ? 2 : 1) + EXP_T - 1) / EXP_T) + 1, hb = (cap + CH_T - 1) / CH_T;
    uint32_t *yh_wipe = (ks.As && !ks.per_item) ? nullptr : d->yh;     // NTT(y) where the unfused transform / product pair leaves it in HBM
    // sc_hip_dilithium_set_tuning bit 10 (one key for the batch: the fused kernel): w = A y travels between the round's kernels in 24 bits per coefficient
    // (mlwe_kernels.h: W24) - a quarter less of what the signer moves most of.  Measured on one box, three alternating pairs of runs:
    // 205 -> 187 KB per signature by the counters, 2.656 -> 2.687 ms per 16384-signature call - the finishing kernel (255 registers, waiting on its loads)
    // pays more for the unpacking than it gains from the shorter rows.  Throughput is the metric, so 32-bit words stay the default and this form stays
    // behind the hook.  Tried on top of it and dropped: the masks y in 24 bits as well (ExpandMask writing q + y at byte 3 j, unaligned word loads in the
    // fused and the finishing kernel): 166 KB per signature, 2.75 ms per call.
    const bool packed = d->tune_w24 && ks.As && !ks.per_item && P.q < (1 << 24) && P.n == N;
    const uint32_t w_words = packed ? (uint32_t)(k * W24) : (uint32_t)(k * P.n);
    uint32_t round = 0, todo = d->tune_sync ? 1 : plan_rounds(batch, cap, s_max), left = B;
    while (left > 0) {
        for (uint32_t i = 0; i < todo; i++, round++) {
            if (round > 4000) return fail("dilithium_sign: rejection loop did not terminate");
            const uint32_t *active = round ? d->act[(round - 1) & 1] : nullptr;     // round 0: identity
            f.active = active; f.next_active = d->act[round & 1];
            hipLaunchKernelGGL(dil_round_kernel, dim3(1), dim3(1), 0, s, d->rs, d->cnt, round ? 0u : B, cap, s_max);
            if (exp2) hipLaunchKernelGGL(dil_expand_kernel<true>, dim3(eb), dim3(EXP_T), 0, s, d->rs, active, d->kappa, ks.K, d->mu, d->y, l * P.n, P.gamma1, P.q, ks.per_item ? 32u : 0u, d->slot_item,
                                         d->w, w_words, yh_wipe);
            else hipLaunchKernelGGL(dil_expand_kernel<false>, dim3(eb), dim3(EXP_T), 0, s, d->rs, active, d->kappa, ks.K, d->mu, d->y, l * P.n, P.gamma1, P.q, ks.per_item ? 32u : 0u, d->slot_item,
                                    d->w, w_words, yh_wipe);
            HIPCHK(hipGetLastError());
            DilNtt a = base_args(d);
            a.in = d->y; a.out_int = d->yh;
            a.njobs_dev = d->rs + RS_NSLOTS; a.w_chunked = packed ? 2 : 1;
            a.key_of = d->slot_item;                                    // per-message keys: the key of a slot is its message's
            if (ks.As && !ks.per_item) {                                // w = A y with NTT(y) kept in registers
                a.key = ks.As; a.keyp = ks.As + (size_t)k * l * N;
                if (mlwe_launch_yaw(d->ctx, a, cap, s)) return SC_HIP_FAILURE;
                a.keyp = nullptr;
            } else {
                if (ks.A_nat ? launch_dil<DOP_KEYPREP>(d, a, cap, l, s) : launch_dil<DOP_FWD_INT>(d, a, cap, l, s)) return SC_HIP_FAILURE;
                a.key = ks.A; a.key_item_stride = ks.per_item ? (size_t)k * l * N : 0; a.key_nat = ks.A_nat ? 1 : 0;
                if (ks.matrix_ready && round == 0) HIPCHK(hipStreamWaitEvent(s, ks.matrix_ready, 0));
                if (launch_dil<DOP_MATVEC>(d, a, cap, k, s)) return SC_HIP_FAILURE;
                a.key_item_stride = 0; a.key_nat = 0;
            }
            const uint32_t two_max = d->tune_exp1 ? 0u : (uint32_t)d->ctx->cus * 4 * 36;      // 36864 attempts on 256 CUs
            const unsigned hb2 = (unsigned)(((size_t)(cap < two_max ? cap : two_max) * 2 + CH_T - 1) / CH_T);
            if (hb2) hipLaunchKernelGGL((dil_challenge_kernel<true, true>), dim3(hb2), dim3(CH_T), 0, s, d->rs, d->slot_item, 0u, d->mu, d->w1, k * P.n, P.weight, d->c8, (const int32_t *)nullptr, (uint32_t *)nullptr, two_max);
            hipLaunchKernelGGL((dil_challenge_kernel<true, false>), dim3(hb), dim3(CH_T), 0, s, d->rs, d->slot_item, 0u, d->mu, d->w1, k * P.n, P.weight, d->c8, (const int32_t *)nullptr, (uint32_t *)nullptr, two_max);
            HIPCHK(hipGetLastError());
            // one wavefront per message that may still be unsigned: the host only knows the count at its last look
            const unsigned fb = (left + 3) / 4;
#define SC_CFIN(M) do { if (l > 4 && k > 4) hipLaunchKernelGGL((dil_cfin_kernel<M, true, true>), dim3(fb), dim3(256), fin_lds, s, a, f); \
                        else if (l > 4) hipLaunchKernelGGL((dil_cfin_kernel<M, true, false>), dim3(fb), dim3(256), fin_lds, s, a, f); \
                        else if (k > 4) hipLaunchKernelGGL((dil_cfin_kernel<M, false, true>), dim3(fb), dim3(256), fin_lds, s, a, f); \
                        else hipLaunchKernelGGL((dil_cfin_kernel<M, false, false>), dim3(fb), dim3(256), fin_lds, s, a, f); } while (0)
            if (sparse_keys) {
                const DilFinS g{ nullptr, (uint32_t)P.eta, ks.s1n, ks.s2n, ks.t0n, refused_dev };
                const unsigned fk = (left + 1) / 2;
                if (k == 6) hipLaunchKernelGGL((dil_cfins_kernel<6, 5, true>), dim3(fk), dim3(128), fink_lds, s, a, f, g);
                else if (k == 5) hipLaunchKernelGGL((dil_cfins_kernel<5, 4, true>), dim3(fk), dim3(128), fink_lds, s, a, f, g);
                else if (k == 4) hipLaunchKernelGGL((dil_cfins_kernel<4, 3, true>), dim3(fk), dim3(128), fink_lds, s, a, f, g);
                else hipLaunchKernelGGL((dil_cfins_kernel<3, 2, true>), dim3(fk), dim3(128), fink_lds, s, a, f, g);
            } else if (sparse_fin) {
                const DilFinS g{ d->d_tabs, d->fin_bias };
                if (k == 6 && l == 5) hipLaunchKernelGGL((dil_cfins_kernel<6, 5>), dim3(fb), dim3(256), fins_lds, s, a, f, g);
                else if (k == 5 && l == 4) hipLaunchKernelGGL((dil_cfins_kernel<5, 4>), dim3(fb), dim3(256), fins_lds, s, a, f, g);
                else if (k == 4 && l == 3) hipLaunchKernelGGL((dil_cfins_kernel<4, 3>), dim3(fb), dim3(256), fins_lds, s, a, f, g);
                else hipLaunchKernelGGL((dil_cfins_kernel<3, 2>), dim3(fb), dim3(256), fins_lds, s, a, f, g);
            }
            else if (a.sg) SC_CFIN(MODE_S32); else SC_CFIN(MODE_U32);
#undef SC_CFIN
            HIPCHK(hipGetLastError());
        }
        // the last queued round's slots (every earlier round's were zeroed by its successor's surplus workgroups) - in front of the look at the count, not behind
        // it: the call usually ends here, and the zeroing then runs while the host waits for the count instead of after it has woken up (a round that follows
        // finds its predecessor's slots zeroed twice)
        hipLaunchKernelGGL(dil_wipe_kernel, dim3(cap < 4096u ? cap : 4096u), dim3(256), 0, s, d->rs, d->y, (uint32_t)(l * P.n), d->w, w_words, yh_wipe);
        HIPCHK(hipGetLastError());
        uint32_t cnt = 0;
        HIPCHK(hipMemcpyAsync(&cnt, d->cnt, 4, hipMemcpyDeviceToHost, s));
        HIPCHK(hipStreamSynchronize(s));
        left = cnt;
        todo = 1;
    }
    if (sparse_keys) {
        uint32_t refused = 0;
        HIPCHK(hipMemcpyAsync(&refused, refused_dev, 4, hipMemcpyDeviceToHost, s));
        HIPCHK(hipStreamSynchronize(s));
        *ks.sparse_refused = refused != 0;
    }
    return SC_HIP_SUCCESS;
}

// d_verdicts[i] = 0 (SC_FUNC_SUCCESS) when signature i verifies, 1 otherwise (test/functional/func_alg_dilithium.c:225)
namespace {
// joins the handle's side streams back into the caller's on every exit path of a call that forked them
struct JoinGuard {
    hipStream_t s, side[2];
    hipEvent_t ev[2];
    ~JoinGuard()
    {
        for (int i = 0; i < 2; i++)
            if (side[i] && ev[i]) { if (hipEventRecord(ev[i], side[i]) == hipSuccess) (void)hipStreamWaitEvent(s, ev[i], 0); else (void)hipGetLastError(); }
    }
};
}

extern "C" int sc_hip_dilithium_verify(sc_hip_dilithium_t *d, size_t batch, const uint8_t *d_msgs, size_t msg_stride, const uint32_t *d_mlens,
                                       uint32_t fixed_len, const uint8_t *d_sigs, size_t sig_stride, const uint32_t *d_siglens,
                                       uint32_t *d_verdicts, void *stream)
{
    if (!d) return fail("dilithium_verify: null handle");
    if (!d->have_pub) return fail("dilithium_verify: no public key loaded");
    if (batch == 0) return SC_HIP_SUCCESS;
    if (batch > 0x7fffffffu / 16) return fail("dilithium_verify: batch too large");
    if (!d_sigs || !d_siglens || !d_verdicts) return fail("dilithium_verify: null buffer");
    if (!d_msgs && (d_mlens || fixed_len)) return fail("dilithium_verify: null message buffer");
    HIPCHK(hipSetDevice(d->ctx->device));
    if (ensure_ws(d, batch, batch)) return SC_HIP_FAILURE;
    hipStream_t s = (hipStream_t)stream;
    const DilP &P = d->P;
    const uint32_t B = (uint32_t)batch;
    const int k = P.k, l = P.l;
    if ((size_t)(l * P.n * P.z_bits) / 8 > sig_stride) return fail("dilithium_verify: sig_stride too small");
    HIPCHK(hipMemsetAsync(d_verdicts, 0, batch * 4, s));
    const unsigned hb = (B + HT - 1) / HT;
    if (l * P.n / 8 > 256) return fail("dilithium_verify: l n above 2048");      // (z's unpacking: a lane per eight coefficients, a workgroup per signature)
    // (measured and dropped: the tail unpacking and mu = CRH(tr, message) on a stream of the handle's own beside the unpacking of z and the matrix product -
    // 39 us of kernels off the critical path, but forking and joining cost as much: 354 -> 359 us per 16384 signatures)
    if (P.z_bits == 20) hipLaunchKernelGGL(dil_tail_mu_kernel<20>, dim3((B + 3) / 4 + hb + B), dim3(256), 0, s, d_sigs, sig_stride, d_siglens, B, d->h, d->c, d_verdicts, P,
                                           d->d_tr_pub, d_msgs, msg_stride, d_mlens, fixed_len, d->mu, d->z);
    else hipLaunchKernelGGL(dil_tail_mu_kernel<0>, dim3((B + 3) / 4 + hb + B), dim3(256), 0, s, d_sigs, sig_stride, d_siglens, B, d->h, d->c, d_verdicts, P,
                            d->d_tr_pub, d_msgs, msg_stride, d_mlens, fixed_len, d->mu, d->z);
    HIPCHK(hipGetLastError());
    DilNtt a = base_args(d);
    // one key for the batch: NTT(c) first, then ONE fused launch - NTT(z), the k rows of A NTT(z) - NTT(c) NTT(t1 2^d), their inverse transforms and UseHint -> w1
    // (round 4: w' is never written, c t1 2^d is never transformed back on its own; 329 -> ~285 us per 16384 signatures).  The signed arithmetic has room for the
    // extra term while 16 (2 l + 1) q < 2^31.
    const bool fused_ct1 = d->d_As && !d->tune_unfused && (!a.sg || 16ull * (2ull * (uint64_t)l + 1) * (uint64_t)P.q < (1ull << 31));
    if (fused_ct1) {
        a.in = d->c; a.out_int = d->ch;
        if (launch_dil<DOP_FWD_INT>(d, a, B, 1, s)) return SC_HIP_FAILURE;
        a.in = d->z; a.out_int = d->yh;
        a.key = d->d_As; a.keyp = d->d_As + (size_t)k * l * N; a.t1s = d->d_t1s;
        if (mlwe_launch_yaw(d->ctx, a, B, s, true)) return SC_HIP_FAILURE;
        a.keyp = nullptr; a.t1s = nullptr;
    } else {
    a.in = d->z; a.out_int = d->yh;
    if (d->d_As && !d->tune_unfused) {
        a.key = d->d_As; a.keyp = d->d_As + (size_t)k * l * N;
        if (mlwe_launch_yaw(d->ctx, a, B, s)) return SC_HIP_FAILURE;
        a.keyp = nullptr;
    } else {
        if (launch_dil<DOP_FWD_INT>(d, a, B, l, s)) return SC_HIP_FAILURE;
        a.key = d->d_A;
        if (launch_dil<DOP_MATVEC>(d, a, B, k, s)) return SC_HIP_FAILURE;
    }
    a.in = d->c; a.out_int = d->ch;
    if (launch_dil<DOP_FWD_INT>(d, a, B, 1, s)) return SC_HIP_FAILURE;
    a.key = d->d_t1s;
    if (launch_dil<DOP_CT1>(d, a, B, k, s)) return SC_HIP_FAILURE;
    }
    // (two lanes per hash while the one-lane form would leave SIMDs without a second wavefront)
    if ((size_t)B <= (size_t)d->ctx->cus * 4 * 64) hipLaunchKernelGGL((dil_challenge_kernel<false, true>), dim3((2 * B + CH_T - 1) / CH_T), dim3(CH_T), 0, s, (const uint32_t *)nullptr, (const uint32_t *)nullptr, B, d->mu, d->w1, k * P.n, P.weight, (int8_t *)nullptr, d->c, d_verdicts, 0u);
    else hipLaunchKernelGGL((dil_challenge_kernel<false, false>), dim3((B + CH_T - 1) / CH_T), dim3(CH_T), 0, s, (const uint32_t *)nullptr, (const uint32_t *)nullptr, B, d->mu, d->w1, k * P.n, P.weight, (int8_t *)nullptr, d->c, d_verdicts, 0u);
    HIPCHK(hipGetLastError());
    return SC_HIP_SUCCESS;
}

// ---- host-buffer convenience (the shape safecrypto_sign / safecrypto_verify have, for a batch) ----

static int stage(void **dev, size_t *cap, size_t bytes)
{
    if (bytes <= *cap) return SC_HIP_SUCCESS;
    if (*dev) (void)sc::wipe_free(*dev);
    *dev = nullptr; *cap = 0;
    HIPCHK(hipMalloc(dev, bytes));
    *cap = bytes;
    return SC_HIP_SUCCESS;
}

extern "C" int sc_hip_dilithium_sign_host(sc_hip_dilithium_t *d, size_t batch, const uint8_t *h_msgs, size_t msg_stride, const uint32_t *h_mlens,
                                          uint32_t fixed_len, uint8_t *h_sigs, size_t sig_stride, uint32_t *h_siglens, uint32_t *h_trials)
{
    if (!d) return fail("dilithium_sign_host: null handle");
    if (batch == 0) return SC_HIP_SUCCESS;
    if (!h_sigs || !h_siglens) return fail("dilithium_sign_host: null output");
    HIPCHK(hipSetDevice(d->ctx->device));
    const size_t moff = (batch * msg_stride + 7) & ~(size_t)7, sbytes = batch * sig_stride, soff = (sbytes + 7) & ~(size_t)7;
    if (stage((void **)&d->msg_dev, &d->msg_cap, moff + batch * 4 + 8) || stage((void **)&d->sig_dev, &d->sig_cap, soff + batch * 8 + 8)) return SC_HIP_FAILURE;
    uint32_t *dl = (uint32_t *)(d->msg_dev + moff), *dsl = (uint32_t *)(d->sig_dev + soff), *dtr = dsl + batch;
    if (h_msgs && batch * msg_stride) HIPCHK(sc::copy_in(d->msg_dev, h_msgs, batch * msg_stride));
    if (h_mlens) HIPCHK(sc::copy_in(dl, h_mlens, batch * 4));
    int rc = sc_hip_dilithium_sign(d, batch, d->msg_dev, msg_stride, h_mlens ? dl : nullptr, fixed_len, d->sig_dev, sig_stride, dsl, dtr, nullptr);
    if (rc) return rc;
    HIPCHK(sc::copy_out(h_sigs, d->sig_dev, sbytes));
    HIPCHK(sc::copy_out(h_siglens, dsl, batch * 4));
    if (h_trials) HIPCHK(sc::copy_out(h_trials, dtr, batch * 4));
    return SC_HIP_SUCCESS;
}

extern "C" int sc_hip_dilithium_verify_host(sc_hip_dilithium_t *d, size_t batch, const uint8_t *h_msgs, size_t msg_stride, const uint32_t *h_mlens,
                                            uint32_t fixed_len, const uint8_t *h_sigs, size_t sig_stride, const uint32_t *h_siglens, uint32_t *h_verdicts)
{
    if (!d) return fail("dilithium_verify_host: null handle");
    if (batch == 0) return SC_HIP_SUCCESS;
    if (!h_sigs || !h_siglens || !h_verdicts) return fail("dilithium_verify_host: null buffer");
    HIPCHK(hipSetDevice(d->ctx->device));
    const size_t moff = (batch * msg_stride + 7) & ~(size_t)7, sbytes = batch * sig_stride, soff = (sbytes + 7) & ~(size_t)7;
    if (stage((void **)&d->msg_dev, &d->msg_cap, moff + batch * 4 + 8) || stage((void **)&d->sig_dev, &d->sig_cap, soff + batch * 8 + 8)) return SC_HIP_FAILURE;
    uint32_t *dl = (uint32_t *)(d->msg_dev + moff), *dsl = (uint32_t *)(d->sig_dev + soff), *dv = dsl + batch;
    if (h_msgs && batch * msg_stride) HIPCHK(sc::copy_in(d->msg_dev, h_msgs, batch * msg_stride));
    if (h_mlens) HIPCHK(sc::copy_in(dl, h_mlens, batch * 4));
    HIPCHK(sc::copy_in(d->sig_dev, h_sigs, sbytes));
    HIPCHK(sc::copy_in(dsl, h_siglens, batch * 4));
    int rc = sc_hip_dilithium_verify(d, batch, d->msg_dev, msg_stride, h_mlens ? dl : nullptr, fixed_len, d->sig_dev, sig_stride, dsl, dv, nullptr);
    if (rc) return rc;
    HIPCHK(hipStreamSynchronize(nullptr));
    HIPCHK(sc::copy_out(h_verdicts, dv, batch * 4));
    return SC_HIP_SUCCESS;
}

// the reference's hash as a test hook: SHAKE128/256 framing around its permutation (standard != 0: FIPS 202 chi)
namespace {
template <bool STD>
__global__ __launch_bounds__(HT) void shake_test_kernel(int rate, const uint8_t *in, uint32_t len, uint8_t *out, uint32_t outlen)
{
    __shared__ uint64_t st[25 * HT];
    Sponge<HT, STD> sp; sp.init(st, threadIdx.x, rate);
    if (threadIdx.x != 0) return;
    sp.absorb(in, len);
    sp.finalize();
    for (uint32_t i = 0; i < outlen; i++) out[i] = sp.squeeze_byte();
}
}  // namespace

// the calls with a key per item fork work onto streams of the handle's own: whatever path leaves such a call, the caller's stream waits for everything
// that was queued on them (an early return must not leave side work running beside the caller's next call on the same workspace)
namespace {
}

// t0 of every key from t (decompose, module_lwe.c:89-104), canonical in [0, q): what sc_hip_dilithium_set_privkey does on the host for one key
__global__ void dil_t0_kernel(const int32_t *__restrict__ t, int32_t *__restrict__ t0, size_t count, int d, int32_t q)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    const int32_t mask = (1 << d) - 1, sign = 1 << d, thresh = 1 << (d - 1);
    int32_t x = t[i] & mask;
    x -= (x > thresh) * sign;
    t0[i] = (x >= 0) ? x : x + q;
}

// replaces: safecrypto_sign -> dilithium_sign (dilithium.c:1397-1698) for `batch` messages, EACH under its own private key (SURVEY 8(d) config 3,
// "16384 distinct keys").  Keys in the layouts sc_hip_dilithium_keygen writes (= the key pair's, dilithium.c:810-823): d_s1 [batch][l][n],
// d_s2, d_t [batch][k][n] int32; d_rho, d_K [batch][32]; d_tr [batch][48].  Per key the matrix is regenerated from rho and s1, s2, t0 are transformed
// on the device; the rejection loop then runs as for one key, every attempt reading its message's key (17 + k l polynomials per attempt instead of a
// shared set, and the unfused NTT(y) / matrix-product pair since the fused kernel keeps ONE matrix in Shoup form).
extern "C" int sc_hip_dilithium_sign_keys(sc_hip_dilithium_t *d, size_t batch, const int32_t *d_s1, const int32_t *d_s2, const int32_t *d_t,
                                          const uint8_t *d_rho, const uint8_t *d_K, const uint8_t *d_tr,
                                          const uint8_t *d_msgs, size_t msg_stride, const uint32_t *d_mlens, uint32_t fixed_len,
                                          uint8_t *d_sigs, size_t sig_stride, uint32_t *d_siglens, uint32_t *d_trials, void *stream)
{
    if (!d) return fail("dilithium_sign_keys: null handle");
    if (batch == 0) return SC_HIP_SUCCESS;
    if (batch > 0x7fffffffu / 4096) return fail("dilithium_sign_keys: batch too large");
    if (!d_s1 || !d_s2 || !d_t || !d_rho || !d_K || !d_tr) return fail("dilithium_sign_keys: null key buffer");
    HIPCHK(hipSetDevice(d->ctx->device));
    hipStream_t s = (hipStream_t)stream;
    const DilP &P = d->P;
    const uint32_t B = (uint32_t)batch;
    const int k = P.k, l = P.l;
    const size_t kn = (size_t)k * N, ln = (size_t)l * N, npoly = batch * (size_t)k * l;
    if (d->kws.reserve(npoly * N * 4 + batch * (ln + 3 * kn) * 4 + 4096)) return SC_HIP_FAILURE;
    int32_t *t0n = d->kws.take<int32_t>(batch * kn);
    uint32_t *s1h = d->kws.take<uint32_t>(batch * ln), *s2h = d->kws.take<uint32_t>(batch * kn), *t0h = d->kws.take<uint32_t>(batch * kn);
    const size_t secret_end = d->kws.mark();                        // what is zeroed behind the call; the matrix below is public
    int32_t *ab = d->kws.take<int32_t>(npoly * N);
    // the matrix comes out of the CTR-DRBG (~0.65 ms of cipher at 16 K keys) and nothing needs it before the first matrix product: it is expanded on a
    // stream of the handle's own while the caller's stream prepares the secrets, hashes the messages and expands the first round's masks (a chain bound by
    // its own length); the first matrix product waits for it.  It stays as the expansion wrote it - natural order, plain form - and NTT(y) carries the
    // Montgomery factor instead (DilNtt::key_nat): no conversion pass over 30 KB per key.
    if (!d->side) {
        HIPCHK(hipStreamCreateWithFlags(&d->side, hipStreamNonBlocking));
        HIPCHK(hipEventCreateWithFlags(&d->ev_fork, hipEventDisableTiming));
        HIPCHK(hipEventCreateWithFlags(&d->ev_join, hipEventDisableTiming));
    }
    HIPCHK(hipEventRecord(d->ev_fork, s));
    HIPCHK(hipStreamWaitEvent(d->side, d->ev_fork, 0));
    JoinGuard join{ s, { d->side, nullptr }, { d->ev_join, nullptr } };
    if (sc_hip_mlwe_expand_matrix(d->ctx, batch, d_rho, 32, 32, k * l, P.q, P.q_bits, ab, d->side)) return SC_HIP_FAILURE;
    HIPCHK(hipEventRecord(d->ev_join, d->side));
    hipLaunchKernelGGL(dil_t0_kernel, dim3((unsigned)((batch * kn + 255) / 256)), dim3(256), 0, s, d_t, t0n, batch * kn, P.d, P.q);
    HIPCHK(hipGetLastError());
    DilNtt a = base_args(d);
    int rc = SC_HIP_SUCCESS;
    a.in = d_s1; a.out_int = s1h; if (launch_dil<DOP_KEYPREP>(d, a, B * (uint32_t)l, 1, s)) rc = SC_HIP_FAILURE;
    a.in = d_s2; a.out_int = s2h; if (rc == SC_HIP_SUCCESS && launch_dil<DOP_KEYPREP>(d, a, B * (uint32_t)k, 1, s)) rc = SC_HIP_FAILURE;
    a.in = t0n; a.out_int = t0h; if (rc == SC_HIP_SUCCESS && launch_dil<DOP_KEYPREP>(d, a, B * (uint32_t)k, 1, s)) rc = SC_HIP_FAILURE;
    DilSignKeys ks = { reinterpret_cast<const uint32_t *>(ab), nullptr, s1h, s2h, t0h, d_K, d_tr, true, true, d->ev_join };
    bool refused = false;
    if (d->tune_sparse_fin) { ks.s1n = d_s1; ks.s2n = d_s2; ks.t0n = t0n; ks.sparse_refused = &refused; }
    if (rc == SC_HIP_SUCCESS) rc = sign_core(d, ks, batch, d_msgs, msg_stride, d_mlens, fixed_len, d_sigs, sig_stride, d_siglens, d_trials, stream);
    if (rc == SC_HIP_SUCCESS && refused) {                          // a key outside the table kernel's range: the whole call again through the transform kernel
        ks.s1n = nullptr; ks.matrix_ready = nullptr;
        rc = sign_core(d, ks, batch, d_msgs, msg_stride, d_mlens, fixed_len, d_sigs, sig_stride, d_siglens, d_trials, stream);
    }
    (void)hipStreamWaitEvent(s, d->ev_join, 0);                    // (again, in case the signing did not get as far as its first matrix product)
    d->kws.wipe_upto(secret_end, s);                                // the per-item t0, NTT(s1), NTT(s2), NTT(t0)
    return rc;
}

// t1 . 2^d for a batch of public keys (dilithium.c:2163-2166), ahead of the key-preparation transform
__global__ void dil_shift_kernel(const int32_t *__restrict__ in, int32_t *__restrict__ out, size_t count, int d)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < count) out[i] = in[i] << d;
}

// replaces: safecrypto_verify -> dilithium_verify (dilithium.c:2073-2260) for `batch` signatures, EACH under its own public key (SURVEY 8(d) config 3,
// "16384 distinct keys"): d_t1 [batch][k][n] int32 and d_rho [batch][32] are the keys as dilithium_pubkey_load leaves them (:904).  Per item the
// matrix is regenerated from rho on the device (the reference's CTR-DRBG, create_rand_product_32_csprng's order, module_lwe.c:588-748), tr =
// CRH(rho, t1) and NTT(t1 . 2^d) are derived, then the same kernels as the one-key path run with a key stride: 4 k l n bytes of matrix are read
// per signature instead of being shared by the batch.  d_verdicts[i] = 0 valid / 1 invalid.
extern "C" int sc_hip_dilithium_verify_keys(sc_hip_dilithium_t *d, size_t batch, const int32_t *d_t1, const uint8_t *d_rho,
                                            const uint8_t *d_msgs, size_t msg_stride, const uint32_t *d_mlens, uint32_t fixed_len,
                                            const uint8_t *d_sigs, size_t sig_stride, const uint32_t *d_siglens, uint32_t *d_verdicts, void *stream)
{
    if (!d) return fail("dilithium_verify_keys: null handle");
    if (batch == 0) return SC_HIP_SUCCESS;
    if (batch > 0x7fffffffu / 4096) return fail("dilithium_verify_keys: batch too large");
    if (!d_t1 || !d_rho || !d_sigs || !d_siglens || !d_verdicts) return fail("dilithium_verify_keys: null buffer");
    if (!d_msgs && (d_mlens || fixed_len)) return fail("dilithium_verify_keys: null message buffer");
    HIPCHK(hipSetDevice(d->ctx->device));
    if (ensure_ws(d, batch, batch)) return SC_HIP_FAILURE;
    hipStream_t s = (hipStream_t)stream;
    const DilP &P = d->P;
    const uint32_t B = (uint32_t)batch;
    const int k = P.k, l = P.l;
    if ((size_t)(l * P.n * P.z_bits) / 8 > sig_stride) return fail("dilithium_verify_keys: sig_stride too small");
    const size_t kn = (size_t)k * N, npoly = batch * (size_t)k * l;
    // grow-only workspace kept by the handle (a per-call hipMalloc of ~60 KB per item costs more than the kernels)
    if (d->kws.reserve(npoly * N * 4 + 2 * batch * kn * 4 + batch * 48 + 2048)) return SC_HIP_FAILURE;
    int32_t *ab = d->kws.take<int32_t>(npoly * N), *t1s = d->kws.take<int32_t>(batch * kn);
    uint32_t *t1h = d->kws.take<uint32_t>(batch * kn);
    uint8_t *trb = d->kws.take<uint8_t>(batch * 48);
    // Three chains that depend on nothing of each other until the matrix product: (1) the matrix of every key out of its CTR-DRBG - the cipher fills the
    // machine's LDS pipes for ~0.65 ms at 16 K keys; (2) tr = CRH(rho, t1) -> mu, one lane per key, 13 + a few sequential permutations: bound by its own
    // length (~0.4 ms), it wants few issue slots but wants them early; (3) the signature side and t1 . 2^d with their transforms: memory-bound kernels
    // that progress beside the cipher at a fifth of their speed and are done when it is.  (1) stays on the caller's stream, (2) and (3) get a stream
    // each of the handle's own; the matrix product waits for (3), the challenge at the very end for (2).
    if (!d->side) {
        HIPCHK(hipStreamCreateWithFlags(&d->side, hipStreamNonBlocking));
        HIPCHK(hipEventCreateWithFlags(&d->ev_fork, hipEventDisableTiming));
        HIPCHK(hipEventCreateWithFlags(&d->ev_join, hipEventDisableTiming));
    }
    if (!d->ev_join2) HIPCHK(hipEventCreateWithFlags(&d->ev_join2, hipEventDisableTiming));
    if (!d->side2) HIPCHK(hipStreamCreateWithFlags(&d->side2, hipStreamNonBlocking));
    hipStream_t s3 = d->side2;
    hipStream_t s2 = d->side;
    HIPCHK(hipMemsetAsync(d_verdicts, 0, batch * 4, s));
    HIPCHK(hipEventRecord(d->ev_fork, s));                               // everything the caller queued before (its inputs) is ahead of both streams
    HIPCHK(hipStreamWaitEvent(s2, d->ev_fork, 0));
    HIPCHK(hipStreamWaitEvent(s3, d->ev_fork, 0));
    JoinGuard join{ s, { s2, s3 }, { d->ev_join, d->ev_join2 } };
    // chain (2) first: its few wavefronts should be resident before the cipher's 7680 take every slot
    const unsigned hb = (B + HT - 1) / HT;
    hipLaunchKernelGGL(dil_crh_t1_kernel, dim3((B + HT - 1) / HT), dim3(HT), 0, s3, d_rho, d_t1, (uint32_t)kn, P.q_bits - P.d, trb, B);
    hipLaunchKernelGGL(dil_mu_kernel, dim3(hb), dim3(HT), 0, s3, trb, d_msgs, msg_stride, d_mlens, fixed_len, d->mu, B, 48u);
    HIPCHK(hipGetLastError());
    HIPCHK(hipEventRecord(d->ev_join2, s3));
    // caller's stream: the matrix.  It is used once per key, so it is never converted: the matrix product reads it as the expansion wrote it (natural order,
    // plain) and the Montgomery factor travels on NTT(z) instead (DilNtt::key_nat) - a pass over 30 KB per key less than the signer's form.
    if (sc_hip_mlwe_expand_matrix(d->ctx, batch, d_rho, 32, 32, k * l, P.q, P.q_bits, ab, s)) return SC_HIP_FAILURE;
    // chain (3): the signature side as in sc_hip_dilithium_verify, t1 . 2^d and its transform; chain (2) below
    const unsigned zblock = (unsigned)((l * P.n / 8 + 63) / 64 * 64);
    if (P.z_bits == 20) hipLaunchKernelGGL(dil_unpackz_kernel<20>, dim3(B), dim3(zblock), 0, s2, d_sigs, sig_stride, B, d->z, l * P.n, P.z_bits, P.q, (uint32_t)(P.gamma1 - P.beta), d_verdicts);
    else hipLaunchKernelGGL(dil_unpackz_kernel<0>, dim3(B), dim3(zblock), 0, s2, d_sigs, sig_stride, B, d->z, l * P.n, P.z_bits, P.q, (uint32_t)(P.gamma1 - P.beta), d_verdicts);
    hipLaunchKernelGGL(dil_unpack_tail_kernel, dim3((B + 3) / 4), dim3(256), 0, s2, d_sigs, sig_stride, d_siglens, B, d->h, d->c, d_verdicts, P);
    hipLaunchKernelGGL(dil_shift_kernel, dim3((unsigned)((batch * kn + 255) / 256)), dim3(256), 0, s2, d_t1, t1s, batch * kn, P.d);
    HIPCHK(hipGetLastError());
    DilNtt a = base_args(d);
    a.in = t1s; a.out_int = t1h;
    if (launch_dil<DOP_KEYPREP>(d, a, B * (uint32_t)k, 1, s2)) return SC_HIP_FAILURE;
    a = base_args(d);
    a.in = d->z; a.out_int = d->yh;
    if (launch_dil<DOP_KEYPREP>(d, a, B * (uint32_t)l, 1, s2)) return SC_HIP_FAILURE;      // NTT(z) R: the factor the plain matrix does not carry
    a = base_args(d);
    a.in = d->c; a.out_int = d->ch;
    if (launch_dil<DOP_FWD_INT>(d, a, B, 1, s2)) return SC_HIP_FAILURE;
    HIPCHK(hipEventRecord(d->ev_join, s2));
    HIPCHK(hipStreamWaitEvent(s, d->ev_join, 0));
    a = base_args(d);
    a.in = d->z; a.out_int = d->yh;
    a.key = reinterpret_cast<const uint32_t *>(ab); a.key_item_stride = (size_t)k * l * N; a.key_nat = 1;
    if (launch_dil<DOP_MATVEC>(d, a, B, k, s)) return SC_HIP_FAILURE;
    a.key_nat = 0;
    a.in = d->c; a.out_int = d->ch;
    a.key = t1h; a.key_item_stride = kn;
    if (launch_dil<DOP_CT1>(d, a, B, k, s)) return SC_HIP_FAILURE;
    HIPCHK(hipStreamWaitEvent(s, d->ev_join2, 0));
    // (two lanes per hash while the one-lane form would leave SIMDs without a second wavefront)
    if ((size_t)B <= (size_t)d->ctx->cus * 4 * 64) hipLaunchKernelGGL((dil_challenge_kernel<false, true>), dim3((2 * B + CH_T - 1) / CH_T), dim3(CH_T), 0, s, (const uint32_t *)nullptr, (const uint32_t *)nullptr, B, d->mu, d->w1, k * P.n, P.weight, (int8_t *)nullptr, d->c, d_verdicts, 0u);
    else hipLaunchKernelGGL((dil_challenge_kernel<false, false>), dim3((B + CH_T - 1) / CH_T), dim3(CH_T), 0, s, (const uint32_t *)nullptr, (const uint32_t *)nullptr, B, d->mu, d->w1, k * P.n, P.weight, (int8_t *)nullptr, d->c, d_verdicts, 0u);
    HIPCHK(hipGetLastError());
    return SC_HIP_SUCCESS;
}

extern "C" int sc_hip_shake_host(sc_hip_ctx_t *ctx, int bits, int standard, const uint8_t *h_in, size_t len, uint8_t *h_out, size_t outlen)
{
    if (!ctx || !h_out || (len && !h_in)) return fail("shake_host: null argument");
    if (bits != 128 && bits != 256) return fail("shake_host: bits must be 128 or 256");
    HIPCHK(hipSetDevice(ctx->device));
    DevBuf ib, ob;
    if (ib.get(len + 8) || ob.get(outlen + 8)) return SC_HIP_FAILURE;
    uint8_t *din = ib.as<uint8_t>(), *dout = ob.as<uint8_t>();
    if (len) HIPCHK(sc::copy_in(din, h_in, len));
    const int rate = 200 - 2 * bits / 8;
    if (standard) hipLaunchKernelGGL(shake_test_kernel<true>, dim3(1), dim3(HT), 0, nullptr, rate, din, (uint32_t)len, dout, (uint32_t)outlen);
    else hipLaunchKernelGGL(shake_test_kernel<false>, dim3(1), dim3(HT), 0, nullptr, rate, din, (uint32_t)len, dout, (uint32_t)outlen);
    int rc = (hipDeviceSynchronize() == hipSuccess) ? SC_HIP_SUCCESS : fail("shake_host: kernel failed");
    if (rc == SC_HIP_SUCCESS) HIPCHK(sc::copy_out(h_out, dout, outlen));
    return rc;
}


// replaces: safecrypto_keygen -> dilithium_keygen (dilithium.c:752-935) for `batch` key pairs.  Key i draws what the reference
// draws from sc->prng_ctx[0] - rho, K, the blocks of s1, the blocks of s2 - from a CTR-DRBG seeded with d_seeds[i]; its matrix
// comes from a second generator seeded with rho.  eta_bits = dilithium_set_t::eta_bits.  Outputs in the key pair's layouts:
// s1 [batch][l][n], s2, t, t1 [batch][k][n] int32; rho, K [batch][32]; tr [batch][48].
extern "C" int sc_hip_dilithium_keygen(sc_hip_dilithium_t *d, size_t batch, int eta_bits, const uint8_t *d_seeds, size_t seed_stride, size_t seed_len,
                                       int32_t *d_s1, int32_t *d_s2, int32_t *d_t, uint8_t *d_rho, uint8_t *d_K, uint8_t *d_tr, int32_t *d_t1, void *stream)
{
    if (!d) return fail("dilithium_keygen: null handle");
    if (batch == 0) return SC_HIP_SUCCESS;
    if (!d_seeds || !d_s1 || !d_s2 || !d_t || !d_rho || !d_K || !d_tr || !d_t1) return fail("dilithium_keygen: null buffer");
    if (eta_bits < 1 || eta_bits > 3) return fail("dilithium_keygen: eta_bits out of range");
    if (batch > 0x7fffffffu / 4096) return fail("dilithium_keygen: batch too large");
    HIPCHK(hipSetDevice(d->ctx->device));
    hipStream_t s = (hipStream_t)stream;
    const DilP &P = d->P;
    const uint32_t B = (uint32_t)batch;
    const int k = P.k, l = P.l;
    const size_t ln = (size_t)l * N, kn = (size_t)k * N;
    const size_t b1 = (ln + 1023) / 1024 * 512, b2 = (kn + 1023) / 1024 * 512, slen = 128 + b1 + b2, npoly = batch * (size_t)k * l;
    DevBuf sb, ab, am, yh, w;
    if (sb.get(batch * slen) || ab.get(npoly * N * 4) || am.get(npoly * N * 4) || yh.get(batch * ln * 4) || w.get(batch * kn * 4)) return SC_HIP_FAILURE;
    uint8_t *strm = sb.as<uint8_t>();
    if (sc_hip_drbg_mem(d->ctx, batch, d_seeds, seed_stride, seed_len, strm, slen, slen, s)) return SC_HIP_FAILURE;
    hipLaunchKernelGGL(dil_kg_slices_kernel, dim3((unsigned)((batch * 64 + 255) / 256)), dim3(256), 0, s, strm, slen, B, d_rho, d_K);
    const uint32_t mask = (1u << (eta_bits + 1)) - 1u;
    hipLaunchKernelGGL(dil_kg_small_kernel, dim3((unsigned)((batch * ln + 255) / 256)), dim3(256), 0, s, strm, slen, (size_t)128, B, (int)ln, P.eta, mask, d_s1);
    hipLaunchKernelGGL(dil_kg_small_kernel, dim3((unsigned)((batch * kn + 255) / 256)), dim3(256), 0, s, strm, slen, (size_t)128 + b1, B, (int)kn, P.eta, mask, d_s2);
    HIPCHK(hipGetLastError());
    if (sc_hip_mlwe_expand_matrix(d->ctx, batch, d_rho, 32, 32, k * l, P.q, P.q_bits, ab.as<int32_t>(), s)) return SC_HIP_FAILURE;
    hipLaunchKernelGGL(dil_to_internal_mont, dim3((unsigned)((npoly * N + 255) / 256)), dim3(256), 0, s, ab.as<int32_t>(), am.as<uint32_t>(), npoly, d->plan->k[1], d->plan->r2, d->plan->r2p, 0, 0);
    HIPCHK(hipGetLastError());
    DilNtt a = base_args(d);
    a.in = d_s1; a.out_int = yh.as<uint32_t>(); a.yh = yh.as<uint32_t>();
    if (launch_dil<DOP_FWD_INT>(d, a, B, l, s)) return SC_HIP_FAILURE;
    a.key = am.as<uint32_t>(); a.key_item_stride = (size_t)k * l * N; a.w = w.as<int32_t>(); a.w1 = nullptr;
    if (launch_dil<DOP_MATVEC>(d, a, B, k, s)) return SC_HIP_FAILURE;
    hipLaunchKernelGGL(dil_kg_t_kernel, dim3((unsigned)((batch * kn + 255) / 256)), dim3(256), 0, s, w.as<int32_t>(), d_s2, d_t, d_t1, batch * kn, P.q, P.d);
    hipLaunchKernelGGL(dil_crh_t1_kernel, dim3((B + HT - 1) / HT), dim3(HT), 0, s, d_rho, d_t1, (uint32_t)kn, P.q_bits - P.d, d_tr, B);
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(s));                                   // the scoped temporaries are released on return
    return SC_HIP_SUCCESS;
}
