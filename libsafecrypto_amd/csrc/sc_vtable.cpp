// sc_vtable.cpp - the reference's function-pointer tables (utils_arith_ntt_t, utils_arith_poly_t, utils_arith_vec_t, utils_sampling_t) served by the
// engine: layout-identical mirrors declared in include/sc_hip_vtable.h, every served slot a host -> device -> host round trip over the batched ABI
// with batch = 1.  Host code only; the kernels are the ones the batched entry points launch.
//
// The reference's slots carry no context argument, so the device is bound per host thread (sc_hip_vtable_bind_device, default device 0) and each
// device has its own context, plan cache and scratch buffers behind one mutex: threads bound to different GPUs do not meet, threads sharing a GPU
// take turns (the reference's own selector, utils_arith_ntt(), is a process-global write and not thread-safe at all, arith.c:360-396).
#include "../../include/sc_hip.h"
#include "../../include/sc_hip_vtable.h"

#include <hip/hip_runtime.h>
#include <atomic>
#include <map>
#include <mutex>
#include <string.h>
#include <utility>
#include <algorithm>
#include <vector>

#include "engine_internal.h"

namespace {

constexpr int MAX_DEV = 64;
std::atomic<unsigned long> g_errors{0};
thread_local int t_dev = 0;

struct PlanX {                                   // a plan plus what the composed slots need
    sc_hip_plan_t *plan = nullptr;
    int32_t *d_untwist = nullptr;                // g^-i, i < n: fft_32_* = forward transform of the un-bit-reversed, untwisted input
};

struct DevState {
    std::mutex mu;
    sc_hip_ctx_t *ctx = nullptr;
    std::map<std::pair<int32_t, int>, PlanX> plans;
    void *buf[4] = { nullptr, nullptr, nullptr, nullptr };
    size_t cap[4] = { 0, 0, 0, 0 };
};
DevState g_dev[MAX_DEV];

inline void note(int rc) { if (rc != SC_HIP_SUCCESS) g_errors++; }

// every helper below is called with the device's mutex held
bool ensure_ctx(DevState &s, int dev)
{
    if (s.ctx) return true;
    return sc_hip_ctx_create(dev, &s.ctx) == SC_HIP_SUCCESS;
}
void *scratch(DevState &s, int i, size_t bytes)
{
    if (bytes <= s.cap[i]) return s.buf[i];
    if (s.buf[i]) (void)sc::wipe_free(s.buf[i]);
    s.buf[i] = nullptr; s.cap[i] = 0;
    const size_t want = bytes < 65536 ? 65536 : bytes;
    if (hipMalloc(&s.buf[i], want) != hipSuccess) { sc::fail("vtable: device scratch allocation failed"); return nullptr; }
    s.cap[i] = want;
    return s.buf[i];
}
PlanX *plan_for(DevState &s, int dev, int32_t q, size_t n)
{
    if (!ensure_ctx(s, dev)) return nullptr;
    const auto key = std::make_pair(q, (int)n);
    auto it = s.plans.find(key);
    if (it != s.plans.end()) return &it->second;
    PlanX px;
    if (sc_hip_plan_create(s.ctx, q, (int)n, &px.plan) != SC_HIP_SUCCESS) return nullptr;
    // untwist table g^-i = -g^(n-i) (g^n = -1), from the plan's reference-format table w[i] = g^i
    std::vector<int32_t> w(n), r(n), u(n);
    if (sc_hip_plan_tables(px.plan, w.data(), r.data()) != SC_HIP_SUCCESS) { sc_hip_plan_destroy(px.plan); return nullptr; }
    u[0] = 1;
    for (size_t i = 1; i < n; i++) u[i] = q - w[n - i];
    if (hipMalloc((void **)&px.d_untwist, n * 4) != hipSuccess || sc::copy_in(px.d_untwist, u.data(), n * 4) != hipSuccess) {
        sc_hip_plan_destroy(px.plan);
        sc::fail("vtable: cannot upload the untwist table");
        return nullptr;
    }
    return &(s.plans[key] = px);
}

struct Bound {                                   // the calling thread's device, locked
    int dev; DevState &s; std::unique_lock<std::mutex> lk;
    Bound() : dev(t_dev), s(g_dev[t_dev]), lk(g_dev[t_dev].mu) { (void)hipSetDevice(dev); }
};

bool up(void *d, const void *h, size_t bytes) { return sc::copy_in(d, h, bytes) == hipSuccess || (sc::fail("vtable: H2D copy failed"), false); }
bool down(void *h, const void *d, size_t bytes) { return sc::copy_out(h, d, bytes) == hipSuccess || (sc::fail("vtable: D2H copy failed"), false); }

// ---- utils_arith_ntt_t slots -----------------------------------------------------------------------------------------------------------------

enum { T_FWD, T_INV };
void transform(int which, int32_t *v, const sc_hip_ntt_params_t *p, const int32_t *t)
{
    if (!p || !v || !t) { note(sc::fail("vtable: null argument")); return; }
    Bound b;
    PlanX *px = plan_for(b.s, b.dev, p->u.ntt32.q, p->n);
    const size_t bytes = p->n * 4;
    int32_t *d = px ? (int32_t *)scratch(b.s, 0, bytes) : nullptr;
    if (!px || !d || !up(d, t, bytes)) { note(SC_HIP_FAILURE); return; }
    const int rc = which == T_FWD ? sc_hip_ntt_fwd(px->plan, 1, d, d, nullptr) : sc_hip_ntt_inv(px->plan, 1, d, d, nullptr);
    note(rc == SC_HIP_SUCCESS && down(v, d, bytes) ? SC_HIP_SUCCESS : SC_HIP_FAILURE);
}
void v_fwd32(int32_t *v, const sc_hip_ntt_params_t *p, const int32_t *t, const int32_t *) { transform(T_FWD, v, p, t); }
void v_inv32(int32_t *v, const sc_hip_ntt_params_t *p, const int32_t *t, const int32_t *, const int32_t *) { transform(T_INV, v, p, t); }
void v_fwd16(int32_t *v, const sc_hip_ntt_params_t *p, const int32_t *t, const int16_t *) { transform(T_FWD, v, p, t); }
void v_inv16(int32_t *v, const sc_hip_ntt_params_t *p, const int32_t *t, const int16_t *, const int16_t *) { transform(T_INV, v, p, t); }

// fft_32_* (ntt_template.c.in:1144-1539): radix-2 DIT of a bit-reversed input with the powers of g^2, X[k] = sum_m x[m] g^(2mk), natural order out,
// in place.  With t[m] = x[m] g^-m that is the forward transform sum_m t[m] g^(m(2k+1)): un-bit-reverse (marshalling, on the host while copying),
// multiply by the untwist table and transform on the device.
void fft_common(int32_t *v, const sc_hip_ntt_params_t *p)
{
    if (!p || !v) { note(sc::fail("vtable: null argument")); return; }
    Bound b;
    PlanX *px = plan_for(b.s, b.dev, p->u.ntt32.q, p->n);
    const size_t n = p->n, bytes = n * 4;
    int32_t *d = px ? (int32_t *)scratch(b.s, 0, bytes) : nullptr;
    if (!px || !d) { note(SC_HIP_FAILURE); return; }
    int bits = 0; while (((size_t)1 << bits) < n) bits++;
    std::vector<int32_t> x(n);
    for (size_t j = 0; j < n; j++) {
        size_t rj = 0;
        for (int k = 0; k < bits; k++) rj |= ((j >> k) & 1) << (bits - 1 - k);
        x[rj] = v[j];
    }
    int rc = up(d, x.data(), bytes) ? SC_HIP_SUCCESS : SC_HIP_FAILURE;
    if (rc == SC_HIP_SUCCESS) rc = sc_hip_pointwise(px->plan, n, d, px->d_untwist, d, nullptr);
    if (rc == SC_HIP_SUCCESS) rc = sc_hip_ntt_fwd(px->plan, 1, d, d, nullptr);
    note(rc == SC_HIP_SUCCESS && down(v, d, bytes) ? SC_HIP_SUCCESS : SC_HIP_FAILURE);
}
void v_fft32(int32_t *v, const sc_hip_ntt_params_t *p, const int32_t *) { fft_common(v, p); }
void v_fft16(int32_t *v, const sc_hip_ntt_params_t *p, const int16_t *) { fft_common(v, p); }

void pointwise(int32_t *v, const sc_hip_ntt_params_t *p, const int32_t *t, const int32_t *u)
{
    if (!p || !v || !t || !u) { note(sc::fail("vtable: null argument")); return; }
    Bound b;
    PlanX *px = plan_for(b.s, b.dev, p->u.ntt32.q, p->n);
    const size_t bytes = p->n * 4;
    int32_t *d0 = px ? (int32_t *)scratch(b.s, 0, bytes) : nullptr, *d1 = px ? (int32_t *)scratch(b.s, 1, bytes) : nullptr;
    if (!px || !d0 || !d1 || !up(d0, t, bytes) || !up(d1, u, bytes)) { note(SC_HIP_FAILURE); return; }
    const int rc = sc_hip_pointwise(px->plan, p->n, d0, d1, d0, nullptr);
    note(rc == SC_HIP_SUCCESS && down(v, d0, bytes) ? SC_HIP_SUCCESS : SC_HIP_FAILURE);
}
void v_pw32(int32_t *v, const sc_hip_ntt_params_t *p, const int32_t *t, const int32_t *u) { pointwise(v, p, t, u); }
void v_pw16(int32_t *v, const sc_hip_ntt_params_t *p, const int32_t *t, const int16_t *u)
{
    if (!p || !u) { note(sc::fail("vtable: null argument")); return; }
    std::vector<int32_t> u32(u, u + p->n);                           // widening is marshalling: the product runs on the device
    pointwise(v, p, t, u32.data());
}
// ntt32_mult_scalar_generic (ntt.c:425-468): v[i] = t[i] * c mod q, made non-negative
void v_mulscalar(int32_t *v, const sc_hip_ntt_params_t *p, const int32_t *t, int32_t c)
{
    if (!p) { note(sc::fail("vtable: null argument")); return; }
    std::vector<int32_t> cc(p->n, c);
    pointwise(v, p, t, cc.data());
}

enum { E_NORM, E_CENT };
void elementwise(int which, int32_t *v, size_t n, const sc_hip_ntt_params_t *p)
{
    if (!p || !v) { note(sc::fail("vtable: null argument")); return; }
    if (n == 0) return;
    Bound b;
    PlanX *px = plan_for(b.s, b.dev, p->u.ntt32.q, p->n);
    const size_t bytes = n * 4;
    int32_t *d = px ? (int32_t *)scratch(b.s, 0, bytes) : nullptr;
    if (!px || !d || !up(d, v, bytes)) { note(SC_HIP_FAILURE); return; }
    const int rc = which == E_NORM ? sc_hip_normalize(px->plan, n, d, d, nullptr) : sc_hip_center(px->plan, n, d, d, nullptr);
    note(rc == SC_HIP_SUCCESS && down(v, d, bytes) ? SC_HIP_SUCCESS : SC_HIP_FAILURE);
}
void v_norm(int32_t *v, size_t n, const sc_hip_ntt_params_t *p) { elementwise(E_NORM, v, n, p); }
void v_cent(int32_t *v, size_t n, const sc_hip_ntt_params_t *p) { elementwise(E_CENT, v, n, p); }

void v_flip(int32_t *v, const sc_hip_ntt_params_t *p)
{
    if (!p || !v) { note(sc::fail("vtable: null argument")); return; }
    Bound b;
    PlanX *px = plan_for(b.s, b.dev, p->u.ntt32.q, p->n);
    const size_t bytes = p->n * 4;
    int32_t *d0 = px ? (int32_t *)scratch(b.s, 0, bytes) : nullptr, *d1 = px ? (int32_t *)scratch(b.s, 1, bytes) : nullptr;
    if (!px || !d0 || !d1 || !up(d0, v, bytes)) { note(SC_HIP_FAILURE); return; }
    const int rc = sc_hip_ntt_flip(px->plan, 1, d0, d1, nullptr);
    note(rc == SC_HIP_SUCCESS && down(v, d1, bytes) ? SC_HIP_SUCCESS : SC_HIP_FAILURE);
}

void sparse(int32_t *v, size_t n, uint16_t omega, const void *t, int is16, const int32_t *c)
{
    Bound b;
    if (!ensure_ctx(b.s, b.dev)) { note(SC_HIP_FAILURE); return; }
    note(sc_hip_mult_sparse_host(b.s.ctx, 1, (int)n, omega, t, is16, c, v));
}
void v_sparse32(int32_t *v, size_t n, uint16_t omega, const int32_t *t, const int32_t *c) { sparse(v, n, omega, t, 0, c); }
void v_sparse16(int32_t *v, size_t n, uint16_t omega, const int16_t *t, const int32_t *c) { sparse(v, n, omega, t, 1, c); }

// invert_32 / div_32 run over n coefficients that need not be the ring degree (the reference passes the length separately)
int32_t v_invert(int32_t *v, const sc_hip_ntt_params_t *p, size_t n)
{
    if (!p || !v) { note(sc::fail("vtable: null argument")); return 1; }
    Bound b;
    PlanX *px = plan_for(b.s, b.dev, p->u.ntt32.q, p->n);
    int32_t rc = 1;
    const int e = px ? sc_hip_ntt_invert_host(px->plan, 1, n, v, &rc) : SC_HIP_FAILURE;
    note(e);
    return e == SC_HIP_SUCCESS ? rc : 1;
}
int32_t v_div(int32_t *num, const int32_t *den, const sc_hip_ntt_params_t *p, size_t n)
{
    if (!p || !num || !den) { note(sc::fail("vtable: null argument")); return 1; }
    Bound b;
    PlanX *px = plan_for(b.s, b.dev, p->u.ntt32.q, p->n);
    int32_t rc = 1;
    const int e = px ? sc_hip_ntt_div_host(px->plan, 1, n, num, den, &rc) : SC_HIP_FAILURE;
    note(e);
    return e == SC_HIP_SUCCESS ? rc : 1;
}

// ---- utils_arith_poly_t / utils_arith_vec_t slots ------------------------------------------------------------------------------------------------

enum { P_ADD, P_SUB, P_MULS, P_NEG };
void poly_op(int which, int32_t *out, size_t n, const int32_t *a, const int32_t *b2, int32_t s)
{
    if (n == 0) return;
    if (!out || !a || ((which == P_ADD || which == P_SUB) && !b2)) { note(sc::fail("vtable: null argument")); return; }
    Bound b;
    const size_t bytes = n * 4;
    if (!ensure_ctx(b.s, b.dev)) { note(SC_HIP_FAILURE); return; }
    int32_t *d0 = (int32_t *)scratch(b.s, 0, bytes), *d1 = (int32_t *)scratch(b.s, 1, bytes);
    if (!d0 || !d1 || !up(d0, a, bytes) || (b2 && !up(d1, b2, bytes))) { note(SC_HIP_FAILURE); return; }
    int rc;
    switch (which) {
    case P_ADD: rc = sc_hip_poly_add(b.s.ctx, n, d0, d1, d0, nullptr); break;
    case P_SUB: rc = sc_hip_poly_sub(b.s.ctx, n, d0, d1, d0, nullptr); break;
    case P_MULS: rc = sc_hip_poly_mul_scalar(b.s.ctx, n, d0, s, d0, nullptr); break;
    default: rc = sc_hip_poly_mod_negate(b.s.ctx, n, s, d0, d0, nullptr); break;
    }
    note(rc == SC_HIP_SUCCESS && down(out, d0, bytes) ? SC_HIP_SUCCESS : SC_HIP_FAILURE);
}
void pl_mul_scalar(int32_t *poly, size_t n, int32_t in) { poly_op(P_MULS, poly, n, poly, nullptr, in); }
void pl_add(int32_t *out, size_t n, const int32_t *in1, const int32_t *in2) { poly_op(P_ADD, out, n, in1, in2, 0); }
void pl_sub(int32_t *out, size_t n, const int32_t *in1, const int32_t *in2) { poly_op(P_SUB, out, n, in1, in2, 0); }
void pl_add_single(int32_t *out, size_t n, const int32_t *in) { poly_op(P_ADD, out, n, out, in, 0); }
void pl_sub_single(int32_t *out, size_t n, const int32_t *in) { poly_op(P_SUB, out, n, out, in, 0); }
void pl_mod_negate(int32_t *out, size_t n, int32_t q, const int32_t *in) { poly_op(P_NEG, out, n, in, nullptr, q); }

enum { R_CMP, R_ABSMAX, R_SCALAR };
int32_t reduce_op(int which, const int32_t *a, const int32_t *b2, size_t n)
{
    if (n == 0) return 0;
    Bound b;
    const size_t bytes = n * 4;
    if (!ensure_ctx(b.s, b.dev)) { note(SC_HIP_FAILURE); return 0; }
    int32_t *d0 = (int32_t *)scratch(b.s, 0, bytes), *d1 = (int32_t *)scratch(b.s, 1, bytes), *dr = (int32_t *)scratch(b.s, 2, 64);
    if (!d0 || !d1 || !dr || !up(d0, a, bytes) || (b2 && !up(d1, b2, bytes))) { note(SC_HIP_FAILURE); return 0; }
    int rc;
    switch (which) {
    case R_CMP: rc = sc_hip_poly_cmp_not_equal(b.s.ctx, 1, n, d0, d1, dr, nullptr); break;
    case R_ABSMAX: rc = sc_hip_vec_absmax(b.s.ctx, 1, n, d0, dr, nullptr); break;
    default: rc = sc_hip_vec_scalar(b.s.ctx, 1, n, d0, d1, dr, nullptr); break;
    }
    int32_t r = 0;
    note(rc == SC_HIP_SUCCESS && down(&r, dr, 4) ? SC_HIP_SUCCESS : SC_HIP_FAILURE);
    return r;
}
int32_t pl_cmp_not_equal(volatile const int32_t *in1, volatile const int32_t *in2, size_t n)
{ return reduce_op(R_CMP, const_cast<const int32_t *>(in1), const_cast<const int32_t *>(in2), n); }
int32_t vc_absmax_32(const int32_t *v, size_t n) { return v ? reduce_op(R_ABSMAX, v, nullptr, n) : 0; }                    // vectors.c:30-32: NULL -> 0
int32_t vc_scalar_32(const int32_t *t, const int32_t *u, size_t n) { return (t && u) ? reduce_op(R_SCALAR, t, u, n) : 0; }
int32_t vc_absmax_16(const int16_t *v, size_t n)
{
    if (!v) return 0;
    std::vector<int32_t> w(v, v + n);
    return reduce_op(R_ABSMAX, w.data(), nullptr, n);
}
int32_t vc_scalar_16(const int16_t *t, const int16_t *u, size_t n)
{
    if (!t || !u) return 0;
    std::vector<int32_t> a(t, t + n), c(u, u + n);
    return reduce_op(R_SCALAR, a.data(), c.data(), n);
}

const sc_hip_ntt32_vtable_t g_vt = {
    v_pw32, v_pw16, v_cent, v_norm,
    v_fwd32, v_inv32, v_fwd32, v_inv32,
    v_fwd16, v_inv16, v_fwd16, v_inv16,
    v_sparse32, v_sparse16, v_invert, v_div,
};

}  // namespace

// ---- sampler -------------------------------------------------------------------------------------------------------------------------------------

struct sc_hip_sampler {
    int dev;
    sc_hip_cdt_t *cdt;
    sc_hip_rng64_fn rng; void *rng_ctx;
    sc_hip_rng32_fn rng32 = nullptr;
    float tail, sigma;
    std::vector<uint64_t> words;
    int hp_words = 0;             // 2 / 3: a 128- / 192-bit sampler (sc_hip_sampler_create_hp)
};

extern "C" int sc_hip_sampler_create(float tail, float sigma, const uint64_t *h_cdf, int size, sc_hip_rng64_fn rng, void *rng_ctx, sc_hip_sampler_t **out)
{
    if (!rng || !out) return sc::fail("sampler_create: null argument");
    Bound b;
    if (!ensure_ctx(b.s, b.dev)) return SC_HIP_FAILURE;
    sc_hip_cdt_t *cdt = nullptr;
    const int rc = h_cdf ? sc_hip_cdt_create_from_table(b.s.ctx, h_cdf, size, &cdt) : sc_hip_cdt_create(b.s.ctx, tail, sigma, &cdt);
    if (rc != SC_HIP_SUCCESS) return SC_HIP_FAILURE;
    sc_hip_sampler *s = new sc_hip_sampler;
    s->dev = b.dev; s->cdt = cdt; s->rng = rng; s->rng_ctx = rng_ctx; s->tail = tail; s->sigma = sigma;
    *out = s;
    return SC_HIP_SUCCESS;
}
// the same object at 128- / 192-bit precision (create_sampler(..., SAMPLING_128BIT / SAMPLING_192BIT, ...)): the caller's table ([size][precision / 64] limbs), precision / 64
// draws of the generator per sample (gaussian_cdf.c:482-483,498-500)
extern "C" int sc_hip_sampler_create_hp(int precision_bits, float tail, float sigma, const uint64_t *h_cdf, int size, sc_hip_rng64_fn rng, void *rng_ctx, sc_hip_sampler_t **out)
{
    if (!rng || !out || !h_cdf) return sc::fail("sampler_create_hp: null argument (the table of a 128- / 192-bit sampler is the caller's: it is MPFR arithmetic)");
    Bound b;
    if (!ensure_ctx(b.s, b.dev)) return SC_HIP_FAILURE;
    sc_hip_cdt_t *cdt = nullptr;
    if (sc_hip_cdt_hp_create_from_table(b.s.ctx, precision_bits, h_cdf, size, &cdt) != SC_HIP_SUCCESS) return SC_HIP_FAILURE;
    sc_hip_sampler *s = new sc_hip_sampler;
    s->dev = b.dev; s->cdt = cdt; s->rng = rng; s->rng_ctx = rng_ctx; s->tail = tail; s->sigma = sigma; s->hp_words = precision_bits / 64;
    *out = s;
    return SC_HIP_SUCCESS;
}
extern "C" void sc_hip_sampler_set_rng(sc_hip_sampler_t *s, sc_hip_rng64_fn rng, void *rng_ctx) { if (s && rng) { s->rng = rng; s->rng_ctx = rng_ctx; } }
extern "C" void sc_hip_sampler_set_rng32(sc_hip_sampler_t *s, sc_hip_rng32_fn rng32) { if (s) s->rng32 = rng32; }
extern "C" void sc_hip_sampler_destroy(sc_hip_sampler_t *s)
{
    if (!s) return;
    { std::lock_guard<std::mutex> lk(g_dev[s->dev].mu); sc_hip_cdt_destroy(s->cdt); }
    delete s;
}

namespace {
// v[i] = CDT(word_i) + centre with word_i = the i-th prng_64 drawn now (sample_vector_32, sampling.c:211-228 with discard = 0)
int sampler_vector(sc_hip_sampler *s, int32_t *v, size_t n, int32_t centre)
{
    if (!s || !v) return sc::fail("sampler: null argument");
    if (n == 0) return SC_HIP_SUCCESS;
    const size_t per = s->hp_words ? (size_t)s->hp_words : 1;
    s->words.resize(n * per);
    for (size_t i = 0; i < n * per; i++) s->words[i] = s->rng(s->rng_ctx);
    std::lock_guard<std::mutex> lk(g_dev[s->dev].mu);
    return s->hp_words ? sc_hip_cdt_hp_sample_host(s->cdt, n, s->words.data(), centre, v) : sc_hip_cdt_sample_host(s->cdt, n, s->words.data(), centre, v);
}
int32_t sm_destroy(void **gauss) { if (gauss && *gauss) { sc_hip_sampler_destroy((sc_hip_sampler *)*gauss); *gauss = nullptr; } return 0; }
int32_t sm_sample(void *gauss)
{
    int32_t v = 0;
    note(sampler_vector((sc_hip_sampler *)gauss, &v, 1, 0));
    return v;
}
// The side-channel fillers (sampling.c:107-228).  Which word goes where is decided by the words themselves - rand_range's rejection, discard_sample's
// repeats - so the host replays the reference's loop over the two callbacks (prng_32 / prng_64 of ONE generator) and only RECORDS: the 64-bit words of the
// samples in draw order and the moves of the shuffle.  The samples - every table search - are then made on the device in one launch, and the moves
// (v[i] = v[j]; v[j] = sample_k + centre) are applied to the vector in order.
int sampler_vector_sca(sc_hip_sampler *s, int32_t *v, size_t n, int32_t centre, uint32_t discard, int blinding)
{
    if (!s || !v) return sc::fail("sampler: null argument");
    if (n == 0) return SC_HIP_SUCCESS;
    if (s->hp_words) return sc::fail("sampler: shuffle / blinding / sample discarding are served at 64-bit precision only");
    if (discard != 0 && discard != 2 && discard != 4 && discard != 6) return sc::fail("sampler: discard must be 0 or SCA_PATTERN_SAMPLE_DISCARD_LO / MD / HI (2 / 4 / 6)");
    if ((discard != 0 || blinding != 0) && !s->rng32)
        return sc::fail("sampler: shuffle / blinding / sample discarding draw prng_32 words between the samples: give the sampler the generator's prng_32 (sc_hip_sampler_set_rng32)");
    const uint32_t thresh = discard == 2 ? 1u << 28 : discard == 4 ? 1u << 30 : discard == 6 ? 1u << 31 : 0;      // set_threshold, sampling.c:85-92
    auto discarded = [&]() -> size_t { return thresh ? (s->rng32(s->rng_ctx) < thresh ? 1 : 0) : 0; };             // discard_sample, :95-105
    struct Move { uint32_t i, j; };                                    // j == ~0u: v[i] = sample (+ centre unless i is the shuffle's first element)
    std::vector<Move> moves;
    std::vector<uint64_t> &w = s->words;
    w.clear();
    if (blinding == 0) {                                               // sample_vector_32, :211-228
        for (size_t i = 0; i < n; i++) {
            w.push_back(s->rng(s->rng_ctx)); moves.push_back({ (uint32_t)i, ~0u });
            i -= discarded();
        }
    } else {                                                           // shuffle_sample_vector_32, :127-145
        w.push_back(s->rng(s->rng_ctx));
        for (size_t i = 1; i < n; i++) {
            const size_t rem = 0xFFFFFFFFu % i;                        // rand_range(ctx, i), :68-83: y % i over the accepted range
            size_t y;
            do y = s->rng32(s->rng_ctx); while (y >= 0xFFFFFFFFu - rem);
            w.push_back(s->rng(s->rng_ctx)); moves.push_back({ (uint32_t)i, (uint32_t)(y % i) });
            i -= discarded();
        }
        if (blinding == 1) for (size_t i = 0; i < n; i++) w.push_back(s->rng(s->rng_ctx));       // blinding_sample_vector_32, :168-191
    }
    std::vector<int32_t> smp(w.size());
    {
        std::lock_guard<std::mutex> lk(g_dev[s->dev].mu);
        if (sc_hip_cdt_sample_host(s->cdt, w.size(), w.data(), 0, smp.data()) != SC_HIP_SUCCESS) return SC_HIP_FAILURE;
    }
    size_t k = 0;
    if (blinding == 0) for (const Move &m : moves) v[m.i] = smp[k++] + centre;
    else {
        v[0] = smp[k++];                                               // (no centre on the first element: sampling.c:134)
        for (const Move &m : moves) { if (m.i != m.j) v[m.i] = v[m.j]; v[m.j] = smp[k++] + centre; }
        if (blinding == 1) for (size_t i = 0; i < n; i++) v[i] -= smp[k++];                      // (the closing swap loop exchanges v[i] with itself)
    }
    std::fill(smp.begin(), smp.end(), 0); std::fill(w.begin(), w.end(), 0);
    return SC_HIP_SUCCESS;
}
template <int BLINDING>
int32_t sm_vector_32_t(const sc_hip_utils_sampling_t *t, int32_t *v, size_t n, int32_t centre)
{
    if (!t) { note(sc::fail("sampler: null table")); return 1; }
    const int rc = (BLINDING == 0 && t->discard == 0) ? sampler_vector((sc_hip_sampler *)t->gauss, v, n, centre)
                                                      : sampler_vector_sca((sc_hip_sampler *)t->gauss, v, n, centre, t->discard, BLINDING);
    note(rc);
    return rc;
}
template <int BLINDING>
int32_t sm_vector_16_t(const sc_hip_utils_sampling_t *t, int16_t *v, size_t n, int32_t centre)
{
    // (the 16-bit fillers run the same loops on SINT16 storage, sampling.c:107-125,147-166,193-206: the moves copy values, the subtraction wraps the same way)
    std::vector<int32_t> w(n);
    const int32_t rc = sm_vector_32_t<BLINDING>(t, w.data(), n, centre);
    if (rc == 0) for (size_t i = 0; i < n; i++) v[i] = (int16_t)w[i];
    return rc;
}
int32_t sm_vector_32(const sc_hip_utils_sampling_t *t, int32_t *v, size_t n, int32_t centre) { return sm_vector_32_t<0>(t, v, n, centre); }
int32_t sm_vector_16(const sc_hip_utils_sampling_t *t, int16_t *v, size_t n, int32_t centre) { return sm_vector_16_t<0>(t, v, n, centre); }
}  // namespace

extern "C" int sc_hip_utils_sampling_fill(sc_hip_utils_sampling_t *dst, sc_hip_sampler_t *s)
{
    if (!dst || !s) return sc::fail("utils_sampling_fill: null argument");
    memset(dst, 0, sizeof *dst);
    dst->destroy = sm_destroy; dst->sample = sm_sample; dst->vector_16 = sm_vector_16; dst->vector_32 = sm_vector_32;
    dst->precision = s->hp_words ? 64 * s->hp_words : 64; dst->dimension = 0; dst->bootstrapped = 0;
    dst->tail = s->tail; dst->sigma = s->sigma; dst->sigma2 = 0.f;
    dst->gauss = s; dst->prng_ctx = s->rng_ctx; dst->discard = 0; dst->bootstrap = nullptr;
    return SC_HIP_SUCCESS;
}

extern "C" int sc_hip_utils_sampling_fill_sca(sc_hip_utils_sampling_t *dst, sc_hip_sampler_t *s, int blinding)
{
    if (blinding < 0 || blinding > 2) return sc::fail("utils_sampling_fill_sca: blinding must be 0 (NORMAL_SAMPLES), 1 (BLINDING_SAMPLES) or 2 (SHUFFLE_SAMPLES)");
    if (s && s->hp_words && blinding) return sc::fail("utils_sampling_fill_sca: the shuffle / blinding fillers are served at 64-bit precision only");
    if (sc_hip_utils_sampling_fill(dst, s) != SC_HIP_SUCCESS) return SC_HIP_FAILURE;
    if (blinding == 1) { dst->vector_32 = sm_vector_32_t<1>; dst->vector_16 = sm_vector_16_t<1>; }
    else if (blinding == 2) { dst->vector_32 = sm_vector_32_t<2>; dst->vector_16 = sm_vector_16_t<2>; }
    return SC_HIP_SUCCESS;
}

// ---- table builders ------------------------------------------------------------------------------------------------------------------------------

static_assert(sizeof(sc_hip_utils_arith_ntt_t) == 77 * sizeof(void *), "utils_arith_ntt_t has 77 slots (ntt.h:217-298)");
static_assert(sizeof(sc_hip_utils_arith_poly_t) == 35 * sizeof(void *), "utils_arith_poly_t has 35 slots (arith.h:74-111)");
static_assert(sizeof(sc_hip_utils_arith_vec_t) == 4 * sizeof(void *), "utils_arith_vec_t has 4 slots (arith.h:123-128)");
static_assert(sizeof(sc_hip_utils_sampling_t) == 6 * sizeof(void *) + 6 * 4 + 2 * sizeof(void *) + 4 + sizeof(void *), "utils_sampling_t layout (sampling.h:68-86)");

extern "C" int sc_hip_utils_arith_ntt_fill(sc_hip_utils_arith_ntt_t *dst, const void *base)
{
    if (!dst) return sc::fail("utils_arith_ntt_fill: null destination");
    if (sc_hip_device_count() < 1) return sc::fail("utils_arith_ntt_fill: no HIP device (the engine has no CPU fallback)");
    if (base) memcpy(dst, base, sizeof *dst); else memset(dst, 0, sizeof *dst);
    dst->mul_32_sparse = v_sparse32;        dst->mul_32_sparse_16 = v_sparse16;
    dst->mul_32_pointwise = v_pw32;         dst->mul_32_pointwise_16 = v_pw16;
    dst->mul_32_scalar = v_mulscalar;
    dst->fft_32_32 = v_fft32;               dst->fft_32_32_large = v_fft32;
    dst->fft_32_16 = v_fft16;               dst->fft_32_16_large = v_fft16;
    dst->invert_32 = v_invert;              dst->div_32 = v_div;
    dst->flip_32 = v_flip;                  dst->center_32 = v_cent;               dst->normalize_32 = v_norm;
    dst->fwd_ntt_32_32 = v_fwd32;           dst->inv_ntt_32_32 = v_inv32;
    dst->fwd_ntt_32_32_large = v_fwd32;     dst->inv_ntt_32_32_large = v_inv32;
    dst->fwd_ntt_32_16 = v_fwd16;           dst->inv_ntt_32_16 = v_inv16;
    dst->fwd_ntt_32_16_large = v_fwd16;     dst->inv_ntt_32_16_large = v_inv16;
    return SC_HIP_SUCCESS;
}

extern "C" int sc_hip_utils_arith_poly_fill(sc_hip_utils_arith_poly_t *dst, const void *base)
{
    if (!dst) return sc::fail("utils_arith_poly_fill: null destination");
    if (sc_hip_device_count() < 1) return sc::fail("utils_arith_poly_fill: no HIP device (the engine has no CPU fallback)");
    if (base) memcpy(dst, base, sizeof *dst); else memset(dst, 0, sizeof *dst);
    dst->mul_32_scalar = pl_mul_scalar;
    dst->add_32 = pl_add;                   dst->sub_32 = pl_sub;
    dst->add_single_32 = pl_add_single;     dst->sub_single_32 = pl_sub_single;
    dst->mod_negate_32 = pl_mod_negate;     dst->cmp_not_equal_32 = pl_cmp_not_equal;
    return SC_HIP_SUCCESS;
}

extern "C" int sc_hip_utils_arith_vec_fill(sc_hip_utils_arith_vec_t *dst)
{
    if (!dst) return sc::fail("utils_arith_vec_fill: null destination");
    if (sc_hip_device_count() < 1) return sc::fail("utils_arith_vec_fill: no HIP device (the engine has no CPU fallback)");
    dst->absmax_32 = vc_absmax_32; dst->scalar_32 = vc_scalar_32; dst->absmax_16 = vc_absmax_16; dst->scalar_16 = vc_scalar_16;
    return SC_HIP_SUCCESS;
}

extern "C" int sc_hip_vtable_bind_device(int device)
{
    if (device < 0 || device >= MAX_DEV || device >= sc_hip_device_count()) return sc::fail("vtable_bind_device: no such HIP device");
    t_dev = device;
    return SC_HIP_SUCCESS;
}

extern "C" const sc_hip_ntt32_vtable_t *sc_hip_utils_arith_ntt32(void)
{
    if (sc_hip_device_count() < 1) return nullptr;
    return &g_vt;
}

extern "C" unsigned long sc_hip_vtable_errors(void) { return g_errors.load(); }
