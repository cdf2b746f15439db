// multi.cpp - the in-library multi-device driver: every polynomial / signature / verification is an independent unit, so a host
// batch is cut into contiguous slices, one per device, each slice is processed by ONE host thread bound to its device (the per-device entry points
// it calls keep >= 2 streams busy: upload of chunk c+1, kernel of chunk c, download of chunk c-1), and the results land in the caller's buffers at
// the slice's offset - the "concatenation" is the address arithmetic.  No collective, no peer traffic; constants (twiddles, keys) are replicated
// into each device's own context at creation.  A device may be listed more than once (two workers on one GPU): that is how a one-GPU box tests it.
#include "engine_internal.h"

#include <memory>
#include <thread>
#include <vector>

struct sc_hip_multi {
    std::vector<int> dev;
    std::vector<sc_hip_ctx_t *> ctx;
};

extern "C" void sc_hip_shard_range(size_t total, int parts, int index, size_t *begin, size_t *count)
{
    size_t b = 0, c = 0;
    if (parts >= 1 && index >= 0 && index < parts) {
        const size_t base = total / (size_t)parts, rem = total % (size_t)parts, i = (size_t)index;
        b = i * base + (i < rem ? i : rem);
        c = base + (i < rem ? 1 : 0);
    }
    if (begin) *begin = b;
    if (count) *count = c;
}

extern "C" int sc_hip_multi_create(const int *devices, int ndev, sc_hip_multi_t **out)
{
    if (!out || ndev < 1 || ndev > 64) return sc::fail("multi_create: 1..64 workers");
    std::unique_ptr<sc_hip_multi> m(new sc_hip_multi);
    for (int i = 0; i < ndev; i++) {
        const int d = devices ? devices[i] : i;
        sc_hip_ctx_t *c = nullptr;
        if (sc_hip_ctx_create(d, &c) != SC_HIP_SUCCESS) { for (auto *x : m->ctx) sc_hip_ctx_destroy(x); return SC_HIP_FAILURE; }
        m->dev.push_back(d); m->ctx.push_back(c);
    }
    *out = m.release();
    return SC_HIP_SUCCESS;
}

extern "C" void sc_hip_multi_destroy(sc_hip_multi_t *m)
{
    if (!m) return;
    for (auto *c : m->ctx) sc_hip_ctx_destroy(c);
    delete m;
}

extern "C" int sc_hip_multi_workers(const sc_hip_multi_t *m) { return m ? (int)m->dev.size() : 0; }
extern "C" sc_hip_ctx_t *sc_hip_multi_ctx(const sc_hip_multi_t *m, int worker) { return (m && worker >= 0 && worker < (int)m->ctx.size()) ? m->ctx[(size_t)worker] : nullptr; }

namespace {
// The caller's buffers are pinned ONCE, on the calling thread, for all workers: sc::HostPins registers the whole pages inside them (portable: usable from
// every device's context; overlapping in / out buffers once) and the workers' per-device host entries copy through those registrations
// (sc::t_driver_pins).  What cannot be registered moves through the runtime's staged path.

// runs job(worker, begin, count) on one host thread per worker; an error message of a worker thread is carried back to the caller's thread.  Nothing a
// worker throws (std::bad_alloc from a temporary, ...) leaves its thread, and threads already started are joined if starting another one fails.
template <class F>
int fan_out(const sc_hip_multi *m, size_t total, F job, const sc::HostPins *pins = nullptr)
{
    const int W = (int)m->dev.size();
    std::vector<int> rc((size_t)W, SC_HIP_SUCCESS);
    std::vector<std::string> err((size_t)W);
    std::vector<std::thread> th;
    bool spawn_failed = false;
    try {
        th.reserve((size_t)W);
        for (int w = 0; w < W; w++) {
            size_t b, c;
            sc_hip_shard_range(total, W, w, &b, &c);
            th.emplace_back([&, w, b, c]() {
                try {
                    if (c == 0) return;
                    if (hipSetDevice(m->dev[(size_t)w]) != hipSuccess) { rc[(size_t)w] = SC_HIP_FAILURE; err[(size_t)w] = "multi: hipSetDevice failed"; return; }
                    sc::t_skip_host_pin = true;                       // slices are never registered by the workers (engine_internal.h)
                    sc::t_driver_pins = pins;
                    rc[(size_t)w] = job(w, b, c);
                    if (rc[(size_t)w] != SC_HIP_SUCCESS) err[(size_t)w] = sc_hip_last_error();
                } catch (const std::exception &e) {
                    rc[(size_t)w] = SC_HIP_FAILURE; err[(size_t)w] = std::string("exception in worker: ") + e.what();
                } catch (...) {
                    rc[(size_t)w] = SC_HIP_FAILURE; err[(size_t)w] = "exception in worker";
                }
            });
        }
    } catch (...) {
        spawn_failed = true;
    }
    for (auto &t : th) if (t.joinable()) t.join();
    if (spawn_failed) return sc::fail("multi: could not start a worker thread");
    for (int w = 0; w < W; w++) if (rc[(size_t)w] != SC_HIP_SUCCESS) return sc::fail("multi (worker " + std::to_string(w) + ", device " + std::to_string(m->dev[(size_t)w]) + "): " + err[(size_t)w]);
    return SC_HIP_SUCCESS;
}
}  // namespace

// ---- the NTT family ---------------------------------------------------------------------------------------------------------------------------

struct sc_hip_multi_plan {
    sc_hip_multi *m;
    std::vector<sc_hip_plan_t *> plan;
};

extern "C" int sc_hip_multi_plan_create(sc_hip_multi_t *m, int32_t q, int n, sc_hip_multi_plan_t **out)
{
    if (!m || !out) return sc::fail("multi_plan_create: null argument");
    std::unique_ptr<sc_hip_multi_plan> p(new sc_hip_multi_plan);
    p->m = m;
    for (auto *c : m->ctx) {
        sc_hip_plan_t *pl = nullptr;
        if (sc_hip_plan_create(c, q, n, &pl) != SC_HIP_SUCCESS) { for (auto *x : p->plan) sc_hip_plan_destroy(x); return SC_HIP_FAILURE; }
        p->plan.push_back(pl);
    }
    *out = p.release();
    return SC_HIP_SUCCESS;
}
extern "C" void sc_hip_multi_plan_destroy(sc_hip_multi_plan_t *p) { if (!p) return; for (auto *x : p->plan) sc_hip_plan_destroy(x); delete p; }

extern "C" int sc_hip_multi_ntt_fwd_host(sc_hip_multi_plan_t *p, size_t batch, const int32_t *h_in, int32_t *h_out)
{
    if (!p || !h_in || !h_out) return sc::fail("multi_ntt_fwd_host: null argument");
    const size_t n = (size_t)sc_hip_plan_n(p->plan[0]);
    sc::HostPins pins; pins.add(h_in, batch * n * 4); pins.add(h_out, batch * n * 4); pins.commit(hipHostRegisterPortable);
    return fan_out(p->m, batch, [&](int w, size_t b, size_t c) { return sc_hip_ntt_fwd_host(p->plan[(size_t)w], c, h_in + b * n, h_out + b * n); }, &pins);
}
extern "C" int sc_hip_multi_ntt_inv_host(sc_hip_multi_plan_t *p, size_t batch, const int32_t *h_in, int32_t *h_out)
{
    if (!p || !h_in || !h_out) return sc::fail("multi_ntt_inv_host: null argument");
    const size_t n = (size_t)sc_hip_plan_n(p->plan[0]);
    sc::HostPins pins; pins.add(h_in, batch * n * 4); pins.add(h_out, batch * n * 4); pins.commit(hipHostRegisterPortable);
    return fan_out(p->m, batch, [&](int w, size_t b, size_t c) { return sc_hip_ntt_inv_host(p->plan[(size_t)w], c, h_in + b * n, h_out + b * n); }, &pins);
}
extern "C" int sc_hip_multi_polymul_ntt_host(sc_hip_multi_plan_t *p, size_t batch, const int32_t *h_a, const int32_t *h_bhat, int32_t *h_out)
{
    if (!p || !h_a || !h_bhat || !h_out) return sc::fail("multi_polymul_ntt_host: null argument");
    const size_t n = (size_t)sc_hip_plan_n(p->plan[0]);
    sc::HostPins pins; pins.add(h_a, batch * n * 4); pins.add(h_bhat, batch * n * 4); pins.add(h_out, batch * n * 4); pins.commit(hipHostRegisterPortable);
    return fan_out(p->m, batch, [&](int w, size_t b, size_t c) { return sc_hip_polymul_ntt_host(p->plan[(size_t)w], c, h_a + b * n, h_bhat + b * n, h_out + b * n); }, &pins);
}

// ---- Dilithium: one key pair replicated to every device, messages sharded -----------------------------------------------------------------------

struct sc_hip_multi_dilithium {
    sc_hip_multi *m;
    std::vector<sc_hip_dilithium_t *> h;
};

extern "C" int sc_hip_multi_dilithium_create(sc_hip_multi_t *m, const sc_hip_dilithium_params_t *params, sc_hip_multi_dilithium_t **out)
{
    if (!m || !params || !out) return sc::fail("multi_dilithium_create: null argument");
    std::unique_ptr<sc_hip_multi_dilithium> d(new sc_hip_multi_dilithium);
    d->m = m;
    for (auto *c : m->ctx) {
        sc_hip_dilithium_t *h = nullptr;
        if (sc_hip_dilithium_create(c, params, &h) != SC_HIP_SUCCESS) { for (auto *x : d->h) sc_hip_dilithium_destroy(x); return SC_HIP_FAILURE; }
        d->h.push_back(h);
    }
    *out = d.release();
    return SC_HIP_SUCCESS;
}
extern "C" void sc_hip_multi_dilithium_destroy(sc_hip_multi_dilithium_t *d) { if (!d) return; for (auto *x : d->h) sc_hip_dilithium_destroy(x); delete d; }

extern "C" int sc_hip_multi_dilithium_set_privkey(sc_hip_multi_dilithium_t *d, const int32_t *s1, const int32_t *s2, const int32_t *t, const uint8_t *K, const uint8_t *tr,
                                                  const int32_t *Ahat)
{
    if (!d) return sc::fail("multi_dilithium_set_privkey: null handle");
    for (auto *h : d->h) if (sc_hip_dilithium_set_privkey(h, s1, s2, t, K, tr, Ahat) != SC_HIP_SUCCESS) return SC_HIP_FAILURE;
    return SC_HIP_SUCCESS;
}
extern "C" int sc_hip_multi_dilithium_set_pubkey(sc_hip_multi_dilithium_t *d, const int32_t *t1, const uint8_t *rho, const int32_t *Ahat)
{
    if (!d) return sc::fail("multi_dilithium_set_pubkey: null handle");
    for (auto *h : d->h) if (sc_hip_dilithium_set_pubkey(h, t1, rho, Ahat) != SC_HIP_SUCCESS) return SC_HIP_FAILURE;
    return SC_HIP_SUCCESS;
}
extern "C" size_t sc_hip_multi_dilithium_sig_max(const sc_hip_multi_dilithium_t *d) { return (d && !d->h.empty()) ? sc_hip_dilithium_sig_max(d->h[0]) : 0; }

extern "C" int sc_hip_multi_dilithium_sign_host(sc_hip_multi_dilithium_t *d, size_t batch, const uint8_t *h_msgs, size_t msg_stride, const uint32_t *h_mlens, uint32_t fixed_len,
                                                uint8_t *h_sigs, size_t sig_stride, uint32_t *h_siglens, uint32_t *h_trials)
{
    if (!d || !h_sigs || !h_siglens) return sc::fail("multi_dilithium_sign_host: null argument");
    return fan_out(d->m, batch, [&](int w, size_t b, size_t c) {
        return sc_hip_dilithium_sign_host(d->h[(size_t)w], c, h_msgs ? h_msgs + b * msg_stride : nullptr, msg_stride, h_mlens ? h_mlens + b : nullptr, fixed_len,
                                          h_sigs + b * sig_stride, sig_stride, h_siglens + b, h_trials ? h_trials + b : nullptr);
    });
}
extern "C" int sc_hip_multi_dilithium_verify_host(sc_hip_multi_dilithium_t *d, size_t batch, const uint8_t *h_msgs, size_t msg_stride, const uint32_t *h_mlens, uint32_t fixed_len,
                                                  const uint8_t *h_sigs, size_t sig_stride, const uint32_t *h_siglens, uint32_t *h_verdicts)
{
    if (!d || !h_sigs || !h_siglens || !h_verdicts) return sc::fail("multi_dilithium_verify_host: null argument");
    return fan_out(d->m, batch, [&](int w, size_t b, size_t c) {
        return sc_hip_dilithium_verify_host(d->h[(size_t)w], c, h_msgs ? h_msgs + b * msg_stride : nullptr, msg_stride, h_mlens ? h_mlens + b : nullptr, fixed_len,
                                            h_sigs + b * sig_stride, sig_stride, h_siglens + b, h_verdicts + b);
    });
}

// ---- BLISS-B, Falcon, ring-TESLA, Kyber: one key replicated to every device, the host batch sharded (BASELINE configs[3], [4]) ---------------------------------

namespace {
// a scheme handle per worker
template <class H> struct Replicas {
    sc_hip_multi *m;
    std::vector<H *> h;
};
template <class H, class P, class Create, class Destroy>
int replicas_create(sc_hip_multi_t *m, const P *params, Replicas<H> **out, Create create, Destroy destroy, const char *what)
{
    if (!m || !params || !out) return sc::fail(std::string(what) + ": null argument");
    std::unique_ptr<Replicas<H>> r(new Replicas<H>);
    r->m = m;
    for (auto *c : m->ctx) {
        H *h = nullptr;
        if (create(c, params, &h) != SC_HIP_SUCCESS) { for (auto *x : r->h) destroy(x); return SC_HIP_FAILURE; }
        r->h.push_back(h);
    }
    *out = r.release();
    return SC_HIP_SUCCESS;
}
template <class H, class F> int for_all(Replicas<H> *r, const char *what, F f)
{
    if (!r) return sc::fail(std::string(what) + ": null handle");
    for (auto *h : r->h) if (f(h) != SC_HIP_SUCCESS) return SC_HIP_FAILURE;
    return SC_HIP_SUCCESS;
}
}  // namespace

struct sc_hip_multi_bliss : Replicas<sc_hip_bliss_t> {};
struct sc_hip_multi_falcon : Replicas<sc_hip_falcon_t> {};
struct sc_hip_multi_tesla : Replicas<sc_hip_tesla_t> {};
struct sc_hip_multi_kyber : Replicas<sc_hip_kyber_t> {};

#define SC_MULTI_CREATE(NAME, PARAMS)                                                                                                                   \
    extern "C" int sc_hip_multi_##NAME##_create(sc_hip_multi_t *m, const PARAMS *params, sc_hip_multi_##NAME##_t **out)                                   \
    {                                                                                                                                                     \
        Replicas<sc_hip_##NAME##_t> *r = nullptr;                                                                                                         \
        const int rc = replicas_create<sc_hip_##NAME##_t>(m, params, &r, sc_hip_##NAME##_create, sc_hip_##NAME##_destroy, "multi_" #NAME "_create");       \
        if (rc == SC_HIP_SUCCESS) *out = static_cast<sc_hip_multi_##NAME##_t *>(r);                                                                        \
        return rc;                                                                                                                                        \
    }                                                                                                                                                     \
    extern "C" void sc_hip_multi_##NAME##_destroy(sc_hip_multi_##NAME##_t *h) { if (!h) return; for (auto *x : h->h) sc_hip_##NAME##_destroy(x); delete static_cast<Replicas<sc_hip_##NAME##_t> *>(h); }
SC_MULTI_CREATE(bliss, sc_hip_bliss_params_t)
SC_MULTI_CREATE(falcon, sc_hip_falcon_params_t)
SC_MULTI_CREATE(tesla, sc_hip_tesla_params_t)
SC_MULTI_CREATE(kyber, sc_hip_kyber_params_t)
#undef SC_MULTI_CREATE

extern "C" int sc_hip_multi_bliss_set_key(sc_hip_multi_bliss_t *h, const int16_t *f, const int16_t *g, const int16_t *a)
{ return for_all(h, "multi_bliss_set_key", [&](sc_hip_bliss_t *x) { return sc_hip_bliss_set_key(x, f, g, a); }); }
// the sampler table of every worker's handle: the reference context's own (64: sc_hip_bliss_set_cdt_table; 128 / 192: sc_hip_bliss_set_cdt_table_hp - the configuration of the
// reference's BLISS-B harness)
extern "C" int sc_hip_multi_bliss_set_cdt_table(sc_hip_multi_bliss_t *h, int precision_bits, const uint64_t *h_cdf, int size)
{
    if (!h_cdf) return sc::fail("multi_bliss_set_cdt_table: null table");
    return for_all(h, "multi_bliss_set_cdt_table", [&](sc_hip_bliss_t *x) { return precision_bits == 64 ? sc_hip_bliss_set_cdt_table(x, h_cdf, size) : sc_hip_bliss_set_cdt_table_hp(x, precision_bits, h_cdf, size); });
}
extern "C" size_t sc_hip_multi_bliss_sig_bytes(const sc_hip_multi_bliss_t *h) { return (h && !h->h.empty()) ? sc_hip_bliss_sig_bytes(h->h[0]) : 0; }
extern "C" int sc_hip_multi_bliss_sign_host(sc_hip_multi_bliss_t *h, size_t batch, const uint8_t *h_msgs, size_t msg_stride, const uint32_t *h_mlens, uint32_t fixed_len,
                                            const uint32_t *h_tapes, size_t tape_stride, const uint32_t *h_tape_lens,
                                            uint8_t *h_sigs, size_t sig_stride, uint32_t *h_siglens, uint32_t *h_trials, uint32_t *h_status)
{
    if (!h || !h_tapes || !h_sigs || !h_siglens || !h_status) return sc::fail("multi_bliss_sign_host: null argument");
    return fan_out(h->m, batch, [&](int w, size_t b, size_t c) {
        return sc_hip_bliss_sign_host(h->h[(size_t)w], c, h_msgs ? h_msgs + b * msg_stride : nullptr, msg_stride, h_mlens ? h_mlens + b : nullptr, fixed_len,
                                      h_tapes + b * tape_stride, tape_stride, h_tape_lens ? h_tape_lens + b : nullptr,
                                      h_sigs + b * sig_stride, sig_stride, h_siglens + b, h_trials ? h_trials + b : nullptr, h_status + b);
    });
}
extern "C" int sc_hip_multi_bliss_verify_host(sc_hip_multi_bliss_t *h, size_t batch, const uint8_t *h_msgs, size_t msg_stride, const uint32_t *h_mlens, uint32_t fixed_len,
                                              const uint8_t *h_sigs, size_t sig_stride, const uint32_t *h_siglens, uint32_t *h_verdicts)
{
    if (!h || !h_sigs || !h_siglens || !h_verdicts) return sc::fail("multi_bliss_verify_host: null argument");
    return fan_out(h->m, batch, [&](int w, size_t b, size_t c) {
        return sc_hip_bliss_verify_host(h->h[(size_t)w], c, h_msgs ? h_msgs + b * msg_stride : nullptr, msg_stride, h_mlens ? h_mlens + b : nullptr, fixed_len,
                                        h_sigs + b * sig_stride, sig_stride, h_siglens + b, h_verdicts + b);
    });
}

extern "C" int sc_hip_multi_falcon_set_privkey(sc_hip_multi_falcon_t *h, const double *sk)
{ return for_all(h, "multi_falcon_set_privkey", [&](sc_hip_falcon_t *x) { return sc_hip_falcon_set_privkey(x, sk); }); }
extern "C" int sc_hip_multi_falcon_set_pubkey(sc_hip_multi_falcon_t *h, const int16_t *pk)
{ return for_all(h, "multi_falcon_set_pubkey", [&](sc_hip_falcon_t *x) { return sc_hip_falcon_set_pubkey(x, pk); }); }
extern "C" size_t sc_hip_multi_falcon_sig_bytes(const sc_hip_multi_falcon_t *h) { return (h && !h->h.empty()) ? sc_hip_falcon_sig_bytes(h->h[0]) : 0; }
extern "C" int sc_hip_multi_falcon_sign_host(sc_hip_multi_falcon_t *h, size_t batch, const uint8_t *h_msgs, size_t msg_stride, const uint32_t *h_mlens, uint32_t fixed_len,
                                             const uint32_t *h_tapes, size_t tape_stride, const uint32_t *h_tape_lens,
                                             uint8_t *h_sigs, size_t sig_stride, uint32_t *h_siglens, uint32_t *h_status)
{
    if (!h || !h_tapes || !h_sigs || !h_siglens || !h_status) return sc::fail("multi_falcon_sign_host: null argument");
    return fan_out(h->m, batch, [&](int w, size_t b, size_t c) {
        return sc_hip_falcon_sign_host(h->h[(size_t)w], c, h_msgs ? h_msgs + b * msg_stride : nullptr, msg_stride, h_mlens ? h_mlens + b : nullptr, fixed_len,
                                       h_tapes + b * tape_stride, tape_stride, h_tape_lens ? h_tape_lens + b : nullptr,
                                       h_sigs + b * sig_stride, sig_stride, h_siglens + b, h_status + b);
    });
}
extern "C" int sc_hip_multi_falcon_verify_host(sc_hip_multi_falcon_t *h, size_t batch, const uint8_t *h_msgs, size_t msg_stride, const uint32_t *h_mlens, uint32_t fixed_len,
                                               const uint8_t *h_sigs, size_t sig_stride, const uint32_t *h_siglens, uint32_t *h_verdicts)
{
    if (!h || !h_sigs || !h_siglens || !h_verdicts) return sc::fail("multi_falcon_verify_host: null argument");
    return fan_out(h->m, batch, [&](int w, size_t b, size_t c) {
        return sc_hip_falcon_verify_host(h->h[(size_t)w], c, h_msgs ? h_msgs + b * msg_stride : nullptr, msg_stride, h_mlens ? h_mlens + b : nullptr, fixed_len,
                                         h_sigs + b * sig_stride, sig_stride, h_siglens + b, h_verdicts + b);
    });
}

extern "C" int sc_hip_multi_tesla_set_key(sc_hip_multi_tesla_t *h, const int32_t *a1, const int32_t *a2, const int16_t *s, const int16_t *e1, const int16_t *e2,
                                          const int32_t *t1, const int32_t *t2)
{ return for_all(h, "multi_tesla_set_key", [&](sc_hip_tesla_t *x) { return sc_hip_tesla_set_key(x, a1, a2, s, e1, e2, t1, t2); }); }
extern "C" size_t sc_hip_multi_tesla_sig_bytes(const sc_hip_multi_tesla_t *h) { return (h && !h->h.empty()) ? sc_hip_tesla_sig_bytes(h->h[0]) : 0; }
extern "C" int sc_hip_multi_tesla_sign_host(sc_hip_multi_tesla_t *h, size_t batch, const uint8_t *h_msgs, size_t msg_stride, const uint32_t *h_mlens, uint32_t fixed_len,
                                            const uint32_t *h_tapes, size_t tape_stride, const uint32_t *h_tape_lens,
                                            uint8_t *h_sigs, size_t sig_stride, uint32_t *h_siglens, uint32_t *h_trials, uint32_t *h_status)
{
    if (!h || !h_tapes || !h_sigs || !h_siglens || !h_status) return sc::fail("multi_tesla_sign_host: null argument");
    return fan_out(h->m, batch, [&](int w, size_t b, size_t c) {
        return sc_hip_tesla_sign_host(h->h[(size_t)w], c, h_msgs ? h_msgs + b * msg_stride : nullptr, msg_stride, h_mlens ? h_mlens + b : nullptr, fixed_len,
                                      h_tapes + b * tape_stride, tape_stride, h_tape_lens ? h_tape_lens + b : nullptr,
                                      h_sigs + b * sig_stride, sig_stride, h_siglens + b, h_trials ? h_trials + b : nullptr, h_status + b);
    });
}
extern "C" int sc_hip_multi_tesla_verify_host(sc_hip_multi_tesla_t *h, size_t batch, const uint8_t *h_msgs, size_t msg_stride, const uint32_t *h_mlens, uint32_t fixed_len,
                                              const uint8_t *h_sigs, size_t sig_stride, const uint32_t *h_siglens, uint32_t *h_verdicts)
{
    if (!h || !h_sigs || !h_siglens || !h_verdicts) return sc::fail("multi_tesla_verify_host: null argument");
    return fan_out(h->m, batch, [&](int w, size_t b, size_t c) {
        return sc_hip_tesla_verify_host(h->h[(size_t)w], c, h_msgs ? h_msgs + b * msg_stride : nullptr, msg_stride, h_mlens ? h_mlens + b : nullptr, fixed_len,
                                        h_sigs + b * sig_stride, sig_stride, h_siglens + b, h_verdicts + b);
    });
}

extern "C" int sc_hip_multi_kyber_set_pubkey(sc_hip_multi_kyber_t *h, const int32_t *t_ntt, const int32_t *Agen, const int32_t *t, const uint8_t *rho)
{
    return for_all(h, "multi_kyber_set_pubkey", [&](sc_hip_kyber_t *x) {
        if (sc_hip_kyber_set_pubkey(x, t_ntt, Agen) != SC_HIP_SUCCESS) return (int)SC_HIP_FAILURE;
        return (t && rho) ? sc_hip_kyber_set_kem_pubkey(x, t, rho) : (int)SC_HIP_SUCCESS;
    });
}
extern "C" int sc_hip_multi_kyber_set_privkey(sc_hip_multi_kyber_t *h, const int32_t *s_ntt)
{ return for_all(h, "multi_kyber_set_privkey", [&](sc_hip_kyber_t *x) { return sc_hip_kyber_set_privkey(x, s_ntt); }); }
extern "C" size_t sc_hip_multi_kyber_ct_bytes(const sc_hip_multi_kyber_t *h) { return (h && !h->h.empty()) ? sc_hip_kyber_ct_bytes(h->h[0]) : 0; }
extern "C" int sc_hip_multi_kyber_kem_encaps_host(sc_hip_multi_kyber_t *h, size_t batch, const uint8_t *h_m, uint8_t *h_ct, size_t ct_stride, uint8_t *h_keys)
{
    if (!h || !h_m || !h_ct || !h_keys) return sc::fail("multi_kyber_kem_encaps_host: null argument");
    return fan_out(h->m, batch, [&](int w, size_t b, size_t c) { return sc_hip_kyber_kem_encaps_host(h->h[(size_t)w], c, h_m + b * 32, h_ct + b * ct_stride, ct_stride, h_keys + b * 32); });
}
extern "C" int sc_hip_multi_kyber_kem_decaps_host(sc_hip_multi_kyber_t *h, size_t batch, const uint8_t *h_ct, size_t ct_stride, uint8_t *h_keys)
{
    if (!h || !h_ct || !h_keys) return sc::fail("multi_kyber_kem_decaps_host: null argument");
    return fan_out(h->m, batch, [&](int w, size_t b, size_t c) { return sc_hip_kyber_kem_decaps_host(h->h[(size_t)w], c, h_ct + b * ct_stride, ct_stride, h_keys + b * 32); });
}
