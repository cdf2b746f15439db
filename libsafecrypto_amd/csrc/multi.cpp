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
// The caller's whole buffer pinned once, on the calling thread, for all workers (portable: usable from every device's context).  A buffer that cannot
// be registered (already pinned by the caller, or not registrable) is left as it is: the copies then take the runtime's staged path.
struct HostPin {
    void *p = nullptr;
    HostPin(const void *ptr, size_t bytes)
    {
        if (ptr && bytes && hipHostRegister(const_cast<void *>(ptr), bytes, hipHostRegisterPortable) == hipSuccess) p = const_cast<void *>(ptr);
        else (void)hipGetLastError();
    }
    ~HostPin() { if (p) (void)hipHostUnregister(p); }
    HostPin(const HostPin &) = delete;
    HostPin &operator=(const HostPin &) = delete;
};

// runs job(worker, begin, count) on one host thread per worker; an error message of a worker thread is carried back to the caller's thread
template <class F>
int fan_out(const sc_hip_multi *m, size_t total, F job)
{
    const int W = (int)m->dev.size();
    std::vector<int> rc((size_t)W, SC_HIP_SUCCESS);
    std::vector<std::string> err((size_t)W);
    std::vector<std::thread> th;
    for (int w = 0; w < W; w++) {
        size_t b, c;
        sc_hip_shard_range(total, W, w, &b, &c);
        th.emplace_back([&, w, b, c]() {
            if (c == 0) return;
            if (hipSetDevice(m->dev[(size_t)w]) != hipSuccess) { rc[(size_t)w] = SC_HIP_FAILURE; err[(size_t)w] = "multi: hipSetDevice failed"; return; }
            sc::t_skip_host_pin = true;                           // slices are never registered by the workers (engine_internal.h)
            rc[(size_t)w] = job(w, b, c);
            if (rc[(size_t)w] != SC_HIP_SUCCESS) err[(size_t)w] = sc_hip_last_error();
        });
    }
    for (auto &t : th) t.join();
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
    HostPin pi(h_in, batch * n * 4), po(h_out != h_in ? h_out : nullptr, batch * n * 4);
    return fan_out(p->m, batch, [&](int w, size_t b, size_t c) { return sc_hip_ntt_fwd_host(p->plan[(size_t)w], c, h_in + b * n, h_out + b * n); });
}
extern "C" int sc_hip_multi_ntt_inv_host(sc_hip_multi_plan_t *p, size_t batch, const int32_t *h_in, int32_t *h_out)
{
    if (!p || !h_in || !h_out) return sc::fail("multi_ntt_inv_host: null argument");
    const size_t n = (size_t)sc_hip_plan_n(p->plan[0]);
    HostPin pi(h_in, batch * n * 4), po(h_out != h_in ? h_out : nullptr, batch * n * 4);
    return fan_out(p->m, batch, [&](int w, size_t b, size_t c) { return sc_hip_ntt_inv_host(p->plan[(size_t)w], c, h_in + b * n, h_out + b * n); });
}
extern "C" int sc_hip_multi_polymul_ntt_host(sc_hip_multi_plan_t *p, size_t batch, const int32_t *h_a, const int32_t *h_bhat, int32_t *h_out)
{
    if (!p || !h_a || !h_bhat || !h_out) return sc::fail("multi_polymul_ntt_host: null argument");
    const size_t n = (size_t)sc_hip_plan_n(p->plan[0]);
    HostPin pa(h_a, batch * n * 4), pb(h_bhat != h_a ? h_bhat : nullptr, batch * n * 4), po((h_out != h_a && h_out != h_bhat) ? h_out : nullptr, batch * n * 4);
    return fan_out(p->m, batch, [&](int w, size_t b, size_t c) { return sc_hip_polymul_ntt_host(p->plan[(size_t)w], c, h_a + b * n, h_bhat + b * n, h_out + b * n); });
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
