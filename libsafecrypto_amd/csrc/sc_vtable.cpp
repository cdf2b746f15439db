// sc_vtable.cpp - reference-shaped single-polynomial entry points over the batched ABI (host code).
#include "../../include/sc_hip.h"
#include "../../include/sc_hip_vtable.h"

#include <atomic>
#include <map>
#include <mutex>
#include <stdlib.h>
#include <utility>
#include <vector>

namespace {

std::mutex g_mu;
sc_hip_ctx_t *g_ctx = nullptr;
std::map<std::pair<int32_t, int>, sc_hip_plan_t *> g_plans;
std::atomic<unsigned long> g_errors{0};

sc_hip_plan_t *plan_for(const sc_hip_ntt_params_t *p)
{
    if (!p) return nullptr;
    std::lock_guard<std::mutex> lk(g_mu);
    if (!g_ctx) {
        const char *d = getenv("SC_HIP_DEVICE");
        if (sc_hip_ctx_create(d ? atoi(d) : 0, &g_ctx) != SC_HIP_SUCCESS) return nullptr;
    }
    const auto key = std::make_pair(p->u.ntt32.q, (int)p->n);
    auto it = g_plans.find(key);
    if (it != g_plans.end()) return it->second;
    sc_hip_plan_t *pl = nullptr;
    if (sc_hip_plan_create(g_ctx, key.first, key.second, &pl) != SC_HIP_SUCCESS) return nullptr;
    g_plans[key] = pl;
    return pl;
}

inline void note(int rc) { if (rc != SC_HIP_SUCCESS) g_errors++; }

void v_fwd32(int32_t *v, const sc_hip_ntt_params_t *p, const int32_t *t, const int32_t *)
{ sc_hip_plan_t *pl = plan_for(p); note(pl ? sc_hip_ntt_fwd_host(pl, 1, t, v) : SC_HIP_FAILURE); }
void v_inv32(int32_t *v, const sc_hip_ntt_params_t *p, const int32_t *t, const int32_t *, const int32_t *)
{ sc_hip_plan_t *pl = plan_for(p); note(pl ? sc_hip_ntt_inv_host(pl, 1, t, v) : SC_HIP_FAILURE); }
void v_fwd16(int32_t *v, const sc_hip_ntt_params_t *p, const int32_t *t, const int16_t *)
{ v_fwd32(v, p, t, nullptr); }
void v_inv16(int32_t *v, const sc_hip_ntt_params_t *p, const int32_t *t, const int16_t *, const int16_t *)
{ v_inv32(v, p, t, nullptr, nullptr); }
void v_pw32(int32_t *v, const sc_hip_ntt_params_t *p, const int32_t *t, const int32_t *u)
{ sc_hip_plan_t *pl = plan_for(p); note(pl ? sc_hip_pointwise_host(pl, p->n, t, u, v) : SC_HIP_FAILURE); }
void v_pw16(int32_t *v, const sc_hip_ntt_params_t *p, const int32_t *t, const int16_t *u)
{
    if (!p) { note(SC_HIP_FAILURE); return; }
    std::vector<int32_t> u32(u, u + p->n);
    v_pw32(v, p, t, u32.data());
}
void v_norm(int32_t *v, size_t n, const sc_hip_ntt_params_t *p)
{ sc_hip_plan_t *pl = plan_for(p); note(pl ? sc_hip_normalize_host(pl, n, v, v) : SC_HIP_FAILURE); }
void v_cent(int32_t *v, size_t n, const sc_hip_ntt_params_t *p)
{ sc_hip_plan_t *pl = plan_for(p); note(pl ? sc_hip_center_host(pl, n, v, v) : SC_HIP_FAILURE); }

sc_hip_ctx_t *ctx_only()
{
    std::lock_guard<std::mutex> lk(g_mu);
    if (!g_ctx) {
        const char *d = getenv("SC_HIP_DEVICE");
        if (sc_hip_ctx_create(d ? atoi(d) : 0, &g_ctx) != SC_HIP_SUCCESS) return nullptr;
    }
    return g_ctx;
}
void v_sparse32(int32_t *v, size_t n, uint16_t omega, const int32_t *t, const int32_t *c)
{ sc_hip_ctx_t *cx = ctx_only(); note(cx ? sc_hip_mult_sparse_host(cx, 1, (int)n, omega, t, 0, c, v) : SC_HIP_FAILURE); }
void v_sparse16(int32_t *v, size_t n, uint16_t omega, const int16_t *t, const int32_t *c)
{ sc_hip_ctx_t *cx = ctx_only(); note(cx ? sc_hip_mult_sparse_host(cx, 1, (int)n, omega, t, 1, c, v) : SC_HIP_FAILURE); }
// invert_32 / div_32 run over n coefficients that need not be the ring degree (the reference passes the length separately)
int32_t v_invert(int32_t *v, const sc_hip_ntt_params_t *p, size_t n)
{
    sc_hip_plan_t *pl = plan_for(p);
    int32_t rc = 1;
    const int e = pl ? sc_hip_ntt_invert_host(pl, 1, n, v, &rc) : SC_HIP_FAILURE;
    note(e);
    return e == SC_HIP_SUCCESS ? rc : 1;
}
int32_t v_div(int32_t *num, const int32_t *den, const sc_hip_ntt_params_t *p, size_t n)
{
    sc_hip_plan_t *pl = plan_for(p);
    int32_t rc = 1;
    const int e = pl ? sc_hip_ntt_div_host(pl, 1, n, num, den, &rc) : SC_HIP_FAILURE;
    note(e);
    return e == SC_HIP_SUCCESS ? rc : 1;
}

const sc_hip_ntt32_vtable_t g_vt = {
    v_pw32, v_pw16, v_cent, v_norm,
    v_fwd32, v_inv32, v_fwd32, v_inv32,
    v_fwd16, v_inv16, v_fwd16, v_inv16,
    v_sparse32, v_sparse16, v_invert, v_div,
};

}  // namespace

extern "C" const sc_hip_ntt32_vtable_t *sc_hip_utils_arith_ntt32(void)
{
    if (sc_hip_device_count() < 1) return nullptr;
    return &g_vt;
}

extern "C" unsigned long sc_hip_vtable_errors(void) { return g_errors.load(); }
