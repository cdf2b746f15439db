// kyber.hip - batched Kyber (round 1: q = 7681, n = 256) CPA encryption / decryption behind the C ABI of sc_hip.h.
//
// Replaces kyber_cpa_enc / kyber_cpa_dec (src/utils/arith/module_lwe.c:1268-1401, 1415-1481) as
// kyber_kem_encapsulation / _decapsulation call them (src/schemes/kem/kyber/kyber_kem.c:866,992-1001; keys hold NTT(t)
// and NTT(s): kyber_kem_params.h:43,46), with binomial_rand_sample_csprng (:380-412), create_rand_product_16_csprng in
// its TRANSPOSED form (:588-748) and mlwe_compress / mlwe_decompress (:43-69, reproduced literally, wrap-around included).
// What the reference derives with its AES-CTR-DRBG is an input: the k*k matrix polynomials in generation order (once per
// key) and, per message, the (2k+1)*n*eta/4 noise bytes its prng_mem calls return for the seed r.
#include "engine_internal.h"
#include "mlwe_kernels.h"

using namespace sc;
using namespace sc_mlwe;

namespace {

struct KyP { int32_t n, q, q_bits; uint32_t q_inv; int32_t q_norm, k, eta, d_u, d_v, d_t, set, pad[5]; };

// mlwe_compress on one value, module_lwe.c:43-56
__device__ __forceinline__ int32_t ky_compress(int32_t x, int d, uint32_t q, uint32_t q_inv, int q_norm)
{
    int32_t t = (int32_t)((uint32_t)x << d);
    t = (int32_t)((uint32_t)t + (q >> 1));
    t = (int32_t)(((uint64_t)(int64_t)t * (uint64_t)q_inv) >> (32 + q_norm));
    return (int32_t)((uint32_t)t & ((1u << d) - 1u));
}

// binomial_rand_sample_csprng: coefficient group g (4 values) of a sampler reads its eta bytes [g*eta, (g+1)*eta)
__global__ void ky_binomial_kernel(const uint8_t *noise, size_t noise_stride, uint32_t count, int32_t *r, int32_t *e1, int32_t *e2, KyP P)
{
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int kn4 = P.k * P.n / 4, n4 = P.n / 4, per = 2 * kn4 + n4;
    const uint32_t item = (uint32_t)(t / per); const int g = (int)(t % per);
    if (item >= count) return;
    const uint8_t *b = noise + (size_t)item * noise_stride + (size_t)g * P.eta;
    uint32_t a[8] = { 0, 0, 0, 0, 0, 0, 0, 0 };
    for (int e = 0; e < P.eta; e++) { const uint32_t v = b[e]; for (int i = 0; i < 8; i++) a[i] += v & (1u << i); }
    int32_t *dst = (g < kn4) ? r + ((size_t)item * kn4 + g) * 4 : (g < 2 * kn4) ? e1 + ((size_t)item * kn4 + (g - kn4)) * 4 : e2 + ((size_t)item * n4 + (g - 2 * kn4)) * 4;
    dst[0] = (int32_t)a[0] - (int32_t)(a[1] >> 1);
    dst[1] = (int32_t)(a[2] >> 2) - (int32_t)(a[3] >> 3);
    dst[2] = (int32_t)(a[4] >> 4) - (int32_t)(a[5] >> 5);
    dst[3] = (int32_t)(a[6] >> 6) - (int32_t)(a[7] >> 7);
}

// u = compress(u0 + e1, d_u) ; v = compress(v0 + [m bit] floor(q/2) + e2, d_v)   (module_lwe.c:1355-1395)
__global__ void ky_enc_finish_kernel(const int32_t *u0, const int32_t *v0, const int32_t *e1, const int32_t *e2, const uint8_t *m, uint32_t count,
                                     int32_t *u, int32_t *v, KyP P)
{
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int per = (P.k + 1) * P.n;
    const uint32_t item = (uint32_t)(t / per); const int e = (int)(t % per);
    if (item >= count) return;
    if (e < P.k * P.n) {
        const size_t i = (size_t)item * P.k * P.n + e;
        u[i] = ky_compress(u0[i] + e1[i], P.d_u, (uint32_t)P.q, P.q_inv, P.q_norm);
    } else {
        const int j = e - P.k * P.n; const size_t i = (size_t)item * P.n + j;
        const int32_t bit = (m[(size_t)item * 32 + (j >> 3)] & (0x80u >> (j & 7))) ? (P.q >> 1) : 0;
        v[i] = ky_compress(v0[i] + bit + e2[i], P.d_v, (uint32_t)P.q, P.q_inv, P.q_norm);
    }
}

// mlwe_decompress(u, d_u): t = u*q >> d  (module_lwe.c:58-69)
__global__ void ky_decompress_kernel(const int32_t *in, int32_t *out, size_t len, int d, int32_t q)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < len) out[i] = (int32_t)(in[i] * q) >> d;
}

// m byte: 8 x compress(decompress(v) - x, 1), most significant bit first (module_lwe.c:1462-1478)
__global__ void ky_dec_finish_kernel(const int32_t *v, const int32_t *x, uint8_t *m, uint32_t count, KyP P)
{
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t item = (uint32_t)(t / 32); const int byte = (int)(t % 32);
    if (item >= count) return;
    uint32_t b = 0;
    for (int j = 0; j < 8; j++) {
        const size_t i = (size_t)item * P.n + 8 * byte + j;
        const int32_t dv = (int32_t)(v[i] * P.q) >> P.d_v;
        b |= (uint32_t)(ky_compress(dv - x[i], 1, (uint32_t)P.q, P.q_inv, P.q_norm) & 1) << (7 - j);
    }
    m[(size_t)item * 32 + byte] = (uint8_t)b;
}

}  // namespace

struct sc_hip_kyber {
    sc_hip_ctx *ctx;
    sc_hip_plan *plan;
    KyP P;
    bool have_pub, have_priv;
    uint32_t *d_A, *d_t, *d_s;            // internal order, Montgomery form: A^T-ready [k][k], t [1][k], s [1][k]
    size_t cap;
    int32_t *r, *e1, *e2, *u0, *v0;
    uint32_t *yh;
};

static int ky_ensure(sc_hip_kyber *h, size_t batch)
{
    if (batch <= h->cap) return SC_HIP_SUCCESS;
    void *old[] = { h->r, h->e1, h->e2, h->u0, h->v0, h->yh };
    for (void *p : old) if (p) (void)hipFree(p);
    h->r = h->e1 = h->e2 = h->u0 = h->v0 = nullptr; h->yh = nullptr; h->cap = 0;
    const size_t kn = (size_t)h->P.k * N;
    HIPCHK(hipMalloc((void **)&h->r, batch * kn * 4)); HIPCHK(hipMalloc((void **)&h->e1, batch * kn * 4)); HIPCHK(hipMalloc((void **)&h->e2, batch * N * 4));
    HIPCHK(hipMalloc((void **)&h->u0, batch * kn * 4)); HIPCHK(hipMalloc((void **)&h->v0, batch * N * 4)); HIPCHK(hipMalloc((void **)&h->yh, batch * kn * 4));
    h->cap = batch;
    return SC_HIP_SUCCESS;
}

extern "C" int sc_hip_kyber_create(sc_hip_ctx_t *ctx, const sc_hip_kyber_params_t *params, sc_hip_kyber_t **out)
{
    if (!ctx || !params || !out) return fail("kyber_create: null argument");
    static_assert(sizeof(KyP) == sizeof(sc_hip_kyber_params_t), "parameter struct layout");
    KyP P; memcpy(&P, params, sizeof P);
    if (P.n != 256) return fail("kyber_create: n must be 256");
    if (P.k < 1 || P.k > 8 || P.eta < 1 || P.eta > 8 || P.d_u < 1 || P.d_u > 16 || P.d_v < 1 || P.d_v > 16 || P.q_norm < 0 || P.q_norm > 31)
        return fail("kyber_create: parameters out of range");
    sc_hip_plan *plan = nullptr;
    if (sc_hip_plan_create(ctx, P.q, P.n, &plan) != SC_HIP_SUCCESS) return SC_HIP_FAILURE;
    sc_hip_kyber *h = new sc_hip_kyber;
    memset(h, 0, sizeof *h);
    h->ctx = ctx; h->plan = plan; h->P = P;
    *out = h;
    return SC_HIP_SUCCESS;
}

extern "C" void sc_hip_kyber_destroy(sc_hip_kyber_t *h)
{
    if (!h) return;
    void *ptrs[] = { h->d_A, h->d_t, h->d_s, h->r, h->e1, h->e2, h->u0, h->v0, h->yh };
    for (void *p : ptrs) if (p) (void)hipFree(p);
    sc_hip_plan_destroy(h->plan);
    delete h;
}

// natural-order NTT-domain polynomials (any representative) -> internal order, Montgomery form; rows/cols: transpose on the way
static int ky_load(sc_hip_kyber *h, const int32_t *host, size_t npoly, uint32_t **dst, int rows, int cols)
{
    const size_t bytes = npoly * N * 4;
    int32_t *tmp = nullptr;
    HIPCHK(hipSetDevice(h->ctx->device));
    HIPCHK(hipMalloc((void **)&tmp, bytes));
    if (!*dst) HIPCHK(hipMalloc((void **)dst, bytes));
    HIPCHK(hipMemcpy(tmp, host, bytes, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(dil_to_internal_mont, dim3((unsigned)((npoly * N + 255) / 256)), dim3(256), 0, nullptr, tmp, *dst, npoly, h->plan->k[1], h->plan->r2, h->plan->r2p, rows, cols);
    int rc = (hipDeviceSynchronize() == hipSuccess) ? SC_HIP_SUCCESS : fail("kyber key load: kernel failed");
    (void)hipFree(tmp);
    return rc;
}

// t_ntt: NTT(t) as the public key stores it (kyber_kem.c:636-650); Agen: the k*k matrix polynomials in the order the
// CSPRNG seeded with rho generates them (module_lwe.c:519-547) - encryption uses them transposed (:700-720)
extern "C" int sc_hip_kyber_set_pubkey(sc_hip_kyber_t *h, const int32_t *t_ntt, const int32_t *Agen)
{
    if (!h || !t_ntt || !Agen) return fail("kyber_set_pubkey: null argument");
    const int k = h->P.k;
    if (ky_load(h, Agen, (size_t)k * k, &h->d_A, k, k)) return SC_HIP_FAILURE;       // A_enc[i][j] = G[j*k + i]
    if (ky_load(h, t_ntt, (size_t)k, &h->d_t, 0, 0)) return SC_HIP_FAILURE;
    h->have_pub = true;
    return SC_HIP_SUCCESS;
}

extern "C" int sc_hip_kyber_set_privkey(sc_hip_kyber_t *h, const int32_t *s_ntt)
{
    if (!h || !s_ntt) return fail("kyber_set_privkey: null argument");
    if (ky_load(h, s_ntt, (size_t)h->P.k, &h->d_s, 0, 0)) return SC_HIP_FAILURE;
    h->have_priv = true;
    return SC_HIP_SUCCESS;
}

extern "C" size_t sc_hip_kyber_noise_bytes(const sc_hip_kyber_t *h) { return h ? (size_t)(2 * h->P.k + 1) * N * h->P.eta / 4 : 0; }

extern "C" int sc_hip_kyber_cpa_enc(sc_hip_kyber_t *h, size_t batch, const uint8_t *d_m, const uint8_t *d_noise, size_t noise_stride,
                                    int32_t *d_u, int32_t *d_v, void *stream)
{
    if (!h) return fail("kyber_cpa_enc: null handle");
    if (!h->have_pub) return fail("kyber_cpa_enc: no public key loaded");
    if (batch == 0) return SC_HIP_SUCCESS;
    if (batch > 0x7fffffffu / 4096) return fail("kyber_cpa_enc: batch too large");
    if (!d_m || !d_noise || !d_u || !d_v) return fail("kyber_cpa_enc: null device pointer");
    if (noise_stride < sc_hip_kyber_noise_bytes(h)) return fail("kyber_cpa_enc: noise_stride smaller than sc_hip_kyber_noise_bytes()");
    HIPCHK(hipSetDevice(h->ctx->device));
    if (ky_ensure(h, batch)) return SC_HIP_FAILURE;
    hipStream_t s = (hipStream_t)stream;
    const KyP &P = h->P;
    const uint32_t B = (uint32_t)batch;
    const int k = P.k;
    const size_t ng = batch * (size_t)((2 * k + 1) * N / 4);
    hipLaunchKernelGGL(ky_binomial_kernel, dim3((unsigned)((ng + 255) / 256)), dim3(256), 0, s, d_noise, noise_stride, B, h->r, h->e1, h->e2, P);
    HIPCHK(hipGetLastError());
    DilNtt a = mlwe_base_args(h->plan, k, k, 1 << 20);
    a.in = h->r; a.out_int = h->yh; a.yh = h->yh;
    if (mlwe_launch<DOP_FWD_INT>(h->ctx, a, B, k, s)) return SC_HIP_FAILURE;
    a.key = h->d_A; a.w = h->u0; a.w1 = nullptr;
    if (mlwe_launch<DOP_MATVEC>(h->ctx, a, B, k, s)) return SC_HIP_FAILURE;
    a.key = h->d_t; a.w = h->v0; a.kk = 1;
    if (mlwe_launch<DOP_MATVEC>(h->ctx, a, B, 1, s)) return SC_HIP_FAILURE;
    const size_t ne = batch * (size_t)(k + 1) * N;
    hipLaunchKernelGGL(ky_enc_finish_kernel, dim3((unsigned)((ne + 255) / 256)), dim3(256), 0, s, h->u0, h->v0, h->e1, h->e2, d_m, B, d_u, d_v, P);
    HIPCHK(hipGetLastError());
    return SC_HIP_SUCCESS;
}

extern "C" int sc_hip_kyber_cpa_dec(sc_hip_kyber_t *h, size_t batch, const int32_t *d_u, const int32_t *d_v, uint8_t *d_m, void *stream)
{
    if (!h) return fail("kyber_cpa_dec: null handle");
    if (!h->have_priv) return fail("kyber_cpa_dec: no private key loaded");
    if (batch == 0) return SC_HIP_SUCCESS;
    if (batch > 0x7fffffffu / 4096) return fail("kyber_cpa_dec: batch too large");
    if (!d_m || !d_u || !d_v) return fail("kyber_cpa_dec: null device pointer");
    HIPCHK(hipSetDevice(h->ctx->device));
    if (ky_ensure(h, batch)) return SC_HIP_FAILURE;
    hipStream_t s = (hipStream_t)stream;
    const KyP &P = h->P;
    const uint32_t B = (uint32_t)batch;
    const int k = P.k;
    const size_t nu = batch * (size_t)k * N;
    hipLaunchKernelGGL(ky_decompress_kernel, dim3((unsigned)((nu + 255) / 256)), dim3(256), 0, s, d_u, h->r, nu, P.d_u, P.q);
    HIPCHK(hipGetLastError());
    DilNtt a = mlwe_base_args(h->plan, 1, k, 1 << 20);
    a.in = h->r; a.out_int = h->yh; a.yh = h->yh;
    if (mlwe_launch<DOP_FWD_INT>(h->ctx, a, B, k, s)) return SC_HIP_FAILURE;
    a.key = h->d_s; a.w = h->v0; a.w1 = nullptr;
    if (mlwe_launch<DOP_MATVEC>(h->ctx, a, B, 1, s)) return SC_HIP_FAILURE;
    hipLaunchKernelGGL(ky_dec_finish_kernel, dim3((unsigned)((batch * 32 + 255) / 256)), dim3(256), 0, s, d_v, h->v0, d_m, B, P);
    HIPCHK(hipGetLastError());
    return SC_HIP_SUCCESS;
}
