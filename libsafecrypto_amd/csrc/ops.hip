// ops.hip - the remaining sc_ntt / sc_poly / sc_vec entries of the hot path, batched, behind the C ABI of include/sc_hip.h:
//   pwr_32 / invert_32 / div_32           src/utils/arith/ntt_template.c.in:1689-1775   (BLISS keygen: bliss_b.c:1193)
//   mul_32_sparse / mul_32_sparse_16      src/utils/arith/ntt.c:381-422                 (ring-TESLA: ring_tesla.c:856-871,1015-1023)
//   mul_32_scalar, add_32_scalar, mod_negate_32, cmp_not_equal_32      src/utils/arith/poly_32.c:38-165
//   absmax_32, scalar_32                  src/utils/arith/vectors.c:25-60               (BLISS norms: bliss_b.c:1010-1040,1416-1417)
// add_single_32 / sub_single_32 (poly_32.c:64-82) are sc_hip_poly_add / sc_hip_poly_sub with the output aliasing the first input.
//
// All of this is HBM-bound elementwise or per-polynomial-reduction work: 16-byte accesses per lane for the flat ops, one
// wavefront per polynomial (coefficient j = lane + 64 c) with shuffle reductions for the per-polynomial ones.  Modular
// results are canonical ([0,q)): the reference's backends return differing representatives, parity is defined after
// normalize_32.  SINT32 sums and products wrap, as they do in the reference's compiled code.
#include "engine_internal.h"

using namespace sc;

namespace {

__device__ __forceinline__ int32_t wave_or(int32_t v) { for (int o = 32; o > 0; o >>= 1) v |= __shfl_xor(v, o); return v; }
__device__ __forceinline__ int32_t wave_max(int32_t v) { for (int o = 32; o > 0; o >>= 1) v = max(v, __shfl_xor(v, o)); return v; }
__device__ __forceinline__ uint32_t wave_min_u(uint32_t v) { for (int o = 32; o > 0; o >>= 1) v = min(v, (uint32_t)__shfl_xor((int)v, o)); return v; }
__device__ __forceinline__ uint32_t wave_add_u(uint32_t v) { for (int o = 32; o > 0; o >>= 1) v += (uint32_t)__shfl_xor((int)v, o); return v; }

// x (canonical) -> x^e mod q, canonical; Montgomery domain, LSB-first square and multiply as ntt32_pwr does
__device__ __forceinline__ uint32_t pow_mod(uint32_t x, uint32_t e, const ModK &k, uint32_t r2, uint32_t r2p)
{
    uint32_t xm = csub(Mod<MODE_U32>::mulc(x, r2, r2p, k), k.q);          // x * 2^32
    uint32_t ym = (e & 1) ? xm : r2;                                      // 1 * 2^32
    e >>= 1;
    while (e > 0) {
        xm = montmul(xm, xm, k);
        if (e & 1) ym = montmul(xm, ym, k);
        e >>= 1;
    }
    return montmul(ym, 1u, k);
}

enum { FL_MUL_SCALAR = 0, FL_MOD_NEGATE = 1, FL_PWR = 2 };

template <int OP>
__global__ __launch_bounds__(256) void flat_kernel(const int32_t *__restrict__ in, int32_t *__restrict__ out, size_t len, int32_t s, ModK k, uint32_t r2, uint32_t r2p)
{
    const size_t nv = len / 4, stride = (size_t)gridDim.x * blockDim.x;
    auto f = [&](int32_t x) -> int32_t {
        if (OP == FL_MUL_SCALAR) return (int32_t)((uint32_t)x * (uint32_t)s);
        if (OP == FL_MOD_NEGATE) return (int32_t)((uint32_t)s - (uint32_t)x);                 // s = q (poly_32.c:149-155); any int32 x: the difference wraps
        return (int32_t)pow_mod(csub(reduce_i32(x, k), k.q), (uint32_t)s, k, r2, r2p);
    };
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < nv; i += stride) {
        const int4 v = reinterpret_cast<const int4 *>(in)[i];
        reinterpret_cast<int4 *>(out)[i] = make_int4(f(v.x), f(v.y), f(v.z), f(v.w));
    }
    if (blockIdx.x == 0 && threadIdx.x < (len & 3)) { const size_t i = nv * 4 + threadIdx.x; out[i] = f(in[i]); }
}

// ntt32_flip_generic (src/utils/arith/ntt.c:584-604; the tail of every inverse transform of the reference, also a slot of its own): v[0] <- -v[0],
// v[i] <-> v[n-i], then ONE conditional +q and ONE conditional -q per coefficient - not a full reduction: inputs are expected in (-q, 2q)
__global__ __launch_bounds__(256) void flip_kernel(const int32_t *__restrict__ in, int32_t *__restrict__ out, size_t batch, uint32_t n, int32_t q)
{
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= batch * n) return;
    const size_t b = t / n; const uint32_t i = (uint32_t)(t % n);
    int32_t x = (i == 0) ? (int32_t)(0u - (uint32_t)in[b * n]) : in[b * n + (n - i)];
    x += q * (int32_t)((uint32_t)x >> 31);
    x -= q * (int32_t)(x >= q);
    out[t] = x;
}

// poly_32.c:38-43: coefficient 0 of every polynomial
__global__ void add_scalar_kernel(int32_t *__restrict__ p, size_t batch, size_t n, const int32_t *__restrict__ scalars)
{
    const size_t b = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (b < batch) p[b * n] = (int32_t)((uint32_t)p[b * n] + (uint32_t)scalars[b]);
}

// poly_32_uniform_rand (poly_32.c:132-156): c[j] coefficients of value +-(c_len - j), one CSPRNG word per placement attempt (bit 0: sign,
// the bits above: position; an occupied position costs the word and is drawn again).  Serial by construction: one lane per polynomial,
// the polynomial as a private byte column in LDS.  used[b]: words consumed; a tape that runs out leaves status 2 and a zero polynomial.
struct UniC { int32_t c[16]; int c_len; };
__global__ __launch_bounds__(64) void uniform_kernel(size_t batch, int n, UniC C, const uint32_t *__restrict__ tapes, size_t tape_stride, const uint32_t *__restrict__ tape_lens,
                                                     int32_t *__restrict__ out, uint32_t *__restrict__ used, uint32_t *__restrict__ status)
{
    extern __shared__ int8_t ucol[];                               // n * 64
    const size_t b = (size_t)blockIdx.x * 64 + threadIdx.x;
    if (b >= batch) return;
    const uint32_t *w = tapes + b * tape_stride;
    const uint32_t len = tape_lens ? tape_lens[b] : (uint32_t)tape_stride;
    int8_t *mine = ucol + threadIdx.x;
    for (int k = 0; k < n; k++) mine[k * 64] = 0;
    uint32_t p = 0;
    bool ok = true;
    for (int j = 0; j < C.c_len && ok; j++) {
        int cnt = 0;
        while (cnt < C.c[j]) {
            if (p >= len) { ok = false; break; }
            const uint32_t x = w[p++];
            const uint32_t idx = (x >> 1) & (uint32_t)(n - 1);
            if (mine[idx * 64] == 0) { mine[idx * 64] = (int8_t)((x & 1) ? j - C.c_len : C.c_len - j); cnt++; }
        }
    }
    for (int k = 0; k < n; k++) out[b * (size_t)n + k] = ok ? (int32_t)mine[k * 64] : 0;
    if (used) used[b] = p;
    if (status) status[b] = ok ? 1u : 2u;
}

enum { RD_CMP = 0, RD_ABSMAX = 1, RD_SCALAR = 2 };

// one wavefront per polynomial
template <int OP>
__global__ __launch_bounds__(256) void reduce_kernel(const int32_t *__restrict__ a, const int32_t *__restrict__ b, int32_t *__restrict__ out, size_t batch, size_t n)
{
    const int lane = threadIdx.x & 63;
    const size_t item = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (item >= batch) return;
    const int32_t *pa = a + item * n, *pb = b ? b + item * n : nullptr;
    int32_t acc = 0; uint32_t sum = 0;
    for (size_t j = lane; j < n; j += 64) {
        if (OP == RD_CMP) acc |= pa[j] ^ pb[j];
        else if (OP == RD_ABSMAX) { const int32_t v = pa[j]; acc = max(acc, max(v, -v)); }     // vectors.c:34-40 (INT_MIN stays negative there too)
        else sum += (uint32_t)pa[j] * (uint32_t)pb[j];
    }
    int32_t r;
    if (OP == RD_CMP) r = wave_or(acc);
    else if (OP == RD_ABSMAX) r = wave_max(acc);
    else r = (int32_t)wave_add_u(sum);
    if (lane == 0) out[item] = r;
}

// invert_32 (den == nullptr: v <- v^(q-2)) / div_32 (num <- num * den^(q-2)), one wavefront per polynomial.  The reference walks
// the coefficients in order and returns SC_FUNC_FAILURE at the first one that is 0 mod q, leaving the earlier ones done and the
// rest untouched: reproduced through the wave-wide minimum index of a zero.
__global__ __launch_bounds__(256) void invert_kernel(int32_t *__restrict__ v, const int32_t *__restrict__ den, int32_t *__restrict__ rc, size_t batch, size_t n,
                                                     ModK k, uint32_t r2, uint32_t r2p)
{
    const int lane = threadIdx.x & 63;
    const size_t item = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (item >= batch) return;
    int32_t *pv = v + item * n;
    const int32_t *pd = den ? den + item * n : pv;
    uint32_t first = 0xffffffffu;
    for (size_t j = lane; j < n; j += 64)
        if (csub(reduce_i32(pd[j], k), k.q) == 0) first = min(first, (uint32_t)j);
    first = wave_min_u(first);
    for (size_t j = lane; j < n && j < first; j += 64) {
        const uint32_t inv = pow_mod(csub(reduce_i32(pd[j], k), k.q), k.q - 2, k, r2, r2p);
        if (den) {
            const uint32_t num = csub(reduce_i32(pv[j], k), k.q);
            pv[j] = (int32_t)csub(Mod<MODE_U32>::mulc(montmul(num, inv, k), r2, r2p, k), k.q);
        } else pv[j] = (int32_t)inv;
    }
    if (lane == 0) rc[item] = first != 0xffffffffu;
}

// ntt.c:381-422: v[j] = sum_k ( j < pos_k ? +t[j + n - pos_k] : -t[j - pos_k] ), exact (wrapping) integers.  One workgroup per
// polynomial, t staged in LDS.
template <typename T>
__global__ __launch_bounds__(256) void sparse_kernel(int32_t *__restrict__ out, const T *__restrict__ t, const int32_t *__restrict__ pos, size_t n, int omega)
{
    extern __shared__ int32_t lds_t[];
    const size_t item = blockIdx.x;
    const T *pt = t + item * n;
    const int32_t *pp = pos + item * (size_t)omega;
    for (size_t j = threadIdx.x; j < n; j += blockDim.x) lds_t[j] = (int32_t)pt[j];
    __syncthreads();
    for (size_t j = threadIdx.x; j < n; j += blockDim.x) {
        uint32_t acc = 0;
        for (int kx = 0; kx < omega; kx++) {
            const size_t p = (size_t)pp[kx];
            acc += (j < p) ? (uint32_t)lds_t[j + n - p] : (uint32_t)-lds_t[j - p];
        }
        out[item * n + j] = (int32_t)acc;
    }
}

template <int OP>
int run_flat(sc_hip_ctx *ctx, const ModK &k, uint32_t r2, uint32_t r2p, size_t len, const int32_t *in, int32_t s, int32_t *out, void *stream, const char *who)
{
    if (!ctx) return fail(std::string(who) + ": null context");
    if (len == 0) return SC_HIP_SUCCESS;
    if (!in || !out) return fail(std::string(who) + ": null buffer");
    if ((((uintptr_t)in | (uintptr_t)out) & 15) != 0) return fail(std::string(who) + ": buffers must be 16-byte aligned");
    HIPCHK(hipSetDevice(ctx->device));
    size_t blocks = (len / 4 + 255) / 256;
    if (blocks == 0) blocks = 1;
    const size_t cap = (size_t)ctx->cus * 8;
    if (blocks > cap) blocks = cap;
    hipLaunchKernelGGL((flat_kernel<OP>), dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, in, out, len, s, k, r2, r2p);
    HIPCHK(hipGetLastError());
    return SC_HIP_SUCCESS;
}

template <int OP>
int run_reduce(sc_hip_ctx *ctx, size_t batch, size_t n, const int32_t *a, const int32_t *b, int32_t *out, void *stream, const char *who)
{
    if (!ctx) return fail(std::string(who) + ": null context");
    if (batch == 0) return SC_HIP_SUCCESS;
    if (!a || !out || (OP != RD_ABSMAX && !b)) return fail(std::string(who) + ": null buffer");
    if (batch > 0x7fffffffu) return fail(std::string(who) + ": batch too large");
    HIPCHK(hipSetDevice(ctx->device));
    hipLaunchKernelGGL((reduce_kernel<OP>), dim3((unsigned)((batch + 3) / 4)), dim3(256), 0, (hipStream_t)stream, a, b, out, batch, n);
    HIPCHK(hipGetLastError());
    return SC_HIP_SUCCESS;
}

int run_invert(sc_hip_plan *p, size_t batch, size_t n, int32_t *v, const int32_t *den, int32_t *rc, void *stream, const char *who)
{
    if (!p) return fail(std::string(who) + ": null plan");
    if (batch == 0) return SC_HIP_SUCCESS;
    if (n == 0 || n > 0x7fffffffu) return fail(std::string(who) + ": bad length");
    if (!v || !rc) return fail(std::string(who) + ": null buffer");
    if (batch > 0x7fffffffu) return fail(std::string(who) + ": batch too large");
    HIPCHK(hipSetDevice(p->ctx->device));
    hipLaunchKernelGGL(invert_kernel, dim3((unsigned)((batch + 3) / 4)), dim3(256), 0, (hipStream_t)stream, v, den, rc, batch, n, p->k[1], p->r2, p->r2p);
    HIPCHK(hipGetLastError());
    return SC_HIP_SUCCESS;
}

template <typename T>
int run_sparse(sc_hip_ctx *ctx, size_t batch, int n, int omega, const T *t, const int32_t *pos, int32_t *out, void *stream, const char *who)
{
    if (!ctx) return fail(std::string(who) + ": null context");
    if (batch == 0) return SC_HIP_SUCCESS;
    if (!t || !pos || !out) return fail(std::string(who) + ": null buffer");
    if (n < 1 || n > 8192 || omega < 0) return fail(std::string(who) + ": n must be in [1, 8192], omega >= 0");
    if (batch > 0x7fffffffu) return fail(std::string(who) + ": batch too large");
    if ((const void *)t == (const void *)out) return fail(std::string(who) + ": the output must not alias t");
    HIPCHK(hipSetDevice(ctx->device));
    hipLaunchKernelGGL((sparse_kernel<T>), dim3((unsigned)batch), dim3(256), sizeof(int32_t) * (size_t)n, (hipStream_t)stream, out, t, pos, (size_t)n, omega);
    HIPCHK(hipGetLastError());
    return SC_HIP_SUCCESS;
}

}  // namespace

extern "C" int sc_hip_ntt_pwr(sc_hip_plan_t *p, size_t len, const int32_t *d_x, int32_t e, int32_t *d_out, void *stream)
{
    if (!p) return fail("ntt_pwr: null plan");
    if (e < 0) return fail("ntt_pwr: negative exponent");
    return run_flat<FL_PWR>(p->ctx, p->k[1], p->r2, p->r2p, len, d_x, e, d_out, stream, "ntt_pwr");
}
extern "C" int sc_hip_ntt_flip(sc_hip_plan_t *p, size_t batch, const int32_t *d_in, int32_t *d_out, void *stream)
{
    if (!p) return fail("ntt_flip: null plan");
    if (batch == 0) return SC_HIP_SUCCESS;
    if (!d_in || !d_out || d_in == d_out) return fail("ntt_flip: null buffer, or output aliasing the input (the permutation is not done in place)");
    HIPCHK(hipSetDevice(p->ctx->device));
    const size_t tot = batch * (size_t)p->n;
    hipLaunchKernelGGL(flip_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, (hipStream_t)stream, d_in, d_out, batch, (uint32_t)p->n, p->q);
    HIPCHK(hipGetLastError());
    return SC_HIP_SUCCESS;
}
extern "C" int sc_hip_ntt_invert(sc_hip_plan_t *p, size_t batch, size_t n, int32_t *d_v, int32_t *d_rc, void *stream)
{ return run_invert(p, batch, n, d_v, nullptr, d_rc, stream, "ntt_invert"); }
extern "C" int sc_hip_ntt_div(sc_hip_plan_t *p, size_t batch, size_t n, int32_t *d_num, const int32_t *d_den, int32_t *d_rc, void *stream)
{
    if (!d_den) return fail("ntt_div: null buffer");
    return run_invert(p, batch, n, d_num, d_den, d_rc, stream, "ntt_div");
}
extern "C" int sc_hip_mult_sparse(sc_hip_ctx_t *ctx, size_t batch, int n, int omega, const int32_t *d_t, const int32_t *d_pos, int32_t *d_out, void *stream)
{ return run_sparse<int32_t>(ctx, batch, n, omega, d_t, d_pos, d_out, stream, "mult_sparse"); }
extern "C" int sc_hip_mult_sparse_16(sc_hip_ctx_t *ctx, size_t batch, int n, int omega, const int16_t *d_t, const int32_t *d_pos, int32_t *d_out, void *stream)
{ return run_sparse<int16_t>(ctx, batch, n, omega, d_t, d_pos, d_out, stream, "mult_sparse_16"); }
extern "C" int sc_hip_poly_mul_scalar(sc_hip_ctx_t *ctx, size_t len, const int32_t *d_in, int32_t scalar, int32_t *d_out, void *stream)
{ ModK k = {}; return run_flat<FL_MUL_SCALAR>(ctx, k, 0, 0, len, d_in, scalar, d_out, stream, "poly_mul_scalar"); }
extern "C" int sc_hip_poly_mod_negate(sc_hip_ctx_t *ctx, size_t len, int32_t q, const int32_t *d_in, int32_t *d_out, void *stream)
{ ModK k = {}; return run_flat<FL_MOD_NEGATE>(ctx, k, 0, 0, len, d_in, q, d_out, stream, "poly_mod_negate"); }
extern "C" int sc_hip_poly_add_scalar(sc_hip_ctx_t *ctx, size_t batch, size_t n, int32_t *d_polys, const int32_t *d_scalars, void *stream)
{
    if (!ctx) return fail("poly_add_scalar: null context");
    if (batch == 0 || n == 0) return SC_HIP_SUCCESS;                       /* poly_32.c:40: nothing happens for n == 0 */
    if (!d_polys || !d_scalars) return fail("poly_add_scalar: null buffer");
    HIPCHK(hipSetDevice(ctx->device));
    hipLaunchKernelGGL(add_scalar_kernel, dim3((unsigned)((batch + 255) / 256)), dim3(256), 0, (hipStream_t)stream, d_polys, batch, n, d_scalars);
    HIPCHK(hipGetLastError());
    return SC_HIP_SUCCESS;
}
extern "C" int sc_hip_poly_cmp_not_equal(sc_hip_ctx_t *ctx, size_t batch, size_t n, const int32_t *d_a, const int32_t *d_b, int32_t *d_out, void *stream)
{ return run_reduce<RD_CMP>(ctx, batch, n, d_a, d_b, d_out, stream, "poly_cmp_not_equal"); }
extern "C" int sc_hip_vec_absmax(sc_hip_ctx_t *ctx, size_t batch, size_t n, const int32_t *d_v, int32_t *d_out, void *stream)
{ return run_reduce<RD_ABSMAX>(ctx, batch, n, d_v, nullptr, d_out, stream, "vec_absmax"); }
extern "C" int sc_hip_vec_scalar(sc_hip_ctx_t *ctx, size_t batch, size_t n, const int32_t *d_t, const int32_t *d_u, int32_t *d_out, void *stream)
{ return run_reduce<RD_SCALAR>(ctx, batch, n, d_t, d_u, d_out, stream, "vec_scalar"); }

// replaces: utils_arith_poly_t uniform_32 = poly_32_uniform_rand (src/utils/arith/poly_32.c:132-156) for `batch` polynomials, polynomial b fed by
// tape b (the prng_32 word stream of the context the reference passes in).  h_c / c_len: the coefficient-count list (BLISS nz: 2 entries, ENS-KEM 12).
extern "C" int sc_hip_poly_uniform(sc_hip_ctx_t *ctx, size_t batch, int n, const uint16_t *h_c, int c_len, const uint32_t *d_tapes, size_t tape_stride,
                                   const uint32_t *d_tape_lens, int32_t *d_out, uint32_t *d_used, uint32_t *d_status, void *stream)
{
    if (!ctx) return fail("poly_uniform: null context");
    if (batch == 0) return SC_HIP_SUCCESS;
    if (!h_c || !d_tapes || !d_out) return fail("poly_uniform: null buffer");
    if (n < 2 || n > 1024 || (n & (n - 1))) return fail("poly_uniform: n must be a power of two up to 1024 (the index is masked with n - 1)");
    if (c_len < 1 || c_len > 16) return fail("poly_uniform: 1..16 coefficient classes");
    UniC C = {};
    C.c_len = c_len;
    size_t total = 0;
    for (int j = 0; j < c_len; j++) { C.c[j] = h_c[j]; total += h_c[j]; }
    if (total > (size_t)n) return fail("poly_uniform: more non-zero coefficients than positions (the reference would not return)");
    HIPCHK(hipSetDevice(ctx->device));
    hipLaunchKernelGGL(uniform_kernel, dim3((unsigned)((batch + 63) / 64)), dim3(64), (size_t)n * 64, (hipStream_t)stream, batch, n, C, d_tapes, tape_stride, d_tape_lens,
                       d_out, d_used, d_status);
    HIPCHK(hipGetLastError());
    return SC_HIP_SUCCESS;
}

// ---- host-buffer variants (H2D, kernel, D2H; synchronous) used by the reference-shaped vtable --------------------------------

static int invert_host(sc_hip_plan_t *p, size_t batch, size_t n, int32_t *h_v, const int32_t *h_den, int32_t *h_rc, const char *who)
{
    if (!p) return fail(std::string(who) + ": null plan");
    if (batch == 0) return SC_HIP_SUCCESS;
    if (n == 0) return fail(std::string(who) + ": bad length");
    if (!h_v || !h_rc) return fail(std::string(who) + ": null host pointer");
    HIPCHK(hipSetDevice(p->ctx->device));
    const size_t bytes = batch * n * 4;
    DevBuf v, d, rc;
    if (v.get(bytes) || rc.get(batch * 4) || (h_den && d.get(bytes))) return SC_HIP_FAILURE;
    HIPCHK(sc::copy_in(v.p, h_v, bytes));
    if (h_den) HIPCHK(sc::copy_in(d.p, h_den, bytes));
    if (run_invert(p, batch, n, (int32_t *)v.p, h_den ? (const int32_t *)d.p : nullptr, (int32_t *)rc.p, nullptr, who)) return SC_HIP_FAILURE;
    HIPCHK(sc::copy_out(h_v, v.p, bytes));
    HIPCHK(sc::copy_out(h_rc, rc.p, batch * 4));
    return SC_HIP_SUCCESS;
}
extern "C" int sc_hip_ntt_invert_host(sc_hip_plan_t *p, size_t batch, size_t n, int32_t *h_v, int32_t *h_rc)
{ return invert_host(p, batch, n, h_v, nullptr, h_rc, "ntt_invert_host"); }
extern "C" int sc_hip_ntt_div_host(sc_hip_plan_t *p, size_t batch, size_t n, int32_t *h_num, const int32_t *h_den, int32_t *h_rc)
{ if (!h_den) return fail("ntt_div_host: null host pointer"); return invert_host(p, batch, n, h_num, h_den, h_rc, "ntt_div_host"); }

extern "C" int sc_hip_mult_sparse_host(sc_hip_ctx_t *ctx, size_t batch, int n, int omega, const void *h_t, int t_is_16, const int32_t *h_pos, int32_t *h_out)
{
    if (!ctx) return fail("mult_sparse_host: null context");
    if (batch == 0) return SC_HIP_SUCCESS;
    if (!h_t || !h_pos || !h_out || n < 1 || omega < 0) return fail("mult_sparse_host: bad argument");
    HIPCHK(hipSetDevice(ctx->device));
    const size_t tb = batch * (size_t)n * (t_is_16 ? 2 : 4), pb = batch * (size_t)omega * 4, ob = batch * (size_t)n * 4;
    DevBuf t, pos, out;
    if (t.get(tb) || pos.get(pb) || out.get(ob)) return SC_HIP_FAILURE;
    HIPCHK(sc::copy_in(t.p, h_t, tb));
    if (pb) HIPCHK(sc::copy_in(pos.p, h_pos, pb));
    const int rc = t_is_16 ? sc_hip_mult_sparse_16(ctx, batch, n, omega, (const int16_t *)t.p, (const int32_t *)pos.p, (int32_t *)out.p, nullptr)
                           : sc_hip_mult_sparse(ctx, batch, n, omega, (const int32_t *)t.p, (const int32_t *)pos.p, (int32_t *)out.p, nullptr);
    if (rc) return rc;
    HIPCHK(sc::copy_out(h_out, out.p, ob));
    return SC_HIP_SUCCESS;
}
