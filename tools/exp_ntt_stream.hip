// exp_ntt_stream.hip - round 4 experiment: how the n = 256 forward / fused NTT kernels should touch HBM when the working set is far past the
// Infinity Cache (fwd 0.48, fused 0.645 of 8 TB/s at 2 GiB / 3 GiB).  Variants of the I/O of ntt_kernel<8, MODE_SMAD, true, OP>
// (csrc/sc_hip.hip) over the SAME register / LDS transform (csrc/ntt_core.h); every variant's output is compared with the library's.
//   build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -I libsafecrypto_amd/csrc tools/exp_ntt_stream.hip -L libsafecrypto_amd/lib -lsc_hip -Wl,-rpath,'$ORIGIN/../libsafecrypto_amd/lib' -o tools/exp_ntt_stream
//   run:   tools/exp_ntt_stream [log2 batch = 20]
#include "engine_internal.h"
#include "ntt_core.h"
#include <stdio.h>
#include <stdlib.h>
#include <time.h>
#include <vector>
#include <algorithm>

using namespace sc;

enum { OP_FWD = 0, OP_MULNTT = 2 };
enum { V_NT_ST = 1, V_NT_LD = 2, V_WIDE_LD = 4, V_WIDE_ST = 8, V_DEEP = 16, V_WIDE_B = 32 };

struct Args {
    const int32_t *in0, *in1; int32_t *out;
    const uint32_t *zf, *zfp, *zi, *zip;
    unsigned long long batch;
    ModK k;
    uint32_t c0, c0p, c1, c1p;
};

__host__ __device__ constexpr int brv4(int r) { return ((r & 1) << 3) | ((r & 2) << 1) | ((r & 4) >> 1) | ((r & 8) >> 3); }

template <int VAR> __device__ __forceinline__ int32_t ld(const int32_t *p)
{ if constexpr (VAR & V_NT_LD) return __builtin_nontemporal_load(p); else return *p; }
template <int VAR> __device__ __forceinline__ void st(int32_t *p, int32_t v)
{ if constexpr (VAR & V_NT_ST) __builtin_nontemporal_store(v, p); else *p = v; }
template <int VAR> __device__ __forceinline__ int4 ld4(const int4 *p)
{
    if constexpr (VAR & V_NT_LD) { int4 v; v.x = __builtin_nontemporal_load(&p->x); v.y = __builtin_nontemporal_load(&p->y); v.z = __builtin_nontemporal_load(&p->z); v.w = __builtin_nontemporal_load(&p->w); return v; }
    else return *p;
}

// n = 256, SMAD.  PW = 4 polynomials per wavefront, LP = 16 lanes per polynomial.
template <int OP, int VAR>
__global__ __launch_bounds__(256) void k_var(Args a)
{
    constexpr int L = 8, MODE = MODE_SMAD;
    using Y = Lay<L>;
    constexpr int N = 256, LP = 16, PW = 4;
    constexpr int PADW = 272;                                             // wide forms: a polynomial image padded by 16 words (bank = 16 pw + lp + 16 r mod 32)
    __shared__ __attribute__((aligned(16))) uint32_t lds[4 * ((VAR & (V_WIDE_LD | V_WIDE_ST | V_WIDE_B)) ? 4 * PADW : 1024)];
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const int pw = lane / LP, lp = lane % LP;
    uint32_t *lds_wave = lds + wid * ((VAR & (V_WIDE_LD | V_WIDE_ST | V_WIDE_B)) ? 4 * PADW : 1024);
    uint32_t *lds_poly = lds_wave + pw * N;                               // the exchange image (unpadded, as in the library)
    const int blp = (int)(__builtin_bitreverse32((uint32_t)lp) >> (32 - (L - 4)));
    const ModK k = a.k;
    TwSet<L> twf, twi;
    twf.load(a.zf, a.zfp, lp);
    if (OP == OP_MULNTT) twi.load(a.zi, a.zip, lp);
    const unsigned long long groups = (a.batch + PW - 1) / PW, gstride = (unsigned long long)gridDim.x * 4;
    unsigned long long g = (unsigned long long)blockIdx.x * 4 + wid;
    if (g >= groups) return;

    int32_t nx[16];
    int4 wx[4];
    auto request_in0 = [&](unsigned long long gg) {
        const bool ok = gg < groups;
        if constexpr (VAR & V_WIDE_LD) {
            const int4 *src = reinterpret_cast<const int4 *>(a.in0 + (size_t)(ok ? gg : 0) * PW * N) + lane;
#pragma unroll
            for (int c = 0; c < 4; c++) wx[c] = ld4<VAR>(src + c * 64);
        } else {
            const int32_t *src = a.in0 + ((size_t)(ok ? gg : 0) * PW + pw) * N + lp;
#pragma unroll
            for (int r = 0; r < 16; r++) nx[r] = ld<VAR>(src + r * LP);
        }
    };
    request_in0(g);
    for (; g < groups; g += gstride) {
        const size_t base = ((size_t)g * PW + pw) * N;
        uint32_t x[16];
        if constexpr (VAR & V_WIDE_LD) {
            // the wavefront's 4 KB arrive as 4 x 16 B per lane (instruction c = polynomial c, lane l = coefficients 4l .. 4l + 3) and are transposed into the
            // round-0 layout (lane (pw, lp) holds coefficients r * 16 + lp) through the padded image
#pragma unroll
            for (int c = 0; c < 4; c++) *reinterpret_cast<int4 *>(lds_wave + c * PADW + 4 * lane) = wx[c];
            wave_lds_fence();
#pragma unroll
            for (int r = 0; r < 16; r++) nx[r] = (int32_t)lds_wave[pw * PADW + r * 16 + lp];
            wave_lds_fence();
        }
#pragma unroll
        for (int r = 0; r < 16; r++) x[r] = reduce_signed(nx[r], k);
        int32_t nb[16];
        int4 wb[4];
        if (OP == OP_MULNTT) {
            if constexpr (VAR & V_WIDE_B) {
                const int4 *srcb = reinterpret_cast<const int4 *>(a.in1 + (size_t)g * PW * N) + lane;
#pragma unroll
                for (int c = 0; c < 4; c++) wb[c] = ld4<VAR>(srcb + c * 64);
            } else {
                const int32_t *srcb = a.in1 + base + brv4(0) * LP + blp;
#pragma unroll
                for (int r = 0; r < 16; r++) nb[r] = ld<VAR>(srcb + brv4(r) * LP);
            }
        }
        if (OP == OP_FWD || (VAR & V_DEEP)) request_in0(g + gstride);
        fwd_core<L, MODE, true>(x, twf, lds_poly, lp, k);
        if (OP == OP_FWD) {
#pragma unroll
            for (int r = 0; r < 16; r++) {
                uint32_t v = x[r];
                v += k.q << 5;
                v += (uint32_t)mul_i24((int32_t)__umulhi(v, k.m32), -(int32_t)k.q);
                x[r] = csub(v, k.q);
            }
            if constexpr (VAR & V_WIDE_ST) {
                // memory order of position (lp, r) is index brv4(r) * 16 + blp: image in memory order, then 16 B per lane out
#pragma unroll
                for (int r = 0; r < 16; r++) lds_wave[pw * PADW + brv4(r) * 16 + blp] = x[r];
                wave_lds_fence();
                int4 *dst = reinterpret_cast<int4 *>(a.out + (size_t)g * PW * N) + lane;
#pragma unroll
                for (int c = 0; c < 4; c++) {
                    const int4 v = *reinterpret_cast<const int4 *>(lds_wave + c * PADW + 4 * lane);
                    if constexpr (VAR & V_NT_ST) { __builtin_nontemporal_store(v.x, &dst[c * 64].x); __builtin_nontemporal_store(v.y, &dst[c * 64].y); __builtin_nontemporal_store(v.z, &dst[c * 64].z); __builtin_nontemporal_store(v.w, &dst[c * 64].w); }
                    else dst[c * 64] = v;
                }
                wave_lds_fence();
            } else {
                int32_t *dst = a.out + base + blp;
#pragma unroll
                for (int r = 0; r < 16; r++) st<VAR>(dst + brv4(r) * LP, (int32_t)x[r]);
            }
            continue;
        }
        if constexpr (VAR & V_WIDE_B) {
#pragma unroll
            for (int c = 0; c < 4; c++) *reinterpret_cast<int4 *>(lds_wave + c * PADW + 4 * lane) = wb[c];
            wave_lds_fence();
#pragma unroll
            for (int r = 0; r < 16; r++) nb[r] = (int32_t)lds_wave[pw * PADW + brv4(r) * 16 + blp];
            wave_lds_fence();
        }
#pragma unroll
        for (int r = 0; r < 16; r++) x[r] = mulmod_ssmall(x[r], reduce_signed(nb[r], k), k);
        if (!(VAR & V_DEEP)) request_in0(g + gstride);
        inv_core<L, MODE>(x, twi, lds_poly, lp, k, a.c0, a.c0p, a.c1, a.c1p);
        if constexpr (VAR & V_WIDE_ST) {
#pragma unroll
            for (int r = 0; r < 16; r++) lds_wave[pw * PADW + r * 16 + lp] = x[r];
            wave_lds_fence();
            int4 *dst = reinterpret_cast<int4 *>(a.out + (size_t)g * PW * N) + lane;
#pragma unroll
            for (int c = 0; c < 4; c++) {
                const int4 v = *reinterpret_cast<const int4 *>(lds_wave + c * PADW + 4 * lane);
                if constexpr (VAR & V_NT_ST) { __builtin_nontemporal_store(v.x, &dst[c * 64].x); __builtin_nontemporal_store(v.y, &dst[c * 64].y); __builtin_nontemporal_store(v.z, &dst[c * 64].z); __builtin_nontemporal_store(v.w, &dst[c * 64].w); }
                else dst[c * 64] = v;
            }
            wave_lds_fence();
        } else {
            int32_t *dst = a.out + base + lp;
#pragma unroll
            for (int r = 0; r < 16; r++) st<VAR>(dst + r * LP, (int32_t)x[r]);
        }
    }
}

// a plain copy in the same shapes: what the memory system gives this traffic with no arithmetic
template <int WIDE, int NT>
__global__ __launch_bounds__(256) void k_copy(const int32_t *in0, const int32_t *in1, int32_t *out, size_t n4)
{
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
        int4 v = reinterpret_cast<const int4 *>(in0)[i];
        if (in1) { const int4 w = reinterpret_cast<const int4 *>(in1)[i]; v.x ^= w.x; v.y ^= w.y; v.z ^= w.z; v.w ^= w.w; }
        if (NT) { int4 *d = reinterpret_cast<int4 *>(out) + i; __builtin_nontemporal_store(v.x, &d->x); __builtin_nontemporal_store(v.y, &d->y); __builtin_nontemporal_store(v.z, &d->z); __builtin_nontemporal_store(v.w, &d->w); }
        else reinterpret_cast<int4 *>(out)[i] = v;
    }
}

#define CK(e) do { hipError_t e_ = (e); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #e, hipGetErrorString(e_)); exit(1); } } while (0)

template <class F> static float time_ms(F f, int iters)
{
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    f(); CK(hipDeviceSynchronize());
    std::vector<float> t;
    for (int rep = 0; rep < 5; rep++) {
        CK(hipEventRecord(e0, nullptr));
        for (int i = 0; i < iters; i++) f();
        CK(hipEventRecord(e1, nullptr)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); t.push_back(ms / iters);
    }
    std::sort(t.begin(), t.end());
    return t[2];
}

int main(int argc, char **argv)
{
    const int lb = argc > 1 ? atoi(argv[1]) : 20;
    const size_t batch = (size_t)1 << lb, N = 256, words = batch * N;
    sc_hip_ctx_t *ctx; sc_hip_plan_t *p;
    if (sc_hip_ctx_create(0, &ctx) || sc_hip_plan_create(ctx, 7681, 256, &p)) { fprintf(stderr, "%s\n", sc_hip_last_error()); return 1; }
    if (p->mode != MODE_SMAD) { fprintf(stderr, "plan is not in MODE_SMAD\n"); return 1; }
    int32_t *a, *b, *o, *ref;
    CK(hipMalloc(&a, words * 4)); CK(hipMalloc(&b, words * 4)); CK(hipMalloc(&o, words * 4)); CK(hipMalloc(&ref, words * 4));
    {
        std::vector<int32_t> h(1 << 22);
        uint64_t s = 88172645463325252ull;
        for (auto &v : h) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; v = (int32_t)(s % 7681); }
        for (size_t off = 0; off < words; off += h.size()) { CK(hipMemcpy(a + off, h.data(), h.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(b + off, h.data() + 1, (h.size() - 1) * 4, hipMemcpyHostToDevice)); }
        CK(hipMemset(b + words - 1, 0, 4));
    }
    Args A{};
    A.in0 = a; A.in1 = b; A.out = o; A.zf = p->d_tab_s; A.zfp = A.zf + N; A.zi = A.zf + 2 * N; A.zip = A.zf + 3 * N; A.batch = batch; A.k = p->k[1];
    const uint32_t *c = p->c_inv[1];
    A.c0 = c[0]; A.c0p = c[1]; A.c1 = c[2]; A.c1p = c[3];
    int cus = 256;
    { hipDeviceProp_t pr; CK(hipGetDeviceProperties(&pr, 0)); cus = pr.multiProcessorCount; }
    const size_t groups = batch / 4;
    auto check = [&](const char *what) {
        std::vector<int32_t> x(1 << 20), y(1 << 20);
        bool ok = true;
        for (size_t off : { (size_t)0, words / 2 - 12345 * 4, words - (1 << 20) }) {
            CK(hipMemcpy(x.data(), o + off, x.size() * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(y.data(), ref + off, y.size() * 4, hipMemcpyDeviceToHost));
            ok = ok && x == y;
        }
        if (!ok) printf("   !!! %s: output differs from the library's\n", what);
        return ok;
    };
    if (lb == 16) {
        // the bench's timed region at the driver's K = 20: a graph replay against plain launches, host clock around launch .. synchronize
        hipStream_t st; CK(hipStreamCreate(&st));
        auto host_us = [&](auto f) {
            std::vector<double> t;
            for (int rep = 0; rep < 9; rep++) {
                CK(hipStreamSynchronize(st));
                timespec t0, t1; clock_gettime(CLOCK_MONOTONIC, &t0);
                f(); CK(hipStreamSynchronize(st));
                clock_gettime(CLOCK_MONOTONIC, &t1);
                t.push_back((t1.tv_sec - t0.tv_sec) * 1e6 + (t1.tv_nsec - t0.tv_nsec) * 1e-3);
            }
            std::sort(t.begin(), t.end());
            return t[4];
        };
        CK(hipSetDeviceFlags(hipDeviceScheduleSpin));
        for (int steps : { 20, 200 }) {
            sc_hip_polymul_ntt_steps(p, batch, a, b, o, steps, st); CK(hipStreamSynchronize(st));
            const double tg = host_us([&]() { sc_hip_polymul_ntt_steps(p, batch, a, b, o, steps, st); });
            const double tp = host_us([&]() { for (int i = 0; i < steps; i++) sc_hip_polymul_ntt(p, batch, a, b, o, st); });
            printf("region of %3d fused steps, batch 2^16: graph replay %.1f us (%.2f us/step), plain launches %.1f us (%.2f us/step)\n", steps, tg, tg / steps, tp, tp / steps);
        }
        CK(hipStreamDestroy(st));
    }
    printf("n = 256, q = 7681, batch 2^%d (%.2f GiB per array), %d CUs; GB/s = algorithmic bytes (8n fwd, 12n fused) / median of 5 timings\n", lb, words * 4.0 / (1 << 30), cus);
    for (int op : { OP_FWD, OP_MULNTT }) {
        const double bytes = (double)batch * N * 4 * (op == OP_FWD ? 2 : 3);
        auto lib = [&]() { if (op == OP_FWD) sc_hip_ntt_fwd(p, batch, a, o, nullptr); else sc_hip_polymul_ntt(p, batch, a, b, o, nullptr); };
        float ms = 0;
        for (int mode : { 0, 2, 1 }) {
            sc_hip_bench_set_ntt_stream(mode);
            ms = time_ms(lib, 5);
            printf("%-6s library kernel, stream mode %d          %8.1f us  %7.1f GB/s  frac %.3f\n", op == OP_FWD ? "fwd" : "fused", mode, ms * 1e3, bytes / ms / 1e6, bytes / ms / 1e6 / 8000.0);
            if (mode == 0) CK(hipMemcpy(ref, o, words * 4, hipMemcpyDeviceToDevice));
            else check("library stream form");
        }
        for (int wide = 0; wide < 2; wide++) for (int nt = 0; nt < 2; nt++) {
            const size_t n4 = words / 4;
            auto cp = [&]() {
                const int32_t *i1 = op == OP_FWD ? nullptr : b;
                if (wide && nt) hipLaunchKernelGGL((k_copy<1, 1>), dim3(cus * 8), dim3(256), 0, nullptr, a, i1, o, n4);
                else hipLaunchKernelGGL((k_copy<1, 0>), dim3(cus * 8), dim3(256), 0, nullptr, a, i1, o, n4);
            };
            if (!wide) continue;
            ms = time_ms(cp, 5);
            printf("%-6s copy of the same bytes, 16 B/lane%s       %8.1f us  %7.1f GB/s  frac %.3f\n", op == OP_FWD ? "fwd" : "fused", nt ? ", nt st" : "       ", ms * 1e3, bytes / ms / 1e6, bytes / ms / 1e6 / 8000.0);
        }
#define RUN(VAR, PER_CU, label) do { \
            const int blocks_ = (PER_CU) > 0 ? (int)std::min<size_t>((groups + 3) / 4, (size_t)(PER_CU) * cus) : (int)((groups + 3) / 4); \
            auto f = [&]() { if (op == OP_FWD) hipLaunchKernelGGL((k_var<OP_FWD, VAR>), dim3(blocks_), dim3(256), 0, nullptr, A); else hipLaunchKernelGGL((k_var<OP_MULNTT, VAR>), dim3(blocks_), dim3(256), 0, nullptr, A); }; \
            CK(hipMemset(o, 0xff, words * 4)); \
            ms = time_ms(f, 5); CK(hipGetLastError()); \
            const bool ok_ = check(label); \
            printf("%-6s %-28s wg/CU %-3d  %8.1f us  %7.1f GB/s  frac %.3f %s\n", op == OP_FWD ? "fwd" : "fused", label, PER_CU, ms * 1e3, bytes / ms / 1e6, bytes / ms / 1e6 / 8000.0, ok_ ? "" : "MISMATCH"); \
        } while (0)
        for (int per_cu : { 2, 3, 4, 5, 8, 0 }) {
            if (per_cu == 2) RUN(0, 2, "as library"); if (per_cu == 3) RUN(0, 3, "as library"); if (per_cu == 4) RUN(0, 4, "as library");
            if (per_cu == 5) RUN(0, 5, "as library"); if (per_cu == 8) RUN(0, 8, "as library"); if (per_cu == 0) RUN(0, 0, "as library, wave per group");
        }
        RUN(V_NT_ST, 4, "nt stores");
        RUN(V_NT_LD, 4, "nt loads");
        RUN(V_NT_ST | V_NT_LD, 4, "nt loads + stores");
        RUN(V_WIDE_LD, 4, "16 B loads");
        RUN(V_WIDE_ST, 4, "16 B stores");
        RUN(V_WIDE_LD | V_WIDE_ST, 4, "16 B loads + stores");
        RUN(V_WIDE_LD | V_WIDE_ST, 3, "16 B loads + stores");
        RUN(V_WIDE_LD | V_WIDE_ST, 5, "16 B loads + stores");
        RUN(V_WIDE_LD | V_WIDE_ST | V_NT_ST, 4, "16 B ld + st, nt st");
        RUN(V_WIDE_LD | V_WIDE_ST | V_NT_ST | V_NT_LD, 4, "16 B ld + st, nt ld + st");
        if (op == OP_MULNTT) {
            RUN(V_WIDE_LD | V_WIDE_ST | V_WIDE_B, 4, "16 B all three");
            RUN(V_WIDE_LD | V_WIDE_ST | V_WIDE_B | V_NT_ST | V_NT_LD, 4, "16 B all three, nt ld + st");
            RUN(V_WIDE_LD | V_WIDE_ST | V_WIDE_B | V_NT_ST | V_NT_LD, 3, "16 B all three, nt ld + st");
            RUN(V_WIDE_LD | V_WIDE_ST | V_WIDE_B | V_NT_ST | V_NT_LD, 2, "16 B all three, nt ld + st");
            RUN(V_WIDE_LD | V_WIDE_ST | V_WIDE_B | V_NT_LD, 4, "16 B all three, nt ld");
            RUN(V_WIDE_LD | V_WIDE_ST | V_WIDE_B | V_NT_ST | V_NT_LD | V_DEEP, 4, "16 B all three, nt, early");
        }
        RUN(V_WIDE_LD | V_WIDE_ST | V_NT_LD, 4, "16 B ld + st, nt ld");
        RUN(V_WIDE_LD | V_WIDE_ST | V_NT_ST | V_NT_LD, 2, "16 B ld + st, nt ld + st");
        RUN(V_WIDE_LD | V_WIDE_ST | V_NT_ST | V_NT_LD, 8, "16 B ld + st, nt ld + st");
        RUN(V_DEEP, 4, "next group requested early");
        RUN(V_DEEP | V_WIDE_LD | V_WIDE_ST, 4, "16 B ld + st, early");
        RUN(V_DEEP | V_WIDE_LD | V_WIDE_ST | V_NT_ST, 4, "16 B ld + st, early, nt st");
    }
    return 0;
}
