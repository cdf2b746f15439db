// falcon_sign_dev.h - the device side of the batched Falcon signer: the hash to the ring, the fast Fourier sampler over the LDL tree in its three
// mappings (a lane per signature, eight lanes per signature, eight signatures per wavefront with the lower tree in LDS - the default), the finishing kernel
// (csrc/falcon.hip fal_sign_core queues them).  Replaces gaussian_sample_with_tree / gaussian_lattice_sample_fft
// (src/utils/arith/gpv.c:2961-3090, 2663-2714), the falcon_fft.c transforms and falcon_sig_sign (src/schemes/sig/falcon/falcon.c:808-935) around them.
// IEEE binary64 with one rounding per operation: the including file sets `#pragma clang fp contract(off)` and is compiled with -ffp-contract=off.
// Included by falcon.hip INSIDE its anonymous namespace, behind keccak_dev.h (hence no includes here); a header so that the kernels' own text can also be
// compiled as host code over tests/emu/wave_emu.h and a signing call be run lane by lane under the sanitizers (tests/test_emu_falcon_sign.py).

constexpr int HT = 64;       // hash kernel: lanes per workgroup (LDS sponge states)
constexpr int FK1 = 4;       // sub-trees of 2^FK coefficients are sampled in registers: 16 with a lane per signature,
constexpr int FK8 = 3;       // 8 when eight lanes share one (the sub-tree runs on one lane of the group)
constexpr int ST = 256;      // sampler kernel: lanes per workgroup

struct FalP { int32_t n, q, q_bits, n_bits, set; float bd; int32_t pad[2]; };   // falcon_set_t fields (falcon_params.h:38-55)

__host__ __device__ constexpr uint32_t treesize(uint32_t logn) { return (logn + 1) << logn; }    // falcon_ldl.h:56-71

// FPC_MUL (falcon_fft.c:101-117): two products then one add/sub per component, each rounded
__device__ __forceinline__ void cmul(double &dr, double &di, double ar, double ai, double br, double bi)
{
    const double r = (ar * br) - (ai * bi), i = (ar * bi) + (ai * br);
    dr = r; di = i;
}

// gaussian_cdf.c:536-553 + :639-659: the halving search returns the largest b with cdf[b] < x (0 if none); the table is
// monotone (all-ones from some index on), so that is the NUMBER of entries b >= 1 with cdf[b] < x.  The row address is
// wave-uniform, so the entries arrive through scalar loads and the count is 2 VALU instructions per entry with no dependent
// memory access (the search would be log2(size) dependent loads per sample).  Sign = LSB of x.
template <int TSL>
__device__ __forceinline__ double cdt_pick(const uint64_t *__restrict__ cdf, uint64_t x)
{
    uint32_t a = 0;
#pragma unroll
    for (int b = 1; b < (1 << TSL); b++) a += (cdf[b] < x) ? 1u : 0u;
    const int32_t v = (x & 1) ? (int32_t)a : -(int32_t)a;
    return (double)v;
}

template <int TSL>
struct Leaf {                 // leaf tables in visiting order + this signature's word stream (both read where they lie, in HBM / L2)
    const uint64_t *__restrict__ tabs;
    const uint32_t *__restrict__ tape;
    uint32_t idx;
    // the leaf's two samples (gpv.c:2671-2686: two prng_64, high word first) and on to the next leaf
    __device__ __forceinline__ void pick2(double &d0, double &d1)
    {
        const uint64_t *cdf = tabs + ((size_t)idx << TSL);
        const uint4 w = *reinterpret_cast<const uint4 *>(tape + 4 * (size_t)idx);      // tape rows are 16-byte aligned
        const uint64_t x0 = ((uint64_t)w.x << 32) | w.y, x1 = ((uint64_t)w.z << 32) | w.w;
        d0 = cdt_pick<TSL>(cdf, x0);
        d1 = cdt_pick<TSL>(cdf, x1);
        idx++;
    }
};

// ---- register-resident bottom of the tree (gpv.c:2663-2714 on 2^L coefficients) ------------------------------------

template <int L, class LF> struct FFS {
    static constexpr int N = 1 << L, HN = N / 2, QN = HN / 2;
    static __device__ __forceinline__ void split(double (&f0)[HN], double (&f1)[HN], const double (&f)[N], const double *__restrict__ gm)
    {
        if (QN == 0) { f0[0] = f[0]; f1[0] = f[HN]; }
#pragma unroll
        for (int u = 0; u < QN; u++) {
            const double ar = f[2 * u], ai = f[2 * u + HN], br = f[2 * u + 1], bi = f[2 * u + 1 + HN];
            double tr = ar + br, ti = ai + bi;
            f0[u] = tr * 0.5; f0[u + QN] = ti * 0.5;
            tr = ar - br; ti = ai - bi;
            cmul(tr, ti, tr, ti, gm[(u + HN) << 1], -gm[((u + HN) << 1) + 1]);
            f1[u] = tr * 0.5; f1[u + QN] = ti * 0.5;
        }
    }
    static __device__ __forceinline__ void merge(double (&f)[N], const double (&f0)[HN], const double (&f1)[HN], const double *__restrict__ gm)
    {
        if (QN == 0) { f[0] = f0[0]; f[HN] = f1[0]; }
#pragma unroll
        for (int u = 0; u < QN; u++) {
            const double ar = f0[u], ai = f0[u + QN];
            double br, bi;
            cmul(br, bi, f1[u], f1[u + QN], gm[(u + HN) << 1], gm[((u + HN) << 1) + 1]);
            f[2 * u] = ar + br; f[2 * u + HN] = ai + bi;
            f[2 * u + 1] = ar - br; f[2 * u + 1 + HN] = ai - bi;
        }
    }
    static __device__ __forceinline__ void run(double (&z0)[N], double (&z1)[N], const double *__restrict__ tree, const double (&t0)[N], const double (&t1)[N],
                                               const double *__restrict__ gm, LF &lf)
    {
        double a0[HN], a1[HN], w0[HN], w1[HN], tmp[N];
        split(a0, a1, t1, gm);
        FFS<L - 1, LF>::run(w0, w1, tree + N + treesize(L - 1), a0, a1, gm, lf);
        merge(z1, w0, w1, gm);
#pragma unroll
        for (int u = 0; u < HN; u++) {
            double tr = t1[u] - z1[u], ti = t1[u + HN] - z1[u + HN];
            cmul(tr, ti, tr, ti, tree[u], tree[u + HN]);
            tmp[u] = tr + t0[u]; tmp[u + HN] = ti + t0[u + HN];
        }
        split(a0, a1, tmp, gm);
        FFS<L - 1, LF>::run(w0, w1, tree + N, a0, a1, gm, lf);
        merge(z0, w0, w1, gm);
    }
};

template <class LF> struct FFS<0, LF> {
    static __device__ __forceinline__ void run(double (&z0)[1], double (&z1)[1], const double *__restrict__, const double (&t0)[1], const double (&t1)[1],
                                               const double *__restrict__, LF &lf)
    {
        double p0, p1;
        lf.pick2(p0, p1);
        z0[0] = floor(t0[0]) + p0;
        z1[0] = floor(t1[0]) + p1;
    }
};

// ---- one signature's share of the interleaved workspace ------------------------------------------------------------------
//
// G lanes work on one signature (G = 1: a lane per signature, best for large batches; G = 8: eight lanes per signature and
// eight signatures per wavefront, which cuts the length of every sweep a lone wavefront has to walk by 8 and puts 8x as many
// wavefronts on the chip - the small-batch variant).  Element e of signature b lives at ((e / G) * B + b) * G + e % G, so the
// G lanes of a group read G consecutive doubles and a wavefront reads one contiguous 512-byte row either way.
template <int G>
struct Col {
    double *p; size_t B;        // p = ws + b * G
    __device__ __forceinline__ size_t at(uint32_t e) const { return G == 1 ? (size_t)e * B : ((size_t)(e / G) * B) * G + (e % G); }
    __device__ __forceinline__ double ld(uint32_t e) const { return p[at(e)]; }
    __device__ __forceinline__ void st(uint32_t e, double v) const { p[at(e)] = v; }
};

// Lanes of a group exchange data through the workspace between sweeps: make the sweep's stores visible to the other lanes of
// the wavefront (workgroup-scope release/acquire = wait for the stores; one CU, one L1) and keep the compiler from moving
// accesses across.  Nothing to do for G = 1.
template <int G> __device__ __forceinline__ void sweep_sync()
{
    if (G > 1) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    }
}

// Every sweep below is a chain "load a few doubles - a handful of flops - store" per iteration, all through one pointer, so the
// compiler keeps the iterations in order and a lone wavefront would pay one full memory latency per iteration.  The iterations
// of one sweep touch disjoint elements, so the loads of PU iterations are issued together before any of them is consumed.
constexpr int PU = 8;
constexpr int SU = 4;     // the tree sweeps: qn >= 4 pairs at the smallest level kept in memory (l = FK + 1 >= 4)

// The transforms run R = 3 (then 2 or 1) butterfly stages per sweep: the 2^R points j0 + a*Q that those stages connect are
// loaded once, all R stages run in registers, and the points are written back - the same operations on the same operands in the
// same order per element as the reference's stage-by-stage loops (each element's value after stage s depends only on its stage
// s-1 inputs), at a third of the HBM traffic.  PG groups are in flight per lane so that a lone wavefront has 2^R * 2 * PG loads
// outstanding.

// falcon_fft.c:203-282, stages u .. u+R-1 (m = 2^u, half-span ht = hn >> u at stage u): group g -> block i1 = g / Q, j0 = i1*P*Q + g % Q
// with Q the half-span of the pass's last stage; point a sits in twiddle block (i1 << s) + (a >> (R - s)) of stage u + s.
template <int G, int R>
__device__ __forceinline__ void fft_pass(const Col<G> &c, uint32_t g, uint32_t f, uint32_t hn, int u, const double *__restrict__ gm)
{
    constexpr int P = 1 << R, PG = (R == 3) ? 2 : 4;
    const int lq = __builtin_ctz(hn) - u - (R - 1);                    // Q = hn >> (u + R - 1)
    const uint32_t Q = 1u << lq, groups = hn >> R, m = 1u << u;
    for (uint32_t g0 = 0; g0 < groups; g0 += PG * G) {
        double xr[PG][P], xi[PG][P]; uint32_t j0[PG], i1[PG];
#pragma unroll
        for (int k = 0; k < PG; k++) {
            const uint32_t grp = g0 + k * G + g;
            i1[k] = grp >> lq; j0[k] = ((i1[k] << R) << lq) + (grp & (Q - 1));
            if (G == 1 || grp < groups) {
#pragma unroll
                for (int a = 0; a < P; a++) { xr[k][a] = c.ld(f + j0[k] + (a << lq)); xi[k][a] = c.ld(f + j0[k] + (a << lq) + hn); }
            }
        }
#pragma unroll
        for (int k = 0; k < PG; k++) {
            const uint32_t grp = g0 + k * G + g;
            if (G == 1 || grp < groups) {
#pragma unroll
                for (int s = 0; s < R; s++) {
                    constexpr int dummy = 0; (void)dummy;
                    const int sp = P >> (s + 1);
#pragma unroll
                    for (int a = 0; a < P; a++) {
                        if (a & sp) continue;
                        const uint32_t tw = (((m << s) + (i1[k] << s) + (uint32_t)(a >> (R - s)))) << 1;
                        double yr, yi;
                        cmul(yr, yi, xr[k][a + sp], xi[k][a + sp], gm[tw], gm[tw + 1]);
                        const double ar = xr[k][a], ai = xi[k][a];
                        xr[k][a] = ar + yr; xi[k][a] = ai + yi;
                        xr[k][a + sp] = ar - yr; xi[k][a + sp] = ai - yi;
                    }
                }
#pragma unroll
                for (int a = 0; a < P; a++) { c.st(f + j0[k] + (a << lq), xr[k][a]); c.st(f + j0[k] + (a << lq) + hn, xi[k][a]); }
            }
        }
    }
    sweep_sync<G>();
}

template <int G>
__device__ void fft_col(const Col<G> &c, uint32_t g, uint32_t f, int logn, const double *__restrict__ gm)
{
    const uint32_t hn = (1u << logn) >> 1;
    int u = 1;
    for (; u + 2 < logn; u += 3) fft_pass<G, 3>(c, g, f, hn, u, gm);
    if (u + 1 < logn) fft_pass<G, 2>(c, g, f, hn, u, gm);
    else if (u < logn) fft_pass<G, 1>(c, g, f, hn, u, gm);
}

// falcon_fft.c:284-378 without the final scaling (the caller applies f*ni while it rounds to integers).  R stages starting at
// half-span t0 = 2^lt and table offset hm0: group g -> block iB = g >> lt, j0 = iB*P*t0 + g % t0; at stage s the pair (a, a + 2^s)
// uses the conjugate of gm[(hm0 >> s) + iB*(P >> (s+1)) + (a >> (s+1))].
template <int G, int R>
__device__ __forceinline__ void ifft_pass(const Col<G> &c, uint32_t g, uint32_t f, uint32_t hn, int lt, uint32_t hm0, const double *__restrict__ gm)
{
    constexpr int P = 1 << R, PG = (R == 3) ? 2 : 4;
    const uint32_t t0 = 1u << lt, groups = hn >> R;
    for (uint32_t g0 = 0; g0 < groups; g0 += PG * G) {
        double xr[PG][P], xi[PG][P]; uint32_t j0[PG], iB[PG];
#pragma unroll
        for (int k = 0; k < PG; k++) {
            const uint32_t grp = g0 + k * G + g;
            iB[k] = grp >> lt; j0[k] = ((iB[k] << R) << lt) + (grp & (t0 - 1));
            if (G == 1 || grp < groups) {
#pragma unroll
                for (int a = 0; a < P; a++) { xr[k][a] = c.ld(f + j0[k] + (a << lt)); xi[k][a] = c.ld(f + j0[k] + (a << lt) + hn); }
            }
        }
#pragma unroll
        for (int k = 0; k < PG; k++) {
            const uint32_t grp = g0 + k * G + g;
            if (G == 1 || grp < groups) {
#pragma unroll
                for (int s = 0; s < R; s++) {
                    const int sp = 1 << s;
#pragma unroll
                    for (int a = 0; a < P; a++) {
                        if (a & sp) continue;
                        const uint32_t tw = ((hm0 >> s) + iB[k] * (uint32_t)(P >> (s + 1)) + (uint32_t)(a >> (s + 1))) << 1;
                        const double ar = xr[k][a], ai = xi[k][a], br = xr[k][a + sp], bi = xi[k][a + sp];
                        xr[k][a] = ar + br; xi[k][a] = ai + bi;
                        const double dr = ar - br, di = ai - bi;
                        cmul(xr[k][a + sp], xi[k][a + sp], dr, di, gm[tw], -gm[tw + 1]);
                    }
                }
#pragma unroll
                for (int a = 0; a < P; a++) { c.st(f + j0[k] + (a << lt), xr[k][a]); c.st(f + j0[k] + (a << lt) + hn, xi[k][a]); }
            }
        }
    }
    sweep_sync<G>();
}

template <int G>
__device__ void ifft_col(const Col<G> &c, uint32_t g, uint32_t f, int logn, const double *__restrict__ gm)
{
    const uint32_t n = 1u << logn, hn = n >> 1;
    int left = logn - 1, lt = 0;                                      // stages to go; half-span 2^lt; first stage uses hm = n/2
    uint32_t hm = n >> 1;
    for (; left >= 3; left -= 3, lt += 3, hm >>= 3) ifft_pass<G, 3>(c, g, f, hn, lt, hm, gm);
    if (left == 2) ifft_pass<G, 2>(c, g, f, hn, lt, hm, gm);
    else if (left == 1) ifft_pass<G, 1>(c, g, f, hn, lt, hm, gm);
}

// falcon_fft.c:694-739, level l >= 4 (qn >= 4 is a multiple of SU; with G > 1 lanes beyond qn idle)
template <int G>
__device__ __forceinline__ void split_col(const Col<G> &c, uint32_t g, uint32_t f0, uint32_t f1, uint32_t f, int l, const double *__restrict__ gm)
{
    const uint32_t hn = (1u << l) >> 1, qn = hn >> 1;
    for (uint32_t u0 = 0; u0 < qn; u0 += SU * G) {
        double ar[SU], ai[SU], br[SU], bi[SU];
#pragma unroll
        for (int k = 0; k < SU; k++) {
            const uint32_t u = u0 + k * G + g;
            if (G == 1 || u < qn) { ar[k] = c.ld(f + 2 * u); ai[k] = c.ld(f + 2 * u + hn); br[k] = c.ld(f + 2 * u + 1); bi[k] = c.ld(f + 2 * u + 1 + hn); }
        }
#pragma unroll
        for (int k = 0; k < SU; k++) {
            const uint32_t u = u0 + k * G + g;
            if (G == 1 || u < qn) {
                double tr = ar[k] + br[k], ti = ai[k] + bi[k];
                c.st(f0 + u, tr * 0.5); c.st(f0 + u + qn, ti * 0.5);
                tr = ar[k] - br[k]; ti = ai[k] - bi[k];
                cmul(tr, ti, tr, ti, gm[(u + hn) << 1], -gm[((u + hn) << 1) + 1]);
                c.st(f1 + u, tr * 0.5); c.st(f1 + u + qn, ti * 0.5);
            }
        }
    }
    sweep_sync<G>();
}

// falcon_fft.c:742-785, level l >= 4
template <int G>
__device__ __forceinline__ void merge_col(const Col<G> &c, uint32_t g, uint32_t f, uint32_t f0, uint32_t f1, int l, const double *__restrict__ gm)
{
    const uint32_t hn = (1u << l) >> 1, qn = hn >> 1;
    for (uint32_t u0 = 0; u0 < qn; u0 += SU * G) {
        double ar[SU], ai[SU], cr[SU], ci[SU];
#pragma unroll
        for (int k = 0; k < SU; k++) {
            const uint32_t u = u0 + k * G + g;
            if (G == 1 || u < qn) { ar[k] = c.ld(f0 + u); ai[k] = c.ld(f0 + u + qn); cr[k] = c.ld(f1 + u); ci[k] = c.ld(f1 + u + qn); }
        }
#pragma unroll
        for (int k = 0; k < SU; k++) {
            const uint32_t u = u0 + k * G + g;
            if (G == 1 || u < qn) {
                double br, bi;
                cmul(br, bi, cr[k], ci[k], gm[(u + hn) << 1], gm[((u + hn) << 1) + 1]);
                c.st(f + 2 * u, ar[k] + br); c.st(f + 2 * u + hn, ai[k] + bi);
                c.st(f + 2 * u + 1, ar[k] - br); c.st(f + 2 * u + 1 + hn, ai[k] - bi);
            }
        }
    }
    sweep_sync<G>();
}

// Workspace layout per signature, in doubles: C0 | C1 | Z0 | Z1 | T (2n).  A node at level l (2^l coefficients) below
// the top writes its results z0, z1 into its parent's scratch and scratches in T beyond it, exactly as the reference's
// pointer arithmetic does (gpv.c:2693-2711): tmp(l) = T + 2n - 2^(l+1), z0(l) = tmp(l+1), z1(l) = tmp(l+1) + 2^l, and
// its targets t0, t1 are the two halves of the parent's z1 (first call) or z0 (second call).
struct Walk {
    uint32_t n; int logn; uint32_t ph;      // ph bit l: the node on the current path at level l is in its second half
    __device__ __forceinline__ uint32_t tmp(int l) const { return 4 * n + 2 * n - (2u << l); }
    __device__ __forceinline__ uint32_t z0(int l) const { return l == logn ? 2 * n : tmp(l + 1); }
    __device__ __forceinline__ uint32_t z1(int l) const { return l == logn ? 3 * n : tmp(l + 1) + (1u << l); }
    __device__ __forceinline__ uint32_t t0(int l) const { return l == logn ? 0 : (((ph >> (l + 1)) & 1) ? z0(l + 1) : z1(l + 1)); }
    __device__ __forceinline__ uint32_t t1(int l) const { return l == logn ? n : t0(l) + (1u << l); }
};

// gaussian_sample_with_tree, G lanes per signature.  c: int32 [batch][n]; tapes: uint32 [batch][tape_stride];
// s1 (may be null), s2: int32 [batch][n]; status[b] = 2 when the tape is shorter than 4n words (nothing is sampled then).
template <int TSL, int G>
__global__ __launch_bounds__(ST) void fal_sample_kernel(uint32_t batch, size_t B, double *__restrict__ ws, const int32_t *__restrict__ c,
                                                        const uint32_t *__restrict__ tapes, size_t tape_stride, const uint32_t *__restrict__ tape_lens,
                                                        const double *__restrict__ sk, const double *__restrict__ gm, const uint64_t *__restrict__ tabs,
                                                        int32_t *__restrict__ s1, int32_t *__restrict__ s2, uint32_t *__restrict__ status, FalP P, double ni,
                                                        const uint32_t *__restrict__ key_of, uint32_t nkeys, size_t sk_stride, size_t tab_stride, int only_deferred)
{
    constexpr int FK = (G == 1) ? FK1 : FK8;
    const uint32_t tid = blockIdx.x * ST + threadIdx.x, b = tid / G, g = tid % G;
    if (b >= batch) return;
    const uint32_t n = (uint32_t)P.n; const int logn = P.n_bits;
    const uint32_t have = tape_lens ? tape_lens[b] : (uint32_t)tape_stride;
    if (have < 4 * n) { if (g == 0) status[b] = 2; return; }
    if (only_deferred) {                           // behind the wavefront-per-8 kernel: only the signatures it handed over (status 7)
        if (status[b] != 7) return;
        if (g == 0) status[b] = 0;
    }
    if (key_of) {                                  // a key per signature out of the handle's key set: basis, tree and leaf tables are this key's
        const uint32_t kx = key_of[b];
        if (kx >= nkeys) { if (g == 0) status[b] = 6; return; }
        sk += (size_t)kx * sk_stride; tabs += (size_t)kx * tab_stride;
    }
    const Col<G> col = { ws + (size_t)b * G, B };
    const double *b00 = sk, *b01 = sk + n, *b10 = sk + 2 * n, *b11 = sk + 3 * n, *tree = sk + 4 * n;
    const uint32_t C0 = 0, C1 = n, Z0 = 2 * n, Z1 = 3 * n, hn = n >> 1;
    const int32_t *cb = c + (size_t)b * n;
    for (uint32_t i0 = 4 * g; i0 < n; i0 += 4 * G) {                  // rows of c are 16-byte aligned (n*4 bytes each)
        const int4 cv = *reinterpret_cast<const int4 *>(cb + i0);
        col.st(C0 + i0, (double)cv.x); col.st(C0 + i0 + 1, (double)cv.y); col.st(C0 + i0 + 2, (double)cv.z); col.st(C0 + i0 + 3, (double)cv.w);
    }
    sweep_sync<G>();
    fft_col<G>(col, g, C0, logn, gm);
    // c1 = c0 * b01 * (-1/q) ; c0 = c0 * b11 * (1/q)   (gpv.c:3049-3054)
    for (uint32_t u0 = 0; u0 < hn; u0 += PU * G) {
        double ar[PU], ai[PU];
#pragma unroll
        for (int k = 0; k < PU; k++) { const uint32_t u = u0 + k * G + g; ar[k] = col.ld(C0 + u); ai[k] = col.ld(C0 + u + hn); }
#pragma unroll
        for (int k = 0; k < PU; k++) {
            const uint32_t u = u0 + k * G + g;
            double r, i;
            cmul(r, i, ar[k], ai[k], b01[u], b01[u + hn]);
            col.st(C1 + u, r * -ni); col.st(C1 + u + hn, i * -ni);
            cmul(r, i, ar[k], ai[k], b11[u], b11[u + hn]);
            col.st(C0 + u, r * ni); col.st(C0 + u + hn, i * ni);
        }
    }
    sweep_sync<G>();

    // ---- fast Fourier sampling, depth first, the second-half tree (tree1) before the first (gpv.c:2693-2711)
    Leaf<TSL> lf = { tabs, tapes + (size_t)b * tape_stride, 0 };
    Walk w = { n, logn, 0 };
    uint32_t toff = 0;                           // offset of the current node's tree block
    int l = logn;
    bool entering = true;
    for (;;) {
        const uint32_t m = 1u << l, hm = m >> 1, qn = hm >> 1;
        if (entering) {
            if (l == FK) {
                // the whole sub-tree in the registers of the group's first lane (its 2 x 8 inputs are one 64-byte line each)
                if (g == 0) {
                    double t0r[1 << FK], t1r[1 << FK], z0r[1 << FK], z1r[1 << FK];
                    const uint32_t pt0 = w.t0(l), pt1 = w.t1(l), pz0 = w.z0(l), pz1 = w.z1(l);
#pragma unroll
                    for (int i = 0; i < (1 << FK); i++) { t0r[i] = col.ld(pt0 + i); t1r[i] = col.ld(pt1 + i); }
                    Leaf<TSL> mine = lf;                               // a copy: lf.idx stays wave-uniform (scalar table loads)
                    FFS<FK, Leaf<TSL>>::run(z0r, z1r, tree + toff, t0r, t1r, gm, mine);
#pragma unroll
                    for (int i = 0; i < (1 << FK); i++) { col.st(pz0 + i, z0r[i]); col.st(pz1 + i, z1r[i]); }
                }
                lf.idx += 1u << FK;
                sweep_sync<G>();
                entering = false; l++;
                continue;
            }
            split_col<G>(col, g, w.z1(l), w.z1(l) + hm, w.t1(l), l, gm);
            w.ph &= ~(1u << l);
            toff += m + treesize(l - 1); l--;
            continue;
        }
        if (!((w.ph >> l) & 1)) {
            // back from tree1: z1 = merge(child); x = (t1 - z1)*L + t0; (z0 halves) = split(x) - one sweep over pairs
            toff -= m + treesize(l - 1);
            const uint32_t ch = w.tmp(l), pz1 = w.z1(l), pz0 = w.z0(l), pt0 = w.t0(l), pt1 = w.t1(l);
            const double *L = tree + toff;
            constexpr int FU = 4;                                  // 12 loads per pair of complex slots
            for (uint32_t u0 = 0; u0 < qn; u0 += FU * G) {
                double ar[FU], ai[FU], cr[FU], ci[FU], p1[FU][4], p0[FU][4];
#pragma unroll
                for (int k = 0; k < FU; k++) {
                    const uint32_t u = u0 + k * G + g;
                    if (G == 1 || u < qn) {
                        ar[k] = col.ld(ch + u); ai[k] = col.ld(ch + u + qn); cr[k] = col.ld(ch + hm + u); ci[k] = col.ld(ch + hm + u + qn);
                        p1[k][0] = col.ld(pt1 + 2 * u); p1[k][1] = col.ld(pt1 + 2 * u + hm); p1[k][2] = col.ld(pt1 + 2 * u + 1); p1[k][3] = col.ld(pt1 + 2 * u + 1 + hm);
                        p0[k][0] = col.ld(pt0 + 2 * u); p0[k][1] = col.ld(pt0 + 2 * u + hm); p0[k][2] = col.ld(pt0 + 2 * u + 1); p0[k][3] = col.ld(pt0 + 2 * u + 1 + hm);
                    }
                }
#pragma unroll
                for (int k = 0; k < FU; k++) {
                    const uint32_t u = u0 + k * G + g;
                    if (G == 1 || u < qn) {
                        const double gr = gm[(u + hm) << 1], gi = gm[((u + hm) << 1) + 1];
                        double br, bi;
                        cmul(br, bi, cr[k], ci[k], gr, gi);
                        const double e_r = ar[k] + br, e_i = ai[k] + bi, o_r = ar[k] - br, o_i = ai[k] - bi;     // z1 at complex slots 2u, 2u+1
                        col.st(pz1 + 2 * u, e_r); col.st(pz1 + 2 * u + hm, e_i);
                        col.st(pz1 + 2 * u + 1, o_r); col.st(pz1 + 2 * u + 1 + hm, o_i);
                        double xr = p1[k][0] - e_r, xi = p1[k][1] - e_i;
                        cmul(xr, xi, xr, xi, L[2 * u], L[2 * u + hm]);
                        xr = xr + p0[k][0]; xi = xi + p0[k][1];
                        double yr = p1[k][2] - o_r, yi = p1[k][3] - o_i;
                        cmul(yr, yi, yr, yi, L[2 * u + 1], L[2 * u + 1 + hm]);
                        yr = yr + p0[k][2]; yi = yi + p0[k][3];
                        double tr = xr + yr, ti = xi + yi;
                        col.st(pz0 + u, tr * 0.5); col.st(pz0 + u + qn, ti * 0.5);
                        tr = xr - yr; ti = xi - yi;
                        cmul(tr, ti, tr, ti, gr, -gi);
                        col.st(pz0 + hm + u, tr * 0.5); col.st(pz0 + hm + u + qn, ti * 0.5);
                    }
                }
            }
            sweep_sync<G>();
            w.ph |= 1u << l;
            toff += m; l--;
            entering = true;
            continue;
        }
        // back from tree0: z0 = merge(child); this node is done
        toff -= m;
        merge_col<G>(col, g, w.z0(l), w.tmp(l), w.tmp(l) + hm, l, gm);
        w.ph &= ~(1u << l);
        if (l == logn) break;
        l++;
    }

    // ---- lattice point (gpv.c:3064-3076): c0 <- z0*b00 + z1*b10 ; c1 <- z1*b11 + z0*b01
    for (uint32_t u0 = 0; u0 < hn; u0 += PU * G) {
        double z0r[PU], z0i[PU], z1r[PU], z1i[PU];
#pragma unroll
        for (int k = 0; k < PU; k++) { const uint32_t u = u0 + k * G + g; z0r[k] = col.ld(Z0 + u); z0i[k] = col.ld(Z0 + u + hn); z1r[k] = col.ld(Z1 + u); z1i[k] = col.ld(Z1 + u + hn); }
#pragma unroll
        for (int k = 0; k < PU; k++) {
            const uint32_t u = u0 + k * G + g;
            double pr, pi, qr, qi;
            cmul(pr, pi, z0r[k], z0i[k], b00[u], b00[u + hn]);
            cmul(qr, qi, z1r[k], z1i[k], b10[u], b10[u + hn]);
            col.st(C0 + u, pr + qr); col.st(C0 + u + hn, pi + qi);
            cmul(pr, pi, z0r[k], z0i[k], b01[u], b01[u + hn]);
            cmul(qr, qi, z1r[k], z1i[k], b11[u], b11[u + hn]);
            col.st(C1 + u, qr + pr); col.st(C1 + u + hn, qi + pi);
        }
    }
    sweep_sync<G>();
    ifft_col<G>(col, g, C0, logn, gm);
    ifft_col<G>(col, g, C1, logn, gm);
    const double sc2 = ldexp(2.0, -logn);                           // falcon_fft.c:370-377
    int32_t *o1 = s1 ? s1 + (size_t)b * n : nullptr, *o2 = s2 + (size_t)b * n;
    for (uint32_t i0 = 0; i0 < n; i0 += PU * G) {
        double v0[PU], v1[PU]; int32_t cv[PU];
#pragma unroll
        for (int k = 0; k < PU; k++) { const uint32_t i = i0 + k * G + g; v0[k] = col.ld(C0 + i); v1[k] = col.ld(C1 + i); cv[k] = cb[i]; }
#pragma unroll
        for (int k = 0; k < PU; k++) {
            const uint32_t i = i0 + k * G + g;
            const long long r0 = (long long)rint(v0[k] * sc2), r1 = (long long)rint(v1[k] * sc2);
            if (o1) o1[i] = (int32_t)((long long)cv[k] - r0);
            o2[i] = (int32_t)-r1;
        }
    }
}

// ---- the production form for batches up to a few tens of thousands: 8 signatures per wavefront, the lower tree in LDS ---------------------
//
// A wavefront owns 8 signatures from beginning to end.  Their working polynomials live in HBM in the plain layout (signature b's element e at
// ws[b*6n + e]) and are touched in two ways:
//   * WAVE mode - all 64 lanes sweep the 8 signatures' polynomials as one index space (signature-major, so every access of the wavefront is
//     one contiguous run inside one signature) - for everything that is >= 64 wide: the three transforms, the basis products and the tree
//     levels above LM (qn = 2^(l-2) >= 64 pairs per sweep).
//   * GROUP mode - 8 lanes per signature, 8 signatures side by side - for the tree from level LM down, whose whole state (the children's
//     inputs and outputs of every level below LM: 3 * 2^LM doubles per signature) lives in LDS: the ~120 node visits below LM, three dependent
//     sweeps each, pay LDS latency instead of an HBM round trip, and none of their traffic reaches HBM.  Only the level-LM node's own targets
//     (read) and results (written) are in HBM.  The 8-coefficient sub-trees at the bottom run in registers, the same on every lane of a group.
// Per signature the tree therefore moves 9 rows of n doubles through HBM per level ABOVE LM only (3 levels for n = 1024, 2 for n = 512)
// instead of 7 / 6 levels.  Values are computed by the same operations in the same order as in the other mappings: the doubles are unchanged.
// Two knobs, both template parameters of the kernel (the launcher picks, fal_launch_sample):
//   LM  - the sub-tree level whose descendants live in LDS (7 or 6).  LDS per signature: 3 * 2^LM doubles.
//   SPW - signatures per wavefront (8 or 4).  Group mode always gives a signature 8 lanes; with SPW = 4 the upper half of the wavefront idles there
//         and joins in for the wave-mode sweeps.
// A lone wavefront issues one VALU instruction per ~10 cycles on this part, two on the same SIMD one per ~5 between them (tools/exp_place.hip,
// profiles/r02_valu_issue_vs_occupancy.txt), and the register sub-trees and LDS levels of this kernel are issue-bound: a second wavefront per SIMD
// is nearly free for them.  That takes a footprint of at most 20 KB of LDS (8 workgroups per CU): (8, 7) needs 37 KB - one wavefront per SIMD however
// large the batch; (8, 6) 19.5 KB.  What (8, 6) gains on the issue-bound phases it partly loses on the sweeps that go through HBM (one more level of
// them, and twice the wavefronts behind the same memory system): +16 % at large batches, -9 % at the BASELINE batch of 8192 (1024 wavefronts either way).
// SPW = 4 doubles the wavefronts of a small batch, but the group-mode phases cost a wavefront the same whether 8 or 4 groups are live, so the
// instructions per signature double with it: no gain (measured, kept behind the measurement hook).
constexpr int SPW_MAX = 8;                     // the workspace is padded to whole wavefronts of the widest variant

struct Lin {                                   // plain layout in HBM
    double *p;
    __device__ __forceinline__ double ld(uint32_t e) const { return p[e]; }
    __device__ __forceinline__ void st(uint32_t e, double v) const { p[e] = v; }
    // elements e, e + 1 for even e: one 16-byte access (every offset the sweeps pass is even, the workspace is 16-byte aligned)
    __device__ __forceinline__ double2 ld2(uint32_t e) const { return *reinterpret_cast<const double2 *>(p + e); }
    __device__ __forceinline__ void st2(uint32_t e, double a, double b) const { *reinterpret_cast<double2 *>(p + e) = make_double2(a, b); }
};
template <int SPW>
struct LdsG {                                  // LDS, group mode: element e of the lane's signature s at (e / 8) * 8 SPW + s * 8 + e % 8 (p = base + s * 8):
    double *p;                                 // the 8 lanes of a group on 8 consecutive elements + SPW groups = 8 SPW consecutive doubles
    __device__ __forceinline__ uint32_t at(uint32_t e) const { return (e >> 3) * (8 * SPW) + (e & 7); }
    __device__ __forceinline__ double ld(uint32_t e) const { return p[at(e)]; }
    __device__ __forceinline__ void st(uint32_t e, double v) const { p[at(e)] = v; }
    __device__ __forceinline__ double2 ld2(uint32_t e) const { return *reinterpret_cast<const double2 *>(p + at(e)); }
    __device__ __forceinline__ void st2(uint32_t e, double a, double b) const { *reinterpret_cast<double2 *>(p + at(e)) = make_double2(a, b); }
};

struct LdsF {                                  // LDS, wave mode: a polynomial being transformed; one double of padding after every 8 keeps the three
    double *p;                                 // access patterns of the radix-8 passes (strides 64, 8 and 1 complex points) off each other's banks
    __device__ __forceinline__ uint32_t at(uint32_t e) const { return e + (e >> 3); }
    __device__ __forceinline__ double ld(uint32_t e) const { return p[at(e)]; }
    __device__ __forceinline__ void st(uint32_t e, double v) const { p[at(e)] = v; }
};

template <int NW> __device__ __forceinline__ void group_sync();   // NW wavefronts of one workgroup hand data to each other between sweeps
__device__ __forceinline__ void wave_sync()   // the lanes of ONE wavefront hand data to each other through memory / LDS between sweeps
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}
template <> __device__ __forceinline__ void group_sync<1>() { wave_sync(); }
template <> __device__ __forceinline__ void group_sync<2>() { __syncthreads(); }       // workgroup-scope release / barrier / acquire

// The sweeps below serve both modes.  G lanes (lane index g) share NS signatures' worth of pairs: pair index v in [0, NS * qn) is pair
// u = v % qn of signature v / qn, whose elements sit `per` doubles behind the previous signature's (group mode: NS = 1, the accessor already
// points at the lane's signature).  SU pairs per lane are loaded before any is used.  GD: the pair count may not fill the last trip (only
// the 16-coefficient level in group mode); everywhere else the loads are unconditional - a guarded load is a branch with a wait inside.

// falcon_fft.c:694-739 with source and destination in (possibly) different places
template <int G, int SU, int NS, bool GD, class D, class S>
__device__ __forceinline__ void split_sw(const D &d, uint32_t f0, uint32_t f1, const S &s, uint32_t f, uint32_t g, int l, uint32_t per, const double *__restrict__ gm)
{
    const uint32_t hn = (1u << l) >> 1, qn = hn >> 1, tot = NS * qn;
    for (uint32_t v0 = 0; v0 < tot; v0 += SU * G) {
        double ar[SU], ai[SU], br[SU], bi[SU], gr[SU], gi[SU];
#pragma unroll
        for (int k = 0; k < SU; k++) {
            const uint32_t v = v0 + k * G + g, u = v & (qn - 1), so = NS == 1 ? 0u : (v >> (l - 2)) * per;
            if (!GD || v < tot) {
                const double2 re = s.ld2(so + f + 2 * u), im = s.ld2(so + f + 2 * u + hn);
                ar[k] = re.x; br[k] = re.y; ai[k] = im.x; bi[k] = im.y;
                gr[k] = gm[(u + hn) << 1]; gi[k] = gm[((u + hn) << 1) + 1];
            }
        }
#pragma unroll
        for (int k = 0; k < SU; k++) {
            const uint32_t v = v0 + k * G + g, u = v & (qn - 1), so = NS == 1 ? 0u : (v >> (l - 2)) * per;
            if (!GD || v < tot) {
                double tr = ar[k] + br[k], ti = ai[k] + bi[k];
                d.st(so + f0 + u, tr * 0.5); d.st(so + f0 + u + qn, ti * 0.5);
                tr = ar[k] - br[k]; ti = ai[k] - bi[k];
                cmul(tr, ti, tr, ti, gr[k], -gi[k]);
                d.st(so + f1 + u, tr * 0.5); d.st(so + f1 + u + qn, ti * 0.5);
            }
        }
    }
}

// falcon_fft.c:742-785
template <int G, int SU, int NS, bool GD, class D, class S>
__device__ __forceinline__ void merge_sw(const D &d, uint32_t f, const S &s, uint32_t f0, uint32_t f1, uint32_t g, int l, uint32_t per, const double *__restrict__ gm)
{
    const uint32_t hn = (1u << l) >> 1, qn = hn >> 1, tot = NS * qn;
    for (uint32_t v0 = 0; v0 < tot; v0 += SU * G) {
        double ar[SU], ai[SU], cr[SU], ci[SU], gr[SU], gi[SU];
#pragma unroll
        for (int k = 0; k < SU; k++) {
            const uint32_t v = v0 + k * G + g, u = v & (qn - 1), so = NS == 1 ? 0u : (v >> (l - 2)) * per;
            if (!GD || v < tot) {
                ar[k] = s.ld(so + f0 + u); ai[k] = s.ld(so + f0 + u + qn); cr[k] = s.ld(so + f1 + u); ci[k] = s.ld(so + f1 + u + qn);
                gr[k] = gm[(u + hn) << 1]; gi[k] = gm[((u + hn) << 1) + 1];
            }
        }
#pragma unroll
        for (int k = 0; k < SU; k++) {
            const uint32_t v = v0 + k * G + g, u = v & (qn - 1), so = NS == 1 ? 0u : (v >> (l - 2)) * per;
            if (!GD || v < tot) {
                double br, bi;
                cmul(br, bi, cr[k], ci[k], gr[k], gi[k]);
                d.st2(so + f + 2 * u, ar[k] + br, ar[k] - br);
                d.st2(so + f + 2 * u + hn, ai[k] + bi, ai[k] - bi);
            }
        }
    }
}

// back from the first child (gpv.c:2701-2707): z1 = merge(child) ; x = (t1 - z1) * L + t0 ; the second child's inputs = split(x) - one sweep.
// ch: the child's results (two halves at cho, cho + hm); t: this node's targets; zo: where z1 goes; ci: where the second child's inputs go.
template <int G, int FU, int NS, bool GD, class CH, class TT, class ZO, class CI>
__device__ __forceinline__ void mid_sw(const CH &ch, uint32_t cho, const TT &t, uint32_t pt0, uint32_t pt1, const ZO &zo, uint32_t pz1, const CI &ci, uint32_t pc,
                                       uint32_t g, int l, uint32_t per, const double *__restrict__ gm, const double *__restrict__ L)
{
    const uint32_t hm = (1u << l) >> 1, qn = hm >> 1, tot = NS * qn;
    for (uint32_t v0 = 0; v0 < tot; v0 += FU * G) {
        double ar[FU], ai[FU], cr[FU], cim[FU], p1[FU][4], p0[FU][4], gr[FU], gi[FU], lv[FU][4];
#pragma unroll
        for (int k = 0; k < FU; k++) {
            const uint32_t v = v0 + k * G + g, u = v & (qn - 1), so = NS == 1 ? 0u : (v >> (l - 2)) * per;
            if (!GD || v < tot) {
                ar[k] = ch.ld(so + cho + u); ai[k] = ch.ld(so + cho + u + qn); cr[k] = ch.ld(so + cho + hm + u); cim[k] = ch.ld(so + cho + hm + u + qn);
                const double2 a1 = t.ld2(so + pt1 + 2 * u), b1 = t.ld2(so + pt1 + 2 * u + hm), a0 = t.ld2(so + pt0 + 2 * u), b0 = t.ld2(so + pt0 + 2 * u + hm);
                p1[k][0] = a1.x; p1[k][1] = b1.x; p1[k][2] = a1.y; p1[k][3] = b1.y;
                p0[k][0] = a0.x; p0[k][1] = b0.x; p0[k][2] = a0.y; p0[k][3] = b0.y;
                gr[k] = gm[(u + hm) << 1]; gi[k] = gm[((u + hm) << 1) + 1];
                lv[k][0] = L[2 * u]; lv[k][1] = L[2 * u + hm]; lv[k][2] = L[2 * u + 1]; lv[k][3] = L[2 * u + 1 + hm];
            }
        }
#pragma unroll
        for (int k = 0; k < FU; k++) {
            const uint32_t v = v0 + k * G + g, u = v & (qn - 1), so = NS == 1 ? 0u : (v >> (l - 2)) * per;
            if (!GD || v < tot) {
                double br, bi;
                cmul(br, bi, cr[k], cim[k], gr[k], gi[k]);
                const double e_r = ar[k] + br, e_i = ai[k] + bi, o_r = ar[k] - br, o_i = ai[k] - bi;     // z1 at complex slots 2u, 2u+1
                zo.st2(so + pz1 + 2 * u, e_r, o_r); zo.st2(so + pz1 + 2 * u + hm, e_i, o_i);
                double xr = p1[k][0] - e_r, xi = p1[k][1] - e_i;
                cmul(xr, xi, xr, xi, lv[k][0], lv[k][1]);
                xr = xr + p0[k][0]; xi = xi + p0[k][1];
                double yr = p1[k][2] - o_r, yi = p1[k][3] - o_i;
                cmul(yr, yi, yr, yi, lv[k][2], lv[k][3]);
                yr = yr + p0[k][2]; yi = yi + p0[k][3];
                double tr = xr + yr, ti = xi + yi;
                ci.st(so + pc + u, tr * 0.5); ci.st(so + pc + u + qn, ti * 0.5);
                tr = xr - yr; ti = xi - yi;
                cmul(tr, ti, tr, ti, gr[k], -gi[k]);
                ci.st(so + pc + hm + u, tr * 0.5); ci.st(so + pc + hm + u + qn, ti * 0.5);
            }
        }
    }
}

// The transforms in wave mode: as fft_pass / ifft_pass above (R butterfly stages per sweep in registers), the groups of NS polynomials (`per`
// elements apart) as one index space shared by the 64 lanes, PG groups per lane in flight.  They run on two polynomials at a time staged in
// LDS (LdsF below): a transform then costs one HBM read and one HBM write of the polynomial instead of one of each per pass.
template <int R, int PG, int NS, int TPB, class A>
__device__ __forceinline__ void fft_pass_w(const A &c, uint32_t lane, uint32_t f, uint32_t per, uint32_t hn, int u, const double *__restrict__ gm)
{
    constexpr int P = 1 << R;
    const int lq = __builtin_ctz(hn) - u - (R - 1), lgr = __builtin_ctz(hn) - R;       // Q = hn >> (u + R - 1); groups per signature = hn >> R
    const uint32_t Q = 1u << lq, groups = 1u << lgr, tot = NS * groups, m = 1u << u;   // tot is a multiple of PG * 64 for n >= 512
    for (uint32_t t0 = 0; t0 < tot; t0 += PG * TPB) {
        double xr[PG][P], xi[PG][P], wr[PG][P - 1], wi[PG][P - 1]; uint32_t j0[PG];
#pragma unroll
        for (int k = 0; k < PG; k++) {
            const uint32_t tt = t0 + k * TPB + lane, t = (TPB > 64 && tt >= tot) ? 0u : tt;      // two wavefronts on a short polynomial: the spare lanes redo group 0 and store nothing
            const uint32_t grp = t & (groups - 1), i1 = grp >> lq;
            j0[k] = (t >> lgr) * per + f + ((i1 << R) << lq) + (grp & (Q - 1));
#pragma unroll
            for (int a = 0; a < P; a++) { xr[k][a] = c.ld(j0[k] + (a << lq)); xi[k][a] = c.ld(j0[k] + (a << lq) + hn); }
#pragma unroll
            for (int e = 0; e < P - 1; e++) {                                             // stage s: 2^s twiddles, block (i1 << s) + j of stage u + s
                const int s = 31 - __builtin_clz((unsigned)e + 1u), j = e + 1 - (1 << s);
                const uint32_t tw = ((m << s) + (i1 << s) + (uint32_t)j) << 1;
                wr[k][e] = gm[tw]; wi[k][e] = gm[tw + 1];
            }
        }
#pragma unroll
        for (int k = 0; k < PG; k++) {
#pragma unroll
            for (int s = 0; s < R; s++) {
                const int sp = P >> (s + 1);
#pragma unroll
                for (int a = 0; a < P; a++) {
                    if (a & sp) continue;
                    const int ti = (1 << s) - 1 + (a >> (R - s));
                    double yr, yi;
                    cmul(yr, yi, xr[k][a + sp], xi[k][a + sp], wr[k][ti], wi[k][ti]);
                    const double ar = xr[k][a], ai = xi[k][a];
                    xr[k][a] = ar + yr; xi[k][a] = ai + yi;
                    xr[k][a + sp] = ar - yr; xi[k][a + sp] = ai - yi;
                }
            }
            if (TPB == 64 || t0 + k * TPB + lane < tot) {
#pragma unroll
                for (int a = 0; a < P; a++) { c.st(j0[k] + (a << lq), xr[k][a]); c.st(j0[k] + (a << lq) + hn, xi[k][a]); }
            }
        }
    }
    group_sync<TPB / 64>();
}

template <int NS, int TPB, class A>
__device__ __forceinline__ void fft_w(const A &c, uint32_t lane, uint32_t f, uint32_t per, int logn, const double *__restrict__ gm)
{
    const uint32_t hn = (1u << logn) >> 1;
    int u = 1;
    for (; u + 2 < logn; u += 3) fft_pass_w<3, 1, NS, TPB>(c, lane, f, per, hn, u, gm);
    if (u + 1 < logn) fft_pass_w<2, 2, NS, TPB>(c, lane, f, per, hn, u, gm);
    else if (u < logn) fft_pass_w<1, 2, NS, TPB>(c, lane, f, per, hn, u, gm);
}

template <int R, int PG, int NS, int TPB, class A>
__device__ __forceinline__ void ifft_pass_w(const A &c, uint32_t lane, uint32_t f, uint32_t per, uint32_t hn, int lt, uint32_t hm0, const double *__restrict__ gm)
{
    constexpr int P = 1 << R;
    const int lgr = __builtin_ctz(hn) - R;
    const uint32_t tt0 = 1u << lt, groups = 1u << lgr, tot = NS * groups;
    for (uint32_t t0 = 0; t0 < tot; t0 += PG * TPB) {
        double xr[PG][P], xi[PG][P], wr[PG][P - 1], wi[PG][P - 1]; uint32_t j0[PG];
#pragma unroll
        for (int k = 0; k < PG; k++) {
            const uint32_t tq = t0 + k * TPB + lane, t = (TPB > 64 && tq >= tot) ? 0u : tq;
            const uint32_t grp = t & (groups - 1), iB = grp >> lt;
            j0[k] = (t >> lgr) * per + f + ((iB << R) << lt) + (grp & (tt0 - 1));
#pragma unroll
            for (int a = 0; a < P; a++) { xr[k][a] = c.ld(j0[k] + (a << lt)); xi[k][a] = c.ld(j0[k] + (a << lt) + hn); }
#pragma unroll
            for (int e = 0; e < P - 1; e++) {                                             // stage s: P >> (s+1) twiddles, stored from the back (stage R-1 first)
                const int sr = 31 - __builtin_clz((unsigned)e + 1u), s = R - 1 - sr, j = e + 1 - (1 << sr);
                const uint32_t tw = ((hm0 >> s) + iB * (uint32_t)(P >> (s + 1)) + (uint32_t)j) << 1;
                wr[k][e] = gm[tw]; wi[k][e] = gm[tw + 1];
            }
        }
#pragma unroll
        for (int k = 0; k < PG; k++) {
#pragma unroll
            for (int s = 0; s < R; s++) {
                const int sp = 1 << s;
#pragma unroll
                for (int a = 0; a < P; a++) {
                    if (a & sp) continue;
                    const int ti = (1 << (R - 1 - s)) - 1 + (a >> (s + 1));
                    const double ar = xr[k][a], ai = xi[k][a], br = xr[k][a + sp], bi = xi[k][a + sp];
                    xr[k][a] = ar + br; xi[k][a] = ai + bi;
                    const double dr = ar - br, di = ai - bi;
                    cmul(xr[k][a + sp], xi[k][a + sp], dr, di, wr[k][ti], -wi[k][ti]);
                }
            }
            if (TPB == 64 || t0 + k * TPB + lane < tot) {
#pragma unroll
                for (int a = 0; a < P; a++) { c.st(j0[k] + (a << lt), xr[k][a]); c.st(j0[k] + (a << lt) + hn, xi[k][a]); }
            }
        }
    }
    group_sync<TPB / 64>();
}

template <int NS, int TPB, class A>
__device__ __forceinline__ void ifft_w(const A &c, uint32_t lane, uint32_t f, uint32_t per, int logn, const double *__restrict__ gm)
{
    const uint32_t n = 1u << logn, hn = n >> 1;
    int left = logn - 1, lt = 0;
    uint32_t hm = n >> 1;
    for (; left >= 3; left -= 3, lt += 3, hm >>= 3) ifft_pass_w<3, 1, NS, TPB>(c, lane, f, per, hn, lt, hm, gm);
    if (left == 2) ifft_pass_w<2, 2, NS, TPB>(c, lane, f, per, hn, lt, hm, gm);
    else if (left == 1) ifft_pass_w<1, 2, NS, TPB>(c, lane, f, per, hn, lt, hm, gm);
}

// Leaf sampling for the group mode.  The 8 leaves of a register sub-tree need 8 CDF tables (the same for every signature) and 8 x 4 tape
// words per signature; both are fetched one sub-tree AHEAD (a 16-byte load or two per lane, issued when the previous sub-tree ends and
// landed long before this one starts) and parked in LDS.  The samples depend on nothing the walk computes, so all 16 of a sub-tree are
// drawn before its recursion starts: the compare-count of cdt_pick is shared by the 8 lanes of the group - lane g takes entries g, g + 8,
// ... - and the partial counts of both samples are added across the group in one register with three DPP steps; every lane ends up with
// the same samples, and the recursion (the same on every lane of the group) finds them in registers.
template <int TSL>
struct LeafPre {
    double d[16];
    int k;
    __device__ __forceinline__ void draw(const uint64_t *tab, const uint32_t *tp, uint32_t g)       // tab: LDS [8][2^TSL]; tp: LDS, this signature's [8][4]
    {
        constexpr int PER = (1 << TSL) / 8;
#pragma unroll
        for (int lf = 0; lf < 8; lf++) {
            const uint4 w = *reinterpret_cast<const uint4 *>(tp + 4 * lf);
            const uint64_t x0 = ((uint64_t)w.x << 32) | w.y, x1 = ((uint64_t)w.z << 32) | w.w;
            const uint64_t *row = tab + ((size_t)lf << TSL) + g;
            uint32_t cnt = 0;
#pragma unroll
            for (int j = 0; j < PER; j++) {
                const uint64_t e = row[8 * j];
                const bool live = j > 0 || g != 0;                       // entry 0 is not part of the search (gaussian_cdf.c:536-553: b starts at 1)
                cnt += ((live && e < x0) ? 1u : 0u) + ((live && e < x1) ? 0x10000u : 0u);
            }
            cnt += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)cnt, 0xB1, 0xF, 0xF, true);      // quad_perm [1,0,3,2]
            cnt += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)cnt, 0x4E, 0xF, 0xF, true);      // quad_perm [2,3,0,1]
            cnt += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)cnt, 0x141, 0xF, 0xF, true);     // row_half_mirror: lane i <-> 7 - i of each 8
            const int32_t a0 = (int32_t)(cnt & 0xFFFFu), a1 = (int32_t)(cnt >> 16);
            d[2 * lf] = (double)((x0 & 1) ? a0 : -a0);
            d[2 * lf + 1] = (double)((x1 & 1) ? a1 : -a1);
        }
        k = 0;
    }
    __device__ __forceinline__ void take(const double *p)             // the 16 samples another wavefront drew (LDS)
    {
#pragma unroll
        for (int i = 0; i < 16; i += 2) { const double2 v = *reinterpret_cast<const double2 *>(p + i); d[i] = v.x; d[i + 1] = v.y; }
        k = 0;
    }
    __device__ __forceinline__ void pick2(double &d0, double &d1)
    {
        // k is a compile-time constant at every use (the recursion is fully unrolled)
        d0 = d[2 * k]; d1 = d[2 * k + 1];
        k++;
    }
};

template <int LM>
struct WalkL {
    static constexpr int LDS_SIG = 3 << LM;   // doubles of LDS per signature: X (2^LM: the level-LM node's children's inputs) | scratch (2 * 2^LM)
    uint32_t n; int logn; uint32_t ph;      // ph bit l: the node on the current path at level l is in its second half
    // HBM offsets inside a signature's 6n doubles (C0 | C1 | Z0 | Z1 | T), levels >= LM: the reference's own pointer layout (gpv.c:2693-2711)
    __device__ __forceinline__ uint32_t tmp(int l) const { return 6 * n - (2u << l); }
    __device__ __forceinline__ uint32_t z0(int l) const { return l == logn ? 2 * n : tmp(l + 1); }
    __device__ __forceinline__ uint32_t z1(int l) const { return l == logn ? 3 * n : tmp(l + 1) + (1u << l); }
    __device__ __forceinline__ uint32_t t0(int l) const { return l == logn ? 0 : (((ph >> (l + 1)) & 1) ? z0(l + 1) : z1(l + 1)); }
    __device__ __forceinline__ uint32_t t1(int l) const { return l == logn ? n : t0(l) + (1u << l); }
    // LDS offsets inside a signature's LDS_SIG doubles: X = [0, 2^LM) holds the inputs of the level-LM node's current child; the scratch
    // behind it is the same stack as T above, for the levels below LM
    __device__ __forceinline__ uint32_t ltmp(int l) const { return (uint32_t)LDS_SIG - (2u << l); }                 // l <= LM
    __device__ __forceinline__ uint32_t lz0(int l) const { return ltmp(l + 1); }                                   // l < LM
    __device__ __forceinline__ uint32_t lz1(int l) const { return ltmp(l + 1) + (1u << l); }
    __device__ __forceinline__ uint32_t lt0(int l) const { return l == LM - 1 ? 0u : (((ph >> (l + 1)) & 1) ? lz0(l + 1) : lz1(l + 1)); }
    __device__ __forceinline__ uint32_t lt1(int l) const { return lt0(l) + (1u << l); }
};

// gaussian_sample_with_tree, SPW signatures per wavefront (one wavefront per workgroup).  ws: 6n doubles per signature, padded to a multiple of 8
// signatures.  tape_stride >= 4n (the launcher answers shorter strides itself): a tape that is short is still read to 4n words - inside its
// own row - and its signature reports status 2 and writes nothing.
// PROF (measurement builds only, sc_hip_falcon_sample_profile): each wavefront adds the shader cycles it spent in the six phases to prof[0..5].
// NW = 2 (SPW = 8, LM = 7 only): the workgroup is TWO wavefronts on the same 8 signatures - the wave-mode sweeps and the transforms are shared by 128
// lanes, group mode gives a signature 16 lanes (8 in each wavefront), the register sub-trees run on the first wavefront while the second waits.  A
// wavefront alone on a SIMD issues one instruction per ~10 cycles, so at the BASELINE batch (8192 signatures = 1024 workgroups) the second wavefront
// of every workgroup lands on a SIMD that would otherwise idle and halves the instruction stream of everything that is not a register sub-tree.
template <int TSL, bool PROF, int SPW, int LM, int NW>
__global__ __launch_bounds__(64 * NW) void fal_sample_w8_kernel(uint32_t batch, double *__restrict__ ws, const int32_t *__restrict__ c,
                                                           const uint32_t *__restrict__ tapes, size_t tape_stride, const uint32_t *__restrict__ tape_lens,
                                                           const double *__restrict__ sk, const double *__restrict__ gm, const uint64_t *__restrict__ tabs,
                                                           int32_t *__restrict__ s1, int32_t *__restrict__ s2, uint32_t *__restrict__ status, FalP P, double ni,
                                                           unsigned long long *__restrict__ prof,
                                                           const uint32_t *__restrict__ key_of, uint32_t nkeys, size_t sk_stride, size_t tab_stride)
{
    constexpr int LDS_SIG = WalkL<LM>::LDS_SIG;
    constexpr int POLY = 1024 + 128;                                   // a padded polynomial (LdsF) for n <= 1024
    constexpr int NSF = (SPW * LDS_SIG >= 2 * POLY) ? 2 : 1;           // polynomials transformed side by side
    constexpr int LDSD = SPW * LDS_SIG > NSF * POLY ? SPW * LDS_SIG : NSF * POLY;
    constexpr int SU_LM = LM >= 6 ? 2 : 1;
    constexpr int TPB = 64 * NW, GL = 8 * NW;                          // lanes of the workgroup; lanes per signature in group mode
    static_assert(NW == 1 || (NW == 2 && SPW == 8 && LM == 7), "two wavefronts per workgroup: 8 signatures, 128-coefficient sub-trees in LDS");
    __shared__ __attribute__((aligned(16))) double lds[LDSD];
    __shared__ uint64_t stab[8 << TSL];                                // the current register sub-tree's 8 leaf tables
    __shared__ __attribute__((aligned(16))) uint32_t stape[SPW * 32];  // and its 8 x 4 tape words per signature
    __shared__ double stree[(LM + 1) << LM];                           // the LDL tree block of the current level-LM node (falcon_ldl.h:56-71: treesize(LM) doubles)
    __shared__ double sgm[2 << LM];                                    // the roots the levels <= LM use: complex entries [0, 2^LM)
    __shared__ __attribute__((aligned(16))) double ssmp[NW == 2 ? 2 * SPW * 16 : 2];   // NW = 2: the 16 samples per signature of the current and the next register sub-tree
    unsigned long long pc[6] = { 0, 0, 0, 0, 0, 0 }, tprev = 0;
    if (PROF) tprev = __builtin_amdgcn_s_memtime();
#define FAL_STAMP(ph) do { if (PROF) { const unsigned long long tn_ = __builtin_amdgcn_s_memtime(); pc[ph] += tn_ - tprev; tprev = tn_; } } while (0)
    constexpr int FK = FK8;
    const uint32_t lane = threadIdx.x, wv_id = lane >> 6, l64 = lane & 63, g = (l64 & 7) + 8 * wv_id;       // lane: index in the workgroup
    const bool act = l64 < 8 * SPW;                                    // group mode: the lanes that have a signature (all of them for SPW = 8)
    const bool w0 = wv_id == 0;
    const bool smp = wv_id == NW - 1;                                  // the wavefront that fetches leaf tables / tape words and (NW = 2) draws the samples
    const uint32_t sg = act ? l64 >> 3 : 0;
    const uint32_t sig0 = blockIdx.x * SPW;
    const uint32_t n = (uint32_t)P.n, hn = n >> 1; const int logn = P.n_bits;
    const uint32_t per = 6 * n;
    const Lin wv = { ws + (size_t)sig0 * per };                        // wave mode: the SPW signatures, `per` doubles apart
    const uint32_t C0 = 0, C1 = n, Z0 = 2 * n, Z1 = 3 * n;
    // group mode: this lane's signature.  A signature past the end of the batch is computed on item 0's inputs (in its own workspace slot).
    const uint32_t mysig = sig0 + sg;
    const bool inb = mysig < batch;
    const uint32_t item = inb ? mysig : 0;
    // a key set: this form serves a workgroup whose signatures all name the same (valid) key - the leaf tables and the lower tree staged in LDS are one
    // key's.  Any other workgroup hands its signatures (status 7) to the 8-lanes-per-signature kernel the launcher runs behind this one.
    if (key_of) {
        const uint32_t k0 = key_of[sig0 < batch ? sig0 : batch - 1];
        const bool odd = act && inb && key_of[mysig] != k0;
        const bool defer = (NW == 2 ? (bool)__syncthreads_or(odd ? 1 : 0) : __ballot(odd) != 0) || k0 >= nkeys;
        if (defer) { if (act && inb && g == 0) status[mysig] = 7; return; }
        sk += (size_t)k0 * sk_stride; tabs += (size_t)k0 * tab_stride;
    }
    const double *b00 = sk, *b01 = sk + n, *b10 = sk + 2 * n, *b11 = sk + 3 * n, *tree = sk + 4 * n;
    const bool shortt = act && inb && tape_lens && tape_lens[mysig] < 4 * n;
    if (shortt && g == 0) status[mysig] = 2;
    const uint64_t runmask = __ballot(act && inb && !shortt);           // bit 8k: signature k of this wavefront writes its results
    const Lin mine = { wv.p + (size_t)sg * per };
    const LdsG<SPW> lg = { lds + sg * 8 };
    for (uint32_t i = lane; i < (2u << LM); i += TPB) sgm[i] = gm[i];
    uint32_t toffLM = 0;                                               // tree offset of the current level-LM node
    const int ln = logn;                                               // log2 of the elements per signature in the flattened loops

    // ---- FFT(c) and the two target vectors (gpv.c:3040-3054): NSF signatures at a time through LDS (n doubles each, padded)
    const LdsF lf2 = { lds };
    for (int pr = 0; pr < SPW; pr += NSF) {
        constexpr int JN = 4 / NW;                                     // 16-byte loads per lane and trip
        for (uint32_t t0 = 0; t0 < NSF * n; t0 += 4 * JN * TPB) {
            int4 cv[JN];
#pragma unroll
            for (int j = 0; j < JN; j++) {
                const uint32_t t = t0 + (j * TPB + lane) * 4, b = sig0 + pr + (t >> ln);
                cv[j] = *reinterpret_cast<const int4 *>(c + (size_t)(b < batch ? b : 0) * n + (t & (n - 1)));
            }
#pragma unroll
            for (int j = 0; j < JN; j++) {
                const uint32_t t = t0 + (j * TPB + lane) * 4;
                if (NSF == 2 || t < n) { lf2.st(t, (double)cv[j].x); lf2.st(t + 1, (double)cv[j].y); lf2.st(t + 2, (double)cv[j].z); lf2.st(t + 3, (double)cv[j].w); }
            }
        }
        group_sync<NW>();
        fft_w<NSF, TPB>(lf2, lane, 0, n, logn, gm);
        for (uint32_t t0 = 0; t0 < NSF * hn; t0 += JN * TPB) {
#pragma unroll
            for (int j = 0; j < JN; j++) {
                const uint32_t t = t0 + j * TPB + lane, k = t >> (ln - 1), u = t & (hn - 1), e = (pr + k) * per + u;
                const double ar = lf2.ld(k * n + u), ai = lf2.ld(k * n + u + hn);
                double r, i;
                cmul(r, i, ar, ai, b01[u], b01[u + hn]);
                wv.st(e + C1, r * -ni); wv.st(e + C1 + hn, i * -ni);
                cmul(r, i, ar, ai, b11[u], b11[u + hn]);
                wv.st(e + C0, r * ni); wv.st(e + C0 + hn, i * ni);
            }
        }
        group_sync<NW>();
    }
    FAL_STAMP(0);

    // ---- fast Fourier sampling, depth first, the second-half tree (tree1) before the first (gpv.c:2693-2711)
    constexpr int TPL = (1 << TSL) / 8;                                // table entries per lane of a sub-tree's 8 tables
    uint64_t ptab[TPL]; uint4 ptape = make_uint4(0, 0, 0, 0);
    const uint32_t *mytape = tapes + (size_t)item * tape_stride;
    uint32_t leaf = 0;                                                 // first leaf of the next register sub-tree (wave-uniform)
    auto prefetch = [&](uint32_t first) {
        if (first < n && smp) {
            const uint64_t *src = tabs + ((size_t)first << TSL) + l64 * TPL;
#pragma unroll
            for (int i = 0; i < TPL; i += 2) { const ulonglong2 v = *reinterpret_cast<const ulonglong2 *>(src + i); ptab[i] = v.x; ptab[i + 1] = v.y; }
            if (act) ptape = *reinterpret_cast<const uint4 *>(mytape + 4 * (size_t)(first + (l64 & 7)));
        }
    };
    // NW = 2: the second wavefront draws the samples, one register sub-tree AHEAD of the first wavefront's recursion (they depend on nothing the walk
    // computes): tables and tape words of sub-tree i + 1 -> LDS -> 16 samples per signature -> ssmp[(i + 1) & 1], while the first wavefront runs sub-tree i
    // on ssmp[i & 1].  The recursion then has half the instructions of a sub-tree visit, the other half runs beside it on another SIMD.
    auto draw_ahead = [&](uint32_t sub) {                              // sampling wavefront only: ptab / ptape hold sub-tree `sub`
#pragma unroll
        for (int i = 0; i < TPL; i++) stab[l64 * TPL + i] = ptab[i];
        *reinterpret_cast<uint4 *>(stape + l64 * 4) = ptape;
        wave_sync();
        LeafPre<TSL> lfl;
        lfl.draw(stab, stape + sg * 32, l64 & 7);
        if ((l64 & 7) == 0) {
            double *dst = ssmp + ((sub & 1) * SPW + sg) * 16;
#pragma unroll
            for (int i = 0; i < 16; i += 2) *reinterpret_cast<double2 *>(dst + i) = make_double2(lfl.d[i], lfl.d[i + 1]);
        }
        wave_sync();                                                   // stab / stape are free again
    };
    prefetch(0);
    if (NW == 2 && smp) { draw_ahead(0); prefetch(1u << FK); }
    uint32_t sub = 0;                                                  // index of the next register sub-tree
    WalkL<LM> w = { n, logn, 0 };
    uint32_t toff = 0;
    int l = logn;
    bool entering = true;
    for (;;) {
        const uint32_t m = 1u << l, hm = m >> 1;
        if (entering) {
            if (l == FK) {
                // an 8-coefficient sub-tree in registers; every lane of a group computes the same thing from the same (broadcast) LDS words
                if (NW == 1) {
#pragma unroll
                    for (int i = 0; i < TPL; i++) stab[lane * TPL + i] = ptab[i];
                    if (act) *reinterpret_cast<uint4 *>(stape + lane * 4) = ptape;
                    wave_sync();
                }
                if (act && w0) {
                    double t0r[1 << FK], t1r[1 << FK], z0r[1 << FK], z1r[1 << FK];
                    const uint32_t pt0 = w.lt0(l), pt1 = w.lt1(l), pz0 = w.lz0(l), pz1 = w.lz1(l);
#pragma unroll
                    for (int i = 0; i < (1 << FK); i++) { t0r[i] = lg.ld(pt0 + i); t1r[i] = lg.ld(pt1 + i); }
                    LeafPre<TSL> lfl;
                    if (NW == 1) lfl.draw(stab, stape + sg * 32, g);
                    else lfl.take(ssmp + ((sub & 1) * SPW + sg) * 16);       // written during the previous visit, before its closing barrier
                    FFS<FK, LeafPre<TSL>>::run(z0r, z1r, stree + (toff - toffLM), t0r, t1r, gm, lfl);
                    if (g == 0) {
#pragma unroll
                        for (int i = 0; i < (1 << FK); i++) { lg.st(pz0 + i, z0r[i]); lg.st(pz1 + i, z1r[i]); }
                    }
                }
                leaf += 1u << FK; sub++;
                if (NW == 1) prefetch(leaf);
                else if (smp && leaf < n) { draw_ahead(sub); prefetch(leaf + (1u << FK)); }      // the samples of the NEXT visit
                group_sync<NW>();
                FAL_STAMP(4);
                entering = false; l++;
                continue;
            }
            if (l > LM) split_sw<TPB, 2, SPW, false>(wv, w.z1(l), w.z1(l) + hm, wv, w.t1(l), lane, l, per, gm);
            else {
                if (l == LM) {
                    toffLM = toff;
                    for (uint32_t i = lane * 2; i < treesize(LM); i += 2 * TPB) *reinterpret_cast<double2 *>(stree + i) = *reinterpret_cast<const double2 *>(tree + toff + i);
                }
                if (act) {
                    // pairs per sweep 2^(l-2): 32, 16, 8, 4 for l = 7..4; GL lanes take SU each, guarded where the lanes outnumber the pairs
                    if (l == LM) split_sw<GL, SU_LM, 1, false>(lg, 0u, hm, mine, w.t1(l), g, l, per, sgm);
                    else if (l >= 6) split_sw<GL, 2 / NW, 1, false>(lg, w.lz1(l), w.lz1(l) + hm, lg, w.lt1(l), g, l, per, sgm);
                    else if (l == 5) split_sw<GL, 1, 1, NW == 2>(lg, w.lz1(l), w.lz1(l) + hm, lg, w.lt1(l), g, l, per, sgm);
                    else split_sw<GL, 1, 1, true>(lg, w.lz1(l), w.lz1(l) + hm, lg, w.lt1(l), g, l, per, sgm);
                }
            }
            group_sync<NW>();
            FAL_STAMP(l > LM ? 1 : l == LM ? 2 : 3);
            w.ph &= ~(1u << l);
            toff += m + treesize(l - 1); l--;
            continue;
        }
        if (!((w.ph >> l) & 1)) {
            toff -= m + treesize(l - 1);
            const double *L = tree + toff;
            const double *LL = stree + (toff - toffLM);                // the same block in LDS (levels <= LM)
            if (l > LM) mid_sw<TPB, 2, SPW, false>(wv, w.tmp(l), wv, w.t0(l), w.t1(l), wv, w.z1(l), wv, w.z0(l), lane, l, per, gm, L);
            else if (act) {
                if (l == LM) mid_sw<GL, SU_LM, 1, false>(lg, w.ltmp(l), mine, w.t0(l), w.t1(l), mine, w.z1(l), lg, 0u, g, l, per, sgm, LL);
                else if (l >= 6) mid_sw<GL, 2 / NW, 1, false>(lg, w.ltmp(l), lg, w.lt0(l), w.lt1(l), lg, w.lz1(l), lg, w.lz0(l), g, l, per, sgm, LL);
                else if (l == 5) mid_sw<GL, 1, 1, NW == 2>(lg, w.ltmp(l), lg, w.lt0(l), w.lt1(l), lg, w.lz1(l), lg, w.lz0(l), g, l, per, sgm, LL);
                else mid_sw<GL, 1, 1, true>(lg, w.ltmp(l), lg, w.lt0(l), w.lt1(l), lg, w.lz1(l), lg, w.lz0(l), g, l, per, sgm, LL);
            }
            group_sync<NW>();
            FAL_STAMP(l > LM ? 1 : l == LM ? 2 : 3);
            w.ph |= 1u << l;
            toff += m; l--;
            entering = true;
            continue;
        }
        toff -= m;
        if (l > LM) merge_sw<TPB, 2, SPW, false>(wv, w.z0(l), wv, w.tmp(l), w.tmp(l) + hm, lane, l, per, gm);
        else if (act) {
            if (l == LM) merge_sw<GL, SU_LM, 1, false>(mine, w.z0(l), lg, w.ltmp(l), w.ltmp(l) + hm, g, l, per, sgm);
            else if (l >= 6) merge_sw<GL, 2 / NW, 1, false>(lg, w.lz0(l), lg, w.ltmp(l), w.ltmp(l) + hm, g, l, per, sgm);
            else if (l == 5) merge_sw<GL, 1, 1, NW == 2>(lg, w.lz0(l), lg, w.ltmp(l), w.ltmp(l) + hm, g, l, per, sgm);
            else merge_sw<GL, 1, 1, true>(lg, w.lz0(l), lg, w.ltmp(l), w.ltmp(l) + hm, g, l, per, sgm);
        }
        group_sync<NW>();
        FAL_STAMP(l > LM ? 1 : l == LM ? 2 : 3);
        w.ph &= ~(1u << l);
        if (l == logn) break;
        l++;
    }

    // ---- lattice point (gpv.c:3064-3076): c0 <- z0*b00 + z1*b10 ; c1 <- z1*b11 + z0*b01, the inverse transforms and the roundings: one
    // signature at a time, its two polynomials side by side in LDS (NSF = 2) or one after the other
    const double sc2 = ldexp(2.0, -logn);                           // falcon_fft.c:370-377
    for (int k = 0; k < SPW; k++) {
        for (int half = 0; half < (NSF == 2 ? 1 : 2); half++) {
            constexpr int JF = 4 / NW;
            for (uint32_t u0 = 0; u0 < hn; u0 += JF * TPB) {
                double z0r[JF], z0i[JF], z1r[JF], z1i[JF];
#pragma unroll
                for (int j = 0; j < JF; j++) {
                    const uint32_t e = k * per + u0 + j * TPB + lane;
                    z0r[j] = wv.ld(e + Z0); z0i[j] = wv.ld(e + Z0 + hn); z1r[j] = wv.ld(e + Z1); z1i[j] = wv.ld(e + Z1 + hn);
                }
#pragma unroll
                for (int j = 0; j < JF; j++) {
                    const uint32_t u = u0 + j * TPB + lane;
                    double pr, pi, qr, qi;
                    if (NSF == 2 || half == 0) {
                        cmul(pr, pi, z0r[j], z0i[j], b00[u], b00[u + hn]);
                        cmul(qr, qi, z1r[j], z1i[j], b10[u], b10[u + hn]);
                        lf2.st(u, pr + qr); lf2.st(u + hn, pi + qi);
                    }
                    if (NSF == 2 || half == 1) {
                        const uint32_t o = NSF == 2 ? n : 0u;
                        cmul(pr, pi, z0r[j], z0i[j], b01[u], b01[u + hn]);
                        cmul(qr, qi, z1r[j], z1i[j], b11[u], b11[u + hn]);
                        lf2.st(o + u, qr + pr); lf2.st(o + u + hn, qi + pi);
                    }
                }
            }
            group_sync<NW>();
            ifft_w<NSF, TPB>(lf2, lane, 0, n, logn, gm);
            if ((runmask >> (8 * k)) & 1) {
                const size_t ob = (size_t)(sig0 + k) * n;
                for (uint32_t i0 = 0; i0 < n; i0 += JF * TPB) {
                    int32_t cv[JF];
#pragma unroll
                    for (int j = 0; j < JF; j++) cv[j] = (NSF == 2 || half == 0) ? c[ob + i0 + j * TPB + lane] : 0;
#pragma unroll
                    for (int j = 0; j < JF; j++) {
                        const uint32_t i = i0 + j * TPB + lane;
                        if (NSF == 2 || half == 0) { const long long r0 = (long long)rint(lf2.ld(i) * sc2); if (s1) s1[ob + i] = (int32_t)((long long)cv[j] - r0); }
                        if (NSF == 2 || half == 1) { const long long r1 = (long long)rint(lf2.ld((NSF == 2 ? n : 0u) + i) * sc2); s2[ob + i] = (int32_t)-r1; }
                    }
                }
            }
            group_sync<NW>();
        }
    }
    FAL_STAMP(5);
    if (PROF && lane == 0) {
#pragma unroll
        for (int i = 0; i < 6; i++) atomicAdd(prof + i, pc[i]);
    }
#undef FAL_STAMP
}

// ---- hash to ring (falcon.c:534-591): SHAKE128 with the reference's permutation, n little-endian 32-bit words --------
// dlp != 0: the DLP-IBE id oracle (dlp_ibe.c:805-816) - same stream, but q is SUBTRACTED when y >= q
__global__ __launch_bounds__(HT) void fal_hash1_kernel(uint32_t batch, const uint8_t *__restrict__ msgs, size_t msg_stride, const uint32_t *__restrict__ mlens, uint32_t fixed_len,
                                                      int32_t *__restrict__ c, FalP P, int dlp)
{
    __shared__ uint64_t st[25 * HT];
    const uint32_t b = blockIdx.x * HT + threadIdx.x;
    Sponge<HT> sp; sp.init(st, threadIdx.x, 168);
    if (b >= batch) return;
    const uint32_t mlen = mlens ? mlens[b] : fixed_len;
    sp.absorb(msgs + (size_t)b * msg_stride, mlen);
    sp.finalize(0x1F);
    const uint32_t mask = (1u << P.q_bits) - 1u, q = (uint32_t)P.q;
    int32_t *cb = c + (size_t)b * P.n;
    auto map = [&](uint32_t y) -> int32_t {
        y &= mask;
        if (dlp) return (int32_t)(y - (y >= q ? q : 0u));
        y -= (uint32_t)((int32_t)(q - y) >> 31) * q;                 // as written in the reference: ADDS q when y > q
        return (int32_t)y;
    };
    // the 168-byte rate is 21 state words = 42 little-endian 32-bit draws per permutation.  The squeeze walks whole blocks, so the state stays in
    // registers from here on (n = 1024 is 25 permutations: no LDS round trip in between, the stores of a block leave while the next is computed)
    uint64_t x[25];
#pragma unroll
    for (int w = 0; w < 25; w++) x[w] = sp.st[w * HT];
    int i = 0;
    for (; i + 42 <= P.n; i += 42) {                                    // whole blocks: unconditional stores
#pragma unroll
        for (int w = 0; w < 21; w++) *reinterpret_cast<int2 *>(cb + i + 2 * w) = make_int2(map((uint32_t)x[w]), map((uint32_t)(x[w] >> 32)));
        if (i + 42 < P.n) keccak_permute_regs<false>(x);
    }
#pragma unroll
    for (int w = 0; w < 21; w++)
        if (i + 2 * w < P.n) *reinterpret_cast<int2 *>(cb + i + 2 * w) = make_int2(map((uint32_t)x[w]), map((uint32_t)(x[w] >> 32)));
}

// The same with two lanes per message (Sponge2, keccak_dev.h): lane h of a pair owns the 32-bit half h of every state word, i.e.
// every second draw.  Used while the one-lane form would leave SIMDs without a wavefront (see fal_launch_hash).
__device__ __forceinline__ void fal_hash2_body(uint32_t block, uint32_t batch, const uint8_t *__restrict__ msgs, size_t msg_stride, const uint32_t *__restrict__ mlens, uint32_t fixed_len,
                                               int32_t *__restrict__ c, const FalP &P, int dlp)
{
    __shared__ uint32_t st[25 * HT];
    // a chain of 25 dependent permutations on a wavefront that issues one instruction in five cycles: in a seeded call the tape expansion (the cipher, which
    // fills every issue slot it is given) runs beside this kernel on the handle's side stream - the chain's instructions go first
    __builtin_amdgcn_s_setprio(3);
    const uint32_t b = (block * HT + threadIdx.x) >> 1;
    const int h = threadIdx.x & 1;
    Sponge2<HT> sp; sp.init(st, threadIdx.x, 168);
    if (b >= batch) return;
    const uint32_t mlen = mlens ? mlens[b] : fixed_len;
    sp.absorb(msgs + (size_t)b * msg_stride, mlen);
    sp.finalize(0x1F);
    const uint32_t mask = (1u << P.q_bits) - 1u, q = (uint32_t)P.q;
    int32_t *cb = c + (size_t)b * P.n;
    auto map = [&](uint32_t y) -> int32_t {
        y &= mask;
        if (dlp) return (int32_t)(y - (y >= q ? q : 0u));
        y -= (uint32_t)((int32_t)(q - y) >> 31) * q;                 // as written in the reference: ADDS q when y > q
        return (int32_t)y;
    };
    // the 168-byte rate is 21 state words = 42 little-endian 32-bit draws per permutation; the state stays in registers over the squeeze (as above)
    uint32_t x[25];
#pragma unroll
    for (int w = 0; w < 25; w++) x[w] = sp.st[w * HT + h];
    int i = 0;
    for (; i + 42 <= P.n; i += 42) {                                    // whole blocks: unconditional stores
#pragma unroll
        for (int w = 0; w < 21; w++) cb[i + 2 * w + h] = map(x[w]);
        if (i + 42 < P.n) keccak_permute_halves<false>(x, h);
    }
#pragma unroll
    for (int w = 0; w < 21; w++)
        if (i + 2 * w < P.n) cb[i + 2 * w + h] = map(x[w]);
}
__global__ __launch_bounds__(HT) void fal_hash2_kernel(uint32_t batch, const uint8_t *__restrict__ msgs, size_t msg_stride, const uint32_t *__restrict__ mlens, uint32_t fixed_len,
                                                      int32_t *__restrict__ c, FalP P, int dlp)
{ fal_hash2_body(blockIdx.x, batch, msgs, msg_stride, mlens, fixed_len, c, P, dlp); }

__device__ __forceinline__ int32_t centre_mod(int32_t x, int32_t q)      // ntt center_32: the representative in [-(q-1)/2, (q-1)/2]
{
    const int32_t q2 = (q - 1) >> 1;
    if (x >= -q2 && x <= q2) return x;                                   // already the representative (every unpacked 12-bit field is): no division
    int32_t r = x % q;
    if (r < -q2) r += q;
    if (r > q2) r -= q;
    return r;
}

// falcon.c:511-521: norm += s1[i]*s1[i] + s2[i]*s2[i] (integer expression, converted to FLOAT) from the top index down, sqrtf,
// ">= bd" fails.  Float addition is not associative and the sum passes 2^24, so the order is kept: the squares are formed in
// parallel into LDS and ONE lane adds them up in the reference's order.
__device__ __forceinline__ bool norm_fails_lds(const int32_t *sq, int n, float bd)
{
    float norm = 0;
    for (int i = n - 4; i >= 0; i -= 4) {
        const int4 v = *reinterpret_cast<const int4 *>(sq + i);
        norm += (float)v.w; norm += (float)v.z; norm += (float)v.y; norm += (float)v.x;
    }
    norm = sqrtf(norm);
    return norm >= bd;
}

constexpr int FW = 4;        // finish kernels: wavefronts (= signatures) per workgroup

struct BitW {                // generic MSB-first packer (any coefficient width)
    uint8_t *p; uint32_t nbytes; uint64_t acc; int nb;
    __device__ __forceinline__ void put(uint32_t v, int bits)
    {
        acc = (acc << bits) | (uint64_t)(v & ((1u << bits) - 1u)); nb += bits;
        while (nb >= 8) { p[nbytes++] = (uint8_t)(acc >> (nb - 8)); nb -= 8; }
    }
    __device__ __forceinline__ void flush() { if (nb) { p[nbytes++] = (uint8_t)(acc << (8 - nb)); nb = 0; } }
};

// falcon.c:879-905: centre s2, norm test (no retry in the reference), raw (q_bits-2)-bit MSB-first packing of s2.
// One wavefront per signature; coefficient i = lane + 64 k.
// The kernel is also the last reader of what a signing call must not leave behind - the lattice point s1 | s2 before packing and, one kernel back,
// the sampler's working polynomials (products with the secret basis; ws) - and zeroes them: wavefront b its rows of s1 and s2 and every batch-th
// kilobyte of the workspace range read as one linear array (whatever layout the sampler's mapping gave it; at any moment the wavefronts write one
// contiguous region, as a memset would).  Order: a wavefront first READS its two rows into registers, then issues its zeroes, then walks its chain
// of dependent steps on registers and LDS alone - the 0.4 GB of zeroes per 8192 signatures take ~55 us of HBM write time wherever they are issued
// (here, as three memsets, from the sampler's epilogue: measured), and a load queued behind them waits that long.
__global__ __launch_bounds__(64 * FW) void fal_finish_kernel(uint32_t batch, int32_t *s1, int32_t *s2, uint8_t *__restrict__ sigs, size_t sig_stride,
                                                             uint32_t *__restrict__ siglens, uint32_t *__restrict__ status, FalP P, double *ws, uint32_t ws_pieces)
{
    extern __shared__ __attribute__((aligned(16))) int32_t fin_lds[];       // per wave: n words (the squares, then the centred s2)
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, n = P.n;
    const uint32_t b = blockIdx.x * FW + wv;
    if (b >= batch) return;
    constexpr int MAXK = 16;                                                // n <= 1024 (falcon_create): coefficients per lane
    const int K = n >> 6;
    int32_t *a1 = s1 + (size_t)b * n, *a2 = s2 + (size_t)b * n;
    int32_t xr[MAXK], yr[MAXK];
    const uint32_t stt = status[b];
#pragma unroll
    for (int k = 0; k < MAXK; k++) { xr[k] = k < K ? a1[lane + 64 * k] : 0; yr[k] = k < K ? a2[lane + 64 * k] : 0; }
    __builtin_amdgcn_sched_barrier(0);                                       // (the zeroes below must not be hoisted over these loads)
#pragma unroll
    for (int k = 0; k < MAXK; k++) if (k < K) { a1[lane + 64 * k] = 0; a2[lane + 64 * k] = 0; }
    if (ws) {
        const size_t total = (size_t)ws_pieces * 3 * n, stride = (size_t)batch * 64;                // 16-byte pieces: 6n doubles per signature slot
        uint4 *z = reinterpret_cast<uint4 *>(ws);
        for (size_t i = (size_t)b * 64 + lane; i < total; i += stride) z[i] = make_uint4(0, 0, 0, 0);
    }
    if (stt == 2 || stt == 6) { if (lane == 0) siglens[b] = 0; return; }      // tape too short / no such key: nothing was sampled
    int32_t *sq = fin_lds + (size_t)wv * n;
#pragma unroll
    for (int k = 0; k < MAXK; k++) if (k < K) {
        const int32_t x = xr[k], y = centre_mod(yr[k], P.q);
        yr[k] = y;
        sq[lane + 64 * k] = x * x + y * y;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    int bad = 0;
    if (lane == 0) bad = norm_fails_lds(sq, n, P.bd) ? 1 : 0;
    bad = __shfl(bad, 0);
    if (bad) { if (lane == 0) { status[b] = 4; siglens[b] = 0; } return; }
#pragma unroll
    for (int k = 0; k < MAXK; k++) if (k < K) sq[lane + 64 * k] = yr[k];    // the centred s2, for the packer (neighbouring coefficients sit in different lanes)
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    uint8_t *out = sigs + (size_t)b * sig_stride;
    const int bits = P.q_bits - 2;
    if (bits == 12 && (reinterpret_cast<uintptr_t>(out) & 3) == 0) {
        // eight coefficients -> twelve bytes, written as three big-endian words (a row that starts on a word boundary)
        for (int gq = lane; gq < n / 8; gq += 64) {
            const int4 lo4 = *reinterpret_cast<const int4 *>(sq + 8 * gq), hi4 = *reinterpret_cast<const int4 *>(sq + 8 * gq + 4);
            const uint32_t u0 = (uint32_t)lo4.x & 0xFFFu, u1 = (uint32_t)lo4.y & 0xFFFu, u2 = (uint32_t)lo4.z & 0xFFFu, u3 = (uint32_t)lo4.w & 0xFFFu;
            const uint32_t u4 = (uint32_t)hi4.x & 0xFFFu, u5 = (uint32_t)hi4.y & 0xFFFu, u6 = (uint32_t)hi4.z & 0xFFFu, u7 = (uint32_t)hi4.w & 0xFFFu;
            uint32_t *ow = reinterpret_cast<uint32_t *>(out) + 3 * gq;
            ow[0] = __builtin_bswap32((u0 << 20) | (u1 << 8) | (u2 >> 4));
            ow[1] = __builtin_bswap32(((u2 & 0xFu) << 28) | (u3 << 16) | (u4 << 4) | (u5 >> 8));
            ow[2] = __builtin_bswap32(((u5 & 0xFFu) << 24) | (u6 << 12) | u7);
        }
        if (lane == 0) { siglens[b] = (uint32_t)(n / 2) * 3; status[b] = 1; }
    } else if (bits == 12) {
        // two coefficients -> three bytes
        for (int p = lane; p < n / 2; p += 64) {
            const uint32_t u = (uint32_t)sq[2 * p] & 0xFFFu, v = (uint32_t)sq[2 * p + 1] & 0xFFFu;
            out[3 * p] = (uint8_t)(u >> 4); out[3 * p + 1] = (uint8_t)(((u & 0xFu) << 4) | (v >> 8)); out[3 * p + 2] = (uint8_t)(v & 0xFFu);
        }
        if (lane == 0) { siglens[b] = (uint32_t)(n / 2) * 3; status[b] = 1; }
    } else if (lane == 0) {
        BitW bw = { out, 0, 0, 0 };
        for (int i = 0; i < n; i++) bw.put((uint32_t)sq[i], bits);
        bw.flush();
        siglens[b] = bw.nbytes;
        status[b] = 1;
    }
}
