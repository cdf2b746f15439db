// ens.hip - batched ENS-KEM encapsulation / decapsulation on MI355X behind the C ABI of include/sc_hip.h.
//
// Replaces, for a batch of operations under one key pair, ens_kem_encapsulation / ens_kem_decapsulation
// (src/schemes/kem/ens/ens_kem.c:785-865, 867-960) with generate_ciphertext (:280-318), poly_32_uniform_rand
// (src/utils/arith/poly_32.c:132-156) and z2_conv_mod2 (src/utils/arith/poly_z2.c:354-376); parameter sets ens_kem_params.c:23-61
// (q = 12289, n = 512 / 1024), raw coding (q_bits per coefficient, MSB first).  Keys are the reference's (its keygen inverts g modulo 2
// and modulo q on the CPU): h and g SINT16 in the NTT domain, 1/g mod 2 bit-reversed and packed (:191-211).
//
// Randomness: the prng_32 word stream ("tape") of sc->prng_ctx[0]: r, then e, each through the sparse placement loop (one word per
// attempt, occupied positions drawn again - so the words consumed vary per operation and are reported).
//
// Mapping: the placement loops are serial - one lane per operation, the polynomial as a byte column in LDS; the product h r (g t) on the
// plan's fused NTT -> pointwise -> INTT kernel with the key as a broadcast operand; one thread per four coefficients packs / unpacks
// the 14-bit fields (7 bytes); decapsulation's convolution in Z2[x] is one workgroup per operation: the parities of g t are gathered with
// __ballot into an LDS bit string, thread i forms output bit i from the n/32 words of the key's bit string rotated by i + 1.
#include "engine_internal.h"

#include <string.h>

using namespace sc;

namespace {

constexpr int HT = 64;

struct EnsP { int32_t set, q, q_bits, n, n_bits, pad0, pad1, coeff_rnd[13]; };      // ens_kem_set_t fields (ens_kem_params.h:30-47)

struct EnsC { int32_t c[12]; };

// r and e of one encapsulation (ens_kem.c:817-818).  status: 1 drawn, 2 tape exhausted.
// A lane draws one operation's two polynomials: for j = 0..11, coeff_rnd[j] positions (tape word >> 1 mod n, redrawn while occupied) get +-(12 - j) - a
// chain of dependent decisions per operation (~300 words per polynomial), so what counts is the latency of one step (no branches: see the window loop).  The polynomial being drawn
// lives in the lane's own LDS row as signed bytes; the tape is read ENS_LOOK words ahead (the draws consume it
// strictly in order, so the words a lane will want next are known: fetched word by word at the moment of use, every one of the ~600 draws waited for
// its own trip to memory - 0.7 ms per batch whatever its size); a finished polynomial leaves through 16-byte stores that the wavefront makes
// together, row by row (a lane writing its own row touched 64 different lines per store instruction).
constexpr int ENS_LOOK = 16;

__global__ __launch_bounds__(HT) void ens_sample_kernel(uint32_t batch, const uint32_t *tapes, size_t tape_stride, const uint32_t *tape_lens, EnsC C,
                                                        int32_t *r, int32_t *e, uint32_t *used, uint32_t *status, int n)
{
    extern __shared__ uint32_t rows[];                             // HT rows of n / 4 words: 64 KB at n = 1024, + the 520 bytes below - above a 64 KB workgroup limit; gfx950 gives a workgroup up to 160 KB, checked when the handle is created
    __shared__ int8_t s_lvl[1024 / 2 + 8];                         // s_lvl[a]: the magnitude the (a + 1)-th accepted draw of a polynomial writes (12 - j while the counts of 0..j cover a)
    const int RS = n / 4;
    const uint32_t base = blockIdx.x * HT, i = base + threadIdx.x;
    const bool live = i < batch;
    const uint32_t *w = tapes + (size_t)(live ? i : 0) * tape_stride;
    const uint32_t len = live ? (tape_lens ? tape_lens[i] : (uint32_t)tape_stride) : 0u;
    uint32_t total = 0;
#pragma unroll
    for (int j = 0; j < 12; j++) total += (uint32_t)C.c[j];         // <= n / 2 (checked when the handle is created)
    for (uint32_t a = threadIdx.x; a <= total + 2; a += HT) {
        int j = 0; uint32_t c = (uint32_t)C.c[0];
#pragma unroll
        for (int t = 1; t <= 12; t++) if (a >= c) { j = t; c += t < 12 ? (uint32_t)C.c[t] : 0u; }
        s_lvl[a] = (int8_t)(12 - j);
    }
    __syncthreads();
    uint32_t *row32 = rows + threadIdx.x * RS;
    int8_t *mine = reinterpret_cast<int8_t *>(row32);
    uint32_t p = 0;
    bool ok = live;
    uint32_t look[ENS_LOOK];                                       // look[t]: word p + t of the tape (clamped reads: never past the row)
    const uint32_t last = len ? len - 1 : 0;
#pragma unroll
    for (int t = 0; t < ENS_LOOK; t++) look[t] = w[min((uint32_t)t, last)];
    for (int which = 0; which < 2; which++) {
        for (int k = 0; k < n / 4; k++) row32[(k + threadIdx.x) & (n / 4 - 1)] = 0;      // (skewed: the rows are a multiple of the bank count apart)
        uint32_t tot = 0;                                          // accepted so far
        int32_t l0 = s_lvl[0], l1 = s_lvl[1];                      // s_lvl[tot], s_lvl[tot + 1]: the level is not looked up on the chain of a draw (below)
        uint32_t busy = (ok && total != 0) ? 1u : 0u, okw = ok ? 1u : 0u;     // flags as words and bitwise logic: written with bool && the loop came out as a branch per draw
        while (__any(busy != 0)) {
            // one window of ENS_LOOK words: consumed in order, the window after it requested first.  No branches inside: a draw is a read of the
            // position, a read of the level, an unconditional write of (accepted ? the new value : what was there) and three counters
            uint32_t nxt[ENS_LOOK];
#pragma unroll
            for (int t = 0; t < ENS_LOOK; t++) nxt[t] = w[min(p + ENS_LOOK + (uint32_t)t, last)];
            uint32_t took = 0;
#pragma unroll
            for (int t = 0; t < ENS_LOOK; t++) {
                const uint32_t out = busy & (uint32_t)(p + took >= len);       // tape exhausted before the polynomial is complete
                okw &= ~out;
                busy &= ~out;
                const uint32_t x = look[t];
                const uint32_t idx = (x >> 1) & (uint32_t)(n - 1);
                const int32_t cur = mine[idx], l2 = s_lvl[tot + 2];             // both reads leave together; l2 is what l1 becomes if this draw is accepted
                const uint32_t acc = busy & (uint32_t)(cur == 0);
                const int32_t val = (x & 1u) ? -l0 : l0;
                mine[idx] = (int8_t)(acc ? val : cur);
                took += busy;
                tot += acc;
                l0 = acc ? l1 : l0;
                l1 = acc ? l2 : l1;
                busy &= (uint32_t)(tot != total);
            }
            ok = okw != 0;
            // a lane that stopped inside the window keeps its position: shift the window by what was taken (whole windows in the common case)
            p += took;
            if (took == ENS_LOOK) {
#pragma unroll
                for (int t = 0; t < ENS_LOOK; t++) look[t] = nxt[t];
            } else if (took) {
#pragma unroll
                for (int t = 0; t < ENS_LOOK; t++) look[t] = w[min(p + (uint32_t)t, last)];
            }
        }
        // the wavefront's rows leave together: 64 lanes x 4 coefficients = 1 KB per store instruction
        const unsigned long long okm = __ballot(ok);
        int32_t *dst = which ? e : r;
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        for (uint32_t it = 0; it < (uint32_t)HT && base + it < batch; it++) {
            const bool good = (okm >> it) & 1;
            for (int k4 = threadIdx.x; k4 < n / 4; k4 += HT) {
                const uint32_t v = good ? rows[it * RS + k4] : 0u;
                // r feeds the transform as 32-bit words; e only meets ens_enc_finish_kernel and stays four signed bytes per word (a quarter of the traffic)
                if (which) reinterpret_cast<uint32_t *>(e)[(size_t)(base + it) * (n / 4) + k4] = v;
                else reinterpret_cast<int4 *>(dst + (size_t)(base + it) * n)[k4] = make_int4((int32_t)(int8_t)v, (int32_t)(int8_t)(v >> 8), (int32_t)(int8_t)(v >> 16), (int32_t)(int8_t)(v >> 24));
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
    if (live) { used[i] = p; status[i] = ok ? 1u : 2u; }
}

// t = normalise(2 centre(h r) + e), packed; key = the parities of e (ens_kem.c:305-315, 833-852).  One thread per four coefficients.
__global__ void ens_enc_finish_kernel(uint32_t batch, const int32_t *hr, const int32_t *e, const uint32_t *status, uint8_t *ct, size_t ct_stride, uint8_t *keys, size_t key_stride,
                                      int n, int32_t q, int q_bits)
{
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int per = n / 4;
    const uint32_t i = (uint32_t)(t >> sc::pow2_shift(per)); const int gq = (int)(t & (size_t)(per - 1));
    if (i >= batch || status[i] != 1) return;
    const sc::FastMod modq((uint32_t)q);
    uint64_t acc = 0; uint32_t kb = 0;
    const uint32_t e4 = reinterpret_cast<const uint32_t *>(e)[(size_t)i * per + gq];       // the four noise values of this group as signed bytes (ens_sample_kernel)
    const int4 h4 = *reinterpret_cast<const int4 *>(hr + (size_t)i * n + 4 * gq);
    const int32_t hv[4] = { h4.x, h4.y, h4.z, h4.w };
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const int32_t ek = (int32_t)(int8_t)(e4 >> (8 * k));
        int32_t v = hv[k];
        if (v > q / 2) v -= q;
        v = 2 * v + ek;                                            // |v| <= q + 12: + 2q makes it an unsigned operand
        v = (int32_t)modq((uint32_t)v + 2u * (uint32_t)q);
        acc = (acc << q_bits) | (uint64_t)(uint32_t)v;
        kb = (kb << 1) | (uint32_t)(ek & 1);
    }
    const int nbytes = 4 * q_bits / 8;                             // 7 for 14-bit coefficients
    uint8_t *o = ct + (size_t)i * ct_stride + (size_t)gq * nbytes;
    for (int b = 0; b < nbytes; b++) o[b] = (uint8_t)(acc >> (8 * (nbytes - 1 - b)));
    // four key bits per thread: two neighbouring threads share a byte - the even one writes it
    const uint32_t other = (uint32_t)__shfl_xor((int)kb, 1);
    if ((gq & 1) == 0) keys[(size_t)i * key_stride + (gq >> 1)] = (uint8_t)((kb << 4) | other);
}

__global__ void ens_unpack_kernel(uint32_t batch, const uint8_t *ct, size_t ct_stride, int32_t *t, int n, int q_bits)
{
    const size_t th = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int per = n / 4;
    const uint32_t i = (uint32_t)(th >> sc::pow2_shift(per)); const int gq = (int)(th & (size_t)(per - 1));
    if (i >= batch) return;
    const int nbytes = 4 * q_bits / 8;
    const uint8_t *s = ct + (size_t)i * ct_stride + (size_t)gq * nbytes;
    uint64_t acc = 0;
    for (int b = 0; b < nbytes; b++) acc = (acc << 8) | s[b];
    const uint32_t mask = (1u << q_bits) - 1u;
    int32_t *o = t + (size_t)i * n + 4 * gq;
    for (int k = 0; k < 4; k++) o[k] = (int32_t)((uint32_t)(acc >> (q_bits * (3 - k))) & mask);
}

// key = ((g t mod q, centred) mod 2) * (1/g mod 2) in Z2[x]/(x^n - 1) as z2_conv_mod2 computes it (ens_kem.c:926-946, poly_z2.c:354-376):
// bit i = parity(x AND rot_{i+1}(b)) with both bit strings MSB first and rot a right rotation of the whole n-bit string.
// One workgroup (n threads) per operation.
__global__ void ens_dec_finish_kernel(const int32_t *gt, const uint32_t *inv_g, uint8_t *keys, size_t key_stride, int n, int32_t q)
{
    __shared__ uint32_t xs[32], bs[32], outw[32], sb[32 * 33];
    const uint32_t i = blockIdx.x;
    const int tid = threadIdx.x, W = n / 32;
    int32_t v = gt[(size_t)i * n + tid] % q;
    if (v < 0) v += q;
    if (v > (q >> 1)) v -= q;
    const unsigned long long m = __ballot(v & 1);                  // lane l of the wavefront <-> coefficient 64*wave + l
    if ((tid & 63) == 0) {
        xs[2 * (tid >> 6)] = __builtin_bitreverse32((uint32_t)m);          // coefficient 32w + k at bit 31 - k
        xs[2 * (tid >> 6) + 1] = __builtin_bitreverse32((uint32_t)(m >> 32));
    }
    if (tid < W) bs[tid] = inv_g[tid];
    __syncthreads();
    // the 32 bit-shifts of b, once per workgroup (32 W words: n threads build one word each, W = n / 32): the n rotations a workgroup needs are these rows
    // read at a word offset, so the inner loop is two reads, an AND and an XOR per word instead of a funnel shift over two reads of b
    {
        const int sh = tid & 31, j = tid >> 5;                             // sb[sh][j] = word j of (b rotated right by sh bits)
        const uint32_t w0 = bs[j], w1 = bs[(j - 1) & (W - 1)];
        sb[sh * (W + 1) + j] = sh ? ((w1 << (32 - sh)) | (w0 >> sh)) : w0;                  // (rows W + 1 apart: the 32 rows a wavefront reads sit on distinct banks)
    }
    __syncthreads();
    const int s = tid + 1, ws = (s >> 5) & (W - 1), bsft = s & 31;     // rotate right by s = tid + 1 positions (W = n / 32 is a power of two: the
    uint32_t acc = 0;                                                  // three runtime modulo operations per word were 90 % of this kernel's instructions)
    const uint32_t *row = sb + bsft * (W + 1);
    for (int j = 0; j < W; j++) acc ^= xs[j] & row[(j - ws) & (W - 1)];
    const uint32_t bit = (uint32_t)__builtin_popcount(acc) & 1u;
    const unsigned long long om = __ballot(bit);
    if ((tid & 63) == 0) {
        outw[2 * (tid >> 6)] = __builtin_bitreverse32((uint32_t)om);
        outw[2 * (tid >> 6) + 1] = __builtin_bitreverse32((uint32_t)(om >> 32));
    }
    __syncthreads();
    if (tid < n / 8) keys[(size_t)i * key_stride + tid] = (uint8_t)(outw[tid >> 2] >> (24 - 8 * (tid & 3)));
}

}  // namespace

struct sc_hip_ens {
    sc_hip_ctx *ctx;
    sc_hip_plan *plan;
    EnsP P;
    int32_t *d_h, *d_g;
    uint32_t *d_inv_g;
    bool have_pub, have_priv;
    sc::Workspace ws;
};

extern "C" void sc_hip_ens_destroy(sc_hip_ens_t *h)
{
    if (!h) return;
    void *ptrs[] = { h->d_h, h->d_g, h->d_inv_g };
    for (void *p : ptrs) if (p) (void)sc::wipe_free(p);
    h->ws.release();
    if (h->plan) sc_hip_plan_destroy(h->plan);
    delete h;
}

extern "C" int sc_hip_ens_create(sc_hip_ctx_t *ctx, const sc_hip_ens_params_t *params, sc_hip_ens_t **out)
{
    if (!ctx || !params || !out) return fail("ens_create: null argument");
    static_assert(sizeof(EnsP) == sizeof(sc_hip_ens_params_t), "parameter struct layout");
    EnsP P; memcpy(&P, params, sizeof P);
    if ((P.n != 512 && P.n != 1024) || (1 << P.n_bits) != P.n) return fail("ens_create: n must be 512 or 1024");
    if (P.q_bits < 8 || P.q_bits > 16 || (4 * P.q_bits) % 8 || P.q < 3 || P.q >= (1 << P.q_bits)) return fail("ens_create: q, q_bits out of range (four coefficients must pack into whole bytes)");
    int total = 0;
    for (int j = 0; j < 12; j++) { if (P.coeff_rnd[j] < 0) return fail("ens_create: negative coefficient count"); total += P.coeff_rnd[j]; }
    if (total > P.n / 2) return fail("ens_create: more than n/2 non-zero coefficients");
    {
        // ens_sample_kernel: a row of n signed bytes per lane + its level table; ens_dec_finish_kernel stays below this
        int lds = 0;
        if (hipDeviceGetAttribute(&lds, hipDeviceAttributeMaxSharedMemoryPerBlock, ctx->device) != hipSuccess) { (void)hipGetLastError(); lds = 65536; }
        if ((size_t)HT * (size_t)P.n + 1024 / 2 + 8 + 64 > (size_t)lds)
            return fail("ens_create: the sampling kernel needs " + std::to_string((size_t)HT * (size_t)P.n + 584) + " bytes of LDS per workgroup, this device allows " + std::to_string(lds));
    }
    sc_hip_ens *h = new sc_hip_ens();
    h->ctx = ctx; h->P = P;
    if (sc_hip_plan_create(ctx, P.q, P.n, &h->plan) != SC_HIP_SUCCESS) { sc_hip_ens_destroy(h); return SC_HIP_FAILURE; }
    *out = h;
    return SC_HIP_SUCCESS;
}

static int ens_upload16(int32_t **dst, const int16_t *src, size_t n)
{
    std::vector<int32_t> t(n);
    for (size_t i = 0; i < n; i++) t[i] = src[i];
    if (!*dst) HIPCHK(hipMalloc((void **)dst, n * 4));
    HIPCHK(sc::copy_in(*dst, t.data(), n * 4));
    return SC_HIP_SUCCESS;
}

extern "C" int sc_hip_ens_set_pubkey(sc_hip_ens_t *h, const int16_t *hh)
{
    if (!h || !hh) return fail("ens_set_pubkey: null argument");
    HIPCHK(hipSetDevice(h->ctx->device));
    if (ens_upload16(&h->d_h, hh, (size_t)h->P.n)) return SC_HIP_FAILURE;
    h->have_pub = true;
    return SC_HIP_SUCCESS;
}

extern "C" int sc_hip_ens_set_privkey(sc_hip_ens_t *h, const int16_t *g, const uint32_t *inv_g)
{
    if (!h || !g || !inv_g) return fail("ens_set_privkey: null argument");
    HIPCHK(hipSetDevice(h->ctx->device));
    if (ens_upload16(&h->d_g, g, (size_t)h->P.n)) return SC_HIP_FAILURE;
    if (!h->d_inv_g) HIPCHK(hipMalloc((void **)&h->d_inv_g, (size_t)h->P.n / 8));
    HIPCHK(sc::copy_in(h->d_inv_g, inv_g, (size_t)h->P.n / 8));
    h->have_priv = true;
    return SC_HIP_SUCCESS;
}

extern "C" size_t sc_hip_ens_ct_bytes(const sc_hip_ens_t *h) { return h ? (size_t)h->P.n * h->P.q_bits / 8 : 0; }
extern "C" size_t sc_hip_ens_key_bytes(const sc_hip_ens_t *h) { return h ? (size_t)h->P.n / 8 : 0; }

extern "C" int sc_hip_ens_encaps(sc_hip_ens_t *h, size_t batch, const uint32_t *d_tapes, size_t tape_stride, const uint32_t *d_tape_lens,
                                 uint8_t *d_ct, size_t ct_stride, uint8_t *d_keys, size_t key_stride, uint32_t *d_used, uint32_t *d_status, void *stream)
{
    if (!h) return fail("ens_encaps: null handle");
    if (!h->have_pub) return fail("ens_encaps: no public key loaded");
    if (batch == 0) return SC_HIP_SUCCESS;
    if (batch > 0x7fffffffu / 2048) return fail("ens_encaps: batch too large");
    if (!d_tapes || !d_ct || !d_keys || !d_status) return fail("ens_encaps: null buffer");
    if (ct_stride < sc_hip_ens_ct_bytes(h) || key_stride < sc_hip_ens_key_bytes(h)) return fail("ens_encaps: stride below the record size");
    HIPCHK(hipSetDevice(h->ctx->device));
    hipStream_t s = (hipStream_t)stream;
    const int n = h->P.n;
    const uint32_t B = (uint32_t)batch;
    const size_t pn = batch * (size_t)n * 4;
    if (h->ws.reserve(3 * (pn + 256) + batch * 4 + 256)) return SC_HIP_FAILURE;
    int32_t *r = h->ws.take<int32_t>(batch * n), *e = h->ws.take<int32_t>(batch * n), *t = h->ws.take<int32_t>(batch * n);
    uint32_t *used = h->ws.take<uint32_t>(batch);
    EnsC C;
    for (int j = 0; j < 12; j++) C.c[j] = h->P.coeff_rnd[j];
    hipLaunchKernelGGL(ens_sample_kernel, dim3((B + HT - 1) / HT), dim3(HT), (size_t)HT * n, s, B, d_tapes, tape_stride, d_tape_lens, C, r, e,
                       d_used ? d_used : used, d_status, n);
    HIPCHK(hipGetLastError());
    if (sc::polymul_ntt_bcast(h->plan, batch, r, h->d_h, t, s)) return SC_HIP_FAILURE;
    const size_t nt = batch * (size_t)(n / 4);
    hipLaunchKernelGGL(ens_enc_finish_kernel, dim3((unsigned)((nt + 255) / 256)), dim3(256), 0, s, B, t, e, d_status, d_ct, ct_stride, d_keys, key_stride,
                       n, h->P.q, h->P.q_bits);
    HIPCHK(hipGetLastError());
    h->ws.wipe(s);                                                  // r and e: e's parities are the shared key
    return SC_HIP_SUCCESS;
}

extern "C" int sc_hip_ens_decaps(sc_hip_ens_t *h, size_t batch, const uint8_t *d_ct, size_t ct_stride, uint8_t *d_keys, size_t key_stride, void *stream)
{
    if (!h) return fail("ens_decaps: null handle");
    if (!h->have_priv) return fail("ens_decaps: no private key loaded");
    if (batch == 0) return SC_HIP_SUCCESS;
    if (batch > 0x7fffffffu / 2048) return fail("ens_decaps: batch too large");
    if (!d_ct || !d_keys) return fail("ens_decaps: null buffer");
    if (ct_stride < sc_hip_ens_ct_bytes(h) || key_stride < sc_hip_ens_key_bytes(h)) return fail("ens_decaps: stride below the record size");
    HIPCHK(hipSetDevice(h->ctx->device));
    hipStream_t s = (hipStream_t)stream;
    const int n = h->P.n;
    const uint32_t B = (uint32_t)batch;
    const size_t pn = batch * (size_t)n * 4;
    if (h->ws.reserve(2 * (pn + 256))) return SC_HIP_FAILURE;
    int32_t *t = h->ws.take<int32_t>(batch * n), *gt = h->ws.take<int32_t>(batch * n);
    const size_t nt = batch * (size_t)(n / 4);
    hipLaunchKernelGGL(ens_unpack_kernel, dim3((unsigned)((nt + 255) / 256)), dim3(256), 0, s, B, d_ct, ct_stride, t, n, h->P.q_bits);
    HIPCHK(hipGetLastError());
    if (sc::polymul_ntt_bcast(h->plan, batch, t, h->d_g, gt, s)) return SC_HIP_FAILURE;
    hipLaunchKernelGGL(ens_dec_finish_kernel, dim3(B), dim3(n), 0, s, gt, h->d_inv_g, d_keys, key_stride, n, h->P.q);
    HIPCHK(hipGetLastError());
    h->ws.wipe(s);                                                  // g t
    return SC_HIP_SUCCESS;
}
