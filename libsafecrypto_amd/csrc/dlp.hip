// dlp.hip - batched DLP-IBE encrypt / decrypt on MI355X behind the C ABI of include/sc_hip.h (extract lives with the tree sampler in falcon.hip).
//
// Replaces, for a batch of n-bit messages, dlp_ibe_encrypt / dlp_ibe_decrypt (src/schemes/ibe/dlp/dlp_ibe.c:1140-1370, 1372-1520) as the
// reference is built: SC_IBE_MESSAGE_LENGTH_N, XOF oracles (id_function :805-816, k_function_xof :850-862; SHAKE128 with the reference's
// permutation, keccak_dev.h), H_NTT_OPTIMISATION (H(id) is an NTT-domain vector), parameter sets dlp_ibe_params.c:22-40 (q = 4206593).
// Encryption may address a different identity per message; decryption uses the one user key loaded into the handle.
//
// Randomness: the prng_32 word stream ("tape") of sc->prng_ctx[0]: n/32 words of key bits (LSB first), then n prng_var(6) draws through
// the 32-bit reservoir of prng.c:1017-1048 (bits 0/1 -> e1 and its sign, 2/3 -> e2, 4/5 -> e3): n/32 + 3n/16 words, one more in flight.
//
// Mapping: one lane per message for the serial parts (the two hashes, the reservoir), which also writes the pad XOR message straight into
// the ciphertext (the pad depends on the key bits only); e3 h and e3 H(id) on the plan's fused NTT -> pointwise -> INTT kernel (broadcast
// public key, per-message H(id)); one thread per eight coefficients for the scaling, shifting and the 23-bit / 5-bit signed packing.
#include "engine_internal.h"
#include "keccak_dev.h"

#include <string.h>

using namespace sc;

namespace {

constexpr int HT = 64;

struct DlpP { int32_t n, q, q_bits, n_bits, set, ubits, m_scale, l; };       // dlp_ibe_set_t fields (dlp_ibe_params.h:106-125) + the user-key width

struct Res {                     // one message's word stream + the prng_var reservoir (prng.c:1017-1048)
    const uint32_t *w; uint32_t len, pos, var_buf; int var_bits; bool underrun;
    __device__ __forceinline__ uint32_t w32() { if (pos >= len) { underrun = true; return 0; } return w[pos++]; }
    __device__ __forceinline__ uint32_t var(int n)
    {
        uint32_t ret = var_buf;
        if (var_bits < n) {
            const int bits = n - var_bits;
            ret <<= bits;
            var_buf = w32();
            ret |= var_buf & ((1u << bits) - 1u);
            var_buf >>= bits;
            var_bits = 32 - bits;
        } else {
            var_buf >>= n;
            var_bits -= n;
        }
        return ret & ((1u << n) - 1u);
    }
};

// eight 0/1 bytes from the low eight bits of x
__device__ __forceinline__ uint64_t spread8(uint32_t x)
{
    uint64_t v = 0;
#pragma unroll
    for (int j = 0; j < 8; j++) v |= (uint64_t)((x >> j) & 1u) << (8 * j);
    return v;
}

// per message: H(id) -> c; key bits k; e1, e2, e3; pad = SHAKE128(k as n bytes)[0, n/8) XOR message -> the tail of the ciphertext
// aux[i] = (e1 + 1) | (e2 + 1) << 2 | k << 4
__global__ __launch_bounds__(HT) void dlp_enc_prep_kernel(uint32_t batch, const uint8_t *ids, size_t id_stride, const uint32_t *idlens, uint32_t fixed_len,
                                                          const uint8_t *msgs, size_t msg_stride, const uint32_t *tapes, size_t tape_stride, const uint32_t *tape_lens,
                                                          int32_t *c, int32_t *e3, uint8_t *aux, uint8_t *ct, size_t ct_stride, uint32_t tail_off, uint32_t *status, DlpP P)
{
    __shared__ uint64_t st[25 * HT];
    __shared__ uint32_t kws[32 * HT];                                  // the key words of a lane (n / 32 <= 32), column per lane: combined with e1, e2 below
    const uint32_t i = blockIdx.x * HT + threadIdx.x;
    Sponge<HT> sp; sp.init(st, threadIdx.x, 168);
    if (i >= batch) return;
    const int n = P.n;
    // id_function (dlp_ibe.c:805-816)
    sp.absorb(ids + (size_t)i * id_stride, idlens ? idlens[i] : fixed_len);
    sp.finalize(0x1F);
    const uint32_t mask = (1u << P.q_bits) - 1u, q = (uint32_t)P.q;
    int32_t *ci = c + (size_t)i * n;
    for (int k = 0; k < n;) {
#pragma unroll 1
        for (int w = 0; w < 21 && k < n; w++, k += 2) {
            const uint64_t v = sp.st[w * HT];
            uint32_t a = (uint32_t)v & mask, b = (uint32_t)(v >> 32) & mask;
            a -= (a >= q) ? q : 0u; b -= (b >= q) ? q : 0u;
            *reinterpret_cast<int2 *>(ci + k) = make_int2((int32_t)a, (int32_t)b);
        }
        if (k < n) sp.permute();
    }
    // key bits (:1215-1222) and the pad (k_function_xof, :850-862)
    Res T{ tapes + (size_t)i * tape_stride, tape_lens ? tape_lens[i] : (uint32_t)tape_stride, 0, 0, 0, false };
    uint8_t *ai = aux + (size_t)i * n;
    sp.init(st, threadIdx.x, 168);
    for (int k = 0; k < n; k += 32) {
        const uint32_t x = T.w32();
        kws[(k >> 5) * HT + threadIdx.x] = x;
#pragma unroll
        for (int b = 0; b < 4; b++) sp.absorb_u64(spread8(x >> (8 * b)));
    }
    sp.finalize(0x1F);
    const uint8_t *mi = msgs + (size_t)i * msg_stride;
    uint8_t *tail = ct + (size_t)i * ct_stride + tail_off;
    for (int b = 0; b < n / 8; b++) tail[b] = (uint8_t)(sp.squeeze_byte() ^ mi[b]);
    // e1, e2, e3 (:1229-1243)
    // eight coefficients at a time: the aux bytes are written once, complete (key bit | e2 + 1 | e1 + 1) - OR-ed into bytes this lane had stored
    // earlier they were n dependent round trips to memory per message, 0.5 ms of the kernel's 0.78 at batch 65536
    int32_t *e3i = e3 + (size_t)i * n;
    for (int k = 0; k < n; k += 8) {
        const uint32_t kb = kws[(k >> 5) * HT + threadIdx.x] >> (k & 31);
        uint32_t lo = 0, hi = 0;
        int32_t a3s[8];
#pragma unroll
        for (int j = 0; j < 8; j++) {
            uint32_t r = T.var(6);
            int32_t a1 = (int32_t)(r & 1u); r >>= 1; if (r & 1u) a1 = -a1; r >>= 1;
            int32_t a2 = (int32_t)(r & 1u); r >>= 1; if (r & 1u) a2 = -a2; r >>= 1;
            int32_t a3 = (int32_t)(r & 1u); r >>= 1; if (r & 1u) a3 = -a3;
            const uint32_t byte = (uint32_t)(a1 + 1) | ((uint32_t)(a2 + 1) << 2) | (((kb >> j) & 1u) << 4);
            if (j < 4) lo |= byte << (8 * j); else hi |= byte << (8 * (j - 4));
            a3s[j] = a3;
        }
        *reinterpret_cast<uint2 *>(ai + k) = make_uint2(lo, hi);
        *reinterpret_cast<int4 *>(e3i + k) = make_int4(a3s[0], a3s[1], a3s[2], a3s[3]);
        *reinterpret_cast<int4 *>(e3i + k + 4) = make_int4(a3s[4], a3s[5], a3s[6], a3s[7]);
    }
    status[i] = T.underrun ? 2u : 1u;
}

// MSB-first packing of eight signed values of `bits` bits into `bits` bytes
__device__ __forceinline__ void pack8(uint8_t *o, const int32_t (&x)[8], int bits)
{
    uint64_t acc = 0; int nb = 0, w = 0;
    const uint32_t mask = (1u << bits) - 1u;
#pragma unroll
    for (int e = 0; e < 8; e++) {
        acc = (acc << bits) | (uint64_t)((uint32_t)x[e] & mask); nb += bits;
        while (nb >= 8) { o[w++] = (uint8_t)(acc >> (nb - 8)); nb -= 8; }
    }
}

// u = normalise(e3 h + q/2 + e1) - q/2 ; v = (normalise(e3 H(id) + q/2 + e2 + k m_scale) - q/2) >> l ; both packed (:1290-1330)
__global__ void dlp_enc_finish_kernel(uint32_t batch, const int32_t *u, const int32_t *v, const uint8_t *aux, const uint32_t *status, uint8_t *ct, size_t ct_stride, DlpP P)
{
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int per = P.n / 8;
    const uint32_t i = (uint32_t)(t >> sc::pow2_shift(per)); const int g = (int)(t & (size_t)(per - 1));
    if (i >= batch || status[i] != 1) return;
    const int32_t q = P.q, q2 = q >> 1;
    const sc::FastMod modq((uint32_t)q);
    int32_t xu[8], xv[8];
    for (int e = 0; e < 8; e++) {
        const size_t idx = (size_t)i * P.n + 8 * g + e;
        const uint32_t a = aux[idx];
        // u, v: products from the transform (canonical, at worst centred): + q keeps the operand non-negative, the sums stay below 2^32 (q < 2^30)
        const int32_t uu = (int32_t)modq((uint32_t)u[idx] + (uint32_t)q + (uint32_t)q2 + (a & 3u) - 1u);
        const int32_t vv = (int32_t)modq((uint32_t)v[idx] + (uint32_t)q + (uint32_t)q2 + ((a >> 2) & 3u) - 1u + ((a >> 4) & 1u) * (uint32_t)P.m_scale);
        xu[e] = uu - q2;
        xv[e] = (vv - q2) >> P.l;
    }
    uint8_t *o = ct + (size_t)i * ct_stride;
    pack8(o + (size_t)g * P.q_bits, xu, P.q_bits);
    pack8(o + (size_t)P.n * P.q_bits / 8 + (size_t)g * (P.q_bits - P.l), xv, P.q_bits - P.l);
}

__device__ __forceinline__ void unpack8(const uint8_t *s, int32_t (&x)[8], int bits)
{
    uint64_t acc = 0; int nb = 0, r = 0;
    for (int e = 0; e < 8; e++) {
        while (nb < bits) { acc = (acc << 8) | s[r++]; nb += 8; }
        const uint32_t val = (uint32_t)(acc >> (nb - bits)) & ((1u << bits) - 1u);
        nb -= bits;
        x[e] = (int32_t)(val << (32 - bits)) >> (32 - bits);
    }
}

__global__ void dlp_dec_unpack_kernel(uint32_t batch, const uint8_t *ct, size_t ct_stride, int32_t *u, int32_t *v, DlpP P)
{
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int per = P.n / 8;
    const uint32_t i = (uint32_t)(t >> sc::pow2_shift(per)); const int g = (int)(t & (size_t)(per - 1));
    if (i >= batch) return;
    const uint8_t *s = ct + (size_t)i * ct_stride;
    int32_t xu[8], xv[8];
    unpack8(s + (size_t)g * P.q_bits, xu, P.q_bits);
    unpack8(s + (size_t)P.n * P.q_bits / 8 + (size_t)g * (P.q_bits - P.l), xv, P.q_bits - P.l);
    for (int e = 0; e < 8; e++) { u[(size_t)i * P.n + 8 * g + e] = xu[e]; v[(size_t)i * P.n + 8 * g + e] = xv[e] << P.l; }
}

// k[i] = [q/2 <= normalise(v - u sk)[i] + q/4 < q] (:1462-1466); message = tail XOR SHAKE128(k as n bytes)[0, n/8).  One lane per message.
__global__ __launch_bounds__(HT) void dlp_dec_finish_kernel(uint32_t batch, const int32_t *v, const int32_t *us, const uint8_t *ct, size_t ct_stride, uint32_t tail_off,
                                                            uint8_t *msgs, size_t msg_stride, DlpP P)
{
    __shared__ uint64_t st[25 * HT];
    const uint32_t i = blockIdx.x * HT + threadIdx.x;
    Sponge<HT> sp; sp.init(st, threadIdx.x, 168);
    if (i >= batch) return;
    const int n = P.n;
    const int32_t q = P.q, q2 = q >> 1, q4 = q >> 2;
    const sc::FastMod modq((uint32_t)q);
    const int4 *pv = reinterpret_cast<const int4 *>(v + (size_t)i * n), *pu = reinterpret_cast<const int4 *>(us + (size_t)i * n);
    for (int k = 0; k < n / 8; k++) {
        uint64_t word = 0;
#pragma unroll
        for (int hlf = 0; hlf < 2; hlf++) {
            const int4 a = pv[2 * k + hlf], b = pu[2 * k + hlf];
            const int32_t d[4] = { a.x - b.x, a.y - b.y, a.z - b.z, a.w - b.w };
#pragma unroll
            for (int e = 0; e < 4; e++) {
                int32_t x = (int32_t)modq((uint32_t)d[e] + 2u * (uint32_t)q);       // differences of canonical (at worst centred) values: above -2q
                x += q4;
                word |= (uint64_t)((x >= q2 && x < q) ? 1u : 0u) << (8 * (4 * hlf + e));
            }
        }
        sp.absorb_u64(word);
    }
    sp.finalize(0x1F);
    const uint8_t *tail = ct + (size_t)i * ct_stride + tail_off;
    uint8_t *o = msgs + (size_t)i * msg_stride;
    for (int b = 0; b < n / 8; b++) o[b] = (uint8_t)(sp.squeeze_byte() ^ tail[b]);
}

}  // namespace

struct sc_hip_dlp {
    sc_hip_ctx *ctx;
    sc_hip_plan *plan;
    DlpP P;
    int32_t *d_h_ntt, *d_uk_ntt;
    bool have_pub, have_uk;
    sc::Workspace ws;
};

extern "C" void sc_hip_dlp_destroy(sc_hip_dlp_t *h)
{
    if (!h) return;
    if (h->d_h_ntt) (void)sc::wipe_free(h->d_h_ntt);
    if (h->d_uk_ntt) (void)sc::wipe_free(h->d_uk_ntt);
    h->ws.release();
    if (h->plan) sc_hip_plan_destroy(h->plan);
    delete h;
}

extern "C" int sc_hip_dlp_create(sc_hip_ctx_t *ctx, const sc_hip_dlp_params_t *params, sc_hip_dlp_t **out)
{
    if (!ctx || !params || !out) return fail("dlp_create: null argument");
    static_assert(sizeof(DlpP) == sizeof(sc_hip_dlp_params_t), "parameter struct layout");
    DlpP P; memcpy(&P, params, sizeof P);
    if ((P.n != 512 && P.n != 1024) || (1 << P.n_bits) != P.n) return fail("dlp_create: n must be 512 or 1024");
    if (P.q_bits < 8 || P.q_bits > 30 || P.q < 3 || P.q >= (1 << P.q_bits) || P.l < 1 || P.l >= P.q_bits || P.m_scale < 0 || P.m_scale >= P.q) return fail("dlp_create: parameters out of range");
    sc_hip_dlp *h = new sc_hip_dlp();
    h->ctx = ctx; h->P = P;
    if (sc_hip_plan_create(ctx, P.q, P.n, &h->plan) != SC_HIP_SUCCESS) { sc_hip_dlp_destroy(h); return SC_HIP_FAILURE; }
    *out = h;
    return SC_HIP_SUCCESS;
}

static int dlp_upload(int32_t **dst, const int32_t *src, size_t n)
{
    if (!*dst) HIPCHK(hipMalloc((void **)dst, n * 4));
    HIPCHK(sc::copy_in(*dst, src, n * 4));
    return SC_HIP_SUCCESS;
}

extern "C" int sc_hip_dlp_set_pubkey(sc_hip_dlp_t *h, const int32_t *h_ntt)
{
    if (!h || !h_ntt) return fail("dlp_set_pubkey: null argument");
    HIPCHK(hipSetDevice(h->ctx->device));
    if (dlp_upload(&h->d_h_ntt, h_ntt, (size_t)h->P.n)) return SC_HIP_FAILURE;
    h->have_pub = true;
    return SC_HIP_SUCCESS;
}

extern "C" int sc_hip_dlp_set_user_key(sc_hip_dlp_t *h, const int32_t *uk_ntt)
{
    if (!h || !uk_ntt) return fail("dlp_set_user_key: null argument");
    HIPCHK(hipSetDevice(h->ctx->device));
    if (dlp_upload(&h->d_uk_ntt, uk_ntt, (size_t)h->P.n)) return SC_HIP_FAILURE;
    h->have_uk = true;
    return SC_HIP_SUCCESS;
}

extern "C" size_t sc_hip_dlp_ct_bytes(const sc_hip_dlp_t *h) { return h ? (size_t)h->P.n * (size_t)(2 * h->P.q_bits - h->P.l) / 8 + (size_t)h->P.n / 8 : 0; }
extern "C" size_t sc_hip_dlp_msg_bytes(const sc_hip_dlp_t *h) { return h ? (size_t)h->P.n / 8 : 0; }
extern "C" size_t sc_hip_dlp_enc_words(const sc_hip_dlp_t *h) { return h ? (size_t)h->P.n / 32 + (6 * (size_t)h->P.n + 31) / 32 + 1 : 0; }

extern "C" int sc_hip_dlp_encrypt(sc_hip_dlp_t *h, size_t batch, const uint8_t *d_ids, size_t id_stride, const uint32_t *d_idlens, uint32_t fixed_len,
                                  const uint8_t *d_msgs, size_t msg_stride, const uint32_t *d_tapes, size_t tape_stride, const uint32_t *d_tape_lens,
                                  uint8_t *d_ct, size_t ct_stride, uint32_t *d_status, void *stream)
{
    if (!h) return fail("dlp_encrypt: null handle");
    if (!h->have_pub) return fail("dlp_encrypt: no public key loaded");
    if (batch == 0) return SC_HIP_SUCCESS;
    if (batch > 0x7fffffffu / 2048) return fail("dlp_encrypt: batch too large");
    if (!d_ids || !d_msgs || !d_tapes || !d_ct || !d_status) return fail("dlp_encrypt: null buffer");
    if (msg_stride < sc_hip_dlp_msg_bytes(h) || ct_stride < sc_hip_dlp_ct_bytes(h)) return fail("dlp_encrypt: stride below the record size");
    HIPCHK(hipSetDevice(h->ctx->device));
    hipStream_t s = (hipStream_t)stream;
    const DlpP &P = h->P;
    const int n = P.n;
    const uint32_t B = (uint32_t)batch;
    const size_t pn = batch * (size_t)n * 4;
    if (h->ws.reserve(4 * (pn + 256) + batch * (size_t)n + 256)) return SC_HIP_FAILURE;
    int32_t *c = h->ws.take<int32_t>(batch * n), *e3 = h->ws.take<int32_t>(batch * n), *u = h->ws.take<int32_t>(batch * n), *v = h->ws.take<int32_t>(batch * n);
    uint8_t *aux = h->ws.take<uint8_t>(batch * n);
    const uint32_t tail_off = (uint32_t)((size_t)n * (size_t)(2 * P.q_bits - P.l) / 8);
    hipLaunchKernelGGL(dlp_enc_prep_kernel, dim3((B + HT - 1) / HT), dim3(HT), 0, s, B, d_ids, id_stride, d_idlens, fixed_len, d_msgs, msg_stride, d_tapes, tape_stride, d_tape_lens,
                       c, e3, aux, d_ct, ct_stride, tail_off, d_status, P);
    HIPCHK(hipGetLastError());
    if (sc::polymul_ntt_bcast(h->plan, batch, e3, h->d_h_ntt, u, s)) return SC_HIP_FAILURE;
    if (sc_hip_polymul_ntt(h->plan, batch, e3, c, v, s)) return SC_HIP_FAILURE;
    const size_t nt = batch * (size_t)(n / 8);
    hipLaunchKernelGGL(dlp_enc_finish_kernel, dim3((unsigned)((nt + 255) / 256)), dim3(256), 0, s, B, u, v, aux, d_status, d_ct, ct_stride, P);
    HIPCHK(hipGetLastError());
    h->ws.wipe(s);                                                  // the ephemeral e3, the key bits (aux) and their products
    return SC_HIP_SUCCESS;
}

extern "C" int sc_hip_dlp_decrypt(sc_hip_dlp_t *h, size_t batch, const uint8_t *d_ct, size_t ct_stride, uint8_t *d_msgs, size_t msg_stride, void *stream)
{
    if (!h) return fail("dlp_decrypt: null handle");
    if (!h->have_uk) return fail("dlp_decrypt: no user key loaded");
    if (batch == 0) return SC_HIP_SUCCESS;
    if (batch > 0x7fffffffu / 2048) return fail("dlp_decrypt: batch too large");
    if (!d_ct || !d_msgs) return fail("dlp_decrypt: null buffer");
    if (msg_stride < sc_hip_dlp_msg_bytes(h) || ct_stride < sc_hip_dlp_ct_bytes(h)) return fail("dlp_decrypt: stride below the record size");
    HIPCHK(hipSetDevice(h->ctx->device));
    hipStream_t s = (hipStream_t)stream;
    const DlpP &P = h->P;
    const int n = P.n;
    const uint32_t B = (uint32_t)batch;
    const size_t pn = batch * (size_t)n * 4;
    if (h->ws.reserve(3 * (pn + 256))) return SC_HIP_FAILURE;
    int32_t *u = h->ws.take<int32_t>(batch * n), *v = h->ws.take<int32_t>(batch * n), *us = h->ws.take<int32_t>(batch * n);
    const size_t nt = batch * (size_t)(n / 8);
    hipLaunchKernelGGL(dlp_dec_unpack_kernel, dim3((unsigned)((nt + 255) / 256)), dim3(256), 0, s, B, d_ct, ct_stride, u, v, P);
    HIPCHK(hipGetLastError());
    if (sc::polymul_ntt_bcast(h->plan, batch, u, h->d_uk_ntt, us, s)) return SC_HIP_FAILURE;
    const uint32_t tail_off = (uint32_t)((size_t)n * (size_t)(2 * P.q_bits - P.l) / 8);
    hipLaunchKernelGGL(dlp_dec_finish_kernel, dim3((B + HT - 1) / HT), dim3(HT), 0, s, B, v, us, d_ct, ct_stride, tail_off, d_msgs, msg_stride, P);
    HIPCHK(hipGetLastError());
    h->ws.wipe(s);                                                  // u * user key
    return SC_HIP_SUCCESS;
}
