// mlwe_dev.h - the device side of mlwe_kernels.h: the NTT-shaped kernels of the module-LWE schemes (Dilithium, Kyber) - n = 256, 16 lanes per polynomial,
// 4 polynomials per wavefront, NTT-domain data in the engine's internal order (position j = lane*16 + r holds natural index bitrev8(j): 16 contiguous
// words per lane).  Nothing of the host runtime in here, so that the kernels' own text can also be compiled as host code over tests/emu/wave_emu.h and
// run lane by lane under the sanitizers (tests/test_emu_dil_sign.py); mlwe_kernels.h adds the launchers.
#pragma once
#include "ntt_core.h"
#include "scalar_hint.h"
#include <string.h>

namespace sc_mlwe {
using namespace sc;

constexpr int L8 = 8, N = 256, LP = 16, PW = 4;

__host__ __device__ constexpr int brv4c(int r) { return ((r & 1) << 3) | ((r & 2) << 1) | ((r & 4) >> 1) | ((r & 8) >> 3); }

#include "mlwe_round.h"

// ------------------------------------------------------------------------------------------------
// NTT-shaped kernels (n = 256: 16 lanes per polynomial, 4 polynomials per wave, 32-bit Shoup path)
// ------------------------------------------------------------------------------------------------

enum { DOP_FWD_INT = 0, DOP_MATVEC = 1, DOP_CT1 = 5, DOP_KEYPREP = 6 };

// The signer's per-attempt w = A y in 24 bits per coefficient (DilNtt::w_chunked == 2; q < 2^24): the 16 values v[0..15] of a lane as 12 words, value e in
// bits [24 e, 24 e + 24) of the little-endian string.  192 words per polynomial instead of 256.
constexpr int W24 = 192;
__device__ __forceinline__ uint32_t pack24_word(const uint32_t (&v)[16], int w)          // w: 0..11, compile-time after unrolling
{
    const int e = (4 * w) / 3, sh = 32 * w - 24 * e;                  // the first value with bits in word w, and how many of its bits lie in earlier words
    uint32_t r = v[e] >> sh;
    if (e + 1 < 16) r |= v[e + 1] << (24 - sh);
    if (24 - sh + 24 < 32 && e + 2 < 16) r |= v[e + 2] << (48 - sh);
    return r;
}
__device__ __forceinline__ uint32_t unpack24(const uint32_t (&p)[12], int e)              // e: 0..15, compile-time after unrolling
{
    const int w = (3 * e) / 4, sh = 24 * e - 32 * w;
    uint32_t r = p[w] >> sh;
    if (sh > 8) r |= p[w + 1] << (32 - sh);
    return r & 0xFFFFFFu;
}

struct DilNtt {
    const uint32_t *zf, *zfp, *zi, *zip;
    ModK k;
    uint32_t c0, c0p, c1, c1p;            // n^-1 and zeta[1]^-1 n^-1 (Shoup pairs): Montgomery-form keys need no 2^32 fix
    uint32_t r2, r2p;                     // 2^32 mod q (Shoup pair)
    uint32_t njobs;                       // slots * per (every per-attempt array is indexed by slot)
    const uint32_t *njobs_dev;            // non-null: the number of ITEMS (slots) is read from device memory (the signer's round state, written by the
                                          // previous round's last kernel: no host read-back between rounds); njobs then only sizes the grid
    int per;                              // polynomials per item handled by this launch
    int l, kk;
    int32_t alpha, am, gamma1_beta, gamma2_beta, d;
    const int32_t *in;                    // DOP_FWD_INT / KEYPREP: natural-order input [item][per][n]
    uint32_t *out_int;                    // DOP_FWD_INT / KEYPREP: internal-order canonical (KEYPREP: Montgomery form)
    const uint32_t *yh;                   // [item][l][n] internal
    const uint32_t *key;                  // matrix [k][l][n] or key vector [per][n], internal order, Montgomery form
    const uint32_t *keyp;                 // dil_yaw_kernel only: Shoup companions floor(key 2^32 / q) of a PLAIN (non-Montgomery) matrix
    size_t key_item_stride;               // words between the matrices / key vectors of consecutive items (0: one key for the whole batch)
    const uint32_t *key_of;               // with a key stride: the key index of each slot (signing: slot -> message); null = the slot index itself
    int key_nat;                          // DOP_MATVEC: the matrix is in NATURAL order and PLAIN form (any int32 representative) - as the CSPRNG expansion and
                                          // the key pair hold it; the other operand (yh) must then carry the Montgomery factor (DOP_KEYPREP's output form).
                                          // 1: polynomial (row, col) at [row][col]; 2: at [col][row] (the transposed product of encryption)
    const uint32_t *ch;                   // [item][n] internal (NTT of c)
    const uint32_t *t1s;                  // dil_yaw_kernel<..., CT1 = true> (verification): NTT(t1 2^d), [kk][n] internal order, Montgomery form
    const int32_t *y;                     // [item][l][n] natural (mask)
    int32_t *w;                           // [item][k][n] natural canonical
    uint8_t *w1;                          // [item][k*n]
    int32_t *z;                           // [item][l][n]
    uint8_t *h;                           // [item][k*n]
    int w_chunked;                        // 2: as 1, in 24 bits per coefficient (W24 below; the fused kernel and the signer's finishing kernel only).
                                          // 1 (the signer's per-attempt w, which only its own kernels read): coefficient j = 16 r + lp of a polynomial is stored at
                                          // 64 (r / 4) + 4 lp + r % 4 - a lane moves its 16 coefficients as four 16-byte pieces and each of the four wave
                                          // instructions covers 256 contiguous bytes per polynomial (sixteen 4-byte stores of 64 bytes each otherwise).
                                          // (Measured and dropped: 16 contiguous words per lane - every instruction then touches 16 lines per polynomial - and
                                          // y in either order: ExpandMask's lane-per-attempt stores want consecutive coefficients at consecutive addresses.)
    int sg;                               // 1: the signed 32-bit arithmetic (MODE_S32, modarith.h) - tables and scale constants are the centred ones
};

// MODE: MODE_S32 (q < 2^25 and 16 l q < 2^31: every reference parameter set) or MODE_U32.  Both produce the same canonical values.
template <int OP, int MODE>
__global__ __launch_bounds__(256) void dil_ntt_kernel(DilNtt a)
{
    constexpr bool SG = MODE == MODE_S32;
    __shared__ __attribute__((aligned(16))) uint32_t lds[4 * 1024];
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const int pw = lane / LP, lp = lane % LP;
    uint32_t *lds_poly = lds + wid * 1024 + pw * N;
    const ModK k = a.k;
    TwSet<L8> tw;
    if (OP == DOP_FWD_INT || OP == DOP_KEYPREP) tw.load(a.zf, a.zfp, lp); else tw.load(a.zi, a.zip, lp);

    const uint32_t njobs = a.njobs_dev ? *a.njobs_dev * (uint32_t)a.per : a.njobs;
    const uint32_t groups = (njobs + PW - 1) / PW;
    for (uint32_t g = blockIdx.x * 4 + wid; g < groups; g += gridDim.x * 4) {
        const uint32_t job = g * PW + pw;
        const bool valid = job < njobs;
        // a job past the end is pointed at job 0 and its results are never stored: every load below stays unconditional (a
        // predicated load compiles to a branch with a full s_waitcnt inside, which serialises the requests)
        const uint32_t aidx = valid ? job / (uint32_t)a.per : 0, sub = valid ? job % (uint32_t)a.per : 0;
        const uint32_t item = aidx;           // slot index
        uint32_t x[16];

        if (OP == DOP_FWD_INT || OP == DOP_KEYPREP) {
            const int32_t *src = a.in + ((size_t)item * a.per + sub) * N + lp;
#pragma unroll
            for (int r = 0; r < 16; r++) x[r] = SG ? reduce_s32(src[r * LP], k) : reduce_i32(src[r * LP], k);        // unconditional: an out-of-range job reads job 0 (see the loop head)
            fwd_core<L8, MODE, true>(x, tw, lds_poly, lp, k);
            uint32_t *dst = a.out_int + ((size_t)item * a.per + sub) * N + lp * 16;
#pragma unroll
            for (int r = 0; r < 16; r++) {
                uint32_t v = SG ? canon_s32(x[r], k) : csub(Mod<MODE_U32>::mulc(x[r], 1u, k.one_p, k), k.q);
                if (OP == DOP_KEYPREP) v = csub(Mod<MODE_U32>::mulc(v, a.r2, a.r2p, k), k.q);
                x[r] = v;
            }
            if (valid) {
#pragma unroll
                for (int c = 0; c < 4; c++) reinterpret_cast<uint4 *>(dst)[c] = make_uint4(x[4 * c], x[4 * c + 1], x[4 * c + 2], x[4 * c + 3]);
            }
            continue;
        }

        // NTT-domain product(s) in internal order (16 contiguous words per lane)
        const uint32_t kidx = (a.key_item_stride && a.key_of) ? a.key_of[item] : item;
        if (OP == DOP_MATVEC) {
#pragma unroll
            for (int r = 0; r < 16; r++) x[r] = 0;
            for (int j = 0; j < a.l; j++) {
                const uint4 *py = reinterpret_cast<const uint4 *>(a.yh + ((size_t)item * a.l + j) * N + lp * 16);
                if (a.key_nat) {
                    // internal index 16 lp + r is natural index bitrev8(16 lp + r) = 16 bitrev4(r) + bitrev4(lp): the 16 lanes of a polynomial read 64
                    // contiguous bytes per load.  montmul(y R, a) = y a: a matrix that is used once per key is never converted or re-laid-out.
                    const size_t poly = a.key_nat == 2 ? (size_t)j * a.kk + sub : (size_t)sub * a.l + j;
                    const uint32_t *pn = a.key + (size_t)kidx * a.key_item_stride + poly * N + (__builtin_bitreverse32((uint32_t)lp) >> 28);
                    uint32_t vy[16];
#pragma unroll
                    for (int c = 0; c < 4; c++) { const uint4 q = py[c]; vy[4 * c] = q.x; vy[4 * c + 1] = q.y; vy[4 * c + 2] = q.z; vy[4 * c + 3] = q.w; }
#pragma unroll
                    for (int r = 0; r < 16; r++) {
                        constexpr int BR4[16] = { 0, 8, 4, 12, 2, 10, 6, 14, 1, 9, 5, 13, 3, 11, 7, 15 };
                        const uint32_t va = SG ? reduce_s32((int32_t)pn[BR4[r] * 16], k) : reduce_i32((int32_t)pn[BR4[r] * 16], k);
                        if constexpr (SG) x[r] += montmul_s32(vy[r], va, k);
                        else x[r] = csub(x[r] + montmul(vy[r], va, k), k.q2);
                    }
                    continue;
                }
                const uint4 *pa = reinterpret_cast<const uint4 *>(a.key + (size_t)kidx * a.key_item_stride + ((size_t)sub * a.l + j) * N + lp * 16);
#pragma unroll
                for (int c = 0; c < 4; c++) {
                    const uint4 vy = py[c], va = pa[c];
                    if constexpr (SG) {          // terms in (-q, q), summed as they are: |x| < l q
                        x[4 * c + 0] += montmul_s32(vy.x, va.x, k); x[4 * c + 1] += montmul_s32(vy.y, va.y, k);
                        x[4 * c + 2] += montmul_s32(vy.z, va.z, k); x[4 * c + 3] += montmul_s32(vy.w, va.w, k);
                        continue;
                    }
                    x[4 * c + 0] = csub(x[4 * c + 0] + montmul(vy.x, va.x, k), k.q2);
                    x[4 * c + 1] = csub(x[4 * c + 1] + montmul(vy.y, va.y, k), k.q2);
                    x[4 * c + 2] = csub(x[4 * c + 2] + montmul(vy.z, va.z, k), k.q2);
                    x[4 * c + 3] = csub(x[4 * c + 3] + montmul(vy.w, va.w, k), k.q2);
                }
            }
        } else {
            const uint4 *pc = reinterpret_cast<const uint4 *>(a.ch + (size_t)item * N + lp * 16);
            const uint4 *pk = reinterpret_cast<const uint4 *>(a.key + (size_t)kidx * a.key_item_stride + (size_t)sub * N + lp * 16);
#pragma unroll
            for (int c = 0; c < 4; c++) {
                const uint4 vc = pc[c], vk = pk[c];
                if constexpr (SG) {
                    x[4 * c + 0] = montmul_s32(vc.x, vk.x, k); x[4 * c + 1] = montmul_s32(vc.y, vk.y, k);
                    x[4 * c + 2] = montmul_s32(vc.z, vk.z, k); x[4 * c + 3] = montmul_s32(vc.w, vk.w, k);
                    continue;
                }
                x[4 * c + 0] = montmul(vc.x, vk.x, k); x[4 * c + 1] = montmul(vc.y, vk.y, k);
                x[4 * c + 2] = montmul(vc.z, vk.z, k); x[4 * c + 3] = montmul(vc.w, vk.w, k);
            }
        }
        const size_t pk_off = ((size_t)item * a.kk + sub) * N + lp;     // [item][k][n] arrays
        // the epilogue's second operand is requested before the transform, not after it
        int32_t pre[16];
        uint8_t preh[16];
#pragma unroll
        for (int r = 0; r < 16; r++) {
            if (OP == DOP_CT1) { pre[r] = a.w[pk_off + r * LP]; preh[r] = a.h[pk_off + r * LP]; }
        }
        inv_core<L8, MODE>(x, tw, lds_poly, lp, k, a.c0, a.c0p, a.c1, a.c1p);    // canonical, natural order j = r*16+lp
        if (!valid) continue;

        if (OP == DOP_MATVEC) {
            if (a.w_chunked) {
                int4 *d4 = reinterpret_cast<int4 *>(a.w + ((size_t)item * a.kk + sub) * N) + lp;
#pragma unroll
                for (int c = 0; c < 4; c++) d4[16 * c] = make_int4((int32_t)x[4 * c], (int32_t)x[4 * c + 1], (int32_t)x[4 * c + 2], (int32_t)x[4 * c + 3]);
            }
#pragma unroll
            for (int r = 0; r < 16; r++) {
                int32_t a0;
                if (!a.w_chunked) a.w[pk_off + r * LP] = (int32_t)x[r];
                if (a.w1) a.w1[pk_off + r * LP] = (uint8_t)round_alpha(x[r], a0, a.alpha, a.am, (int32_t)k.q);   // high_order_bits
            }
        } else if (OP == DOP_CT1) {           // verify: w' = A z - c t1 2^d ; w1 = UseHint(h, w')   (dilithium.c:1286-1312)
            const int32_t m = ((int32_t)k.q - 1) / a.alpha, m_mask = m - 1;
#pragma unroll
            for (int r = 0; r < 16; r++) {
                const uint32_t ww = (uint32_t)pre[r];
                const uint32_t t = csub(ww + k.q - x[r], k.q);
                int32_t a0;
                int32_t a1 = (int32_t)round_alpha(t, a0, a.alpha, a.am, (int32_t)k.q);
                if (preh[r] == 1) { if (a0 > 0) a1++; else a1--; }
                if (a1 < 0) a1 += m;
                a.w1[pk_off + r * LP] = (uint8_t)(a1 & m_mask);
            }
        }
    }
}

// w = A y for one attempt per 16-lane group, NTT(y) never leaving the registers: the LL forward transforms of y, then for each of the kk rows
// the LL products with the (constant, L2-resident) matrix row, one inverse transform and the MATVEC epilogue (w, high_order_bits).
// Replaces the pair <FWD_INT> + <MATVEC> where the matrix is the same for the whole batch: that pair writes NTT(y) to HBM and reads it back
// kk times (the product kernel runs at the L2 bandwidth of those re-reads); here the four attempts of a wavefront also share every matrix load.
// The matrix is in PLAIN form with Shoup companions (4-instruction products, terms in [0, 2q) summed without reduction: LL <= 5 terms stay
// below 2^27), which the separate kernels could not afford (it doubles the matrix traffic they are bound by).
// MODE_S32: the transforms run on signed lazy residues; NTT(y) is kept as x + 32q (positive, below 2^31 - the unsigned Shoup product with the
// plain matrix takes any 32-bit operand) instead of canonical, and the LL <= 5 terms in [0, 2q) go into the inverse transform unreduced.
// ALDS: the matrix (plain values | Shoup companions, 2 kk LL KB) is copied into LDS once per (persistent) workgroup and every product reads it from there:
// with two wavefronts per SIMD the L2 round trip of the 8 matrix loads per (row, column) was what the wavefronts waited for (49 % of their cycles in
// s_waitcnt, tools/prof_kernel_pmc.sh).  The host picks it when the matrix and the transforms' exchange image fit twice into a CU's 160 KB.
// CT1 (round 4, verification): the row's c t1 2^d is subtracted IN the NTT domain (NTT(c) from the launch before, NTT(t1 2^d) a key constant) and the row's epilogue
// is UseHint -> w1: w' = A z - c t1 2^d never exists in HBM, the k inverse transforms of c t1 2^d and the DOP_CT1 launch are gone (dilithium.c:1286-1312).
template <int LL, int MODE, bool ALDS, bool CT1 = false>
__global__ __launch_bounds__(256, 2) void dil_yaw_kernel(DilNtt a)
{
    constexpr bool SG = MODE == MODE_S32;
    extern __shared__ __attribute__((aligned(16))) uint32_t lds[];          // [4 * 1024] exchange images, then (ALDS) the matrix and its companions
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const int pw = lane / LP, lp = lane % LP;
    uint32_t *lds_poly = lds + wid * 1024 + pw * N;
    const ModK k = a.k;
    const uint32_t *Am = a.key, *Ap = a.keyp;
    if constexpr (ALDS) {
        const uint32_t words = (uint32_t)a.kk * LL * N;
        uint4 *dm = reinterpret_cast<uint4 *>(lds + 4 * 1024), *dp = dm + words / 4;
        const uint4 *sm = reinterpret_cast<const uint4 *>(a.key), *sp = reinterpret_cast<const uint4 *>(a.keyp);
        for (uint32_t i = threadIdx.x; i < words / 4; i += 256) { dm[i] = sm[i]; dp[i] = sp[i]; }
        __syncthreads();
        Am = lds + 4 * 1024; Ap = Am + words;
    }
    TwSet<L8> twi;
    twi.load(a.zi, a.zip, lp);
    const uint32_t njobs = a.njobs_dev ? *a.njobs_dev : a.njobs;   // attempts
    const uint32_t groups = (njobs + PW - 1) / PW;
    for (uint32_t g = blockIdx.x * 4 + wid; g < groups; g += gridDim.x * 4) {
        const uint32_t slot = g * PW + pw;
        const bool valid = slot < njobs;
        const uint32_t s = valid ? slot : 0;                        // unconditional loads: an attempt past the end reads attempt 0, stores nothing
        uint32_t yh[LL][16];
        // the forward twiddles are (re)loaded per attempt group (L1/L2 hits) and die before the row loop: 30 registers the LL * 16 words of NTT(y)
        // need more; the opaque copy of the pointer keeps the compiler from hoisting the loads out of the persistent loop
        const uint32_t *zf = a.zf, *zfp = a.zfp;
        keep_scalar(zf, zfp);
        TwSet<L8> twf;
        twf.load(zf, zfp, lp);
#pragma unroll
        for (int j = 0; j < LL; j++) {
            const int32_t *src = a.in + ((size_t)s * LL + j) * N + lp;
            uint32_t x[16];
#pragma unroll
            for (int r = 0; r < 16; r++) x[r] = SG ? reduce_s32(src[r * LP], k) : reduce_i32(src[r * LP], k);
            fwd_core<L8, MODE, true>(x, twf, lds_poly, lp, k);
#pragma unroll
            for (int r = 0; r < 16; r++) yh[j][r] = SG ? x[r] + (k.q << 5) : csub(Mod<MODE_U32>::mulc(x[r], 1u, k.one_p, k), k.q);
        }
        for (int i = 0; i < a.kk; i++) {
            uint32_t x[16];
#pragma unroll
            for (int r = 0; r < 16; r++) x[r] = 0;
#pragma unroll
            for (int j = 0; j < LL; j++) {
                const size_t ko = ((size_t)i * LL + j) * N + lp * 16;
                const uint4 *pa = reinterpret_cast<const uint4 *>(Am + ko), *pp = reinterpret_cast<const uint4 *>(Ap + ko);
#pragma unroll
                for (int c = 0; c < 4; c++) {
                    const uint4 va = pa[c], vp = pp[c];
                    x[4 * c + 0] += Mod<MODE_U32>::mulc(yh[j][4 * c + 0], va.x, vp.x, k);
                    x[4 * c + 1] += Mod<MODE_U32>::mulc(yh[j][4 * c + 1], va.y, vp.y, k);
                    x[4 * c + 2] += Mod<MODE_U32>::mulc(yh[j][4 * c + 2], va.z, vp.z, k);
                    x[4 * c + 3] += Mod<MODE_U32>::mulc(yh[j][4 * c + 3], va.w, vp.w, k);
                }
            }
            if constexpr (!SG) {
#pragma unroll
                for (int r = 0; r < 16; r++) x[r] = Mod<MODE_U32>::mulc(x[r], 1u, k.one_p, k);
            }
            uint32_t hbits = 0;                                         // CT1: the row's sixteen hints of this lane, one bit each (sixteen live bytes spilled)
            if constexpr (CT1) {
                const size_t hoff = ((size_t)s * a.kk + i) * N + lp;     // requested first, packed behind the products: in flight under them, one register across the transform
                uint8_t preh[16];
#pragma unroll
                for (int r = 0; r < 16; r++) preh[r] = a.h[hoff + r * LP];
                const uint4 *pc = reinterpret_cast<const uint4 *>(a.ch + (size_t)s * N + lp * 16), *pk = reinterpret_cast<const uint4 *>(a.t1s + (size_t)i * N + lp * 16);
#pragma unroll
                for (int c = 0; c < 4; c++) {
                    const uint4 vc = pc[c], vk = pk[c];
                    if constexpr (SG) {                                 // one more term of magnitude below q beside the LL terms in [0, 2q): the host checks 16 (2 LL + 1) q < 2^31
                        x[4 * c + 0] -= montmul_s32(vc.x, vk.x, k); x[4 * c + 1] -= montmul_s32(vc.y, vk.y, k);
                        x[4 * c + 2] -= montmul_s32(vc.z, vk.z, k); x[4 * c + 3] -= montmul_s32(vc.w, vk.w, k);
                    } else {
                        x[4 * c + 0] = csub(x[4 * c + 0] + k.q2 - montmul(vc.x, vk.x, k), k.q2); x[4 * c + 1] = csub(x[4 * c + 1] + k.q2 - montmul(vc.y, vk.y, k), k.q2);
                        x[4 * c + 2] = csub(x[4 * c + 2] + k.q2 - montmul(vc.z, vk.z, k), k.q2); x[4 * c + 3] = csub(x[4 * c + 3] + k.q2 - montmul(vc.w, vk.w, k), k.q2);
                    }
                }
#pragma unroll
                for (int r = 0; r < 16; r++) hbits |= (uint32_t)(preh[r] == 1) << r;
            }
            inv_core<L8, MODE>(x, twi, lds_poly, lp, k, a.c0, a.c0p, a.c1, a.c1p);    // canonical, natural order j = r*16+lp
            if (!valid) continue;
            const size_t off = ((size_t)slot * a.kk + i) * N + lp;
            if constexpr (CT1) {                                        // w1 = UseHint(h, w')
                const int32_t m = ((int32_t)k.q - 1) / a.alpha, m_mask = m - 1;
#pragma unroll
                for (int r = 0; r < 16; r++) {
                    int32_t a0;
                    int32_t a1 = (int32_t)round_alpha(x[r], a0, a.alpha, a.am, (int32_t)k.q);
                    if ((hbits >> r) & 1u) { if (a0 > 0) a1++; else a1--; }
                    if (a1 < 0) a1 += m;
                    a.w1[off + r * LP] = (uint8_t)(a1 & m_mask);
                }
                continue;
            }
            if (a.w_chunked == 2) {                                     // 24 bits per coefficient (q < 2^24): the lane's 16 values as three 16-byte pieces, 768 bytes per polynomial
                int4 *d4 = reinterpret_cast<int4 *>(a.w + ((size_t)slot * a.kk + i) * W24) + lp;
#pragma unroll
                for (int c = 0; c < 3; c++) {
                    uint32_t pk[4];
#pragma unroll
                    for (int e = 0; e < 4; e++) pk[e] = pack24_word(x, 4 * c + e);
                    d4[16 * c] = make_int4((int32_t)pk[0], (int32_t)pk[1], (int32_t)pk[2], (int32_t)pk[3]);
                }
            } else if (a.w_chunked) {
                int4 *d4 = reinterpret_cast<int4 *>(a.w + ((size_t)slot * a.kk + i) * N) + lp;
#pragma unroll
                for (int c = 0; c < 4; c++) d4[16 * c] = make_int4((int32_t)x[4 * c], (int32_t)x[4 * c + 1], (int32_t)x[4 * c + 2], (int32_t)x[4 * c + 3]);
            }
#pragma unroll
            for (int r = 0; r < 16; r++) {
                int32_t a0;
                if (!a.w_chunked) a.w[off + r * LP] = (int32_t)x[r];
                if (a.w1) a.w1[off + r * LP] = (uint8_t)round_alpha(x[r], a0, a.alpha, a.am, (int32_t)k.q);   // high_order_bits
            }
        }
    }
}

// the constant matrix for dil_yaw_kernel: natural-order NTT-domain input -> internal order, plain canonical values + Shoup companions
static __global__ void dil_to_internal_shoup(const int32_t *in, uint32_t *out, uint32_t *outp, size_t npoly, ModK k)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= npoly * N) return;
    const size_t p = i / N; const uint32_t j = (uint32_t)(i % N);
    const uint32_t nat = __builtin_bitreverse32(j) >> 24;
    const uint32_t v = csub(reduce_i32(in[p * N + nat], k), k.q);
    out[i] = v;
    outp[i] = (uint32_t)(((uint64_t)v << 32) / k.q);
}

// natural-order NTT-domain array -> internal order, Montgomery form (for the expanded matrix A)
// rows/cols != 0: the input is a sequence of [cols][rows][n] matrices, each transposed on the way (out: [rows][cols][n] each)
static __global__ void dil_to_internal_mont(const int32_t *in, uint32_t *out, size_t npoly, ModK k, uint32_t r2, uint32_t r2p, int rows = 0, int cols = 0)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= npoly * N) return;
    size_t p = i / N; const uint32_t j = (uint32_t)(i % N);
    if (rows) { const size_t rc = (size_t)rows * cols, q = p % rc; p = p - q + (q % (size_t)cols) * (size_t)rows + q / (size_t)cols; }   // per matrix
    const uint32_t nat = __builtin_bitreverse32(j) >> 24;
    uint32_t v = reduce_i32(in[p * N + nat], k);
    out[i] = csub(Mod<MODE_U32>::mulc(v, r2, r2p, k), k.q);
}

}  // namespace sc_mlwe
