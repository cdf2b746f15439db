// mlwe_kernels.h - the NTT-shaped kernels of the module-LWE schemes (Dilithium, Kyber): n = 256, 16 lanes per
// polynomial, 4 polynomials per wavefront, 32-bit Shoup arithmetic, NTT-domain data kept in the engine's internal order
// (position j = lane*16 + r holds natural index bitrev8(j): 16 contiguous words per lane).  Included by dilithium.hip
// and kyber.hip; everything has internal linkage.
#pragma once
#include "engine_internal.h"
#include "mlwe_dev.h"
#include <string.h>

namespace sc_mlwe {

// kernel arguments common to every launch for plan p (k x l module, rounding modulus alpha = 2*gamma_2)
static inline DilNtt mlwe_base_args(sc_hip_plan *p, int kk, int l, int32_t gamma2)
{
    DilNtt a;
    memset(&a, 0, sizeof a);
    const int n = p->n;
    // signed 32-bit arithmetic where its bounds hold: q < 2^25, and the largest value entering an inverse transform (l terms below q from the
    // Montgomery products, or l terms below 2q from the fused kernel's Shoup products) times 16 stays below 2^31
    const int lm = l > kk ? l : kk;
    a.sg = (p->s32_ok && p->d_tab_s && 32ull * (uint64_t)lm * (uint64_t)p->q < (1ull << 31)) ? 1 : 0;
    a.zf = a.sg ? p->d_tab_s : p->d_tab[1]; a.zfp = a.zf + n; a.zi = a.zf + 2 * n; a.zip = a.zf + 3 * n;
    a.k = p->k[1];
    const uint32_t *c = a.sg ? p->c_inv_s : p->c_inv[1];
    a.c0 = c[0]; a.c0p = c[1]; a.c1 = c[2]; a.c1p = c[3];
    a.r2 = p->r2; a.r2p = p->r2p;
    a.l = l; a.kk = kk;
    a.alpha = 2 * gamma2; a.am = (1 << 30) / a.alpha;
    return a;
}

template <int OP>
static int mlwe_launch(sc_hip_ctx *ctx, DilNtt a, uint32_t nitems, int per, hipStream_t s)
{
    a.per = per; a.njobs = nitems * (uint32_t)per;
    if (a.njobs == 0) return SC_HIP_SUCCESS;
    size_t blocks = ((size_t)a.njobs + 15) / 16;
    const size_t cap = (size_t)ctx->cus * 8;
    if (blocks > cap) blocks = cap;
    if (a.sg) hipLaunchKernelGGL((dil_ntt_kernel<OP, MODE_S32>), dim3((unsigned)blocks), dim3(256), 0, s, a);
    else hipLaunchKernelGGL((dil_ntt_kernel<OP, MODE_U32>), dim3((unsigned)blocks), dim3(256), 0, s, a);
    HIPCHK(hipGetLastError());
    return SC_HIP_SUCCESS;
}

// fused NTT(y) -> A NTT(y) -> INTT for `nitems` attempts (l = 2..5); returns SC_HIP_FAILURE for other l (the caller falls back to the pair)
static int mlwe_launch_yaw(sc_hip_ctx *ctx, DilNtt a, uint32_t nitems, hipStream_t s, bool ct1 = false)
{
    a.njobs = nitems; a.per = 1;
    if (nitems == 0) return SC_HIP_SUCCESS;
    size_t blocks = ((size_t)nitems + 15) / 16;
    const size_t cap = (size_t)ctx->cus * 2;                       // two 256-thread workgroups per CU (256 registers per lane)
    if (blocks > cap) blocks = cap;
    const size_t ex = 4 * 1024 * sizeof(uint32_t), mat = (size_t)2 * a.kk * a.l * N * sizeof(uint32_t);
    const bool alds = 2 * (ex + mat) <= 160 * 1024;              // both resident workgroups of a CU hold a copy
#define SC_YAW_(LLv, M, AL) do { \
        static bool attr_done = false; \
        if (!attr_done) { HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void *>(&dil_yaw_kernel<LLv, M, AL>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)(ex + (AL ? 80 * 1024 - ex : 0)))); attr_done = true; } \
        hipLaunchKernelGGL((dil_yaw_kernel<LLv, M, AL>), dim3((unsigned)blocks), dim3(256), AL ? ex + mat : ex, s, a); } while (0)
#define SC_YAWC_(LLv, M, AL) do { \
        static bool attr_done = false; \
        if (!attr_done) { HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void *>(&dil_yaw_kernel<LLv, M, AL, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)(ex + (AL ? 80 * 1024 - ex : 0)))); attr_done = true; } \
        hipLaunchKernelGGL((dil_yaw_kernel<LLv, M, AL, true>), dim3((unsigned)blocks), dim3(256), AL ? ex + mat : ex, s, a); } while (0)
#define SC_YAW(LLv) do { if (ct1) { if (a.sg) { if (alds) SC_YAWC_(LLv, MODE_S32, true); else SC_YAWC_(LLv, MODE_S32, false); } \
                                   else { if (alds) SC_YAWC_(LLv, MODE_U32, true); else SC_YAWC_(LLv, MODE_U32, false); } } \
                        else if (a.sg) { if (alds) SC_YAW_(LLv, MODE_S32, true); else SC_YAW_(LLv, MODE_S32, false); } \
                        else { if (alds) SC_YAW_(LLv, MODE_U32, true); else SC_YAW_(LLv, MODE_U32, false); } } while (0)
    switch (a.l) {
    case 2: SC_YAW(2); break;
    case 3: SC_YAW(3); break;
    case 4: SC_YAW(4); break;
    case 5: SC_YAW(5); break;
    default: return SC_HIP_FAILURE;
    }
#undef SC_YAW
#undef SC_YAW_
#undef SC_YAWC_
    HIPCHK(hipGetLastError());
    return SC_HIP_SUCCESS;
}

}  // namespace sc_mlwe
