// emu_falcon_sign.cpp - a Falcon signing call of the engine run on the CPU lane by lane: the kernels' own text (libsafecrypto_amd/csrc/falcon_sign_dev.h,
// keccak_dev.h) compiled as host code over wave_emu.h - binary64 with one rounding per operation, as on the device (-ffp-contract=off) - and queued as
// fal_sign_core (csrc/falcon.hip) queues them: the hash to the ring (one lane or two per message), the fast Fourier sampler over the LDL tree in the mapping
// fal_launch_sample would pick or one it can be told to (eight signatures per wavefront with the lower tree in LDS and a second wavefront that draws the
// samples ahead - the default -, the same with one wavefront, with 64-coefficient sub-trees, a lane per signature, eight lanes per signature), the
// finishing kernel (centring, the float norm test, the packing, the zeroing of the per-call secrets).
// Test infrastructure (tests/test_emu_falcon_sign.py); nothing here is part of the product.
#pragma clang fp contract(off)
#define EMU_SHARED static
#define EMU_HAVE_NTT_CORE
#include "wave_emu.h"
#include "keccak_dev.h"
#include "ntt_kernel.h"

namespace sc { static inline int pow2_shift(int per) { return 31 - __builtin_clz((unsigned)per); } }      // csrc/engine_internal.h:47 (a header of the host runtime)

namespace {
#include "falcon_sign_dev.h"
#include "falcon_verify_dev.h"
}

namespace {
template <typename F> void launch(unsigned grid, unsigned block, F &&f) { if (grid) emu::launch(grid, block, std::function<void()>(f)); }

struct Args {
    uint32_t batch; double *ws; const int32_t *c; const uint32_t *tapes; size_t tape_stride; const uint32_t *tape_lens; const double *sk, *gm; const uint64_t *tabs;
    int32_t *s1, *s2; uint32_t *status; FalP P; double ni; size_t sk_stride, tab_stride;
};
template <int TSL, int SPW, int LM, int NW> void w8(const Args &a)
{
    launch((a.batch + SPW - 1) / SPW, 64 * NW, [&]() { fal_sample_w8_kernel<TSL, false, SPW, LM, NW>(a.batch, a.ws, a.c, a.tapes, a.tape_stride, a.tape_lens, a.sk, a.gm, a.tabs, a.s1, a.s2, a.status, a.P, a.ni,
                                                                                                  nullptr, nullptr, 1u, a.sk_stride, a.tab_stride); });
}
template <int TSL, int G> void cols(const Args &a)
{
    const size_t threads = G == 8 ? (size_t)a.batch * 8 : a.batch;
    launch((unsigned)((threads + ST - 1) / ST), ST, [&]() { fal_sample_kernel<TSL, G>(a.batch, (size_t)a.batch, a.ws, a.c, a.tapes, a.tape_stride, a.tape_lens, a.sk, a.gm, a.tabs, a.s1, a.s2, a.status, a.P, a.ni,
                                                                                   nullptr, 1u, a.sk_stride, a.tab_stride, 0); });
}
template <int TSL> int sample(const Args &a, int variant)
{
    switch (variant) {
    case 0: w8<TSL, 8, 7, 2>(a); return 0;          // what fal_launch_sample launches
    case 1: w8<TSL, 8, 7, 1>(a); return 0;
    case 2: w8<TSL, 8, 6, 1>(a); return 0;
    case 3: cols<TSL, 1>(a); return 0;
    case 4: cols<TSL, 8>(a); return 0;
    }
    return 1;
}
}  // namespace

// par: the 8 words of FalP; sk: the expanded private key ((n_bits + 5) n doubles); gm: the 2048 doubles of the root table; tabs: the n leaf tables,
// 2^ts_log entries each, in visiting order (as sc_hip_falcon_set_privkeys tabulates them).  variant: see sample() above; two_lane: the hash kernel's form.
extern "C" int emu_falcon_sign(const int32_t *par, const double *sk, const double *gm, const uint64_t *tabs, int ts_log,
                               uint32_t batch, const uint8_t *msgs, size_t msg_stride, const uint32_t *mlens,
                               const uint32_t *tapes, size_t tape_stride, const uint32_t *tape_lens, int variant, int two_lane,
                               uint8_t *sigs, size_t sig_stride, uint32_t *siglens, uint32_t *status, int32_t *s1_out, int32_t *s2_out, uint32_t *residue)
{
    FalP P;
    memcpy(&P, par, sizeof P);
    const size_t n = (size_t)P.n;
    if ((n != 512 && n != 1024) || tape_stride < 4 * n || (tape_stride & 3)) return 1;
    const uint32_t B = batch;
    const size_t rows = (B + SPW_MAX - 1) / SPW_MAX * SPW_MAX;
    std::vector<double> ws(rows * 6 * n, 0.0);
    std::vector<int32_t> c((size_t)B * n), s1((size_t)B * n), s2((size_t)B * n);
    for (uint32_t i = 0; i < B; i++) status[i] = 0;
    if (two_lane) launch((2 * B + HT - 1) / HT, HT, [&]() { fal_hash2_kernel(B, msgs, msg_stride, mlens, 0u, c.data(), P, 0); });
    else launch((B + HT - 1) / HT, HT, [&]() { fal_hash1_kernel(B, msgs, msg_stride, mlens, 0u, c.data(), P, 0); });
    Args a{ B, ws.data(), c.data(), tapes, tape_stride, tape_lens, sk, gm, tabs, s1.data(), s2.data(), status, P, 1.0 / (double)(long)P.q,
            (size_t)(P.n_bits + 5) * n, n << ts_log };
    int rc;
    switch (ts_log) {
    case 4: rc = sample<4>(a, variant); break;
    case 5: rc = sample<5>(a, variant); break;
    case 6: rc = sample<6>(a, variant); break;
    default: return 2;
    }
    if (rc) return 3;
    if (s1_out) memcpy(s1_out, s1.data(), s1.size() * 4);
    if (s2_out) memcpy(s2_out, s2.data(), s2.size() * 4);
    if (sizeof(int32_t) * FW * n > sizeof emu::dyn_lds) return 4;
    launch((B + FW - 1) / FW, 64 * FW, [&]() { fal_finish_kernel(B, s1.data(), s2.data(), sigs, sig_stride, siglens, status, P, ws.data(), (uint32_t)rows); });
    uint64_t res = 0;
    for (double v : ws) { uint64_t b; memcpy(&b, &v, 8); res |= b; }
    for (int32_t v : s1) res |= (uint32_t)v;
    for (int32_t v : s2) res |= (uint32_t)v;
    *residue = res ? 1u : 0u;
    return 0;
}

// A verification call, one key for the batch: sc_hip_falcon_set_pubkey (h reduced and transformed on the NTT kernel) and fal_verify_core - the hash to the ring
// and the signature's unpacking as one launch, s2 h on the NTT kernel with the key broadcast, the finishing kernel (s1 = c - s2 h centred, the float norm
// test).  tab / modk / cf / cm: the plan's table, ModK and the scale constants for OP_FWD and for OP_MULNTT in `mode`.  verdicts[i] = 0: valid.
extern "C" int emu_falcon_verify(const int32_t *par, int mode, const uint32_t *tab, const uint32_t *modk, const uint32_t *cf, const uint32_t *cm, const int16_t *h,
                                 uint32_t batch, const uint8_t *msgs, size_t msg_stride, const uint32_t *mlens,
                                 const uint8_t *sigs, size_t sig_stride, const uint32_t *siglens, uint32_t *verdicts)
{
    FalP P;
    memcpy(&P, par, sizeof P);
    const int n = P.n;
    if ((n != 512 && n != 1024) || (mode != MODE_SMAD && mode != MODE_S32)) return 1;
    const uint32_t B = batch;
    static int32_t sink[1024];
    const ModK mk{ modk[0], modk[1], modk[2], modk[3], modk[4], modk[5] };
    auto args = [&](const uint32_t *c) { NttArgs a{}; a.sink = sink; a.zf = tab; a.zfp = tab + n; a.zi = tab + 2 * n; a.zip = tab + 3 * n; a.k = mk; a.c0 = c[0]; a.c0p = c[1]; a.c1 = c[2]; a.c1p = c[3]; return a; };
    // ---- sc_hip_falcon_set_pubkeys
    std::vector<int32_t> hhat(n);
    for (int i = 0; i < n; i++) { int32_t v = h[i] % P.q; if (v < 0) v += P.q; hhat[i] = v; }
    {
        NttArgs a = args(cf);
        a.in0 = hhat.data(); a.out = hhat.data(); a.batch = 1;
        if (n == 512) { if (mode == MODE_SMAD) launch(1, 256, [&]() { ntt_kernel<9, MODE_SMAD, true, OP_FWD, false>(a); }); else launch(1, 256, [&]() { ntt_kernel<9, MODE_S32, true, OP_FWD, false>(a); }); }
        else { if (mode == MODE_SMAD) launch(1, 256, [&]() { ntt_kernel<10, MODE_SMAD, true, OP_FWD, false>(a); }); else launch(1, 256, [&]() { ntt_kernel<10, MODE_S32, true, OP_FWD, false>(a); }); }
    }
    // ---- fal_verify_core
    std::vector<int32_t> c((size_t)B * n), s1((size_t)B * n), s2((size_t)B * n);
    for (uint32_t i = 0; i < B; i++) verdicts[i] = 0;
    const size_t ne = (size_t)B * n;
    launch((2 * B + HT - 1) / HT + (unsigned)((ne / 2 + 255) / 256), 256, [&]() { fal_hash2_unpack_kernel(B, msgs, msg_stride, mlens, 0u, c.data(), P, 0, sigs, sig_stride, siglens, s2.data(), verdicts, P.q_bits - 2, 1); });
    {
        NttArgs a = args(cm);
        a.in0 = s2.data(); a.in1 = hhat.data(); a.out = s1.data(); a.in1_bcast = 1u; a.batch = B;
        const int pw = 1024 / n;
        const unsigned blocks = (unsigned)(((B + pw - 1) / pw + 3) / 4);
        if (n == 512) { if (mode == MODE_SMAD) launch(blocks, 256, [&]() { ntt_kernel<9, MODE_SMAD, true, OP_MULNTT, false>(a); }); else launch(blocks, 256, [&]() { ntt_kernel<9, MODE_S32, true, OP_MULNTT, false>(a); }); }
        else { if (mode == MODE_SMAD) launch(blocks, 256, [&]() { ntt_kernel<10, MODE_SMAD, true, OP_MULNTT, false>(a); }); else launch(blocks, 256, [&]() { ntt_kernel<10, MODE_S32, true, OP_MULNTT, false>(a); }); }
    }
    if (sizeof(int32_t) * FW * (size_t)n > sizeof emu::dyn_lds) return 2;
    launch((B + FW - 1) / FW, 64 * FW, [&]() { fal_vfinish_kernel(B, c.data(), s1.data(), s2.data(), verdicts, P); });
    return 0;
}
