// emu_dil_sign.cpp - a whole Dilithium signing call of the engine, one key for the batch, run on the CPU lane by lane: the kernels' own text
// (libsafecrypto_amd/csrc/dil_sign_dev.h, mlwe_dev.h, keccak_dev.h, ntt_core.h, dil_cfins.h) compiled as host code over wave_emu.h, queued in the order
// and with the grids sc_hip_dilithium_set_privkey / sign_core (csrc/dilithium.hip) queue them: key preparation (the key vectors through the transform
// kernel, the matrix into plain + Shoup form), mu, and per round: the round state, ExpandMask, the fused NTT(y) -> A NTT(y) -> INTT kernel, both forms of the
// challenge hash, the finishing kernel (the transform one - the product's default - or the table one), the wipe.  The round LOOP here is a restatement
// of sign_core's (that function is HIP API calls around these launches and cannot be compiled off the device); everything inside a launch is the product's text.
// Test infrastructure (tests/test_emu_dil_sign.py); nothing here is part of the product.
#define EMU_SHARED static
#define EMU_HAVE_NTT_CORE
#include "wave_emu.h"
#include "keccak_dev.h"
#include "mlwe_dev.h"

using namespace sc;
using namespace sc_mlwe;

namespace {
#include "dil_sign_dev.h"
#include "dil_verify_dev.h"
}

namespace {
template <typename F> void launch(unsigned grid, unsigned block, F &&f) { if (grid) emu::launch(grid, block, std::function<void()>(f)); }

struct Plan { const uint32_t *tab; ModK k; uint32_t c[4], r2, r2p; int sg; };

DilNtt base_args(const Plan &p, const DilP &P)
{
    DilNtt a;
    memset(&a, 0, sizeof a);
    const int n = P.n;
    a.sg = p.sg;
    a.zf = p.tab; a.zfp = a.zf + n; a.zi = a.zf + 2 * n; a.zip = a.zf + 3 * n;
    a.k = p.k;
    a.c0 = p.c[0]; a.c0p = p.c[1]; a.c1 = p.c[2]; a.c1p = p.c[3];
    a.r2 = p.r2; a.r2p = p.r2p;
    a.l = P.l; a.kk = P.k;
    a.alpha = 2 * P.gamma2; a.am = (1 << 30) / a.alpha;
    a.gamma1_beta = P.gamma1 - P.beta; a.gamma2_beta = P.gamma2 - P.beta; a.d = P.d;
    return a;
}

template <int OP> void launch_dil(DilNtt a, uint32_t nitems, int per)          // mlwe_launch
{
    a.per = per; a.njobs = nitems * (uint32_t)per;
    if (!a.njobs) return;
    const unsigned blocks = (a.njobs + 15) / 16;
    if (a.sg) launch(blocks, 256, [&]() { dil_ntt_kernel<OP, MODE_S32>(a); });
    else launch(blocks, 256, [&]() { dil_ntt_kernel<OP, MODE_U32>(a); });
}

template <int LL> void launch_yawc_l(const DilNtt &a, unsigned blocks, bool alds)          // the verifier's form: c t1 2^d and UseHint fused in (CT1)
{
    if (a.sg) { if (alds) launch(blocks, 256, [&]() { dil_yaw_kernel<LL, MODE_S32, true, true>(a); }); else launch(blocks, 256, [&]() { dil_yaw_kernel<LL, MODE_S32, false, true>(a); }); }
    else { if (alds) launch(blocks, 256, [&]() { dil_yaw_kernel<LL, MODE_U32, true, true>(a); }); else launch(blocks, 256, [&]() { dil_yaw_kernel<LL, MODE_U32, false, true>(a); }); }
}
template <int LL> void launch_yaw_l(const DilNtt &a, unsigned blocks, bool alds)
{
    if (a.sg) { if (alds) launch(blocks, 256, [&]() { dil_yaw_kernel<LL, MODE_S32, true>(a); }); else launch(blocks, 256, [&]() { dil_yaw_kernel<LL, MODE_S32, false>(a); }); }
    else { if (alds) launch(blocks, 256, [&]() { dil_yaw_kernel<LL, MODE_U32, true>(a); }); else launch(blocks, 256, [&]() { dil_yaw_kernel<LL, MODE_U32, false>(a); }); }
}
int launch_yaw(DilNtt a, uint32_t nitems, unsigned max_blocks, bool ct1 = false)  // mlwe_launch_yaw
{
    a.njobs = nitems; a.per = 1;
    if (!nitems) return 0;
    unsigned blocks = (nitems + 15) / 16;
    if (blocks > max_blocks) blocks = max_blocks;                                  // (the device's two workgroups per CU: a persistent loop)
    const size_t ex = 4 * 1024 * sizeof(uint32_t), mat = (size_t)2 * a.kk * a.l * N * sizeof(uint32_t);
    const bool alds = 2 * (ex + mat) <= 160 * 1024;
    switch (a.l) {
    case 2: if (ct1) launch_yawc_l<2>(a, blocks, alds); else launch_yaw_l<2>(a, blocks, alds); break;
    case 3: if (ct1) launch_yawc_l<3>(a, blocks, alds); else launch_yaw_l<3>(a, blocks, alds); break;
    case 4: if (ct1) launch_yawc_l<4>(a, blocks, alds); else launch_yaw_l<4>(a, blocks, alds); break;
    case 5: if (ct1) launch_yawc_l<5>(a, blocks, alds); else launch_yaw_l<5>(a, blocks, alds); break;
    default: return 1;
    }
    return 0;
}
}  // namespace

// flags: bit 0 = the finishing kernel without transforms (tuning bit 11; tabs / bias from sc_hip_dilithium_fin_tables_host), bit 1 = one lane per hash
// everywhere (tuning bit 9), bit 2 = a persistent fused kernel (one workgroup: several trips of its loop).
// tab = the plan's table for the arithmetic the product selects ([zf | zfp | zi | zip]), modk = the plan's k[1], cinv = the n^-1 scale constants of that arithmetic.
// Returns 0, or a positive code; *rounds = rounds run.
extern "C" int emu_dil_sign(const int32_t *par, const uint32_t *tab, const uint32_t *modk, const uint32_t *cinv, uint32_t r2, uint32_t r2p, int sg,
                            const int32_t *s1, const int32_t *s2, const int32_t *t, const uint8_t *K, const uint8_t *tr, const int32_t *Ahat,
                            const uint32_t *tabs, uint32_t bias,
                            uint32_t batch, const uint8_t *msgs, size_t msg_stride, const uint32_t *mlens, uint32_t fixed_len,
                            uint32_t cap, uint32_t s_max, int flags,
                            uint8_t *sigs, size_t sig_stride, uint32_t *siglens, uint32_t *trials, uint32_t *rounds, uint32_t *residue)
{
    DilP P;
    memcpy(&P, par, sizeof P);
    const int k = P.k, l = P.l, q = P.q, dd = P.d;
    if (P.n != N || l < 2 || l > 5) return 1;
    Plan pl{ tab, ModK{ modk[0], modk[1], modk[2], modk[3], modk[4], modk[5] }, { cinv[0], cinv[1], cinv[2], cinv[3] }, r2, r2p, sg };

    // ---- sc_hip_dilithium_set_privkey: decompose(t) -> t0, the key vectors through DOP_KEYPREP, the matrix into plain | Shoup form
    std::vector<int32_t> t0((size_t)k * N);
    const int32_t mask = (1 << dd) - 1, sign = 1 << dd, thresh = 1 << (dd - 1);
    for (size_t i = 0; i < t0.size(); i++) { int32_t x = t[i] & mask; x -= (x > thresh) * sign; t0[i] = (x >= 0) ? x : x + q; }
    std::vector<uint32_t> d_s1((size_t)l * N), d_s2((size_t)k * N), d_t0((size_t)k * N), d_As((size_t)2 * k * l * N);
    {
        DilNtt a = base_args(pl, P);
        a.in = s1; a.out_int = d_s1.data(); launch_dil<DOP_KEYPREP>(a, (uint32_t)l, 1);
        a.in = s2; a.out_int = d_s2.data(); launch_dil<DOP_KEYPREP>(a, (uint32_t)k, 1);
        a.in = t0.data(); a.out_int = d_t0.data(); launch_dil<DOP_KEYPREP>(a, (uint32_t)k, 1);
        const size_t np = (size_t)k * l;
        launch((unsigned)((np * N + 255) / 256), 256, [&]() { dil_to_internal_shoup(Ahat, d_As.data(), d_As.data() + np * N, np, pl.k); });
    }

    // ---- sign_core: the workspace (ensure_ws) ...
    const uint32_t B = batch;
    std::vector<uint8_t> mu((size_t)B * 48), w1((size_t)cap * k * N);
    std::vector<uint32_t> kappa(B, 0u), act0(B + 4), act1(B + 4), cnt(16, 0u), rs(16, 0u), slot_item(cap);
    std::vector<int8_t> c8((size_t)cap * N);
    std::vector<int32_t> y((size_t)cap * l * N), w((size_t)cap * k * N);
    launch((B + HT - 1) / HT, HT, [&]() { dil_mu_kernel(tr, msgs, msg_stride, mlens, fixed_len, mu.data(), B, 0u); });

    // ... and the rounds
    const bool sparse_fin = (flags & 1) && tabs && l == k - 1 && k >= 3 && k <= 6;
    const bool one_lane = (flags & 2) != 0;
    const size_t fin_words = 4 + 4 * cfin_wave_words(P), fins_words = 4 + cfins_table_words(P) + 4 * cfins_wave_words(P);
    if ((sparse_fin ? fins_words : fin_words) * 4 > sizeof emu::dyn_lds) return 2;
    DilFin f;
    memset(&f, 0, sizeof f);
    f.c8 = c8.data(); f.s1 = d_s1.data(); f.s2 = d_s2.data(); f.t0 = d_t0.data(); f.s1_stride = 0; f.k_stride = 0;
    f.rs = rs.data(); f.kappa = kappa.data(); f.sigs = sigs; f.sig_stride = sig_stride; f.siglens = siglens; f.trials = trials; f.next_count = cnt.data(); f.P = P;
    const bool exp2 = !one_lane;
    const unsigned eb = (unsigned)(((size_t)cap * (exp2 ? 2 : 1) + EXP_T - 1) / EXP_T) + 1, hb = (cap + CH_T - 1) / CH_T;
    const uint32_t w_words = (uint32_t)(k * P.n);
    uint32_t round = 0, left = B;
    while (left > 0) {
        if (round > 200) return 3;
        const uint32_t *active = round ? (((round - 1) & 1) ? act1.data() : act0.data()) : nullptr;
        f.active = active; f.next_active = (round & 1) ? act1.data() : act0.data();
        launch(1, 64, [&]() { dil_round_kernel(rs.data(), cnt.data(), round ? 0u : B, cap, s_max); });
        if (exp2) launch(eb, EXP_T, [&]() { dil_expand_kernel<true>(rs.data(), active, kappa.data(), K, mu.data(), y.data(), l * P.n, P.gamma1, P.q, 0u, slot_item.data(), w.data(), w_words, nullptr); });
        else launch(eb, EXP_T, [&]() { dil_expand_kernel<false>(rs.data(), active, kappa.data(), K, mu.data(), y.data(), l * P.n, P.gamma1, P.q, 0u, slot_item.data(), w.data(), w_words, nullptr); });
        DilNtt a = base_args(pl, P);
        a.y = y.data(); a.w = w.data(); a.w1 = w1.data();
        a.in = y.data(); a.njobs_dev = rs.data() + RS_NSLOTS; a.w_chunked = 1; a.key_of = slot_item.data();
        a.key = d_As.data(); a.keyp = d_As.data() + (size_t)k * l * N;
        if (launch_yaw(a, cap, (flags & 4) ? 1u : 512u)) return 4;
        a.keyp = nullptr;
        const uint32_t two_max = one_lane ? 0u : cap / 2 + 1;                      // (the device's threshold is 36 attempts per SIMD; here: both forms get their turn)
        const unsigned hb2 = (unsigned)(((size_t)(cap < two_max ? cap : two_max) * 2 + CH_T - 1) / CH_T);
        if (hb2) launch(hb2, CH_T, [&]() { dil_challenge_kernel<true, true>(rs.data(), slot_item.data(), 0u, mu.data(), w1.data(), k * P.n, P.weight, c8.data(), nullptr, nullptr, two_max); });
        launch(hb, CH_T, [&]() { dil_challenge_kernel<true, false>(rs.data(), slot_item.data(), 0u, mu.data(), w1.data(), k * P.n, P.weight, c8.data(), nullptr, nullptr, two_max); });
        const unsigned fb = (left + 3) / 4;
        if (sparse_fin) {
            const DilFinS g{ tabs, bias, nullptr, nullptr, nullptr, nullptr };
            if (k == 6) launch(fb, 256, [&]() { dil_cfins_kernel<6, 5>(a, f, g); });
            else if (k == 5) launch(fb, 256, [&]() { dil_cfins_kernel<5, 4>(a, f, g); });
            else if (k == 4) launch(fb, 256, [&]() { dil_cfins_kernel<4, 3>(a, f, g); });
            else launch(fb, 256, [&]() { dil_cfins_kernel<3, 2>(a, f, g); });
        } else if (a.sg) {
            if (l > 4 && k > 4) launch(fb, 256, [&]() { dil_cfin_kernel<MODE_S32, true, true>(a, f); });
            else if (l > 4) launch(fb, 256, [&]() { dil_cfin_kernel<MODE_S32, true, false>(a, f); });
            else if (k > 4) launch(fb, 256, [&]() { dil_cfin_kernel<MODE_S32, false, true>(a, f); });
            else launch(fb, 256, [&]() { dil_cfin_kernel<MODE_S32, false, false>(a, f); });
        } else {
            if (l > 4 && k > 4) launch(fb, 256, [&]() { dil_cfin_kernel<MODE_U32, true, true>(a, f); });
            else if (l > 4) launch(fb, 256, [&]() { dil_cfin_kernel<MODE_U32, true, false>(a, f); });
            else if (k > 4) launch(fb, 256, [&]() { dil_cfin_kernel<MODE_U32, false, true>(a, f); });
            else launch(fb, 256, [&]() { dil_cfin_kernel<MODE_U32, false, false>(a, f); });
        }
        round++;
        // (sign_core queues the rounds its schedule predicts before it looks; here every round is followed by the look)
        launch(cap < 64u ? cap : 64u, 256, [&]() { dil_wipe_kernel(rs.data(), y.data(), (uint32_t)(l * P.n), w.data(), w_words, nullptr); });
        left = cnt[0];
    }
    *rounds = round;
    // what sc_hip_dilithium_workspace_residue reports: the per-call secrets y and w = A y are zero behind the call
    uint32_t res = 0;
    for (int32_t v : y) res |= (uint32_t)v;
    for (int32_t v : w) res |= (uint32_t)v;
    *residue = res;
    return 0;
}

// A verification call, one key for the batch: sc_hip_dilithium_set_pubkey (t1 2^d through the transform kernel, the matrix, tr = CRH(rho, t1)) and
// sc_hip_dilithium_verify's default path - the three readers of the inputs as one launch, NTT(c), the fused NTT(z) -> A NTT(z) - NTT(c) NTT(t1 2^d) -> INTT ->
// UseHint kernel, the challenge hash compared with the signature's c.  flags bit 1: one lane per challenge hash.  verdicts[i] = 0: signature i verifies.
extern "C" int emu_dil_verify(const int32_t *par, const uint32_t *tab, const uint32_t *modk, const uint32_t *cinv, uint32_t r2, uint32_t r2p, int sg,
                              const int32_t *t1, const uint8_t *rho, const int32_t *Ahat,
                              uint32_t batch, const uint8_t *msgs, size_t msg_stride, const uint32_t *mlens, uint32_t fixed_len,
                              const uint8_t *sigs, size_t sig_stride, const uint32_t *siglens, int flags, uint32_t *verdicts)
{
    DilP P;
    memcpy(&P, par, sizeof P);
    const int k = P.k, l = P.l;
    if (P.n != N || l < 2 || l > 5 || l * P.n / 8 > 256) return 1;
    if ((size_t)(l * P.n * P.z_bits) / 8 > sig_stride) return 2;
    Plan pl{ tab, ModK{ modk[0], modk[1], modk[2], modk[3], modk[4], modk[5] }, { cinv[0], cinv[1], cinv[2], cinv[3] }, r2, r2p, sg };
    const uint32_t B = batch;
    // ---- sc_hip_dilithium_set_pubkey
    std::vector<int32_t> t1s((size_t)k * N);
    for (size_t i = 0; i < t1s.size(); i++) t1s[i] = t1[i] << P.d;
    std::vector<uint32_t> d_t1s((size_t)k * N), d_As((size_t)2 * k * l * N);
    std::vector<uint8_t> tr_pub(48);
    {
        DilNtt a = base_args(pl, P);
        a.in = t1s.data(); a.out_int = d_t1s.data(); launch_dil<DOP_KEYPREP>(a, (uint32_t)k, 1);
        const size_t np = (size_t)k * l;
        launch((unsigned)((np * N + 255) / 256), 256, [&]() { dil_to_internal_shoup(Ahat, d_As.data(), d_As.data() + np * N, np, pl.k); });
        launch(1, HT, [&]() { dil_crh_t1_kernel(rho, t1, (uint32_t)t1s.size(), P.q_bits - P.d, tr_pub.data(), 1u); });
    }
    // ---- sc_hip_dilithium_verify
    std::vector<uint8_t> mu((size_t)B * 48), h((size_t)B * k * N), w1((size_t)B * k * N);
    std::vector<int32_t> c((size_t)B * N), z((size_t)B * l * N);
    std::vector<uint32_t> ch((size_t)B * N), yh((size_t)B * l * N);
    for (uint32_t i = 0; i < B; i++) verdicts[i] = 0;
    const unsigned hb = (B + HT - 1) / HT;
    if (P.z_bits == 20) launch((B + 3) / 4 + hb + B, 256, [&]() { dil_tail_mu_kernel<20>(sigs, sig_stride, siglens, B, h.data(), c.data(), verdicts, P, tr_pub.data(), msgs, msg_stride, mlens, fixed_len, mu.data(), z.data()); });
    else launch((B + 3) / 4 + hb + B, 256, [&]() { dil_tail_mu_kernel<0>(sigs, sig_stride, siglens, B, h.data(), c.data(), verdicts, P, tr_pub.data(), msgs, msg_stride, mlens, fixed_len, mu.data(), z.data()); });
    DilNtt a = base_args(pl, P);
    a.yh = yh.data(); a.ch = ch.data(); a.w1 = w1.data(); a.z = z.data(); a.h = h.data();
    if (a.sg && !(16ull * (2ull * (uint64_t)l + 1) * (uint64_t)P.q < (1ull << 31))) return 3;      // (the fused form's bound; every reference parameter set meets it)
    a.in = c.data(); a.out_int = ch.data();
    launch_dil<DOP_FWD_INT>(a, B, 1);
    a.in = z.data(); a.out_int = yh.data();
    a.key = d_As.data(); a.keyp = d_As.data() + (size_t)k * l * N; a.t1s = d_t1s.data();
    if (launch_yaw(a, B, 512u, true)) return 4;
    if (!(flags & 2)) launch((2 * B + CH_T - 1) / CH_T, CH_T, [&]() { dil_challenge_kernel<false, true>(nullptr, nullptr, B, mu.data(), w1.data(), k * P.n, P.weight, nullptr, c.data(), verdicts, 0u); });
    else launch((B + CH_T - 1) / CH_T, CH_T, [&]() { dil_challenge_kernel<false, false>(nullptr, nullptr, B, mu.data(), w1.data(), k * P.n, P.weight, nullptr, c.data(), verdicts, 0u); });
    return 0;
}
