// emu_tesla_sign.cpp - a ring-TESLA signing call of the engine run on the CPU lane by lane: the kernels' own text (libsafecrypto_amd/csrc/tesla_sign_dev.h,
// keccak_dev.h, and the NTT kernel's two-product form of ntt_kernel.h) compiled as host code over wave_emu.h, queued per round as ts_sign_core
// (csrc/tesla.hip) queues a planned round: the round state, the masks from the tapes, v1 = a1 y and v2 = a2 y from ONE forward transform (OP_MULNTT2),
// poly_rounding, the oracle (SHA3-512 on the reference's permutation, resuming from the all-zero prefix state), f_function (the byte-swapped Salsa20
// stream), the deciding kernel (sparse products through LDS shift tables, test_w / test_rejection), the selection and packing, and the closing wipe.
// Here every round is followed by the look at the count (ts_sign_core queues the rounds its schedule predicts first); the round state stays on the "device".
// Test infrastructure (tests/test_emu_tesla_sign.py); nothing here is part of the product.
#define EMU_SHARED static
#define EMU_HAVE_NTT_CORE
#include "wave_emu.h"
#include "keccak_dev.h"
#include "drbg_dev.h"
#include "ntt_kernel.h"

namespace {
#include "tesla_sign_dev.h"
#include "tesla_verify_dev.h"
}

namespace {
template <typename F> void launch(unsigned grid, unsigned block, F &&f) { if (grid) emu::launch(grid, block, std::function<void()>(f)); }
}

// par: the 12 words of TsP; tab / modk / c / lazy: the plan's table, ModK, scale constants and lazy flag for `mode` and OP_MULNTT (tests/test_emu_ntt.py PlanConsts)
extern "C" int emu_tesla_sign(const int32_t *par, int mode, int lazy, const uint32_t *tab, const uint32_t *modk, const uint32_t *c,
                              const int32_t *a1, const int32_t *a2, const int16_t *ks, const int16_t *ke1, const int16_t *ke2,
                              uint32_t batch, const uint8_t *msgs, size_t msg_stride, const uint32_t *mlens,
                              const uint32_t *tapes, size_t tape_stride, const uint32_t *tape_lens, uint32_t cap,
                              uint8_t *sigs, size_t sig_stride, uint32_t *siglens, uint32_t *trials, uint32_t *status, uint32_t *rounds, uint32_t *residue)
{
    TsP P;
    memcpy(&P, par, sizeof P);
    const int n = P.n;
    if (n != 512 || (mode != MODE_SMAD && mode != MODE_S32 && mode != MODE_U32)) return 1;
    const uint32_t B = batch;
    // ---- sc_hip_tesla_create: the sponge behind the all-zero blocks; sc_hip_tesla_set_key: the shift tables
    const uint32_t zblocks = (uint32_t)(8 * n / 72);
    std::vector<uint64_t> zstate(25);
    launch(1, HT, [&]() { ts_zero_prefix_kernel(zblocks, zstate.data()); });
    std::vector<int16_t> key(6 * (size_t)n);
    const int16_t *src[3] = { ks, ke1, ke2 };
    for (int p = 0; p < 3; p++)
        for (int m = 0; m < n; m++) { key[2 * n * p + m] = src[p][m]; key[2 * n * p + n + m] = (int16_t)-src[p][m]; }
    // ---- ts_ensure_ws
    std::vector<uint32_t> pos(B, 0u), stat(B, 0u), act0(B), act1(B), cnt(16, 0u), rs(16, 0u), need(cap), sbase(cap), nz(cap), acc(cap);
    std::vector<int32_t> y((size_t)cap * n), v1((size_t)cap * n), v2((size_t)cap * n), r1((size_t)cap * n), r2((size_t)cap * n), zs((size_t)cap * n), cidx((size_t)cap * P.omega);
    std::vector<uint8_t> md((size_t)cap * 64);
    St st{ pos.data(), stat.data() };
    for (uint32_t i = 0; i < B; i++) trials[i] = 0;
    static int32_t sink[1024];
    uint32_t round = 0, left = B;
    while (left > 0) {
        if (round > 300) return 3;
        const uint32_t *active = round ? (((round - 1) & 1) ? act1.data() : act0.data()) : nullptr;
        uint32_t *next = (round & 1) ? act1.data() : act0.data();
        const uint32_t nact = B, S = 1, nslots = cap;                       // a planned round: the host's upper bounds, the sizes come from rs
        launch(1, 64, [&]() { if (threadIdx.x == 0) ts_rs_kernel(rs.data(), cnt.data(), round ? 0u : B, cap); });
        const size_t ne = (size_t)nslots * n;
        launch((unsigned)((ne / 4 + 255) / 256), 256, [&]() { ts_y_kernel(active, nslots, S, need.data(), sbase.data(), tapes, tape_stride, y.data(), P, rs.data(), st, tape_lens); });
        {   // sc::polymul_ntt_bcast2: both products from one forward transform of y, over the whole slot budget
            NttArgs a{};
            a.in0 = y.data(); a.in1 = a1; a.in2 = a2; a.out = v1.data(); a.out2 = v2.data(); a.sink = sink; a.in1_bcast = 1u;
            a.batch = nslots;
            a.zf = tab; a.zfp = tab + n; a.zi = tab + 2 * n; a.zip = tab + 3 * n;
            a.k = ModK{ modk[0], modk[1], modk[2], modk[3], modk[4], modk[5] };
            a.c0 = c[0]; a.c0p = c[1]; a.c1 = c[2]; a.c1p = c[3];
            const int pw = 1024 / n;
            const unsigned blocks = (unsigned)(((nslots + pw - 1) / pw + 3) / 4);
            if (mode == MODE_SMAD) launch(blocks, 256, [&]() { ntt_kernel<9, MODE_SMAD, true, OP_MULNTT2, false>(a); });
            else if (mode == MODE_S32) launch(blocks, 256, [&]() { ntt_kernel<9, MODE_S32, true, OP_MULNTT2, false>(a); });
            else {                                                              // a modulus outside the signed modes (ring-TESLA-II): the two single products
                for (int which = 0; which < 2; which++) {
                    NttArgs b = a;
                    if (which) { b.in1 = a2; b.out = v2.data(); }
                    if (lazy) launch(blocks, 256, [&]() { ntt_kernel<9, MODE_U32, true, OP_MULNTT, false>(b); });
                    else launch(blocks, 256, [&]() { ntt_kernel<9, MODE_U32, false, OP_MULNTT, false>(b); });
                }
            }
        }
        launch((nslots + 3) / 4, 256, [&]() { ts_round_kernel(nslots, v1.data(), v2.data(), r1.data(), r2.data(), nz.data(), 1, P, rs.data()); });
        launch((nslots + HT - 1) / HT, HT, [&]() { ts_oracle_kernel(active, nslots, S, need.data(), msgs, msg_stride, mlens, 0u, r1.data(), r2.data(), nz.data(), zstate.data(), zblocks, md.data(), 0,
                                                                    nullptr, nullptr, (size_t)0, 0u, P, rs.data()); });
        launch((nslots + HT - 1) / HT, HT, [&]() { ts_f_kernel(nullptr, nslots, need.data(), md.data(), (size_t)64, 0u, 0, cidx.data(), P, rs.data()); });
        launch((nslots + 3) / 4, 256, [&]() { ts_decide_kernel(nslots, need.data(), key.data(), v1.data(), v2.data(), y.data(), cidx.data(), zs.data(), acc.data(), P, rs.data()); });
        launch((nact + 4 * SEL_PER - 1) / (4 * SEL_PER), 256, [&]() { ts_select_kernel(active, nact, S, need.data(), acc.data(), st, zs.data(), md.data(), sigs, sig_stride, siglens, trials, next, cnt.data(), P, rs.data()); });
        round++;
        left = cnt[0];
    }
    {   // the closing wipe (the six per-attempt arrays as far as the slot budget goes)
        const size_t used = (size_t)cap * n * 4;
        ZeroSix z6{ { y.data(), v1.data(), v2.data(), r1.data(), r2.data(), zs.data() }, used / 16 };
        launch((unsigned)((used / 16 + 255) / 256), 256, [&]() { ts_zero6_kernel(z6); });
    }
    uint32_t res = 0;
    for (auto *arr : { &y, &v1, &v2, &r1, &r2, &zs }) for (int32_t v : *arr) res |= (uint32_t)v;
    *residue = res;
    for (uint32_t i = 0; i < B; i++) status[i] = stat[i];
    *rounds = round;
    return 0;
}

// A verification call: sc_hip_tesla_verify's path - z's unpacking with test_rejection and F from the signature's digest as one launch, a1 z and a2 z from one
// forward transform, w = a z - t c through the public key's shift tables, rounding, the oracle compared with the signature's digest.  verdicts[i] = 0: valid.
extern "C" int emu_tesla_verify(const int32_t *par, int mode, int lazy, const uint32_t *tab, const uint32_t *modk, const uint32_t *c,
                                const int32_t *a1, const int32_t *a2, const int32_t *t1, const int32_t *t2,
                                uint32_t batch, const uint8_t *msgs, size_t msg_stride, const uint32_t *mlens,
                                const uint8_t *sigs, size_t sig_stride, const uint32_t *siglens, uint32_t *verdicts)
{
    TsP P;
    memcpy(&P, par, sizeof P);
    const int n = P.n;
    if (n != 512 || (mode != MODE_SMAD && mode != MODE_S32 && mode != MODE_U32)) return 1;
    const uint32_t B = batch;
    const uint32_t zblocks = (uint32_t)(8 * n / 72);
    std::vector<uint64_t> zstate(25);
    launch(1, HT, [&]() { ts_zero_prefix_kernel(zblocks, zstate.data()); });
    std::vector<int32_t> ttab(4 * (size_t)n);                               // sc_hip_tesla_set_key: t1 | -t1 | t2 | -t2
    for (int m = 0; m < n; m++) { ttab[m] = t1[m]; ttab[n + m] = -t1[m]; ttab[2 * n + m] = t2[m]; ttab[3 * n + m] = -t2[m]; }
    std::vector<int32_t> y((size_t)B * n), v1((size_t)B * n), v2((size_t)B * n), r1((size_t)B * n), r2((size_t)B * n), cidx((size_t)B * P.omega);
    std::vector<uint32_t> nz(B);
    for (uint32_t i = 0; i < B; i++) verdicts[i] = 0;
    static int32_t sink[1024];
    const uint32_t md_off = ((uint32_t)P.n * P.q_bits + 7) / 8;
    if (sig_stride < md_off + 64) return 2;
    launch((B + HT - 1) / HT + (B + 3) / 4, 256, [&]() { ts_unpack_f_kernel(B, sigs, sig_stride, siglens, y.data(), verdicts, P, md_off, cidx.data()); });
    const ModK mk{ modk[0], modk[1], modk[2], modk[3], modk[4], modk[5] };
    {
        NttArgs a{};
        a.in0 = y.data(); a.in1 = a1; a.in2 = a2; a.out = v1.data(); a.out2 = v2.data(); a.sink = sink; a.in1_bcast = 1u; a.batch = B;
        a.zf = tab; a.zfp = tab + n; a.zi = tab + 2 * n; a.zip = tab + 3 * n; a.k = mk;
        a.c0 = c[0]; a.c0p = c[1]; a.c1 = c[2]; a.c1p = c[3];
        const int pw = 1024 / n;
        const unsigned blocks = (unsigned)(((B + pw - 1) / pw + 3) / 4);
        if (mode == MODE_SMAD) launch(blocks, 256, [&]() { ntt_kernel<9, MODE_SMAD, true, OP_MULNTT2, false>(a); });
        else if (mode == MODE_S32) launch(blocks, 256, [&]() { ntt_kernel<9, MODE_S32, true, OP_MULNTT2, false>(a); });
        else {
            for (int which = 0; which < 2; which++) {
                NttArgs b = a;
                if (which) { b.in1 = a2; b.out = v2.data(); }
                if (lazy) launch(blocks, 256, [&]() { ntt_kernel<9, MODE_U32, true, OP_MULNTT, false>(b); });
                else launch(blocks, 256, [&]() { ntt_kernel<9, MODE_U32, false, OP_MULNTT, false>(b); });
            }
        }
    }
    if (sizeof(int32_t) * 4 * (size_t)n > sizeof emu::dyn_lds) return 3;
    launch((B + 3) / 4, 256, [&]() { ts_vw_kernel(B, v1.data(), v2.data(), ttab.data(), cidx.data(), P, mk); });
    launch((B + 3) / 4, 256, [&]() { ts_round_kernel(B, v1.data(), v2.data(), r1.data(), r2.data(), nz.data(), 0, P, nullptr); });
    launch((B + HT - 1) / HT, HT, [&]() { ts_oracle_kernel(nullptr, B, 1u, nullptr, msgs, msg_stride, mlens, 0u, r1.data(), r2.data(), nz.data(), zstate.data(), zblocks, nullptr, 1, verdicts,
                                                           sigs, sig_stride, md_off, P, nullptr); });
    return 0;
}
