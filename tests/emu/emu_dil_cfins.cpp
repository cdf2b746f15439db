// emu_dil_cfins.cpp - Dilithium's finishing kernel without transforms (libsafecrypto_amd/csrc/dil_cfins.h), compiled as host code over wave_emu.h and
// run for one round of the signer: the kernel's own text, the product's own table builder (sc_hip_dilithium_fin_tables_host), inputs from the oracle's
// attempts.  Test infrastructure (tests/test_emu_dil_cfins.py); nothing here is part of the product.
#include "wave_emu.h"

// what dilithium.hip has in scope where it includes the kernel (csrc/dilithium.hip:38-40,147,472-482; csrc/mlwe_kernels.h:14,54-72; csrc/modarith.h:40)
constexpr int N = 256;
enum { RS_NACT = 0, RS_S = 1, RS_NSLOTS = 2, RS_PREV = 3 };
struct DilP { int32_t n, q, q_bits, k, l, d, weight, gamma1, gamma1_bits, gamma2, eta, z_bits, beta, omega, omega_bits, set; };
struct DilNtt { const int32_t *y; int32_t *w; int32_t alpha, am, gamma1_beta, gamma2_beta; };
struct DilFin {
    const int8_t *c8;
    const uint32_t *active, *rs;
    uint32_t *kappa;
    uint8_t *sigs; size_t sig_stride; uint32_t *siglens, *trials;
    uint32_t *next_active, *next_count;
    DilP P;
};
static inline uint32_t csub(uint32_t x, uint32_t m) { return std::min(x, x - m); }
#include "mlwe_round.h"

alignas(16) uint32_t fin_lds[40960];                  // the workgroup's LDS (160 KB)

#include "dil_cfins.h"

// One launch of the kernel as sign_core makes it (csrc/dilithium.hip).  w_nat [nact * S][k][n] canonical in NATURAL order: stored here in the 16-byte-piece
// layout the fused kernel writes (word 64 c + 4 lp + e of a row holds coefficient 64 c + 16 e + lp; DilNtt::w_chunked).  Returns 0, or 1 when the
// parameter set has no instance.
extern "C" int emu_dil_cfins_round(const int32_t *par, const uint32_t *tabs, uint32_t bias, uint32_t nact, uint32_t S, const uint32_t *active,
                                   const int32_t *y, const int32_t *w_nat, const int8_t *c8, uint32_t *kappa,
                                   uint8_t *sigs, size_t sig_stride, uint32_t *siglens, uint32_t *trials, uint32_t *next_active, uint32_t *next_count)
{
    DilP P;
    memcpy(&P, par, sizeof P);
    const int k = P.k, l = P.l;
    std::vector<int32_t> w((size_t)nact * S * k * N);
    for (size_t row = 0; row < (size_t)nact * S * k; row++)
        for (int c = 0; c < 4; c++)
            for (int lp = 0; lp < 16; lp++)
                for (int e = 0; e < 4; e++) w[row * N + 64 * c + 4 * lp + e] = w_nat[row * N + 64 * c + 16 * e + lp];
    uint32_t rs[4] = { nact, S, nact * S, 0 };
    DilNtt a{};
    a.y = y; a.w = w.data();
    a.alpha = 2 * P.gamma2; a.am = (1 << 30) / a.alpha;                 // mlwe_base_args
    a.gamma1_beta = P.gamma1 - P.beta; a.gamma2_beta = P.gamma2 - P.beta;
    DilFin f{};
    f.c8 = c8; f.active = active; f.rs = rs; f.kappa = kappa; f.sigs = sigs; f.sig_stride = sig_stride; f.siglens = siglens; f.trials = trials;
    f.next_active = next_active; f.next_count = next_count; f.P = P;
    const DilFinS g{ tabs, bias };
    if ((4 + cfins_table_words(P) + 4 * cfins_wave_words(P)) > sizeof fin_lds / 4) return 1;
    const unsigned fb = (nact + 3) / 4;
    if (k == 6 && l == 5) emu::launch(fb, 256, [&]() { dil_cfins_kernel<6, 5>(a, f, g); });
    else if (k == 5 && l == 4) emu::launch(fb, 256, [&]() { dil_cfins_kernel<5, 4>(a, f, g); });
    else if (k == 4 && l == 3) emu::launch(fb, 256, [&]() { dil_cfins_kernel<4, 3>(a, f, g); });
    else if (k == 3 && l == 2) emu::launch(fb, 256, [&]() { dil_cfins_kernel<3, 2>(a, f, g); });
    else return 1;
    return 0;
}

// The variant with a key per message (template argument PERKEY): two wavefronts a workgroup, every wavefront building its message's tables from the raw
// key rows.  s1 [B][l][n], s2 [B][k][n] as the key pairs hold them, t0 [B][k][n] canonical (what dil_t0_kernel leaves); *refused as the kernel leaves it.
extern "C" int emu_dil_cfins_keys_round(const int32_t *par, uint32_t bias, uint32_t nact, uint32_t S, const uint32_t *active,
                                        const int32_t *y, const int32_t *w_nat, const int8_t *c8, uint32_t *kappa,
                                        uint8_t *sigs, size_t sig_stride, uint32_t *siglens, uint32_t *trials, uint32_t *next_active, uint32_t *next_count,
                                        const int32_t *s1, const int32_t *s2, const int32_t *t0, uint32_t *refused)
{
    DilP P;
    memcpy(&P, par, sizeof P);
    const int k = P.k, l = P.l;
    std::vector<int32_t> w((size_t)nact * S * k * N);
    for (size_t row = 0; row < (size_t)nact * S * k; row++)
        for (int c = 0; c < 4; c++)
            for (int lp = 0; lp < 16; lp++)
                for (int e = 0; e < 4; e++) w[row * N + 64 * c + 4 * lp + e] = w_nat[row * N + 64 * c + 16 * e + lp];
    uint32_t rs[4] = { nact, S, nact * S, 0 };
    DilNtt a{};
    a.y = y; a.w = w.data();
    a.alpha = 2 * P.gamma2; a.am = (1 << 30) / a.alpha;
    a.gamma1_beta = P.gamma1 - P.beta; a.gamma2_beta = P.gamma2 - P.beta;
    DilFin f{};
    f.c8 = c8; f.active = active; f.rs = rs; f.kappa = kappa; f.sigs = sigs; f.sig_stride = sig_stride; f.siglens = siglens; f.trials = trials;
    f.next_active = next_active; f.next_count = next_count; f.P = P;
    const DilFinS g{ nullptr, bias, s1, s2, t0, refused };
    if ((4 + 2 * (cfins_table_words(P) + cfins_wave_words(P))) > sizeof fin_lds / 4) return 1;
    const unsigned fb = (nact + 1) / 2;
    if (k == 6 && l == 5) emu::launch(fb, 128, [&]() { dil_cfins_kernel<6, 5, true>(a, f, g); });
    else if (k == 5 && l == 4) emu::launch(fb, 128, [&]() { dil_cfins_kernel<5, 4, true>(a, f, g); });
    else if (k == 4 && l == 3) emu::launch(fb, 128, [&]() { dil_cfins_kernel<4, 3, true>(a, f, g); });
    else if (k == 3 && l == 2) emu::launch(fb, 128, [&]() { dil_cfins_kernel<3, 2, true>(a, f, g); });
    else return 1;
    return 0;
}
