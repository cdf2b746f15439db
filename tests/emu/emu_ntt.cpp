// emu_ntt.cpp - the batched NTT kernel (libsafecrypto_amd/csrc/ntt_kernel.h over ntt_core.h / modarith.h: the headline kernel of bench.py and the transform
// inside every scheme) compiled as host code over wave_emu.h and launched as sc_hip.hip's launch_ntt_l would launch it: the kernel's own text, every
// instantiation the host can select.  Test infrastructure (tests/test_emu_ntt.py); nothing here is part of the product.
#define EMU_SHARED static
#define EMU_HAVE_NTT_CORE
#include "wave_emu.h"
#include "ntt_kernel.h"

template <int L, int MODE, bool LAZY, int OP, bool STREAM = false>
static int run(const NttArgs &a, unsigned blocks) { emu::launch(blocks, 256, [&]() { ntt_kernel<L, MODE, LAZY, OP, STREAM>(a); }); return 0; }

// the selection of csrc/sc_hip.hip launch_ntt_l, instantiation for instantiation
template <int L, int OP>
static int run_l(const NttArgs &a, int mode, bool lazy, unsigned blocks, bool stream_io)
{
    if constexpr (OP == OP_MULNTT2) {
        if (mode == MODE_SMAD) return run<L, MODE_SMAD, true, OP>(a, blocks);
        if (mode == MODE_S32) return run<L, MODE_S32, true, OP>(a, blocks);
        return 2;
    } else {
        if (mode == MODE_SMAD) return stream_io ? run<L, MODE_SMAD, true, OP, true>(a, blocks) : run<L, MODE_SMAD, true, OP>(a, blocks);
        if (mode == MODE_S32) return stream_io ? run<L, MODE_S32, true, OP, true>(a, blocks) : run<L, MODE_S32, true, OP>(a, blocks);
        if (stream_io) return 2;
        if constexpr (OP == OP_INV) {
            if (mode == MODE_MAD) return run<L, MODE_MAD, false, OP>(a, blocks);
            return mode == MODE_U24 ? run<L, MODE_U24, false, OP>(a, blocks) : run<L, MODE_U32, false, OP>(a, blocks);
        } else {
            if (mode == MODE_MAD) return run<L, MODE_MAD, true, OP>(a, blocks);
            if (mode == MODE_U24) return lazy ? run<L, MODE_U24, true, OP>(a, blocks) : run<L, MODE_U24, false, OP>(a, blocks);
            return lazy ? run<L, MODE_U32, true, OP>(a, blocks) : run<L, MODE_U32, false, OP>(a, blocks);
        }
    }
}
template <int OP>
static int run_op(int L, const NttArgs &a, int mode, bool lazy, unsigned blocks, bool stream_io)
{
    switch (L) {
    case 8: return run_l<8, OP>(a, mode, lazy, blocks, stream_io);
    case 9: return run_l<9, OP>(a, mode, lazy, blocks, stream_io);
    default: return run_l<10, OP>(a, mode, lazy, blocks, stream_io);
    }
}

extern "C" void emu_set_wave_serial(int on) { emu::wave_serial = on != 0; }      // the kernel has no workgroup barrier: its wavefronts may run one after the other

// tab = [zf | zfp | zi | zip], n words each; modk = q, q2, qinv, m32, one_p, half; c = c0, c0p, c1, c1p.  Returns 0, 2 for a combination the host never selects.
extern "C" int emu_ntt(int L, int op, int mode, int lazy, int stream_io, unsigned blocks, const uint32_t *tab, const uint32_t *modk, const uint32_t *c,
                       unsigned long long batch, const uint32_t *batch_dev, int bcast, const int32_t *in0, const int32_t *in1, const int32_t *in2, int32_t *out, int32_t *out2)
{
    static int32_t sink[1024];
    const int n = 1 << L;
    NttArgs a{};
    a.in0 = in0; a.in1 = in1; a.in2 = in2; a.out = out; a.out2 = out2; a.sink = sink;
    a.zf = tab; a.zfp = tab + n; a.zi = tab + 2 * n; a.zip = tab + 3 * n;
    a.batch = batch; a.batch_dev = batch_dev; a.in1_bcast = bcast ? 1u : 0u;
    a.k.q = modk[0]; a.k.q2 = modk[1]; a.k.qinv = modk[2]; a.k.m32 = modk[3]; a.k.one_p = modk[4]; a.k.half = modk[5];
    a.c0 = c[0]; a.c0p = c[1]; a.c1 = c[2]; a.c1p = c[3];
    switch (op) {
    case OP_FWD: return run_op<OP_FWD>(L, a, mode, lazy != 0, blocks, stream_io != 0);
    case OP_INV: return run_op<OP_INV>(L, a, mode, lazy != 0, blocks, stream_io != 0);
    case OP_MULNTT: return run_op<OP_MULNTT>(L, a, mode, lazy != 0, blocks, stream_io != 0);
    case OP_MUL: return run_op<OP_MUL>(L, a, mode, lazy != 0, blocks, stream_io != 0);
    case OP_MULNTT2: return run_op<OP_MULNTT2>(L, a, mode, lazy != 0, blocks, stream_io != 0);
    }
    return 2;
}
