// bliss_sign_dev.h - the device side of the batched BLISS-B signer: every kernel of a signing round (csrc/bliss.hip bl_round_launch queues them).
// Replaces bliss_b_sign (src/schemes/sig/bliss_b/bliss_b.c:1266-1535) an attempt per item: get_vector_32 x2, round_and_drop, oracle, greedy_sc, the
// exp/cosh rejection, signature_gen, check_norms and the bit packing.  Included by bliss.hip INSIDE its anonymous namespace, behind keccak_dev.h and
// cdt_guide.h (hence no includes here); a header so that the kernels' own text can also be compiled as host code over tests/emu/wave_emu.h and a signing
// call be run lane by lane under the sanitizers the GPU pool cannot offer (tests/test_emu_bliss_sign.py).

constexpr int HT = 64;

struct BlissP {              // bliss_set_t (bliss_params.h:43-75), the fields used here
    int32_t n, q, q_bits, n_bits, d, p, kappa, b_inf, b_l2, z1_bits, z2_bits, set;
    float sigma, m, tail;
    int32_t pad;
};

// bliss_b.c:58-64
__device__ __forceinline__ int32_t barrett(int64_t x, int64_t m, int64_t q) { int64_t t = (x * m) >> 30, c = x - t * q; if (q <= c) c -= q; return (int32_t)c; }

struct Tape {                // one message's CSPRNG word stream + the prng_var reservoir (prng.c:1017-1048)
    const uint32_t *w; uint32_t len, pos, var_buf; int var_bits; bool underrun;
    // An attempt's decision draws 1 + 27 + 26 reservoir bits: at most three words, the three at the position the attempt starts from.  They are
    // requested when the kernel starts (prefetch) instead of one dependent round trip each in the middle of a chain of dependent steps.
    uint32_t pre[3], pre_at; bool have_pre;
    __device__ __forceinline__ void prefetch()
    {
        pre_at = pos; have_pre = true;
#pragma unroll
        for (int i = 0; i < 3; i++) { const uint32_t at = pos + i < len ? pos + i : (len ? len - 1 : 0); pre[i] = len ? w[at] : 0u; }
    }
    __device__ __forceinline__ uint32_t w32()
    {
        if (pos >= len) { underrun = true; return 0; }
        const uint32_t d = pos - pre_at;
        pos++;
        if (have_pre && d < 3) return d == 0 ? pre[0] : d == 1 ? pre[1] : pre[2];
        return w[pos - 1];
    }
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
    __device__ __forceinline__ double dbl() { const uint32_t a = var(27), b = var(26); return (a * 67108864.0 + b) * 1.11022302462516e-16; }
};

struct St {                  // per-item state arrays (device).  An item is a message - or, in the speculative phase of a signing call, one later PAIR of
    uint32_t *pos, *var_buf, *iter, *cur, *status;      // attempts of a message (bl_sign_core): msg_of then maps the item to the message whose inputs
    int32_t *var_bits;                                  // (text, tape, key) it reads; null = the item is the message
    const uint32_t *msg_of;
};

// start of an attempt (bliss_b.c:1366-1373): odd attempts swap t and u, even ones draw 2n fresh samples
__global__ void bl_begin_kernel(const uint32_t *active, uint32_t nact, St s, const uint32_t *tape_lens, uint32_t tape_stride, int n,
                                uint32_t *need, uint32_t *sbase, uint32_t *siglens, const uint32_t *mlens, const uint32_t *nd)
{
    // nd: the round's item count where the device knows it and the host does not (the round behind one whose outcome was not read back): nact is
    // then the host's upper bound, the grid is sized for it and the surplus leaves at once
    if (nd) nact = min(nact, *nd);
    const uint32_t aidx = blockIdx.x * blockDim.x + threadIdx.x;
    if (aidx >= nact) return;
    const uint32_t item = active ? active[aidx] : aidx, src = s.msg_of ? s.msg_of[item] : item;
    if (mlens && mlens[src] == 0) { s.status[item] = 3; siglens[item] = 0; need[aidx] = 2; return; }        // oracle() refuses an empty message
    if (s.iter[item] & 1) { s.cur[item] ^= 1; need[aidx] = 0; return; }
    const uint32_t len = tape_lens ? tape_lens[src] : tape_stride, pos = s.pos[item];
    if (pos + 4u * (uint32_t)n > len) { s.status[item] = 2; siglens[item] = 0; need[aidx] = 2; return; }     // tape exhausted
    need[aidx] = 1; sbase[aidx] = pos; s.pos[item] = pos + 4u * (uint32_t)n;
}

// get_vector_32(t), get_vector_32(u): sample s of the attempt uses tape words 2s (high) and 2s+1 (low).  A wavefront takes 256 consecutive samples of
// one item per step - a lane four of them: 32 bytes of tape in, one 16-byte store out - so everything about the item (does it draw, where on its tape,
// which of its two vectors is current) is wave-uniform and read once per 2 KB of tape rather than once per lane and pair of samples; the table search is
// the guide + S probes of cdt_guide.h (S = 0: the halving search).
template <int S>
__global__ __launch_bounds__(256) void bl_sample_kernel(const uint32_t *active, uint32_t nact, const uint32_t *need, const uint32_t *sbase, const uint32_t *cur,
                                                        const uint32_t *tapes, size_t tape_stride, const uint64_t *cdf, const uint16_t *gd, CdtGuide g, int size, int32_t *tu, int n,
                                                        const uint32_t *msg_of, const uint32_t *nd)
{
    extern __shared__ __attribute__((aligned(16))) uint64_t tab[];
    if (nd) nact = min(nact, *nd);
    const int lps = 31 - __clz(2 * n / 256);                          // log2 of the steps per item (n = 256: 2 steps, n = 512: 4)
    const size_t steps = (size_t)nact << lps, nwaves = (size_t)gridDim.x * 4;
    if ((size_t)blockIdx.x * 4 >= steps) return;
    uint16_t *guide = reinterpret_cast<uint16_t *>(tab + size + CDT_PAD);
    for (int i = threadIdx.x; i < size + CDT_PAD; i += blockDim.x) tab[i] = cdf[i];
    if (S) for (uint32_t i = threadIdx.x; i < (g.total + 1) / 2; i += blockDim.x) reinterpret_cast<uint32_t *>(guide)[i] = reinterpret_cast<const uint32_t *>(gd)[i];
    __syncthreads();
    const int lane = threadIdx.x & 63;
    for (size_t ws = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6); ws < steps; ws += nwaves) {
        const uint32_t aidx = (uint32_t)__builtin_amdgcn_readfirstlane((int)(ws >> lps));
        if (need[aidx] != 1) continue;
        const uint32_t item = active ? active[aidx] : aidx;
        const int sidx = (((int)ws & ((1 << lps) - 1)) << 8) + 4 * lane;              // first of the lane's four samples, 0 .. 2n-1
        const uint32_t *tp = tapes + (size_t)(msg_of ? msg_of[item] : item) * tape_stride + sbase[aidx] + 2 * sidx;
        uint32_t w[8];
        if ((reinterpret_cast<uintptr_t>(tp) & 15) == 0) {
            const uint4 q0 = reinterpret_cast<const uint4 *>(tp)[0], q1 = reinterpret_cast<const uint4 *>(tp)[1];
            w[0] = q0.x; w[1] = q0.y; w[2] = q0.z; w[3] = q0.w; w[4] = q1.x; w[5] = q1.y; w[6] = q1.z; w[7] = q1.w;
        } else {
#pragma unroll
            for (int e = 0; e < 8; e++) w[e] = tp[e];
        }
        int32_t v[4];
#pragma unroll
        for (int e = 0; e < 4; e++) {
            const uint64_t u = ((uint64_t)w[2 * e] << 32) | w[2 * e + 1];
            uint32_t a = 0;
            if (S) a = cdt_lookup<(S > 3 ? 0 : S)>(u, tab, guide, g);
            else for (uint32_t st = (uint32_t)size >> 1; st > 0; st >>= 1) { const uint32_t b = a + st; if (tab[b] < u) a = b; }
            v[e] = (u & 1) ? (int32_t)a : -(int32_t)a;
        }
        const uint32_t c = cur[item];
        const int which = sidx < n ? (int)c : (int)(c ^ 1);
        *reinterpret_cast<int4 *>(tu + ((size_t)item * 2 + which) * n + (sidx & (n - 1))) = make_int4(v[0], v[1], v[2], v[3]);
    }
}

// ---- the same two kernels for a sampler at 128- / 192-bit precision (SC_FLAG_0_SAMPLE_128BIT / _192BIT: gaussian_cdf_sample_128 / _192, gaussian_cdf.c:468-509) -----------
// A sample is W = 2 / 3 prng_64 draws, i.e. 2 W tape words: words 0 / 1 the first draw's high / low half (limb 0: its bit 0 is the sign), words 2 W - 2 / 2 W - 1 the last
// draw's (the most significant limb: the only one the reference's comparison reads, ">=" - sc_hip.hip cdt_hp_kernel has the derivation); the handle's table holds the top
// limbs, searched with top + 1.  A fresh attempt is 2 n samples = 4 n W words.  Separate kernels, so that the text of the 64-bit ones above stays what the GPU suite ran.
__global__ void bl_begin_hp_kernel(const uint32_t *active, uint32_t nact, St s, const uint32_t *tape_lens, uint32_t tape_stride, int n, uint32_t wpa,
                                   uint32_t *need, uint32_t *sbase, uint32_t *siglens, const uint32_t *mlens, const uint32_t *nd)
{
    if (nd) nact = min(nact, *nd);
    const uint32_t aidx = blockIdx.x * blockDim.x + threadIdx.x;
    if (aidx >= nact) return;
    const uint32_t item = active ? active[aidx] : aidx, src = s.msg_of ? s.msg_of[item] : item;
    if (mlens && mlens[src] == 0) { s.status[item] = 3; siglens[item] = 0; need[aidx] = 2; return; }
    if (s.iter[item] & 1) { s.cur[item] ^= 1; need[aidx] = 0; return; }
    const uint32_t len = tape_lens ? tape_lens[src] : tape_stride, pos = s.pos[item];
    if (pos + wpa > len) { s.status[item] = 2; siglens[item] = 0; need[aidx] = 2; return; }                 // tape exhausted
    need[aidx] = 1; sbase[aidx] = pos; s.pos[item] = pos + wpa;
}

// a lane takes one sample per step (a wavefront 64 consecutive ones of one item): 16 or 24 bytes of tape in, one word out
template <int S>
__global__ __launch_bounds__(256) void bl_sample_hp_kernel(const uint32_t *active, uint32_t nact, const uint32_t *need, const uint32_t *sbase, const uint32_t *cur,
                                                           const uint32_t *tapes, size_t tape_stride, const uint64_t *cdf, const uint16_t *gd, CdtGuide g, int size, int32_t *tu, int n,
                                                           int W, const uint32_t *msg_of, const uint32_t *nd)
{
    extern __shared__ __attribute__((aligned(16))) uint64_t tab[];
    if (nd) nact = min(nact, *nd);
    const int lps = 31 - __clz(2 * n / 64);                           // log2 of the steps per item
    const size_t steps = (size_t)nact << lps, nwaves = (size_t)gridDim.x * 4;
    if ((size_t)blockIdx.x * 4 >= steps) return;
    uint16_t *guide = reinterpret_cast<uint16_t *>(tab + size + CDT_PAD);
    for (int i = threadIdx.x; i < size + CDT_PAD; i += blockDim.x) tab[i] = cdf[i];
    if (S) for (uint32_t i = threadIdx.x; i < (g.total + 1) / 2; i += blockDim.x) reinterpret_cast<uint32_t *>(guide)[i] = reinterpret_cast<const uint32_t *>(gd)[i];
    __syncthreads();
    const int lane = threadIdx.x & 63;
    for (size_t ws = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6); ws < steps; ws += nwaves) {
        const uint32_t aidx = (uint32_t)__builtin_amdgcn_readfirstlane((int)(ws >> lps));
        if (need[aidx] != 1) continue;
        const uint32_t item = active ? active[aidx] : aidx;
        const int sidx = (((int)ws & ((1 << lps) - 1)) << 6) + lane;                   // the lane's sample, 0 .. 2n-1
        const uint32_t *tp = tapes + (size_t)(msg_of ? msg_of[item] : item) * tape_stride + sbase[aidx] + (size_t)2 * W * sidx;
        const uint32_t sign = tp[1] & 1u;
        const uint64_t top = ((uint64_t)tp[2 * W - 2] << 32) | tp[2 * W - 1], u = top + 1;
        uint32_t a = 0;
        if (S) a = cdt_lookup<(S > 3 ? 0 : S)>(u, tab, guide, g);
        else for (uint32_t st = (uint32_t)size >> 1; st > 0; st >>= 1) { const uint32_t b = a + st; if (tab[b] < u) a = b; }
        if (top == ~0ull) a = (uint32_t)size - 1u;
        const uint32_t c = cur[item];
        const int which = sidx < n ? (int)c : (int)(c ^ 1);
        tu[((size_t)item * 2 + which) * n + (sidx & (n - 1))] = sign ? (int32_t)a : -(int32_t)a;
    }
}

// dense copy of the current t for the NTT product
// (a key per message: the active messages' public keys a, SINT16 rows, are gathered beside it as the product's second operand)
__global__ void bl_stage_kernel(const uint32_t *active, uint32_t nact, const uint32_t *need, const uint32_t *cur, const int32_t *tu, int32_t *tin, int n,
                                const int16_t *a_items = nullptr, int32_t *ain = nullptr, const uint32_t *msg_of = nullptr, const uint32_t *nd = nullptr)
{
    if (nd) nact = min(nact, *nd);
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int ln = 31 - __clz(n);                                         // n is a power of two (bliss_create): no 64-bit division per element
    const uint32_t aidx = (uint32_t)(t >> ln); const int i = (int)(t & (size_t)(n - 1));
    if (aidx >= nact) return;
    const uint32_t item = active ? active[aidx] : aidx;
    tin[t] = (need[aidx] == 2) ? 0 : tu[((size_t)item * 2 + cur[item]) * n + i];
    if (a_items) ain[t] = (int32_t)a_items[(size_t)(msg_of ? msg_of[item] : item) * n + i];
}

// round_and_drop, bliss_b.c:681-703
__global__ void bl_round_kernel(const uint32_t *active, uint32_t nact, const uint32_t *cur, const int32_t *tu, int32_t *v, int16_t *z, BlissP P, const uint32_t *nd)
{
    if (nd) nact = min(nact, *nd);
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int n = P.n;
    const int ln = 31 - __clz(n);                                         // n is a power of two (bliss_create): no 64-bit division per element
    const uint32_t aidx = (uint32_t)(t >> ln); const int i = (int)(t & (size_t)(n - 1));
    if (aidx >= nact) return;
    const uint32_t item = active ? active[aidx] : aidx;
    const int32_t u = tu[((size_t)item * 2 + (cur[item] ^ 1)) * n + i];
    const int32_t two_q = 2 * P.q;
    int32_t tmp = v[t];
    tmp += (tmp & 1) * P.q;
    tmp = barrett((int64_t)tmp + u, (1 << 30) / two_q, two_q);
    tmp += (tmp < 0) * two_q;
    v[t] = tmp;
    z[t] = (int16_t)barrett((tmp + (1 << (P.d - 1))) >> P.d, (1 << 30) / P.p, P.p);
}

// oracle, bliss_b.c:552-620: SHA3-512(m || z (int16, little endian) || r_be16), r = 0,1,...; 9-bit indices from big-endian
// byte pairs; the first kappa distinct ones.  mode 0: write them; mode 1 (verify): compare with the signature's.
__global__ __launch_bounds__(HT) void bl_oracle1_kernel(const uint32_t *active, uint32_t nact, const uint32_t *need, const uint8_t *msgs, size_t msg_stride,
                                                       const uint32_t *mlens, uint32_t fixed_len, const int16_t *z, int32_t *cidx, uint32_t *ok,
                                                       int mode, uint32_t *verdict, BlissP P, const uint32_t *msg_of = nullptr, const uint32_t *nd = nullptr)
{
    __shared__ uint64_t st[25 * HT];
    __shared__ uint32_t fl[32 * HT];
    if (nd) nact = min(nact, *nd);
    const uint32_t aidx = blockIdx.x * HT + threadIdx.x;
    Sponge<HT> sp; sp.init(st, threadIdx.x, 72);
    if (aidx >= nact) return;
    if (need && need[aidx] == 2) { if (ok) ok[aidx] = 0; return; }
    const uint32_t item = active ? active[aidx] : aidx, src = msg_of ? msg_of[item] : item;
    const uint32_t mlen = mlens ? mlens[src] : fixed_len;
    if (mlen == 0) { if (mode == 0) ok[aidx] = 0; else verdict[item] = 1; return; }
    const uint8_t *m = msgs + (size_t)src * msg_stride;
    const uint8_t *zb = reinterpret_cast<const uint8_t *>(z + (size_t)aidx * P.n);
    uint32_t *myfl = fl + threadIdx.x;
    for (int w = 0; w < (P.n + 31) / 32; w++) myfl[w * HT] = 0;
    const int mask = (1 << P.n_bits) - 1;
    int cnt = 0;
    uint32_t neq = 0;
    bool done = false;
    for (int r = 0; r < 65536 && !done; r++) {
        sp.init(st, threadIdx.x, 72);
        sp.absorb(m, mlen);
        sp.absorb(zb, 2 * (size_t)P.n);
        sp.absorb_byte((uint8_t)(r >> 8)); sp.absorb_byte((uint8_t)(r & 0xFF));
        sp.finalize(0x06);
        for (int i = 0; i < 64 && !done; i += 2) {
            const uint32_t hi = sp.squeeze_byte(), lo = sp.squeeze_byte();
            const int idx = (int)((hi << 8) + lo) & mask;
            const uint32_t bit = 1u << (idx & 31), wv = myfl[(idx >> 5) * HT];
            if (!(wv & bit)) {
                if (mode == 0) cidx[(size_t)aidx * P.kappa + cnt] = idx; else neq |= (uint32_t)(cidx[(size_t)aidx * P.kappa + cnt] != idx);
                cnt++;
                if (cnt == P.kappa) done = true;
                myfl[(idx >> 5) * HT] = wv | bit;
            }
        }
    }
    if (mode == 0) ok[aidx] = done ? 1u : 0u;
    else if (!done || neq) verdict[item] = 1;
}

// The same with two lanes per message (Sponge2, keccak_dev.h): a batch of 16 K messages is only 256 wavefronts lane-per-message.
// Used while the one-lane form has no more wavefronts than the chip has SIMDs.
__global__ __launch_bounds__(HT) void bl_oracle2_kernel(const uint32_t *active, uint32_t nact, const uint32_t *need, const uint8_t *msgs, size_t msg_stride,
                                                       const uint32_t *mlens, uint32_t fixed_len, const int16_t *z, int32_t *cidx, uint32_t *ok,
                                                       int mode, uint32_t *verdict, BlissP P, const uint32_t *msg_of = nullptr, const uint32_t *nd = nullptr)
{
    __shared__ uint32_t st[25 * HT];
    __shared__ uint32_t fl[32 * (HT / 2)];
    if (nd) nact = min(nact, *nd);
    const uint32_t aidx = (blockIdx.x * HT + threadIdx.x) >> 1;
    const bool first = (threadIdx.x & 1) == 0;                     // the lane of the pair that writes results
    Sponge2<HT> sp; sp.init(st, threadIdx.x, 72);
    if (aidx >= nact) return;
    if (need && need[aidx] == 2) { if (ok && first) ok[aidx] = 0; return; }
    const uint32_t item = active ? active[aidx] : aidx, src = msg_of ? msg_of[item] : item;
    const uint32_t mlen = mlens ? mlens[src] : fixed_len;
    if (mlen == 0) { if (first) { if (mode == 0) ok[aidx] = 0; else verdict[item] = 1; } return; }
    const uint8_t *m = msgs + (size_t)src * msg_stride;
    const uint8_t *zb = reinterpret_cast<const uint8_t *>(z + (size_t)aidx * P.n);
    uint32_t *myfl = fl + (threadIdx.x >> 1);                      // both lanes keep the same bitmap (same values, same addresses)
    for (int w = 0; w < (P.n + 31) / 32; w++) myfl[w * (HT / 2)] = 0;
    const int mask = (1 << P.n_bits) - 1;
    int cnt = 0;
    uint32_t neq = 0;
    bool done = false;
    for (int r = 0; r < 65536 && !done; r++) {
        sp.init(st, threadIdx.x, 72);
        sp.absorb(m, mlen);
        if ((sp.pos & 7) == 0) sp.template absorb_words<9>(reinterpret_cast<const uint32_t *>(zb), P.n / 4);      // z rows are n int16: 8-byte aligned
        else sp.absorb(zb, 2 * (size_t)P.n);
        sp.absorb_byte((uint8_t)(r >> 8)); sp.absorb_byte((uint8_t)(r & 0xFF));
        sp.finalize(0x06);
        for (int i = 0; i < 64 && !done; i += 2) {
            const uint32_t hi = sp.squeeze_byte(), lo = sp.squeeze_byte();
            const int idx = (int)((hi << 8) + lo) & mask;
            const uint32_t bit = 1u << (idx & 31), wv = myfl[(idx >> 5) * (HT / 2)];
            if (!(wv & bit)) {
                if (mode == 0) { if (first) cidx[(size_t)aidx * P.kappa + cnt] = idx; } else neq |= (uint32_t)(cidx[(size_t)aidx * P.kappa + cnt] != idx);
                cnt++;
                if (cnt == P.kappa) done = true;
                myfl[(idx >> 5) * (HT / 2)] = wv | bit;
            }
        }
    }
    if (!first) return;
    if (mode == 0) ok[aidx] = done ? 1u : 0u;
    else if (!done || neq) verdict[item] = 1;
}

// Wave-wide sum / maximum, the same value in every lane.  Four DPP steps fold a row of 16 lanes (swap neighbours, swap pairs, mirror the halves of 8,
// mirror the row) and four v_readlane + scalar adds fold the rows: no LDS crossbar round trip (six dependent ds_bpermute per reduction as __shfl_xor
// compiles them; greedy_sc does one reduction per challenge index on the critical path of the one-wavefront-per-message decision).
template <typename F>
__device__ __forceinline__ int32_t wave_fold(int32_t v, F op)
{
    v = op(v, __builtin_amdgcn_update_dpp(0, v, 0xB1, 0xF, 0xF, true));          // quad_perm [1,0,3,2]
    v = op(v, __builtin_amdgcn_update_dpp(0, v, 0x4E, 0xF, 0xF, true));          // quad_perm [2,3,0,1]
    v = op(v, __builtin_amdgcn_update_dpp(0, v, 0x141, 0xF, 0xF, true));         // row_half_mirror
    v = op(v, __builtin_amdgcn_update_dpp(0, v, 0x140, 0xF, 0xF, true));         // row_mirror
    return op(op(__builtin_amdgcn_readlane(v, 0), __builtin_amdgcn_readlane(v, 16)), op(__builtin_amdgcn_readlane(v, 32), __builtin_amdgcn_readlane(v, 48)));
}
__device__ __forceinline__ int32_t wave_sum(int32_t v) { return wave_fold(v, [](int32_t a, int32_t b) { return a + b; }); }
__device__ __forceinline__ int32_t wave_max(int32_t v) { return wave_fold(v, [](int32_t a, int32_t b) { return max(a, b); }); }

// A[d] = <f, x^d f> + <g, x^d g> in Z[x] / (x^n + 1): (x^d f)[m] = f[m - d] for m >= d, -f[m - d + n] below.  Once per key load; a thread per d.
__global__ void bl_acorr_kernel(const int16_t *__restrict__ f, const int16_t *__restrict__ g, int32_t *__restrict__ out, int n)
{
    const int d = blockIdx.x * blockDim.x + threadIdx.x;
    if (d >= n) return;
    int32_t acc = 0;
    for (int m = 0; m < n; m++) {
        const int k = m - d;
        const int32_t fv = k >= 0 ? (int32_t)f[k] : -(int32_t)f[k + n], gv = k >= 0 ? (int32_t)g[k] : -(int32_t)g[k + n];
        acc += (int32_t)f[m] * fv + (int32_t)g[m] * gv;
    }
    out[d] = acc;
}

struct BitW {
    uint8_t *p; uint32_t nbytes; uint64_t acc; int nb;
    __device__ __forceinline__ void put(uint32_t v, int bits)
    {
        acc = (acc << bits) | (uint64_t)(v & ((1u << bits) - 1u)); nb += bits;
        while (nb >= 8) { p[nbytes++] = (uint8_t)(acc >> (nb - 8)); nb -= 8; }
    }
    __device__ __forceinline__ void flush() { if (nb) { p[nbytes++] = (uint8_t)(acc << (8 - nb)); nb = 0; } }
};

// one wavefront per message: greedy_sc, the +-(x,y) choice, the rejection test, signature_gen, check_norms, packing
// (bliss_b.c:1399-1463).  Coefficient j = lane + 64 c.
template <int KMAX>                                                       // kappa <= KMAX (32: BLISS-B-0 .. III, 64: anything bliss_create admits)
__global__ __launch_bounds__(256) void bl_decide_kernel(const uint32_t *active, uint32_t nact, const uint32_t *need, const uint32_t *ok, St s,
                                                        const uint32_t *tapes, size_t tape_stride, const uint32_t *tape_lens,
                                                        const int16_t *f, const int16_t *g, int32_t *tu, const int32_t *v, const int16_t *z,
                                                        const int32_t *cidx, uint8_t *sigs, size_t sig_stride, uint32_t *siglens, uint32_t *trials,
                                                        uint32_t *next_active, uint32_t *next_count, BlissP P, double tie_rel, size_t key_stride, uint32_t max_iter,
                                                        const uint32_t *nd, const int32_t *acorr)
{
    extern __shared__ __attribute__((aligned(16))) int32_t xy[];         // per wave: x[n], y[n]; behind the four waves' pairs: per wave the key's 2n-word shift table (one-key path)
    if (nd) nact = min(nact, *nd);
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, n = P.n, C = n / 64;
    const uint32_t aidx = blockIdx.x * 4 + wid;
    if (aidx >= nact) return;
    if (need[aidx] == 2) return;                                          // tape exhausted: already reported
    const uint32_t item = active ? active[aidx] : aidx, src = s.msg_of ? s.msg_of[item] : item;
    int32_t *x = xy + (size_t)wid * 2 * n, *y = x + n;
    const uint32_t it = s.iter[item];
    // max_iter = 2: the item is one pair of attempts (a fresh one and the one that re-uses its vectors); when both fail the item ends with status 7 and the
    // host resolves the message from its later pairs (bl_sign_core).  max_iter = 0: the reference's open-ended loop.
    auto reject = [&]() {
        if (lane != 0) return;
        s.iter[item] = it + 1;
        if (max_iter && it + 1 >= max_iter) { s.status[item] = 7; siglens[item] = 0; }
        else next_active[atomicAdd(next_count, 1u)] = item;
    };
    if (!ok[aidx]) {                                                      // oracle failed: continue (bliss_b.c:1395-1397) WITHOUT the attempt's reservoir draws -
        if (max_iter) { if (lane == 0) { s.status[item] = 8; siglens[item] = 0; } }   // the later pairs' tape positions no longer hold: status 8, the message is redone in order
        else reject();
        return;
    }
    f += (size_t)src * key_stride; g += (size_t)src * key_stride;        // key_stride = n: a private key per message

    constexpr int MAXC = 8;
    int32_t xr[MAXC], yr[MAXC];                                           // the lane's coefficients of greedy_sc's x and y (j = lane + 64 c)
    int32_t num = 0;
    // everything the decision reads from global memory besides the key - the reservoir's state and tape words, the attempt's vectors t and u (v and its
    // rounding z further down) - is requested here, in front of greedy_sc, not after it
    Tape T{ tapes + (size_t)src * tape_stride, tape_lens ? tape_lens[src] : (uint32_t)tape_stride, s.pos[item], s.var_buf[item], s.var_bits[item], false };
    T.prefetch();
    const uint32_t cur = s.cur[item];
    int32_t *t = tu + ((size_t)item * 2 + cur) * n, *u = tu + ((size_t)item * 2 + (cur ^ 1)) * n;
    int32_t tr[MAXC], ur[MAXC];
#pragma unroll
    for (int c = 0; c < MAXC; c++) {
        const int j = lane + 64 * c;
        tr[c] = c < C ? t[j] : 0; ur[c] = c < C ? u[j] : 0;
    }
    const int pos_v = lane < P.kappa ? cidx[(size_t)aidx * P.kappa + lane] : 0;      // the challenge's positions, lane k holds position k (kappa <= 64): read with v_readlane wherever a step needs one
    if (acorr) {
        // greedy_sc (bliss_b.c:622-679) under ONE key, without its kappa dependent sweeps over x and y.  x = sum_k s_k x^{i_k} f and y likewise with g (x^i: the
        // negacyclic shift the two inner loops of a step apply); multiplication by x^i preserves inner products, so the quantity step k tests is
        //     sgn_k = <x^{i_k} f, x> + <x^{i_k} g, y> = sum_{k' < k} s_k' A[|i_k - i_k'|],   A[d] = <f, x^d f> + <g, x^d g>
        // - the key's autocorrelation, n integers tabulated when the key is loaded (bl_acorr_kernel).  The kappa signs are therefore kappa steps of ONE
        // multiply-add per lane (lane k carries sgn_k) instead of kappa sweeps of 2n products with a wavefront reduction each, and x, y are then built in one
        // pass with no dependence between its terms.  All integers, the same sums: the signatures are the reference's byte for byte (tests/test_gpu_bliss.py).
        // the key in LDS as this wavefront's shift table: word n + m = (f[m], g[m]) as two int16, word m = their negatives, m = 0 .. n-1 - then
        // (x^i f, x^i g)[j] = word n + j - i for every j, the sign of the wrapped part included, and a lane's eight coefficients of a term are eight reads
        // at fixed offsets from one address
        uint32_t *fg2 = reinterpret_cast<uint32_t *>(xy + (size_t)4 * 2 * n) + (size_t)wid * 2 * n;
        for (int c = 0; c < C; c++) {
            const int j = lane + 64 * c;
            const int32_t fv = f[j], gv = g[j];
            fg2[n + j] = ((uint32_t)fv & 0xFFFFu) | ((uint32_t)gv << 16);
            fg2[j] = ((uint32_t)(-fv) & 0xFFFFu) | ((uint32_t)(-gv) << 16);
        }
        const int kap = P.kappa;
        const int my_i = pos_v;
        int32_t m[KMAX];                                                  // m[k'] = A[|i_lane - i_k'|]: independent look-ups, all requested before the chain below
#pragma unroll
        for (int kp = 0; kp < KMAX; kp++) {
            if (kp < kap) { const int d = my_i - __builtin_amdgcn_readlane(my_i, kp); m[kp] = acorr[d < 0 ? -d : d]; }
            else m[kp] = 0;
        }
        int32_t acc = 0, my_s = 0;
#pragma unroll
        for (int kp = 0; kp < KMAX; kp++) {
            if (kp < kap) {
                const int32_t sg = __builtin_amdgcn_readlane(acc, kp) > 0 ? -1 : 1;      // step kp's sign: its sum is complete (contributions of the steps before it)
                if (lane == kp) my_s = sg;
                if (lane > kp) acc += sg * m[kp];
            }
        }
        // |x|^2 + |y|^2 = sum_k A[0] + 2 sum_k s_k sgn_k
        num = wave_sum(lane < kap ? acorr[0] + 2 * my_s * acc : 0);
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier();      // fg2 is written
        // x, y as packed int16 pairs (|x|, |y| <= kappa max|f|, max|g|: far below 2^15): one packed multiply-add per term and coefficient
        typedef short s2 __attribute__((ext_vector_type(2)));
        s2 xy16[MAXC];
#pragma unroll
        for (int c = 0; c < MAXC; c++) xy16[c] = s2{ 0, 0 };
        for (int kp = 0; kp < kap; kp++) {
            const int i = __builtin_amdgcn_readlane(my_i, kp), sg = __builtin_amdgcn_readlane(my_s, kp);
            const s2 sgp = s2{ (short)sg, (short)sg };
            const uint32_t *row = fg2 + (n + lane - i);
#pragma unroll
            for (int c = 0; c < MAXC; c++) {
                if (c < C) {
                    const uint32_t w = row[64 * c];
                    s2 fg; __builtin_memcpy(&fg, &w, 4);
                    xy16[c] = fg * sgp + xy16[c];
                }
            }
        }
#pragma unroll
        for (int c = 0; c < MAXC; c++) { xr[c] = (int32_t)xy16[c].x; yr[c] = (int32_t)xy16[c].y; }
    } else {
    // greedy_sc as the reference runs it (a private key per message: its autocorrelation would cost more than the sweeps)
    for (int c = 0; c < C; c++) { x[lane + 64 * c] = 0; y[lane + 64 * c] = 0; }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier();
    // the lane's coefficients of f and g do not change over the kappa steps: read once (n <= 512: eight per lane), as is the next step's index
    int32_t fr[MAXC], gr[MAXC];
#pragma unroll
    for (int c = 0; c < MAXC; c++) { const int j = lane + 64 * c; fr[c] = c < C ? (int32_t)f[j] : 0; gr[c] = c < C ? (int32_t)g[j] : 0; }
    int inext = cidx[(size_t)aidx * P.kappa];
    for (int k = 0; k < P.kappa; k++) {
        const int i = inext;
        if (k + 1 < P.kappa) inext = cidx[(size_t)aidx * P.kappa + k + 1];
        int32_t sgn = 0;
#pragma unroll
        for (int c = 0; c < MAXC; c++) {
            if (c < C) {
                int idx = i + lane + 64 * c; int32_t sg = 1;
                if (idx >= n) { idx -= n; sg = -1; }
                sgn += sg * (fr[c] * x[idx] + gr[c] * y[idx]);
            }
        }
        sgn = wave_sum(sgn);
#pragma unroll
        for (int c = 0; c < MAXC; c++) {
            if (c < C) {
                int idx = i + lane + 64 * c; int32_t sg = 1;
                if (idx >= n) { idx -= n; sg = -1; }
                if (sgn > 0) sg = -sg;
                x[idx] += sg * fr[c];
                y[idx] += sg * gr[c];
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier();
    }
#pragma unroll
    for (int c = 0; c < MAXC; c++) {
        const int j = lane + 64 * c;
        xr[c] = c < C ? x[j] : 0; yr[c] = c < C ? y[j] : 0;
        num += xr[c] * xr[c] + yr[c] * yr[c];
    }
    num = wave_sum(num);
    }

    // the CSPRNG reservoir of this message (every lane computes the same scalar sequence; lane 0 writes it back)
    // (T, t, u and the registers tr, ur, vr, zr: loaded above)
    const bool sub = T.var(1) != 0;
    int32_t den = 0;
#pragma unroll
    for (int c = 0; c < MAXC; c++) {
        if (c < C) {
            const int j = lane + 64 * c;
            const int32_t xx = xr[c], yy = yr[c];
            const int32_t tt = sub ? tr[c] - xx : tr[c] + xx, uu = sub ? ur[c] - yy : ur[c] + yy;
            t[j] = tt; u[j] = uu;
            tr[c] = tt; ur[c] = uu;
            den += tt * xx + uu * yy;
        }
    }
    den = wave_sum(den);
    // v and its rounding z are needed behind the rejection test: requested here, in front of the exp / cosh arithmetic
    int32_t vr[MAXC], zr[MAXC];
#pragma unroll
    for (int c = 0; c < MAXC; c++) {
        const int j = lane + 64 * c;
        vr[c] = c < C ? v[(size_t)aidx * n + j] : 0; zr[c] = c < C ? (int32_t)z[(size_t)aidx * n + j] : 0;
    }
    const double inv_sig2 = 1.0 / ((double)P.sigma * P.sigma);
    const double thresh = 1.0 / (P.m * exp(-0.5 * inv_sig2 * num) * cosh(inv_sig2 * den));
    const double rr = T.dbl();
    if (lane == 0) { s.pos[item] = T.pos; s.var_buf[item] = T.var_buf; s.var_bits[item] = T.var_bits; }
    if (T.underrun) { if (lane == 0) { s.status[item] = 2; siglens[item] = 0; } return; }
    // exp / cosh here are the device's f64 functions, the reference's are glibc's: the two thresholds may differ in the last bits.  A draw closer to
    // the threshold than tie_rel (2^-40 relative: ~13 bits above what either library can be off by) is NOT decided here - it would decide the rest
    // of the signature's tape - but reported (status 5) for the caller to sign that message on the reference path.  Expected once per ~10^12 attempts.
    if (fabs(rr - thresh) <= tie_rel * thresh) { if (lane == 0) { s.status[item] = 5; siglens[item] = 0; } return; }
    if (rr > thresh) { reject(); return; }

    // signature_gen (bliss_b.c:1045-1075; its Barrett multiplier is built from q, not p - kept) + check_norms (:1010-1043)
    const int32_t mm = (1 << 30) / P.q, q2 = 2 * P.q, p2 = P.p >> 1;
    int32_t maxt = 0, maxz = 0, nt = 0, nz = 0;
#pragma unroll
    for (int c = 0; c < MAXC; c++) {
        if (c >= C) continue;
        const int j = lane + 64 * c;
        int32_t tmp = vr[c] - ur[c];
        tmp += (tmp < 0) * q2;
        tmp -= (tmp >= q2) * q2;
        tmp = ((tmp + (1 << (P.d - 1))) >> P.d);
        tmp = tmp - ((tmp * mm) >> 30) * P.p;
        tmp -= (P.p <= tmp) * P.p;
        tmp = zr[c] - tmp;
        tmp += (tmp <= -p2) * P.p;
        tmp -= (tmp > p2) * P.p;
        const int32_t zz = (int32_t)(int16_t)tmp, tt = tr[c];
        x[j] = tt; y[j] = zz;                                            // staged for the packer
        maxt = max(maxt, max(tt, -tt)); maxz = max(maxz, max(zz, -zz));
        nt += tt * tt; nz += zz * zz;
    }
    maxt = wave_max(maxt); maxz = wave_max(maxz); nt = wave_sum(nt); nz = wave_sum(nz);
    if (maxt > P.b_inf || (maxz << P.d) > P.b_inf || nt + (nz << (2 * P.d)) > P.b_l2) { reject(); return; }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier();
    uint8_t *out = sigs + (size_t)item * sig_stride;
    if (P.z1_bits == 12 && P.z2_bits == 3) {
        // BLISS-B-I..IV coding: two 12-bit t coefficients are three bytes, eight 3-bit z coefficients are three bytes - every
        // lane packs its share; the kappa 9-bit indices (byte aligned start) are left to one lane
        for (int p = lane; p < n / 2; p += 64) {
            const uint32_t a = (uint32_t)x[2 * p] & 0xFFFu, b = (uint32_t)x[2 * p + 1] & 0xFFFu;
            out[3 * p] = (uint8_t)(a >> 4); out[3 * p + 1] = (uint8_t)(((a & 0xFu) << 4) | (b >> 8)); out[3 * p + 2] = (uint8_t)b;
        }
        uint8_t *oz = out + n * 12 / 8;
        for (int gq = lane; gq < n / 8; gq += 64) {
            uint32_t acc = 0;
#pragma unroll
            for (int e = 0; e < 8; e++) acc = (acc << 3) | ((uint32_t)y[8 * gq + e] & 7u);
            oz[3 * gq] = (uint8_t)(acc >> 16); oz[3 * gq + 1] = (uint8_t)(acc >> 8); oz[3 * gq + 2] = (uint8_t)acc;
        }
        if (lane == 0) {
            BitW bw{ out, (uint32_t)(n * 15 / 8), 0, 0 };
            for (int k = 0; k < P.kappa; k++) bw.put((uint32_t)__builtin_amdgcn_readlane(pos_v, k), P.n_bits);      // (no load per index: the lane that packs them had kappa round trips in a row here)
            bw.flush();
            siglens[item] = bw.nbytes;
        }
    } else if (lane == 0) {
        BitW bw{ out, 0, 0, 0 };
        for (int j = 0; j < n; j++) bw.put((uint32_t)x[j], P.z1_bits);
        for (int j = 0; j < n; j++) bw.put((uint32_t)y[j], P.z2_bits);
        for (int k = 0; k < P.kappa; k++) bw.put((uint32_t)__builtin_amdgcn_readlane(pos_v, k), P.n_bits);
        bw.flush();
        siglens[item] = bw.nbytes;
    }
    if (lane == 0) { trials[item] = it + 1; s.status[item] = 1; s.iter[item] = it + 1; }
}
