// dil_verify_dev.h - the device side of the batched Dilithium verifier that is not shared with the signer: tr = CRH(rho, t1) at key load, and the three
// readers of a verification call's inputs - z's unpacking, the tail's (hints and c, with the structural checks of dilithium.c:2065-2123), mu = CRH(tr, message) -
// as roles of one launch.  Replaces those parts of dilithium_verify (src/schemes/sig/dilithium/dilithium.c:2002-2240); the matrix product with c t1 2^d and
// UseHint fused in is dil_yaw_kernel<..., CT1> (mlwe_dev.h), the challenge dil_challenge_kernel<false, ...> (dil_sign_dev.h).  Included by dilithium.hip
// INSIDE its anonymous namespace behind dil_sign_dev.h; a header so that a verification call can also be run lane by lane on the CPU (tests/test_emu_dil_sign.py).

// tr = SHAKE256( pack(rho: 32 x 8 bits, t1: k*n x bits) )  (collision_resistant_hash_t1, module_lwe.c:140-180)
// tr = CRH(rho, t1) (collision_resistant_hash_t1, dilithium.c:1038-1075): SHAKE256 over rho and the bit-packed t1 -> 48 bytes;
// one lane per key (count keys: rho [count][32], t1 [count][len], out [count][48])
__global__ __launch_bounds__(HT) void dil_crh_t1_kernel(const uint8_t *rho, const int32_t *t1, uint32_t len, int bits, uint8_t *out48, uint32_t count)
{
    __shared__ uint64_t st[25 * HT];
    Sponge<HT> sp; sp.init(st, threadIdx.x, 136);
    const uint32_t key = blockIdx.x * HT + threadIdx.x;
    if (key >= count) return;
    rho += (size_t)key * 32; t1 += (size_t)key * len; out48 += (size_t)key * 48;
    for (int i = 0; i < 32; i++) sp.absorb_byte(rho[i]);
    // the MSB-first bit string is cut into bytes; eight bytes at a time go into the state as one little-endian word (the stream stays 8-byte
    // aligned: 32 bytes of rho, a rate of 17 words) instead of eight read-modify-write rounds on the LDS state
    uint32_t acc = 0; int nb = 0;
    uint64_t word = 0; int wb = 0;
    // (a lane walks its own key: its loads are 16 bytes wide and the next one is in flight while the current four coefficients are packed - the
    // walk is otherwise a chain of dependent 4-byte loads 6 KB apart between lanes, which tripled the kernel's time over its 13 permutations)
    auto put = [&](int32_t c) {
        acc = (acc << bits) | ((uint32_t)c & ((1u << bits) - 1)); nb += bits;
        while (nb >= 8) {
            word |= (uint64_t)(uint8_t)(acc >> (nb - 8)) << (8 * wb); nb -= 8;
            if (++wb == 8) { sp.absorb_u64(word); word = 0; wb = 0; }
        }
        acc &= (1u << nb) - 1;
    };
    uint32_t i = 0;
    if ((len & 3) == 0 && (reinterpret_cast<uintptr_t>(t1) & 15) == 0 && len >= 4) {
        const int4 *t4 = reinterpret_cast<const int4 *>(t1);
        int4 cur = t4[0];
        for (; i + 4 <= len; i += 4) {
            const int4 nxt = t4[i + 4 < len ? (i >> 2) + 1 : (i >> 2)];
            put(cur.x); put(cur.y); put(cur.z); put(cur.w);
            cur = nxt;
        }
    }
    for (; i < len; i++) put(t1[i]);
    for (int i = 0; i < wb; i++) sp.absorb_byte((uint8_t)(word >> (8 * i)));
    if (nb) sp.absorb_byte((uint8_t)(acc << (8 - nb)));
    sp.finalize();
    for (int i = 0; i < 48; i++) out48[i] = sp.squeeze_byte();
}

// verify: unpack z (-> canonical), hints, c; structural checks (dilithium.c:2065-2123)
// One workgroup per signature, a lane per EIGHT coefficients: 8 z_bits bits are z_bits whole bytes (20 for every reference parameter set: five words
// where the row starts on a word boundary), out as two 16-byte stores.  (One lane per pair of coefficients - five byte loads, two 4-byte stores, a
// 64-bit division for its indices - took 51 of the verifier's 378 us at batch 16384.)
// ZB: z_bits at compile time (the word a field starts in is then a constant: the group's words stay in registers); 0 = any width, at run time
template <int ZB>
__device__ __forceinline__ void dil_unpackz_body(uint32_t block, const uint8_t *sigs, size_t sig_stride, uint32_t count, int32_t *z, int ln, int z_bits_rt, int32_t q,
                                                 uint32_t gamma1_beta, uint32_t *verdict)
{
    const int z_bits = ZB ? ZB : z_bits_rt;
    const uint32_t item = block;
    const int grp = threadIdx.x;
    if (item >= count || grp >= ln / 8) return;
    const uint8_t *src = sigs + (size_t)item * sig_stride + (size_t)grp * z_bits;
    uint32_t W[8];                                                      // the group's bytes as big-endian words (z_bits <= 24 bytes -> 6 words, one spare)
    const int nw = (z_bits + 3) / 4;
    if ((reinterpret_cast<uintptr_t>(src) & 3) == 0 && (z_bits & 3) == 0) {
#pragma unroll
        for (int w = 0; w < 6; w++) W[w] = w < nw ? __builtin_bswap32(reinterpret_cast<const uint32_t *>(src)[w]) : 0u;
    } else {
#pragma unroll
        for (int w = 0; w < 6; w++) {
            uint32_t v = 0;
#pragma unroll
            for (int b = 0; b < 4; b++) v = (v << 8) | (4 * w + b < z_bits ? (uint32_t)src[4 * w + b] : 0u);
            W[w] = v;
        }
    }
    W[6] = W[7] = 0;
    int32_t out[8];
    bool bad = false;
#pragma unroll
    for (int e = 0; e < 8; e++) {
        const int pos = e * z_bits, w = pos >> 5, o = pos & 31;        // field e: bits [pos, pos + z_bits) of the big-endian string
        const uint64_t V = ((uint64_t)W[w] << 32) | W[w + 1];
        const uint32_t raw = (uint32_t)(V >> (64 - o - z_bits)) & ((1u << z_bits) - 1u);
        const int32_t sv = (int32_t)(raw << (32 - z_bits)) >> (32 - z_bits);           // signed raw coefficient
        const uint32_t zz = (uint32_t)(sv < 0 ? sv + q : sv);                          // normalize_32
        out[e] = (int32_t)zz;
        bad |= norm_bad(zz, (uint32_t)q, gamma1_beta);
    }
    int4 *dst = reinterpret_cast<int4 *>(z + (size_t)item * ln + 8 * grp);
    dst[0] = make_int4(out[0], out[1], out[2], out[3]); dst[1] = make_int4(out[4], out[5], out[6], out[7]);
    if (bad) verdict[item] = 1;
}
template <int ZB>
__global__ void dil_unpackz_kernel(const uint8_t *sigs, size_t sig_stride, uint32_t count, int32_t *z, int ln, int z_bits_rt, int32_t q,
                                   uint32_t gamma1_beta, uint32_t *verdict)
{ dil_unpackz_body<ZB>(blockIdx.x, sigs, sig_stride, count, z, ln, z_bits_rt, q, gamma1_beta, verdict); }

// the tail of a signature (dilithium.c:2085-2123): omega_bits count, `count` hint positions of h_bits each, then c as n 2-bit
// codes (3 = -1).  One wavefront per signature: every field has a fixed width, so lane i reads hint i and four codes of c; the
// number of DISTINCT hint positions (what the reference's "cnt += h[p] == 0; h[p] = 1" loop counts) comes from an LDS bitmap.
__device__ __forceinline__ void dil_unpack_tail_body(uint32_t block, const uint8_t *sigs, size_t sig_stride, const uint32_t *siglens, uint32_t count,
                                                     uint8_t *h, int32_t *c, uint32_t *verdict, const DilP &P)
{
    __shared__ uint32_t seen[4][64];                                   // k*n <= 2048 hint positions per signature
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const uint32_t item = block * 4 + wv;
    if (item >= count) return;
    const int kn = P.k * P.n, h_bits = 8 + ((P.k + 1) >> 1);
    const uint8_t *s = sigs + (size_t)item * sig_stride;
    // an (untrusted) length beyond the row is invalid, and nothing is ever read past the row: positions are bounded by `total`
    const uint32_t slen = siglens[item];
    const bool too_long = slen > sig_stride;
    const uint32_t total = (uint32_t)(too_long ? sig_stride : slen) * 8;
    const uint32_t zbits = (uint32_t)(P.l * P.n * P.z_bits);
    auto get = [&](uint32_t pos, int bits) -> uint32_t {                // bits <= 16, the caller has checked pos + bits <= total
        const uint32_t b0 = pos >> 3;
        const uint32_t w = ((uint32_t)s[b0] << 16) | ((uint32_t)(b0 + 1 < sig_stride ? s[b0 + 1] : 0) << 8) | (b0 + 2 < sig_stride ? s[b0 + 2] : 0);
        return (w >> (24 - (pos & 7) - bits)) & ((1u << bits) - 1u);
    };
    uint8_t *hh = h + (size_t)item * kn;
    for (int i = lane * 16; i < kn; i += 64 * 16) *reinterpret_cast<uint4 *>(hh + i) = make_uint4(0, 0, 0, 0);
    seen[wv][lane] = 0;
    bool err = too_long || total < zbits + (uint32_t)P.omega_bits;
    uint32_t no = err ? 0 : get(zbits, P.omega_bits);
    if (no > (uint32_t)P.omega) { err = true; no = 0; }                 // more hints than check_hint_ones allows: invalid whatever they are
    const uint32_t hbase = zbits + (uint32_t)P.omega_bits, cbase = hbase + no * (uint32_t)h_bits;
    if (!err && total < cbase + 2u * (uint32_t)P.n) err = true;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    uint32_t fresh = 0, oob = 0;
    if (!err) {
        for (uint32_t i = lane; i < no; i += 64) {
            const uint32_t p = get(hbase + i * (uint32_t)h_bits, h_bits);
            if (p >= (uint32_t)kn) { oob = 1; continue; }                // the reference writes h[p] unchecked; out-of-range = invalid
            const uint32_t bit = 1u << (p & 31), old = atomicOr(&seen[wv][p >> 5], bit);
            fresh += (old & bit) ? 0u : 1u;
            hh[p] = 1;
        }
    }
    for (int o = 32; o > 0; o >>= 1) { fresh += (uint32_t)__shfl_xor((int)fresh, o); oob |= (uint32_t)__shfl_xor((int)oob, o); }
    int32_t *ci = c + (size_t)item * P.n;
    for (int i = lane * 4; i < P.n; i += 256) {
        const uint32_t v = err ? 0u : get(cbase + 2u * (uint32_t)i, 8);
        int32_t o4[4];
#pragma unroll
        for (int e = 0; e < 4; e++) { const uint32_t t = (v >> (6 - 2 * e)) & 3u; o4[e] = (t == 3) ? -1 : (int32_t)t; }
        *reinterpret_cast<int4 *>(ci + i) = make_int4(o4[0], o4[1], o4[2], o4[3]);
    }
    if (lane == 0 && (err || oob || fresh > (uint32_t)P.omega)) verdict[item] = 1;     // check_hint_ones (dilithium.c:2119-2123)
}
__global__ __launch_bounds__(256) void dil_unpack_tail_kernel(const uint8_t *sigs, size_t sig_stride, const uint32_t *siglens, uint32_t count,
                                                              uint8_t *h, int32_t *c, uint32_t *verdict, DilP P)
{ dil_unpack_tail_body(blockIdx.x, sigs, sig_stride, siglens, count, h, c, verdict, P); }

// The verifier's tail unpacking and mu = CRH(tr, message) as ONE launch: the first workgroups hash (a workgroup's first HT lanes: a lane per message), the rest unpack.  The two have nothing to do with each other, and mu is a chain of one permutation that took a launch of its own to wait
// for (16 us of a 300 us verification call at batch 16384); forking a stream for it costs as much as it hides (measured in round 3), a second role in a
// launch that is running anyway costs nothing.
// (and z's unpacking, a workgroup per signature, behind them: the three readers of a verification call's inputs are one launch)
template <int ZB>
__global__ __launch_bounds__(256) void dil_tail_mu_kernel(const uint8_t *sigs, size_t sig_stride, const uint32_t *siglens, uint32_t count, uint8_t *h, int32_t *c, uint32_t *verdict, DilP P,
                                                          const uint8_t *tr, const uint8_t *msgs, size_t msg_stride, const uint32_t *mlens, uint32_t fixed_len, uint8_t *mu, int32_t *z)
{
    const uint32_t nb_mu = (count + HT - 1) / HT, nb_tail = (count + 3) / 4;      // the hashing workgroups first: their chain starts with the launch and runs under the unpacking
    if (blockIdx.x >= nb_mu + nb_tail) dil_unpackz_body<ZB>(blockIdx.x - nb_mu - nb_tail, sigs, sig_stride, count, z, P.l * P.n, P.z_bits, P.q, (uint32_t)(P.gamma1 - P.beta), verdict);
    else if (blockIdx.x >= nb_mu) dil_unpack_tail_body(blockIdx.x - nb_mu, sigs, sig_stride, siglens, count, h, c, verdict, P);
    else if (threadIdx.x < HT) dil_mu_body(blockIdx.x, tr, msgs, msg_stride, mlens, fixed_len, mu, count, 0u);
}
