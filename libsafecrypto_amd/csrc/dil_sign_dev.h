// dil_sign_dev.h - the device side of the batched Dilithium signer: every kernel of a signing call (csrc/dilithium.hip sign_core queues them).
// Replaces dilithium_sign (src/schemes/sig/dilithium/dilithium.c:1397-1958) and the module_lwe.c helpers it calls, an attempt per slot.
// Included by dilithium.hip INSIDE its anonymous namespace, behind keccak_dev.h and mlwe_dev.h (hence no includes here); a header so that the kernels'
// own text can also be compiled as host code over tests/emu/wave_emu.h and a whole signing call be run lane by lane under the sanitizers the GPU pool
// cannot offer (tests/test_emu_dil_sign.py).

constexpr int HT = 64;                    // threads per workgroup of the one-thread-per-message kernels
constexpr int EXP_T = 64;                 // ExpandMask (state in registers).  Measured: 256-thread workgroups (a wavefront per SIMD of a CU) are SLOWER - 400 against 375 us at 49152
                                          // attempts, 361 against 298 us at 12800 - a lone wavefront wants a CU to itself
constexpr int CH_T = 64;                  // challenge hash (128: 182 against 126 us at 49152 attempts)

struct DilP {                             // dilithium_set_t (dilithium_params.h:60-93), the fields used here
    int32_t n, q, q_bits, k, l, d, weight, gamma1, gamma1_bits, gamma2, eta, z_bits, beta, omega, omega_bits, set;
};

// mu = SHAKE256(tr || M)  (collision_resistant_hash_message, module_lwe.c:182-198)
__device__ __forceinline__ void dil_mu_body(uint32_t block, const uint8_t *tr, const uint8_t *msgs, size_t stride, const uint32_t *lens,
                                            uint32_t fixed_len, uint8_t *mu, uint32_t count, uint32_t tr_stride)
{
    __shared__ uint64_t st[25 * HT];
    const uint32_t item = block * HT + threadIdx.x;
    const bool valid = item < count;
    Sponge<HT> sp; sp.init(st, threadIdx.x, 136);
    if (!valid) return;
    sp.absorb(tr + (size_t)item * tr_stride, 48);                 // tr_stride = 48: a key per item
    const uint32_t len = lens ? lens[item] : fixed_len;
    sp.absorb(msgs + (size_t)item * stride, len);
    sp.finalize();
    for (int i = 0; i < 48; i++) mu[(size_t)item * 48 + i] = sp.squeeze_byte();
}
__global__ __launch_bounds__(HT) void dil_mu_kernel(const uint8_t *tr, const uint8_t *msgs, size_t stride, const uint32_t *lens,
                                                    uint32_t fixed_len, uint8_t *mu, uint32_t count, uint32_t tr_stride = 0)
{ dil_mu_body(blockIdx.x, tr, msgs, stride, lens, fixed_len, mu, count, tr_stride); }

// The signer's round state lives on the device: rs[0] = messages still unsigned, rs[1] = attempts per message this round ("slots" per
// message), rs[2] = rs[0] * rs[1].  Written by dil_round_kernel at the head of a round from the count the previous round's last kernel left,
// so a signing call queues its rounds back to back without a host read-back between them.
enum { RS_NACT = 0, RS_S = 1, RS_NSLOTS = 2, RS_PREV = 3 };          // RS_PREV: the previous round's slot count (what may still hold its y and w)

__global__ void dil_round_kernel(uint32_t *rs, uint32_t *next_count, uint32_t first_nact, uint32_t slot_cap, uint32_t s_max)
{
    if (threadIdx.x | blockIdx.x) return;
    const uint32_t nact = first_nact ? first_nact : *next_count;
    uint32_t S = nact ? slot_cap / nact : 0;
    if (S > s_max) S = s_max;
    if (nact && S < 1) S = 1;
    rs[RS_PREV] = first_nact ? 0u : rs[RS_NSLOTS];
    rs[RS_NACT] = nact; rs[RS_S] = S; rs[RS_NSLOTS] = nact * S;
    *next_count = 0;
}

// zeroes y, w (and NTT(y)) of the slots of the round the state describes: behind the last round of a call
__global__ void dil_wipe_kernel(const uint32_t *rs, int32_t *y, uint32_t y_words, int32_t *w, uint32_t w_words, uint32_t *yh)
{
    const uint32_t nslots = rs[RS_NSLOTS];
    const size_t ny = (size_t)nslots * y_words / 4, nw = (size_t)nslots * w_words / 4, stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < ny; i += stride) {
        reinterpret_cast<uint4 *>(y)[i] = make_uint4(0, 0, 0, 0);
        if (yh) reinterpret_cast<uint4 *>(yh)[i] = make_uint4(0, 0, 0, 0);
    }
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < nw; i += stride) reinterpret_cast<uint4 *>(w)[i] = make_uint4(0, 0, 0, 0);
}

// y = ExpandMask(K, mu, kappa)  (module_lwe.c:201-244): SHAKE256(mu || K || kappa_hi || kappa_lo) -> 20-bit fields, rejection below 2 gamma_1 - 2.
// One lane per attempt.  The input is 82 bytes - one block - and the squeeze walks whole blocks, so the sponge needs no dynamic addressing:
// the state stays in registers from beginning to end (no LDS round trip per permutation, no LDS limit on residency).
// TWO: two lanes per attempt - lane h of a pair holds the 32-bit half h of every state word (keccak_permute_halves: 125 instead of 190 vector
// instructions per round).  The kernel is a chain of ~25 permutations per attempt at less than one wavefront per SIMD for any batch BASELINE names,
// i.e. it runs at the issue rate of a lone wavefront: what counts is the length of that chain, not the work.  Both lanes parse the squeezed block
// (each fetches the partner's halves with two DPP moves per word); the even lane stores.
template <bool TWO>
__global__ __launch_bounds__(EXP_T) void dil_expand_kernel(const uint32_t *rs, const uint32_t *active, const uint32_t *kappa, const uint8_t *K,
                                                        const uint8_t *mu, int32_t *y, int limit, int32_t gamma1, int32_t q, uint32_t K_stride, uint32_t *slot_item,
                                                        int32_t *wipe_w, uint32_t wipe_w_words, uint32_t *wipe_yh)
{
    constexpr uint32_t APB = TWO ? EXP_T / 2 : EXP_T;                       // attempts per workgroup
    const uint32_t aidx = TWO ? (blockIdx.x * EXP_T + threadIdx.x) >> 1 : blockIdx.x * EXP_T + threadIdx.x;
    const int h = TWO ? (int)(threadIdx.x & 1) : 0;
    const uint32_t S = rs[RS_S], nslots = rs[RS_NSLOTS];
    const uint32_t nhb = (nslots + APB - 1) / APB;                          // workgroups that hash
    if (blockIdx.x >= nhb) {
        // A workgroup with no attempt to expand zeroes the masks y and the products w = A y that the PREVIOUS round left in slots this round does
        // not overwrite, [nslots, prev) - the per-call secrets (dilithium.c:1964-1975).  Every slot is zeroed once per call, by workgroups that would
        // otherwise leave at once, while the hashing workgroups keep the vector units busy and leave the memory system idle.
        const uint32_t prev = rs[RS_PREV], surplus = gridDim.x - nhb;
        for (uint32_t s0 = nslots + (blockIdx.x - nhb) * EXP_T; s0 < prev; s0 += surplus * EXP_T) {
            const uint32_t cnt = prev - s0 < (uint32_t)EXP_T ? prev - s0 : (uint32_t)EXP_T;
            uint4 *zy = reinterpret_cast<uint4 *>(y + (size_t)s0 * limit), *zw = reinterpret_cast<uint4 *>(wipe_w + (size_t)s0 * wipe_w_words);
            const uint32_t ny = cnt * (uint32_t)limit / 4, nw = cnt * wipe_w_words / 4;
            for (uint32_t i = threadIdx.x; i < ny; i += EXP_T) zy[i] = make_uint4(0, 0, 0, 0);
            for (uint32_t i = threadIdx.x; i < nw; i += EXP_T) zw[i] = make_uint4(0, 0, 0, 0);
            if (wipe_yh) {
                uint4 *zh = reinterpret_cast<uint4 *>(wipe_yh + (size_t)s0 * limit);
                for (uint32_t i = threadIdx.x; i < ny; i += EXP_T) zh[i] = make_uint4(0, 0, 0, 0);
            }
        }
        return;
    }
    if (aidx >= nslots) return;
    // slot aidx = (message index in the round's list) * S + attempt: message active[aidx / S], kappa = the message's base + aidx % S
    const uint32_t item = active ? active[aidx / S] : aidx / S;
    if (h == 0) slot_item[aidx] = item;
    const uint32_t kp = (kappa[item] + aidx % S) & 0xFFFFu;            // UINT16 kappa (dilithium.c:1421)
    uint64_t s[TWO ? 1 : 25];
    uint32_t x[TWO ? 25 : 1];
    const uint8_t *kb = K + (size_t)item * K_stride;                   // K_stride = 32: a key per message (the caller's buffer: any alignment)
    if constexpr (TWO) {
        const uint32_t *m4 = reinterpret_cast<const uint32_t *>(mu + (size_t)item * 48) + h;     // the workspace's own array: aligned
#pragma unroll
        for (int w = 0; w < 6; w++) x[w] = m4[2 * w];
        if ((reinterpret_cast<uintptr_t>(kb) & 3) == 0) {
#pragma unroll
            for (int w = 0; w < 4; w++) x[6 + w] = reinterpret_cast<const uint32_t *>(kb)[2 * w + h];
        } else {
#pragma unroll
            for (int w = 0; w < 4; w++) {
                uint32_t v = 0;
#pragma unroll
                for (int b = 0; b < 4; b++) v |= (uint32_t)kb[8 * w + 4 * h + b] << (8 * b);
                x[6 + w] = v;
            }
        }
        x[10] = h ? 0u : ((kp >> 8) | ((kp & 0xFFu) << 8) | (0x1Fu << 16));        // kappa high byte first, then the SHAKE pad
#pragma unroll
        for (int w = 11; w < 25; w++) x[w] = 0;
        if (h) x[16] = 0x80u << 24;                                          // rate 136: last byte of word 16
        keccak_permute_halves<false>(x, h);
    } else {
        const uint64_t *m8 = reinterpret_cast<const uint64_t *>(mu + (size_t)item * 48);
#pragma unroll
        for (int w = 0; w < 6; w++) s[w] = m8[w];
        if ((reinterpret_cast<uintptr_t>(kb) & 7) == 0) {
#pragma unroll
            for (int w = 0; w < 4; w++) s[6 + w] = reinterpret_cast<const uint64_t *>(kb)[w];
        } else {
#pragma unroll
            for (int w = 0; w < 4; w++) {
                uint64_t v = 0;
#pragma unroll
                for (int b = 0; b < 8; b++) v |= (uint64_t)kb[8 * w + b] << (8 * b);
                s[6 + w] = v;
            }
        }
        s[10] = (uint64_t)(kp >> 8) | ((uint64_t)(kp & 0xFF) << 8) | (0x1Full << 16);     // kappa high byte first, then the SHAKE pad
#pragma unroll
        for (int w = 11; w < 25; w++) s[w] = 0;
        s[16] ^= 0x80ull << 56;                                              // rate 136: last byte of word 16
        keccak_permute_regs<false>(s);
    }
    const uint32_t thresh = (uint32_t)(2 * gamma1 - 2);
    const int32_t add = q + gamma1 - 1;
    int32_t *yy = y + (size_t)aidx * limit;
    // The 5-byte groups of the reference (two 20-bit samples: b0 | b1<<8 | (b2&15)<<16 and b2>>4 | b3<<4 | b4<<12, module_lwe.c:218-236)
    // are consecutive 20-bit fields of the squeezed stream read as a little-endian bit string.  A 136-byte block is 54.4 fields, so where the fields
    // sit in a block repeats every five blocks (272 fields = 68 groups of four): the loop body is five blocks with every field's position - and its
    // place in its group of four - known at compile time (one or two instructions to cut a field out, no bit buffer, no per-field branch).
    // Rejection is rare (2^20 - (2 gamma_1 - 2) values of 2^20: 0.1 %): a group whose four fields are all accepted leaves as one 16-byte store
    // (every lane of a store instruction writes to a line of its own, and a 4-byte piece of it costs the memory system as much as a 16-byte one);
    // only a lane that rejected something, or is four samples from the end, goes field by field.
    struct __attribute__((packed, aligned(4))) Four { int32_t a, b, c, d; };       // a lane that has rejected a field stores its later groups off 16-byte boundaries
    int j = 0;
    int32_t gv[4] = { 0, 0, 0, 0 };
    bool gok[4] = { false, false, false, false };
    uint32_t pc[34], carry = 0;
    auto pieces = [&]() {
#pragma unroll
        for (int w = 0; w < 17; w++) {
            if constexpr (TWO) {                                         // half 0 / half 1 of the word, each from the lane of the pair that holds it
                pc[2 * w] = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x[w], 0xA0, 0xF, 0xF, true);          // quad_perm [0,0,2,2]
                pc[2 * w + 1] = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x[w], 0xF5, 0xF, 0xF, true);      // quad_perm [1,1,3,3]
            } else { pc[2 * w] = (uint32_t)s[w]; pc[2 * w + 1] = (uint32_t)(s[w] >> 32); }
        }
    };
    auto flush = [&]() {
        if (gok[0] & gok[1] & gok[2] & gok[3] & (j + 4 <= limit)) {
            if (h == 0) *reinterpret_cast<Four *>(yy + j) = Four{ gv[0], gv[1], gv[2], gv[3] };
            j += 4;
        } else {
#pragma unroll
            for (int e = 0; e < 4; e++)
                if (gok[e] && j < limit) { if (h == 0) yy[j] = gv[e]; j++; }
        }
    };
    // block PH of the five: r bits of its first field are the top r bits of the previous block's last piece (carry); F0 fields came before it
    auto block = [&](auto ph) {
        constexpr int PH = decltype(ph)::value;
        constexpr int r = (8 * PH) % 20, nf = (1088 + r) / 20, F0 = PH == 0 ? 0 : PH == 1 ? 54 : PH == 2 ? 108 : PH == 3 ? 163 : 217;
        pieces();
#pragma unroll
        for (int i = 0; i < nf; i++) {
            const int st = 20 * i - r;
            uint32_t f;
            if (st < 0) f = __builtin_amdgcn_alignbit(pc[0], carry, 32 - r);
            else {
                const int p = st >> 5, o = st & 31;
                f = o <= 12 ? (pc[p] >> o) : __builtin_amdgcn_alignbit(pc[p < 33 ? p + 1 : 33], pc[p], o);
            }
            f &= 0xFFFFFu;
            const int slot = (F0 + i) & 3;
            gv[slot] = add - (int32_t)f; gok[slot] = f < thresh;
            if (slot == 3) flush();
        }
        carry = pc[33];
        if (j < limit) { if constexpr (TWO) keccak_permute_halves<false>(x, h); else keccak_permute_regs<false>(s); }
    };
    while (j < limit) {
        block(std::integral_constant<int, 0>{});
        if (j < limit) block(std::integral_constant<int, 1>{});
        if (j < limit) block(std::integral_constant<int, 2>{});
        if (j < limit) block(std::integral_constant<int, 3>{});
        if (j < limit) block(std::integral_constant<int, 4>{});
    }
}

// c = H(mu, w1): SHAKE128 -> 8 sign bytes + 60 position bytes -> kyber_oracle_core
// (dilithium.c:1011-1036, module_lwe.c:1115-1141).  SIGN: one lane per attempt of the round (count from the round state), c written as n signed
// bytes per attempt; !SIGN (verify): one lane per signature, compared with the signature's c.
// The hash input mu || w1 is 6 + 32 k whole 64-bit words: it is absorbed 21 words (one SHAKE128 block) at a time with the state in registers, and the
// next block's words are requested before the current block is permuted - a lane walks its own 1.5 KB of w1, and a sponge that fetched word by
// word paid the latency of every one of those loads in sequence.  Only the 72 output bytes are addressed dynamically (through LDS).
// TWO: two lanes per hash - lane h of a pair holds the 32-bit half h of every state word (keccak_permute_halves), reads its half of every input word,
// and the pair shares the column of LDS the challenge polynomial is built in: its zeroing, its conversion and (verify) its comparison are split
// between the two lanes, the 60 dependent steps of the oracle run on the even one.  Like ExpandMask this kernel is a chain of permutations
// (11 for Dilithium-III) at a fraction of a wavefront per SIMD: the two-lane form shortens the chain by a third.
template <bool SIGN, bool TWO>
__global__ __launch_bounds__(CH_T) void dil_challenge_kernel(const uint32_t *rs, const uint32_t *slot_item, uint32_t nact, const uint8_t *mu, const uint8_t *w1,
                                                           int w1len, int weight, int8_t *c8, const int32_t *c32, uint32_t *verdict, uint32_t two_max)
{
    constexpr int APB = TWO ? CH_T / 2 : CH_T;                                  // hashes per workgroup
    __shared__ uint32_t ob[18 * APB];                                          // the 72 output bytes: half hh of word w at ob[(2 w + hh) * APB + column]
    __shared__ int8_t cc[APB * N];
    const uint32_t col = TWO ? threadIdx.x >> 1 : threadIdx.x;
    const int h = TWO ? (int)(threadIdx.x & 1) : 0;
    const uint32_t aidx = blockIdx.x * APB + col;
    if (SIGN) {
        // the signer queues both forms every round and the round's size, which only the device knows, picks one: two lanes per hash while that keeps
        // the kernel at no more than about a wavefront per SIMD (measured at 49152 attempts: 125 us one lane, 133 two; at 13 K: 112 and 82)
        nact = rs[RS_NSLOTS];
        if (TWO ? nact > two_max : nact <= two_max) return;
    }
    if (blockIdx.x * APB >= nact) return;                                        // (whole workgroup)
    const bool valid = aidx < nact;
    const uint32_t item = valid ? aidx : 0;                                    // slot: index of w1 / c
    const uint32_t mitem = valid ? (slot_item ? slot_item[aidx] : aidx) : 0;   // message: index of mu
    const int total = 6 + w1len / 8;                                           // words of input
    if constexpr (TWO) {
        const uint32_t *m4 = reinterpret_cast<const uint32_t *>(mu + (size_t)mitem * 48) + h, *w4 = reinterpret_cast<const uint32_t *>(w1 + (size_t)item * w1len) + h;
        auto word = [&](int gi) -> uint32_t {                                  // unconditional loads (a guarded one is a branch with a wait inside)
            const int ci = gi < total ? gi : total - 1;
            const uint32_t v = ci < 6 ? m4[2 * ci] : w4[2 * (ci - 6)];
            return gi < total ? v : 0;
        };
        uint32_t x[25], cur[21];
#pragma unroll
        for (int w = 0; w < 25; w++) x[w] = 0;
#pragma unroll
        for (int w = 0; w < 21; w++) cur[w] = word(w);
        int base = 0;
        for (; base + 21 <= total; base += 21) {
            uint32_t nxt[21];
#pragma unroll
            for (int w = 0; w < 21; w++) nxt[w] = word(base + 21 + w);
#pragma unroll
            for (int w = 0; w < 21; w++) x[w] ^= cur[w];
            keccak_permute_halves<false>(x, h);
#pragma unroll
            for (int w = 0; w < 21; w++) cur[w] = nxt[w];
        }
        const int rem = total - base;                                          // 0..20 words of the last block, then the pad
#pragma unroll
        for (int w = 0; w < 21; w++) { x[w] ^= cur[w]; if (w == rem && h == 0) x[w] ^= 0x1F; }
        if (h) x[20] ^= 0x80u << 24;
        keccak_permute_halves<false>(x, h);
#pragma unroll
        for (int w = 0; w < 9; w++) ob[(2 * w + h) * APB + col] = x[w];
    } else {
        const uint64_t *m8 = reinterpret_cast<const uint64_t *>(mu + (size_t)mitem * 48), *w8 = reinterpret_cast<const uint64_t *>(w1 + (size_t)item * w1len);
        auto word = [&](int gi) -> uint64_t {                                  // unconditional loads (a guarded one is a branch with a wait inside)
            const int ci = gi < total ? gi : total - 1;
            const uint64_t v = ci < 6 ? m8[ci] : w8[ci - 6];
            return gi < total ? v : 0;
        };
        uint64_t s[25], cur[21];
#pragma unroll
        for (int w = 0; w < 25; w++) s[w] = 0;
#pragma unroll
        for (int w = 0; w < 21; w++) cur[w] = word(w);
        int base = 0;
        for (; base + 21 <= total; base += 21) {
            uint64_t nxt[21];
#pragma unroll
            for (int w = 0; w < 21; w++) nxt[w] = word(base + 21 + w);
#pragma unroll
            for (int w = 0; w < 21; w++) s[w] ^= cur[w];
            keccak_permute_regs<false>(s);
#pragma unroll
            for (int w = 0; w < 21; w++) cur[w] = nxt[w];
        }
        const int rem = total - base;                                          // 0..20 words of the last block, then the pad
#pragma unroll
        for (int w = 0; w < 21; w++) { s[w] ^= cur[w]; if (w == rem) s[w] ^= 0x1F; }
        s[20] ^= 0x80ull << 56;
        keccak_permute_regs<false>(s);
#pragma unroll
        for (int w = 0; w < 9; w++) { ob[(2 * w) * APB + col] = (uint32_t)s[w]; ob[(2 * w + 1) * APB + col] = (uint32_t)(s[w] >> 32); }
    }
    auto out_byte = [&](int i) -> uint32_t { return (ob[(i >> 2) * APB + col] >> (8 * (i & 3))) & 0xFFu; };
    int8_t *mine = cc + col;                  // the hash's private column: coefficient i at mine[i*APB] (no systematic bank conflict)
    constexpr int LN = TWO ? 2 : 1;           // lanes that share the column: lane h takes every LN-th piece of the parallel steps
    for (int i = h; i < N; i += LN) mine[i * APB] = 0;
    sc_mlwe::wave_lds_fence();
    if (valid && h == 0) {
        const int nwb = (weight + 7) >> 3;
        int pos = nwb;
        uint32_t sj = 0, mask = 1;
        for (int i = N - weight; i < N; i++) {
            uint32_t b = out_byte(pos++);
            while (b > (uint32_t)i) b >>= 1;
            mine[i * APB] = mine[b * APB];
            mine[b * APB] = (out_byte((int)sj) & mask) ? -1 : 1;
            mask <<= 1;
            if (mask == 256) { mask = 1; sj++; }
        }
    }
    sc_mlwe::wave_lds_fence();
    if (valid) {
        if (SIGN) {
            uint32_t *dst = reinterpret_cast<uint32_t *>(c8 + (size_t)item * N);
            for (int i = 4 * h; i < N; i += 4 * LN)
                dst[i >> 2] = (uint32_t)(uint8_t)mine[i * APB] | ((uint32_t)(uint8_t)mine[(i + 1) * APB] << 8) | ((uint32_t)(uint8_t)mine[(i + 2) * APB] << 16) |
                              ((uint32_t)(uint8_t)mine[(i + 3) * APB] << 24);
        } else {
            uint32_t neq = 0;
            for (int i = h; i < N; i += LN) neq |= (uint32_t)(c32[(size_t)item * N + i] != mine[i * APB]);
            if (neq) verdict[item] = 1;
        }
    }
}

// MSB-first bit writer: bits are collected in a register and leave as whole bytes (packer.c:299-351 semantics:
// values masked to `bits`, the last byte zero padded)
struct BitW {
    uint8_t *p; uint32_t nbytes; uint64_t acc; int nb;
    __device__ __forceinline__ void put(uint32_t v, int bits)
    {
        acc = (acc << bits) | (uint64_t)(v & ((1u << bits) - 1u)); nb += bits;
        while (nb >= 8) { p[nbytes++] = (uint8_t)(acc >> (nb - 8)); nb -= 8; }
    }
    __device__ __forceinline__ void flush() { if (nb) { p[nbytes++] = (uint8_t)(acc << (8 - nb)); nb = 0; } }
};

// Everything of an attempt behind the challenge hash, and the decision over a message's attempts, in ONE kernel - one wavefront per message that is
// still unsigned, walking the S attempts ("slots") of its round in kappa order and stopping at the first that passes (the attempt the reference's
// sequential loop stops at, dilithium.c:1566-1725):
//   NTT(c) once, kept in registers; z = y + c s1 (l inverse transforms, four polynomials side by side on the 16-lane groups of the wavefront) and
//   ||z||inf < gamma_1 - beta; r = w - c s2 and ||LowBits(r)||inf < gamma_2 - beta; c t0 with ||c t0||inf < gamma_2 - beta and the hints
//   [HighBits(r + c t0) != HighBits(r)], at most omega of them; then the signature - z centred in z_bits per coefficient, the hint count and
//   positions, c in two bits per coefficient, MSB first (dilithium.c:1803, 1914-1945; packer.c:299-351) - straight from LDS / registers.
// An attempt that fails a test costs nothing after that test, attempts behind the accepted one cost nothing at all, and z, r, the hints and NTT(c)
// never exist in HBM (the five kernels this replaces passed them between each other: 25 KB per attempt).  The masks y and w = A y of the
// message's S slots - the per-call secrets (dilithium.c:1964-1975) - are zeroed here, by their last reader, instead of by memsets behind the call.
// Messages with no accepted attempt advance kappa by S and are appended to the next round's list (one global atomic per workgroup).
struct DilFin {
    const int8_t *c8;                     // [slot][n]
    const uint32_t *s1, *s2, *t0;         // NTT-domain key vectors, internal order, Montgomery form
    size_t s1_stride, k_stride;           // words between the key vectors of consecutive MESSAGES (0: one key for the batch)
    const uint32_t *active;               // the round's message list (null: identity)
    const uint32_t *rs;
    uint32_t *kappa;
    uint8_t *sigs; size_t sig_stride; uint32_t *siglens, *trials;
    uint32_t *next_active, *next_count;
    DilP P;
};

static inline size_t cfin_wave_words(const DilP &P) { return 1024u + 256u + 256u * (size_t)P.l + 128u + 64u + 64u * (size_t)P.k; }

// the operands of one pass (four polynomials side by side): a key polynomial's 16 words and the 16 coefficients of w or y the epilogue needs
struct FinOps { uint4 key[4]; int32_t pre[16]; };

// L2P / K2P: l > 4 / k > 4, i.e. the z stage / the r and c t0 stages take two passes.  The passes of an attempt run in the order r, z, c t0 (the
// r test rejects most often); each pass requests the NEXT pass's operands before its own inverse transform - also across a stage boundary, where
// they are wasted if the test fails - because a wavefront walking its message's attempts is otherwise a chain of exposed HBM / L2 round trips.
template <int MODE, bool L2P, bool K2P>
__global__ __launch_bounds__(256, 2) void dil_cfin_kernel(DilNtt a, DilFin f)
{
    constexpr bool SG = MODE == MODE_S32;
    extern __shared__ __attribute__((aligned(16))) uint32_t fin_lds[];
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const int pw = lane / LP, lp = lane % LP;
    const DilP &P = f.P;
    const int l = P.l, kk = P.k;
    uint32_t *wbase = fin_lds + 4 + (size_t)wid * (1024u + 256u + 256u * l + 128u + 64u + 64u * kk);
    uint32_t *lds_poly = wbase + pw * N;                                // the transforms' exchange image (ntt_core.h)
    uint32_t *chl = wbase + 1024;                                       // NTT(c), internal order (the four groups hold the same 16 words per lane: kept once, in LDS)
    uint32_t *zl = chl + 256;                                           // z of the current attempt, natural order [l][n]
    uint32_t *tw = zl + 256 * l;                                        // signature tail, big-endian words
    uint32_t *cb = tw + 128;                                            // c as n signed bytes
    uint8_t *hl = reinterpret_cast<uint8_t *>(cb + 64);                 // hints [k][n]
    const ModK k = a.k;
    const uint32_t nact = f.rs[RS_NACT], S = f.rs[RS_S];
    const uint32_t aidx = blockIdx.x * 4 + wid;
    const bool live = aidx < nact;                                      // wave-uniform
    if (threadIdx.x == 0) fin_lds[0] = 0;
    __syncthreads();
    int found = -1;
    uint32_t item = 0;
    if (live) {
        item = f.active ? f.active[aidx] : aidx;
        TwSet<L8> twi;
        twi.load(a.zi, a.zip, lp);
        const uint32_t *s1k = f.s1 + (size_t)item * f.s1_stride + lp * 16, *s2k = f.s2 + (size_t)item * f.k_stride + lp * 16, *t0k = f.t0 + (size_t)item * f.k_stride + lp * 16;
        const uint32_t *c32 = reinterpret_cast<const uint32_t *>(f.c8) + (size_t)aidx * S * 64 + lane;
        uint32_t creg = c32[0];
        for (uint32_t at = 0; at < S && found < 0; at++) {
            const uint32_t slot = aidx * S + at;
            FinOps o[2];
            uint32_t x[16];
            // pass (stage, p): group pw handles polynomial t = 4 p + pw of the stage; a group without one repeats polynomial 0 and stores nothing
            auto issue = [&](FinOps &d, const uint32_t *key, const int32_t *arr, int per, int p) {
                const int t = 4 * p + pw, tt = t < per ? t : 0;
                const uint4 *pk = reinterpret_cast<const uint4 *>(key + (size_t)tt * N);
#pragma unroll
                for (int c = 0; c < 4; c++) d.key[c] = pk[c];
                if (arr == a.w && a.w_chunked == 2) {                   // ... 24 bits per coefficient: three pieces, unpacked into the same sixteen registers
                    const int4 *pa = reinterpret_cast<const int4 *>(arr + ((size_t)slot * per + tt) * sc_mlwe::W24) + lp;
                    uint32_t pk[12];
#pragma unroll
                    for (int c = 0; c < 3; c++) { const int4 v = pa[16 * c]; pk[4 * c] = (uint32_t)v.x; pk[4 * c + 1] = (uint32_t)v.y; pk[4 * c + 2] = (uint32_t)v.z; pk[4 * c + 3] = (uint32_t)v.w; }
#pragma unroll
                    for (int e = 0; e < 16; e++) d.pre[e] = (int32_t)sc_mlwe::unpack24(pk, e);
                } else if (arr == a.w) {                                // w is stored in 16-byte pieces (DilNtt::w_chunked)
                    const int4 *pa = reinterpret_cast<const int4 *>(arr + ((size_t)slot * per + tt) * N) + lp;
#pragma unroll
                    for (int c = 0; c < 4; c++) { const int4 v = pa[16 * c]; d.pre[4 * c] = v.x; d.pre[4 * c + 1] = v.y; d.pre[4 * c + 2] = v.z; d.pre[4 * c + 3] = v.w; }
                } else if (arr) {
                    const int32_t *pa = arr + ((size_t)slot * per + tt) * N + lp;
#pragma unroll
                    for (int r = 0; r < 16; r++) d.pre[r] = pa[r * LP];
                }
            };
            auto product = [&](const FinOps &d) {                       // x = c o key (internal order: 16 contiguous words per lane)
#pragma unroll
                for (int c = 0; c < 4; c++) {
                    const uint4 vk = d.key[c], vc = reinterpret_cast<const uint4 *>(chl + lp * 16)[c];
                    if constexpr (SG) {
                        x[4 * c + 0] = montmul_s32(vc.x, vk.x, k); x[4 * c + 1] = montmul_s32(vc.y, vk.y, k);
                        x[4 * c + 2] = montmul_s32(vc.z, vk.z, k); x[4 * c + 3] = montmul_s32(vc.w, vk.w, k);
                    } else {
                        x[4 * c + 0] = montmul(vc.x, vk.x, k); x[4 * c + 1] = montmul(vc.y, vk.y, k);
                        x[4 * c + 2] = montmul(vc.z, vk.z, k); x[4 * c + 3] = montmul(vc.w, vk.w, k);
                    }
                }
            };
            issue(o[0], s2k, a.w, kk, 0);                                   // both passes of the r stage are requested before NTT(c) is computed
            if constexpr (K2P) issue(o[1], s2k, a.w, kk, 1);
            cb[lane] = creg;
            if (at + 1 < S) creg = c32[(size_t)(at + 1) * 64];              // the next attempt's c, should this one fail
            wave_lds_fence();
            {
                const int8_t *cbb = reinterpret_cast<const int8_t *>(cb);
#pragma unroll
                for (int r = 0; r < 16; r++) {
                    const int32_t v = cbb[r * LP + lp];
                    x[r] = SG ? (uint32_t)v : (v < 0 ? k.q - 1u : (uint32_t)v);
                }
                // the forward twiddles are loaded per attempt and die before the passes below (as in dil_yaw_kernel)
                const uint32_t *zf = a.zf, *zfp = a.zfp;
                keep_scalar(zf, zfp);
                TwSet<L8> twf;
                twf.load(zf, zfp, lp);
                fwd_core<L8, MODE, true>(x, twf, lds_poly, lp, k);
#pragma unroll
                for (int r = 0; r < 16; r++) x[r] = SG ? canon_s32(x[r], k) : csub(Mod<MODE_U32>::mulc(x[r], 1u, k.one_p, k), k.q);
                if (pw == 0) {
#pragma unroll
                    for (int c = 0; c < 4; c++) reinterpret_cast<uint4 *>(chl + lp * 16)[c] = make_uint4(x[4 * c], x[4 * c + 1], x[4 * c + 2], x[4 * c + 3]);
                }
                wave_lds_fence();
            }
            bool bad = false;
            uint32_t rr[2][16];
            // ---- r = w - c s2; ||LowBits(r)||inf < gamma_2 - beta (dilithium.c:1671-1700)
            auto epi_r = [&](const FinOps &d, int p) {
                const bool act = 4 * p + pw < kk;
#pragma unroll
                for (int r = 0; r < 16; r++) {
                    const uint32_t v = csub((uint32_t)d.pre[r] + k.q - x[r], k.q);
                    rr[p][r] = v;
                    int32_t a0;
                    round_alpha(v, a0, a.alpha, a.am, (int32_t)k.q);
                    const uint32_t lo = a0 < 0 ? (uint32_t)(a0 + (int32_t)k.q) : (uint32_t)a0;       // low_order_bits
                    bad |= act && norm_bad(lo, k.q, (uint32_t)a.gamma2_beta);
                }
            };
            constexpr int Z0 = K2P ? 0 : 1, Z1 = 1 - Z0, T0 = L2P ? Z0 : Z1, T1 = 1 - T0;     // which operand set the stages' first passes find their operands in
            product(o[0]);
            if constexpr (!K2P) issue(o[1], s1k, a.y, l, 0);
            inv_core<L8, MODE>(x, twi, lds_poly, lp, k, a.c0, a.c0p, a.c1, a.c1p);
            epi_r(o[0], 0);
            if constexpr (K2P) {
                if (__ballot(bad) == 0) {                                   // an attempt the first four rows already reject skips the rest of the stage
                    product(o[1]);
                    issue(o[0], s1k, a.y, l, 0);
                    inv_core<L8, MODE>(x, twi, lds_poly, lp, k, a.c0, a.c0p, a.c1, a.c1p);
                    epi_r(o[1], 1);
                }
            }
            bool ok = __ballot(bad) == 0;
            if (ok) {
                // ---- z = y + c s1, normalised; ||z||inf < gamma_1 - beta (dilithium.c:1655-1669)
                auto epi_z = [&](const FinOps &d, int p) {
                    const int t = 4 * p + pw;
#pragma unroll
                    for (int r = 0; r < 16; r++) {
                        const uint32_t zz = csub(csub(x[r] + reduce_i32(d.pre[r], k), k.q2), k.q);
                        if (t < l) { zl[t * N + r * LP + lp] = zz; bad |= norm_bad(zz, k.q, (uint32_t)a.gamma1_beta); }
                    }
                };
                product(o[Z0]);
                if constexpr (L2P) issue(o[Z1], s1k, a.y, l, 1); else issue(o[Z1], t0k, nullptr, kk, 0);
                inv_core<L8, MODE>(x, twi, lds_poly, lp, k, a.c0, a.c0p, a.c1, a.c1p);
                epi_z(o[Z0], 0);
                if constexpr (L2P) {
                    if (__ballot(bad) == 0) {                               // likewise: four of the five rows decide most rejections
                        product(o[Z1]);
                        issue(o[Z0], t0k, nullptr, kk, 0);
                        inv_core<L8, MODE>(x, twi, lds_poly, lp, k, a.c0, a.c0p, a.c1, a.c1p);
                        epi_z(o[Z1], 1);
                    }
                }
                ok = __ballot(bad) == 0;
            }
            uint32_t ones = 0;
            if (ok) {
                // ---- c t0; ||c t0||inf < gamma_2 - beta; hints (make_hint, dilithium.c:1258-1284, 1702-1725)
                uint32_t cnt = 0;
                auto epi_t = [&](int p) {
                    const int t = 4 * p + pw;
#pragma unroll
                    for (int r = 0; r < 16; r++) {
                        const uint32_t sum = csub(rr[p][r] + x[r], k.q);        // (w - c s2) + c t0
                        int32_t a0;
                        const uint32_t hb = round_alpha(sum, a0, a.alpha, a.am, (int32_t)k.q) != round_alpha(rr[p][r], a0, a.alpha, a.am, (int32_t)k.q);
                        if (t < kk) { bad |= norm_bad(x[r], k.q, (uint32_t)a.gamma2_beta); hl[t * N + r * LP + lp] = (uint8_t)hb; cnt += hb; }
                    }
                };
                product(o[T0]);
                if constexpr (K2P) issue(o[T1], t0k, nullptr, kk, 1);
                inv_core<L8, MODE>(x, twi, lds_poly, lp, k, a.c0, a.c0p, a.c1, a.c1p);
                epi_t(0);
                if constexpr (K2P) {
                    product(o[T1]);
                    inv_core<L8, MODE>(x, twi, lds_poly, lp, k, a.c0, a.c0p, a.c1, a.c1p);
                    epi_t(1);
                }
                for (int o_ = 32; o_ > 0; o_ >>= 1) cnt += (uint32_t)__shfl_xor((int)cnt, o_);
                ones = cnt;
                ok = __ballot(bad) == 0 && ones <= (uint32_t)P.omega;
            }
            if (!ok) continue;
            // ---- accepted: the signature
            found = (int)at;
            wave_lds_fence();
            uint8_t *out = f.sigs + (size_t)item * f.sig_stride;
            {
                const int32_t half = ((int32_t)k.q - 1) >> 1;
                const int pairs = l * N / 2, nbytes = 2 * P.z_bits / 8;
                const uint64_t m = (1ull << P.z_bits) - 1;
                for (int pr = lane; pr < pairs; pr += 64) {
                    int32_t v0 = (int32_t)zl[2 * pr], v1 = (int32_t)zl[2 * pr + 1];
                    if (v0 > half) v0 -= (int32_t)k.q;
                    if (v1 > half) v1 -= (int32_t)k.q;
                    const uint64_t bits = (((uint64_t)(uint32_t)v0 & m) << P.z_bits) | ((uint64_t)(uint32_t)v1 & m);     // 2 z_bits bits, MSB first
                    uint8_t *dst = out + (size_t)pr * nbytes;
                    for (int b = 0; b < nbytes; b++) dst[b] = (uint8_t)(bits >> (8 * (nbytes - 1 - b)));
                }
            }
            // tail: count, hint positions in ascending order, c - assembled MSB-first in LDS as big-endian 32-bit words, every field's bit offset
            // from a prefix sum of the per-lane hint counts
            const int kn = kk * P.n, h_bits = 8 + ((kk + 1) >> 1);
            tw[lane] = 0; tw[lane + 64] = 0;
            wave_lds_fence();
            auto put = [&](uint32_t off, uint32_t v, int bits) {                // bits <= 16: at most two words
                const uint32_t w = off >> 5; const int sh = 32 - (int)(off & 31) - bits;
                if (sh >= 0) atomicOr(&tw[w], v << sh);
                else { atomicOr(&tw[w], v >> (-sh)); atomicOr(&tw[w + 1], v << (32 + sh)); }
            };
            if (lane == 0) put(0, ones, P.omega_bits);
            const int chunk = kn / 64;                                          // kn = k * 256
            const uint8_t *hh = hl + lane * chunk;
            uint32_t mine = 0;
            for (int i = 0; i < chunk; i++) mine += hh[i];
            uint32_t incl = mine;
            for (int o = 1; o < 64; o <<= 1) { const uint32_t t = (uint32_t)__shfl_up((int)incl, o); if (lane >= o) incl += t; }
            uint32_t rank = incl - mine;
            for (int i = 0; i < chunk; i++)
                if (hh[i]) { put((uint32_t)P.omega_bits + rank * (uint32_t)h_bits, (uint32_t)(lane * chunk + i), h_bits); rank++; }
            const uint32_t cbase = (uint32_t)P.omega_bits + ones * (uint32_t)h_bits;
            {
                const uint32_t v = cb[lane];                                    // c[4 lane .. 4 lane + 3]: two bits each, -1 -> 3 (dilithium.c:1939-1945)
                const uint32_t b = ((v & 3u) << 6) | (((v >> 8) & 3u) << 4) | (((v >> 16) & 3u) << 2) | ((v >> 24) & 3u);
                put(cbase + 8u * (uint32_t)lane, b, 8);
            }
            wave_lds_fence();
            const uint32_t tbits = cbase + 2u * (uint32_t)P.n, tbytes = (tbits + 7) / 8, zbytes = (uint32_t)(l * P.n * P.z_bits / 8);
            for (uint32_t i = lane; i < tbytes; i += 64) out[zbytes + i] = (uint8_t)(tw[i >> 2] >> (24 - 8 * (i & 3)));
            if (lane == 0) { f.trials[item] = f.kappa[item] + at + 1; f.siglens[item] = zbytes + tbytes; }
        }
        // (the masks y and w = A y of the round's slots are zeroed by the next round's dil_expand_kernel / the call's dil_wipe_kernel)
    }
    const bool again = live && found < 0 && lane == 0;
    uint32_t rpos = 0;
    if (again) rpos = atomicAdd(&fin_lds[0], 1u);
    __syncthreads();
    if (threadIdx.x == 0 && fin_lds[0]) fin_lds[1] = atomicAdd(f.next_count, fin_lds[0]);
    __syncthreads();
    if (again) { f.kappa[item] += S; f.next_active[fin_lds[1] + rpos] = item; }
}

#include "dil_cfins.h"
