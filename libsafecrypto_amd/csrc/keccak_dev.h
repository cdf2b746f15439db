// keccak_dev.h - device sponge compatible with the reference's xof layer
// (src/utils/crypto/xof.c over src/utils/crypto/sha3/tinysha3.c).
//
// Framing is SHAKE128/256 (rate 168/136, pad 0x1F ... 0x80, incremental absorb / squeeze,
// tinysha3.c:308-330,382-404).  The permutation is the one the reference actually runs: the library
// is always built with -DSHA3_UNROLLED (src/utils/crypto/Makefile.am:94) and that variant's chi step
// (tinysha3.c:186-235) updates the overlapping windows st[j..j+4], j = 0,1,2,3,4, one after the other,
// instead of the five rows at 0,5,10,15,20 - so it is NOT Keccak-f[1600].  Signatures, challenge
// polynomials and masks of the reference are only reproducible with the same permutation, hence it is
// reproduced here bit for bit (template flag STD selects the FIPS 202 chi for self-tests).
//
// One hash instance per lane.  The 25-word state lives in LDS, word w of lane t at st[w*NT + t]
// (conflict-free 8-byte accesses); absorb/squeeze address it dynamically, the permutation pulls it into
// registers with static indices.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace sc {

// 64-bit rotate by a compile-time amount as two v_alignbit_b32 on the halves (full rate); the generic (x << n) | (x >> (64 - n))
// becomes v_lshlrev_b64 + v_lshrrev_b64 + 2 v_or, and 64-bit shifts issue at a fraction of the 32-bit rate on gfx950 - the
// permutation spends a quarter of its instructions on rotates.
__device__ __forceinline__ uint64_t rol64(uint64_t x, int n)
{
    uint32_t lo = (uint32_t)x, hi = (uint32_t)(x >> 32);
    if (n >= 32) { const uint32_t t = lo; lo = hi; hi = t; n -= 32; }
    if (n == 0) return ((uint64_t)hi << 32) | lo;
    const uint32_t nlo = __builtin_amdgcn_alignbit(lo, hi, 32 - n), nhi = __builtin_amdgcn_alignbit(hi, lo, 32 - n);
    return ((uint64_t)nhi << 32) | nlo;
}

// gfx950 has V_BITOP3_B32 (any three-input boolean function in one instruction): the five-way column parity of theta is two of
// them per 32-bit half instead of four XORs, and a chi term a ^ (~b & c) is one instead of two.  The compiler forms them from
// 32-bit expressions but not from the split halves of 64-bit ones, so the permutation is written on explicit halves:
// about 190 instructions per round instead of 270.
__device__ __forceinline__ uint32_t xor3(uint32_t a, uint32_t b, uint32_t c) { return __builtin_amdgcn_bitop3_b32(a, b, c, 0x96); }
__device__ __forceinline__ uint32_t chi3(uint32_t a, uint32_t b, uint32_t c) { return __builtin_amdgcn_bitop3_b32(a, b, c, 0xD2); }   // a ^ (~b & c)

template <bool STD>
__device__ __forceinline__ void keccak_permute_regs(uint64_t (&s)[25])
{
    constexpr uint64_t RC[24] = {
        0x0000000000000001ULL, 0x0000000000008082ULL, 0x800000000000808aULL, 0x8000000080008000ULL, 0x000000000000808bULL,
        0x0000000080000001ULL, 0x8000000080008081ULL, 0x8000000000008009ULL, 0x000000000000008aULL, 0x0000000000000088ULL,
        0x0000000080008009ULL, 0x000000008000000aULL, 0x000000008000808bULL, 0x800000000000008bULL, 0x8000000000008089ULL,
        0x8000000000008003ULL, 0x8000000000008002ULL, 0x8000000000000080ULL, 0x000000000000800aULL, 0x800000008000000aULL,
        0x8000000080008081ULL, 0x8000000000008080ULL, 0x0000000080000001ULL, 0x8000000080008008ULL };
    constexpr int ROTC[24] = { 1, 3, 6, 10, 15, 21, 28, 36, 45, 55, 2, 14, 27, 41, 56, 8, 25, 43, 62, 18, 39, 61, 20, 44 };
    constexpr int PILN[24] = { 10, 7, 11, 17, 18, 3, 5, 16, 8, 21, 24, 4, 15, 23, 19, 13, 12, 2, 20, 14, 22, 9, 6, 1 };
    uint32_t lo[25], hi[25];
#pragma unroll
    for (int i = 0; i < 25; i++) { lo[i] = (uint32_t)s[i]; hi[i] = (uint32_t)(s[i] >> 32); }
#pragma unroll 1
    for (int r = 0; r < 24; r++) {
        uint32_t bl[5], bh[5];
#pragma unroll
        for (int i = 0; i < 5; i++) {
            bl[i] = xor3(xor3(lo[i], lo[i + 5], lo[i + 10]), lo[i + 15], lo[i + 20]);
            bh[i] = xor3(xor3(hi[i], hi[i + 5], hi[i + 10]), hi[i + 15], hi[i + 20]);
        }
#pragma unroll
        for (int i = 0; i < 5; i++) {
            const int p = (i + 4) % 5, n = (i + 1) % 5;
            // D[i] = C[i-1] ^ rol(C[i+1], 1) is never formed: each word takes both terms in one three-way XOR (5 instructions fewer per half and round)
            const uint32_t rl = __builtin_amdgcn_alignbit(bl[n], bh[n], 31), rh = __builtin_amdgcn_alignbit(bh[n], bl[n], 31);   // rol 1
#pragma unroll
            for (int j = 0; j < 25; j += 5) { lo[j + i] = xor3(lo[j + i], bl[p], rl); hi[j + i] = xor3(hi[j + i], bh[p], rh); }
        }
        uint32_t tl = lo[1], th = hi[1];
#pragma unroll
        for (int i = 0; i < 24; i++) {
            const uint32_t nl = lo[PILN[i]], nh = hi[PILN[i]];
            uint32_t a = tl, b = th;
            int n = ROTC[i];
            if (n >= 32) { a = th; b = tl; n -= 32; }
            if (n == 0) { lo[PILN[i]] = a; hi[PILN[i]] = b; }
            else { lo[PILN[i]] = __builtin_amdgcn_alignbit(a, b, 32 - n); hi[PILN[i]] = __builtin_amdgcn_alignbit(b, a, 32 - n); }
            tl = nl; th = nh;
        }
#pragma unroll
        for (int jj = 0; jj < 5; jj++) {
            const int j = STD ? 5 * jj : jj;                    // tinysha3.c:186-235 uses 0,1,2,3,4
            const uint32_t c0 = lo[j], c1 = lo[j + 1], c2 = lo[j + 2], c3 = lo[j + 3], c4 = lo[j + 4];
            lo[j] = chi3(c0, c1, c2); lo[j + 1] = chi3(c1, c2, c3); lo[j + 2] = chi3(c2, c3, c4); lo[j + 3] = chi3(c3, c4, c0); lo[j + 4] = chi3(c4, c0, c1);
            const uint32_t d0 = hi[j], d1 = hi[j + 1], d2 = hi[j + 2], d3 = hi[j + 3], d4 = hi[j + 4];
            hi[j] = chi3(d0, d1, d2); hi[j + 1] = chi3(d1, d2, d3); hi[j + 2] = chi3(d2, d3, d4); hi[j + 3] = chi3(d3, d4, d0); hi[j + 4] = chi3(d4, d0, d1);
        }
        lo[0] ^= (uint32_t)RC[r]; hi[0] ^= (uint32_t)(RC[r] >> 32);
    }
#pragma unroll
    for (int i = 0; i < 25; i++) s[i] = ((uint64_t)hi[i] << 32) | lo[i];
}

// NT = threads per workgroup (LDS stride of the state words)
template <int NT, bool STD = false>
struct Sponge {
    uint64_t *st;          // LDS: &state[threadIdx.x]
    int rate, pos;

    __device__ __forceinline__ void init(uint64_t *lds_state, int tid, int rate_bytes)
    {
        st = lds_state + tid; rate = rate_bytes; pos = 0;
#pragma unroll
        for (int w = 0; w < 25; w++) st[w * NT] = 0;
    }
    // resume from a stored state (25 words) taken at a block boundary
    __device__ __forceinline__ void load(uint64_t *lds_state, int tid, int rate_bytes, const uint64_t *__restrict__ src)
    {
        st = lds_state + tid; rate = rate_bytes; pos = 0;
#pragma unroll
        for (int w = 0; w < 25; w++) st[w * NT] = src[w];
    }
    __device__ __forceinline__ void save(uint64_t *dst) const
    {
#pragma unroll
        for (int w = 0; w < 25; w++) dst[w] = st[w * NT];
    }
    __device__ __forceinline__ void permute()
    {
        uint64_t s[25];
#pragma unroll
        for (int w = 0; w < 25; w++) s[w] = st[w * NT];
        keccak_permute_regs<STD>(s);
#pragma unroll
        for (int w = 0; w < 25; w++) st[w * NT] = s[w];
    }
    __device__ __forceinline__ void absorb_byte(uint8_t b)
    {
        st[(pos >> 3) * NT] ^= (uint64_t)b << (8 * (pos & 7));
        if (++pos == rate) { permute(); pos = 0; }
    }
    // pos must be a multiple of 8 (rates are multiples of 8)
    __device__ __forceinline__ void absorb_u64(uint64_t v)
    {
        st[(pos >> 3) * NT] ^= v;
        pos += 8;
        if (pos == rate) { permute(); pos = 0; }
    }
    __device__ __forceinline__ void absorb(const uint8_t *p, size_t len)
    {
        size_t i = 0;
        while (i < len && (pos & 7)) absorb_byte(p[i++]);
        if ((((uintptr_t)(p + i)) & 7) == 0)
            for (; i + 8 <= len; i += 8) absorb_u64(*reinterpret_cast<const uint64_t *>(p + i));
        for (; i < len; i++) absorb_byte(p[i]);
    }
    // pad = 0x1F: SHAKE (tinysha3.c:382-388); pad = 0x06: fixed-output SHA-3 (tinysha3.c:334-340)
    __device__ __forceinline__ void finalize(uint32_t pad = 0x1F)
    {
        st[(pos >> 3) * NT] ^= (uint64_t)pad << (8 * (pos & 7));
        st[((rate - 1) >> 3) * NT] ^= (uint64_t)0x80 << (8 * ((rate - 1) & 7));
        permute();
        pos = 0;
    }
    __device__ __forceinline__ uint8_t squeeze_byte()
    {
        if (pos == rate) { permute(); pos = 0; }
        const uint8_t b = (uint8_t)(st[(pos >> 3) * NT] >> (8 * (pos & 7)));
        pos++;
        return b;
    }
};

// ------------------------------------------------------------------------------------------------------------------------
// Two lanes per hash instance.  A lane-per-instance kernel over B messages runs B/64 wavefronts; when that is fewer than half
// the SIMDs of the chip (16 K messages = 256 wavefronts on 1024 SIMDs) the kernel runs at the latency of one wavefront's
// serial chain of permutations while most SIMDs idle.  Sponge2 gives each instance the lanes (2m, 2m+1): lane h holds the 32-bit
// half h of every state word, so a round is 126 instructions per wavefront instead of 192 (the column parities and chi act on
// halves independently; a rotate needs the partner's half - one DPP move - and the SAME v_alignbit form on both lanes), and
// twice as many wavefronts carry the batch.  Total work goes up by ~30 %, so this only pays while SIMDs are idle.
// Both lanes of a pair call every method with identical arguments.  State in LDS: half h of word w of pair m at st[w*NT + 2m + h].
__device__ __forceinline__ uint32_t partner_half(uint32_t x)
{
    // quad_perm [1,0,3,2]: swap adjacent lanes (bound_ctrl: no 'old' value to set up).  Measured and dropped: the same exchange through the LDS crossbar
    // (ds_swizzle, which would take the 29 moves per round off the vector ALU) - BLISS sign 1.21 -> 1.42 ms, Dilithium 2.64 -> 2.86, Falcon 1.03 -> 1.10.
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0xB1, 0xF, 0xF, true);
}

template <bool STD>
__device__ __forceinline__ void keccak_permute_halves(uint32_t (&x)[25], int h)
{
    constexpr uint64_t RC[24] = {
        0x0000000000000001ULL, 0x0000000000008082ULL, 0x800000000000808aULL, 0x8000000080008000ULL, 0x000000000000808bULL,
        0x0000000080000001ULL, 0x8000000080008081ULL, 0x8000000000008009ULL, 0x000000000000008aULL, 0x0000000000000088ULL,
        0x0000000080008009ULL, 0x000000008000000aULL, 0x000000008000808bULL, 0x800000000000008bULL, 0x8000000000008089ULL,
        0x8000000000008003ULL, 0x8000000000008002ULL, 0x8000000000000080ULL, 0x000000000000800aULL, 0x800000008000000aULL,
        0x8000000080008081ULL, 0x8000000000008080ULL, 0x0000000080000001ULL, 0x8000000080008008ULL };
    constexpr int ROTC[24] = { 1, 3, 6, 10, 15, 21, 28, 36, 45, 55, 2, 14, 27, 41, 56, 8, 25, 43, 62, 18, 39, 61, 20, 44 };
    constexpr int PILN[24] = { 10, 7, 11, 17, 18, 3, 5, 16, 8, 21, 24, 4, 15, 23, 19, 13, 12, 2, 20, 14, 22, 9, 6, 1 };
    // 64-bit rotate left by n (0 < n < 64, n != 32) of the word whose halves are (own, partner): lo' = alignbit(lo, hi, 32-n),
    // hi' = alignbit(hi, lo, 32-n) for n < 32 - the same expression in terms of (own, partner) on both lanes; operands swap for n > 32
    auto rot = [](uint32_t own, uint32_t par, int n) -> uint32_t {
        return n < 32 ? __builtin_amdgcn_alignbit(own, par, 32 - n) : __builtin_amdgcn_alignbit(par, own, 64 - n);
    };
#pragma unroll 1
    for (int r = 0; r < 24; r++) {
        uint32_t bc[5];
#pragma unroll
        for (int i = 0; i < 5; i++) bc[i] = xor3(xor3(x[i], x[i + 5], x[i + 10]), x[i + 15], x[i + 20]);
#pragma unroll
        for (int i = 0; i < 5; i++) {
            const int p = (i + 4) % 5, n = (i + 1) % 5;
            const uint32_t rt = rot(bc[n], partner_half(bc[n]), 1);         // (as above: the two terms of D[i] go into the words directly)
#pragma unroll
            for (int j = 0; j < 25; j += 5) x[j + i] = xor3(x[j + i], bc[p], rt);
        }
        uint32_t t = x[1];
#pragma unroll
        for (int i = 0; i < 24; i++) {
            const uint32_t nx = x[PILN[i]];
            x[PILN[i]] = rot(t, partner_half(t), ROTC[i]);
            t = nx;
        }
#pragma unroll
        for (int jj = 0; jj < 5; jj++) {
            const int j = STD ? 5 * jj : jj;                    // tinysha3.c:186-235 uses 0,1,2,3,4
            const uint32_t c0 = x[j], c1 = x[j + 1], c2 = x[j + 2], c3 = x[j + 3], c4 = x[j + 4];
            x[j] = chi3(c0, c1, c2); x[j + 1] = chi3(c1, c2, c3); x[j + 2] = chi3(c2, c3, c4); x[j + 3] = chi3(c3, c4, c0); x[j + 4] = chi3(c4, c0, c1);
        }
        x[0] ^= h ? (uint32_t)(RC[r] >> 32) : (uint32_t)RC[r];
    }
}

// NT = threads per workgroup (even); NT/2 instances per workgroup
template <int NT, bool STD = false>
struct Sponge2 {
    uint32_t *st;          // LDS: &state[pair base], i.e. word w of half h at st[w*NT + h]
    int rate, pos, h;

    __device__ __forceinline__ void init(uint32_t *lds_state, int tid, int rate_bytes)
    {
        st = lds_state + (tid & ~1); rate = rate_bytes; pos = 0; h = tid & 1;
#pragma unroll
        for (int w = 0; w < 25; w++) st[w * NT + h] = 0;
    }
    // resume from a stored state (25 64-bit words) taken at a block boundary
    __device__ __forceinline__ void load(uint32_t *lds_state, int tid, int rate_bytes, const uint64_t *__restrict__ src)
    {
        st = lds_state + (tid & ~1); rate = rate_bytes; pos = 0; h = tid & 1;
#pragma unroll
        for (int w = 0; w < 25; w++) st[w * NT + h] = (uint32_t)(src[w] >> (32 * h));
    }
    __device__ __forceinline__ void permute()
    {
        uint32_t x[25];
#pragma unroll
        for (int w = 0; w < 25; w++) x[w] = st[w * NT + h];
        keccak_permute_halves<STD>(x, h);
#pragma unroll
        for (int w = 0; w < 25; w++) st[w * NT + h] = x[w];
    }
    __device__ __forceinline__ void absorb_byte(uint8_t b)
    {
        if (((pos >> 2) & 1) == h) st[(pos >> 3) * NT + h] ^= (uint32_t)b << (8 * (pos & 3));
        if (++pos == rate) { permute(); pos = 0; }
    }
    // pos must be a multiple of 8 (rates are multiples of 8)
    __device__ __forceinline__ void absorb_u64(uint64_t v)
    {
        st[(pos >> 3) * NT + h] ^= (uint32_t)(v >> (32 * h));
        pos += 8;
        if (pos == rate) { permute(); pos = 0; }
    }
    __device__ __forceinline__ void absorb(const uint8_t *p, size_t len)
    {
        size_t i = 0;
        while (i < len && (pos & 7)) absorb_byte(p[i++]);
        if ((((uintptr_t)(p + i)) & 7) == 0)
            for (; i + 8 <= len; i += 8) absorb_u64(*reinterpret_cast<const uint64_t *>(p + i));
        for (; i < len; i++) absorb_byte(p[i]);
    }
    // `nwords` 64-bit words from an 8-byte-aligned source, absorbed at a word-aligned position of a sponge whose rate is RW words: the words of the
    // NEXT block are requested before the current block is permuted (word by word, a lone wavefront pays a memory round trip per word: 9 per block
    // of SHA3-512).  Loads are unconditional - an index past the end reads the last word again.
    template <int RW>
    __device__ __forceinline__ void absorb_words(const uint32_t *__restrict__ src32, int nwords)
    {
        int sw = pos >> 3, w0 = 0;
        const int last = nwords - 1;
        uint32_t cur[RW], nx[RW];
        auto fetch = [&](uint32_t (&d)[RW], int from) {
#pragma unroll
            for (int j = 0; j < RW; j++) d[j] = src32[2 * (from + j < last ? from + j : last) + h];
        };
        fetch(cur, 0);
        if (sw != 0) {                                               // up to the end of the block the sponge stands in: through the LDS image
            const int take = (RW - sw) < nwords ? (RW - sw) : nwords;
            fetch(nx, take);
#pragma unroll
            for (int j = 0; j < RW; j++) if (j < take) st[(sw + j) * NT + h] ^= cur[j];
            sw += take; w0 = take;
            if (sw == RW) { permute(); sw = 0; }
#pragma unroll
            for (int j = 0; j < RW; j++) cur[j] = nx[j];
        }
        if (sw == 0 && nwords - w0 >= RW) {                          // whole blocks: the state stays in registers from block to block
            uint32_t x[25];
#pragma unroll
            for (int w = 0; w < 25; w++) x[w] = st[w * NT + h];
            while (nwords - w0 >= RW) {
                fetch(nx, w0 + RW);
#pragma unroll
                for (int j = 0; j < RW; j++) x[j] ^= cur[j];
                keccak_permute_halves<STD>(x, h);
                w0 += RW;
#pragma unroll
                for (int j = 0; j < RW; j++) cur[j] = nx[j];
            }
#pragma unroll
            for (int w = 0; w < 25; w++) st[w * NT + h] = x[w];
        }
        if (w0 < nwords) {                                           // fewer than a block left (sw = 0 here, or the source ended inside the first block)
            const int take = nwords - w0;
#pragma unroll
            for (int j = 0; j < RW; j++) if (j < take) st[(sw + j) * NT + h] ^= cur[j];
            sw += take;
        }
        pos = sw * 8;
    }
    __device__ __forceinline__ void finalize(uint32_t pad = 0x1F)
    {
        if (((pos >> 2) & 1) == h) st[(pos >> 3) * NT + h] ^= pad << (8 * (pos & 3));
        if ((((rate - 1) >> 2) & 1) == h) st[((rate - 1) >> 3) * NT + h] ^= 0x80u << (8 * ((rate - 1) & 3));
        permute();
        pos = 0;
    }
    // both lanes return the byte (read from the owning half in LDS)
    __device__ __forceinline__ uint8_t squeeze_byte()
    {
        if (pos == rate) { permute(); pos = 0; }
        const uint8_t b = (uint8_t)(st[(pos >> 3) * NT + ((pos >> 2) & 1)] >> (8 * (pos & 3)));
        pos++;
        return b;
    }
};

}  // namespace sc
