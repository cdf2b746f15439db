// drbg_dev.h - the reference's default CSPRNG on the device: AES-256 CTR-DRBG as create_csprng() instantiates it
// (src/utils/arith/module_lwe.c:914-940 -> src/utils/crypto/ctr_drbg.c:37-197, prng_get_func.c:108-119,159-193,
// prng.c:95-143,963-993,1050-1107).  One generator per lane.
//
// Instantiation (ctr_drbg.c:37-62,94-149): the context starts zeroed; counters 1, 2, 3 - each replicated into the four
// little-endian words of a block - are encrypted under the ALL-ZERO key into bytes[32..48), [16..32), [0..16) (a constant,
// computed once on the host: DrbgConst); 4 seed bytes -> ctr, the next 32 (the seed is read circularly) -> key;
// key ^= bytes[12..44); counter = 3 ^ ctr; key schedule.  Output (ctr_drbg.c:164-197): block i = AES_key(counter + i, replicated
// x4).  prng_mem hands these bytes out in order, whole 64-byte units per call (prng.c:1050-1107); prng_32 / prng_64 read the
// same stream through a pool filled with 64-bit little-endian draws split HIGH word first (prng.c:95-143), i.e. the words of
// every 16-byte block come out as w1, w0, w3, w2.
//
// AES is FIPS-197 AES-256 (the reference uses Gladman's implementation, src/utils/crypto/aes): T-table form, the table
// Te0[x] = (2s, s, s, 3s) in LDS (filled by the workgroup from the S-box passed in DrbgConst; the S-box is read out of its second byte), round keys in
// registers.  State columns are little-endian words (byte r of a word = row r).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace sc {

struct DrbgConst {
    uint8_t sbox[256];
    uint8_t init_bytes[48];       // AES_0(3) | AES_0(2) | AES_0(1)
};

// REP copies of the T-table, copy (lane mod REP) for lane: with REP = 32 every lane of a half-wavefront reads its own LDS bank and the 208 table
// look-ups of a block are conflict-free (two passes per wavefront) instead of ~5 deep on random indices - the bulk expansions (a matrix per key) are
// bound by exactly these look-ups.  REP = 1 keeps the 2 KB form for kernels whose LDS goes to hash states.  The S-box is byte 1 of Te0[x].
template <int REP>
struct AesLdsT {
    uint32_t te0[256 * REP];
    // every lane of the workgroup takes part; call once, then __syncthreads()
    __device__ void fill(const DrbgConst &c)
    {
        for (int e = threadIdx.x; e < 256 * REP; e += blockDim.x) {
            const uint32_t s = c.sbox[e / REP], s2 = ((s << 1) ^ ((s >> 7) * 0x11b)) & 0xff, s3 = s2 ^ s;
            te0[e] = s2 | (s << 8) | (s << 16) | (s3 << 24);
        }
    }
    __device__ __forceinline__ uint32_t t(uint32_t x) const { return te0[x * REP + (threadIdx.x & (REP - 1))]; }
    __device__ __forceinline__ uint32_t sb(uint32_t x) const { return (t(x) >> 8) & 0xffu; }
    // the entry for byte K of the state word s
    template <int K> __device__ __forceinline__ uint32_t tk(uint32_t s) const { return t((s >> (8 * K)) & 0xffu); }
};
using AesLds = AesLdsT<1>;
using AesLdsWide = AesLdsT<32>;

// The table for the bulk expansions (a tape per message: 34 G blocks/s wanted): 64 copies, one per lane, entry x of copy c at byte 256 x + 4 c.  A look-up's
// LDS address is then (byte K of the state word) in address byte 1 and the lane's 4 c in byte 0: ONE v_perm_b32 instead of an extract, a shift and an OR
// (the 32-copy form spends 2.1 vector instructions per look-up on the address, 464 of the ~750 of a block), and every lane of a half-wavefront still has a
// bank of its own.  64 KB: two 512-thread workgroups per CU.  Must be the kernel's only LDS object placed first (its addresses are used as offsets from it).
struct AesLds64 {
    uint32_t te0[256 * 64];
    __device__ void fill(const DrbgConst &c)
    {
        for (int e = threadIdx.x; e < 256 * 64; e += blockDim.x) {
            const uint32_t s = c.sbox[e >> 6], s2 = ((s << 1) ^ ((s >> 7) * 0x11b)) & 0xff, s3 = s2 ^ s;
            te0[e] = s2 | (s << 8) | (s << 16) | (s3 << 24);
        }
    }
};
// per-lane view of an AesLds64 (the table itself is shared; the lane's copy offset lives in a register)
struct AesLane64 {
    const uint32_t *te0; uint32_t off;
    __device__ __forceinline__ AesLane64(const AesLds64 &L) : te0(L.te0), off((threadIdx.x & 63u) * 4u) {}
    template <int K> __device__ __forceinline__ uint32_t tk(uint32_t s) const
    {
        const uint32_t a = __builtin_amdgcn_perm(s, off, 0x0c0c0000u | ((4u + K) << 8));       // bytes: [0] = off, [1] = byte K of s, [2] = [3] = 0
        return *reinterpret_cast<const uint32_t *>(reinterpret_cast<const char *>(te0) + a);
    }
    __device__ __forceinline__ uint32_t t(uint32_t x) const { return *reinterpret_cast<const uint32_t *>(reinterpret_cast<const char *>(te0) + ((x << 8) | off)); }
    __device__ __forceinline__ uint32_t sb(uint32_t x) const { return (t(x) >> 8) & 0xffu; }
};

__device__ __forceinline__ uint32_t rotl32(uint32_t x, int n) { return __builtin_amdgcn_alignbit(x, x, 32 - n); }
__device__ __forceinline__ uint32_t x3(uint32_t a, uint32_t b, uint32_t c) { return __builtin_amdgcn_bitop3_b32(a, b, c, 0x96); }      // three-way XOR in one instruction

struct Aes256 {
    uint32_t rk[60];
    template <class LT>
    __device__ __forceinline__ void set_key(const uint32_t key[8], const LT &L)
    {
#pragma unroll
        for (int i = 0; i < 8; i++) rk[i] = key[i];
        uint32_t rc = 1;
#pragma unroll
        for (int i = 8; i < 60; i++) {
            uint32_t t = rk[i - 1];
            if (i % 8 == 0) {
                t = (t >> 8) | (t << 24);                                                     // RotWord on a little-endian word
                t = L.sb(t & 0xff) | (L.sb((t >> 8) & 0xff) << 8) | (L.sb((t >> 16) & 0xff) << 16) | (L.sb(t >> 24) << 24);
                t ^= rc;
                rc = ((rc << 1) ^ ((rc >> 7) * 0x11b)) & 0xff;
            } else if (i % 8 == 4) {
                t = L.sb(t & 0xff) | (L.sb((t >> 8) & 0xff) << 8) | (L.sb((t >> 16) & 0xff) << 16) | (L.sb(t >> 24) << 24);
            }
            rk[i] = rk[i - 8] ^ t;
        }
    }
    template <class LT>
    __device__ __forceinline__ void encrypt(const uint32_t in[4], uint32_t out[4], const LT &L) const
    {
        uint32_t s0 = in[0] ^ rk[0], s1 = in[1] ^ rk[1], s2 = in[2] ^ rk[2], s3 = in[3] ^ rk[3];
#pragma unroll
        for (int r = 1; r < 14; r++) {
            const uint32_t t0 = x3(x3(L.template tk<0>(s0), rotl32(L.template tk<1>(s1), 8), rotl32(L.template tk<2>(s2), 16)), rotl32(L.template tk<3>(s3), 24), rk[4 * r]);
            const uint32_t t1 = x3(x3(L.template tk<0>(s1), rotl32(L.template tk<1>(s2), 8), rotl32(L.template tk<2>(s3), 16)), rotl32(L.template tk<3>(s0), 24), rk[4 * r + 1]);
            const uint32_t t2 = x3(x3(L.template tk<0>(s2), rotl32(L.template tk<1>(s3), 8), rotl32(L.template tk<2>(s0), 16)), rotl32(L.template tk<3>(s1), 24), rk[4 * r + 2]);
            const uint32_t t3 = x3(x3(L.template tk<0>(s3), rotl32(L.template tk<1>(s0), 8), rotl32(L.template tk<2>(s1), 16)), rotl32(L.template tk<3>(s2), 24), rk[4 * r + 3]);
            s0 = t0; s1 = t1; s2 = t2; s3 = t3;
        }
        // last round: the S-box is byte 1 of an entry
        auto sbk = [&](uint32_t e, int shift) -> uint32_t { return ((e >> 8) & 0xffu) << shift; };
        out[0] = (sbk(L.template tk<0>(s0), 0) | sbk(L.template tk<1>(s1), 8) | sbk(L.template tk<2>(s2), 16) | sbk(L.template tk<3>(s3), 24)) ^ rk[56];
        out[1] = (sbk(L.template tk<0>(s1), 0) | sbk(L.template tk<1>(s2), 8) | sbk(L.template tk<2>(s3), 16) | sbk(L.template tk<3>(s0), 24)) ^ rk[57];
        out[2] = (sbk(L.template tk<0>(s2), 0) | sbk(L.template tk<1>(s3), 8) | sbk(L.template tk<2>(s0), 16) | sbk(L.template tk<3>(s1), 24)) ^ rk[58];
        out[3] = (sbk(L.template tk<0>(s3), 0) | sbk(L.template tk<1>(s0), 8) | sbk(L.template tk<2>(s1), 16) | sbk(L.template tk<3>(s2), 24)) ^ rk[59];
    }
};

// two blocks under one key, round by round side by side: the sixteen look-ups of one block's round are in flight while the other block's round is being
// combined - a wavefront that walks ONE block alternates between waiting for LDS and issuing vector instructions, and the bulk expansion ran at the SUM of
// its LDS time and its vector time (197 us for 134 MB where either alone is ~100 us)
template <class LT>
__device__ __forceinline__ void aes256_encrypt2(const Aes256 &A, const uint32_t in0[4], const uint32_t in1[4], uint32_t out0[4], uint32_t out1[4], const LT &L)
{
    const uint32_t *rk = A.rk;
    uint32_t a0 = in0[0] ^ rk[0], a1 = in0[1] ^ rk[1], a2 = in0[2] ^ rk[2], a3 = in0[3] ^ rk[3];
    uint32_t b0 = in1[0] ^ rk[0], b1 = in1[1] ^ rk[1], b2 = in1[2] ^ rk[2], b3 = in1[3] ^ rk[3];
#pragma unroll
    for (int r = 1; r < 14; r++) {
        // all 32 look-ups of the two rounds first, then the combining
        const uint32_t p00 = L.template tk<0>(a0), p01 = L.template tk<1>(a1), p02 = L.template tk<2>(a2), p03 = L.template tk<3>(a3);
        const uint32_t p10 = L.template tk<0>(a1), p11 = L.template tk<1>(a2), p12 = L.template tk<2>(a3), p13 = L.template tk<3>(a0);
        const uint32_t p20 = L.template tk<0>(a2), p21 = L.template tk<1>(a3), p22 = L.template tk<2>(a0), p23 = L.template tk<3>(a1);
        const uint32_t p30 = L.template tk<0>(a3), p31 = L.template tk<1>(a0), p32 = L.template tk<2>(a1), p33 = L.template tk<3>(a2);
        const uint32_t q00 = L.template tk<0>(b0), q01 = L.template tk<1>(b1), q02 = L.template tk<2>(b2), q03 = L.template tk<3>(b3);
        const uint32_t q10 = L.template tk<0>(b1), q11 = L.template tk<1>(b2), q12 = L.template tk<2>(b3), q13 = L.template tk<3>(b0);
        const uint32_t q20 = L.template tk<0>(b2), q21 = L.template tk<1>(b3), q22 = L.template tk<2>(b0), q23 = L.template tk<3>(b1);
        const uint32_t q30 = L.template tk<0>(b3), q31 = L.template tk<1>(b0), q32 = L.template tk<2>(b1), q33 = L.template tk<3>(b2);
        a0 = x3(x3(p00, rotl32(p01, 8), rotl32(p02, 16)), rotl32(p03, 24), rk[4 * r]);
        a1 = x3(x3(p10, rotl32(p11, 8), rotl32(p12, 16)), rotl32(p13, 24), rk[4 * r + 1]);
        a2 = x3(x3(p20, rotl32(p21, 8), rotl32(p22, 16)), rotl32(p23, 24), rk[4 * r + 2]);
        a3 = x3(x3(p30, rotl32(p31, 8), rotl32(p32, 16)), rotl32(p33, 24), rk[4 * r + 3]);
        b0 = x3(x3(q00, rotl32(q01, 8), rotl32(q02, 16)), rotl32(q03, 24), rk[4 * r]);
        b1 = x3(x3(q10, rotl32(q11, 8), rotl32(q12, 16)), rotl32(q13, 24), rk[4 * r + 1]);
        b2 = x3(x3(q20, rotl32(q21, 8), rotl32(q22, 16)), rotl32(q23, 24), rk[4 * r + 2]);
        b3 = x3(x3(q30, rotl32(q31, 8), rotl32(q32, 16)), rotl32(q33, 24), rk[4 * r + 3]);
    }
    auto sbk = [&](uint32_t e, int shift) -> uint32_t { return ((e >> 8) & 0xffu) << shift; };
    out0[0] = (sbk(L.template tk<0>(a0), 0) | sbk(L.template tk<1>(a1), 8) | sbk(L.template tk<2>(a2), 16) | sbk(L.template tk<3>(a3), 24)) ^ rk[56];
    out0[1] = (sbk(L.template tk<0>(a1), 0) | sbk(L.template tk<1>(a2), 8) | sbk(L.template tk<2>(a3), 16) | sbk(L.template tk<3>(a0), 24)) ^ rk[57];
    out0[2] = (sbk(L.template tk<0>(a2), 0) | sbk(L.template tk<1>(a3), 8) | sbk(L.template tk<2>(a0), 16) | sbk(L.template tk<3>(a1), 24)) ^ rk[58];
    out0[3] = (sbk(L.template tk<0>(a3), 0) | sbk(L.template tk<1>(a0), 8) | sbk(L.template tk<2>(a1), 16) | sbk(L.template tk<3>(a2), 24)) ^ rk[59];
    out1[0] = (sbk(L.template tk<0>(b0), 0) | sbk(L.template tk<1>(b1), 8) | sbk(L.template tk<2>(b2), 16) | sbk(L.template tk<3>(b3), 24)) ^ rk[56];
    out1[1] = (sbk(L.template tk<0>(b1), 0) | sbk(L.template tk<1>(b2), 8) | sbk(L.template tk<2>(b3), 16) | sbk(L.template tk<3>(b0), 24)) ^ rk[57];
    out1[2] = (sbk(L.template tk<0>(b2), 0) | sbk(L.template tk<1>(b3), 8) | sbk(L.template tk<2>(b0), 16) | sbk(L.template tk<3>(b1), 24)) ^ rk[58];
    out1[3] = (sbk(L.template tk<0>(b3), 0) | sbk(L.template tk<1>(b0), 8) | sbk(L.template tk<2>(b1), 16) | sbk(L.template tk<3>(b2), 24)) ^ rk[59];
}

struct Drbg {
    Aes256 aes;
    uint32_t counter;
    // seed: len >= 1 bytes, read circularly (prng_get_func.c:108-119)
    template <class LT>
    __device__ __forceinline__ void init(const uint8_t *seed, uint32_t len, const DrbgConst &c, const LT &L)
    {
        uint32_t idx = 0, ctr = 0, key[8];
        for (int i = 0; i < 4; i++) { ctr |= (uint32_t)seed[idx] << (8 * i); if (++idx == len) idx = 0; }
#pragma unroll
        for (int w = 0; w < 8; w++) {
            uint32_t v = 0;
            for (int i = 0; i < 4; i++) { v |= (uint32_t)(seed[idx] ^ c.init_bytes[12 + 4 * w + i]) << (8 * i); if (++idx == len) idx = 0; }
            key[w] = v;
        }
        counter = 3u ^ ctr;
        aes.set_key(key, L);
    }
    // the next 16 bytes of the stream as four little-endian words
    template <class LT>
    __device__ __forceinline__ void block(uint32_t out[4], const LT &L)
    {
        const uint32_t in[4] = { counter, counter, counter, counter };
        counter++;
        aes.encrypt(in, out, L);
    }
};

}  // namespace sc
