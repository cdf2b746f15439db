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
};
using AesLds = AesLdsT<1>;
using AesLdsWide = AesLdsT<32>;

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
            const uint32_t t0 = x3(x3(L.t(s0 & 0xff), rotl32(L.t((s1 >> 8) & 0xff), 8), rotl32(L.t((s2 >> 16) & 0xff), 16)), rotl32(L.t(s3 >> 24), 24), rk[4 * r]);
            const uint32_t t1 = x3(x3(L.t(s1 & 0xff), rotl32(L.t((s2 >> 8) & 0xff), 8), rotl32(L.t((s3 >> 16) & 0xff), 16)), rotl32(L.t(s0 >> 24), 24), rk[4 * r + 1]);
            const uint32_t t2 = x3(x3(L.t(s2 & 0xff), rotl32(L.t((s3 >> 8) & 0xff), 8), rotl32(L.t((s0 >> 16) & 0xff), 16)), rotl32(L.t(s1 >> 24), 24), rk[4 * r + 2]);
            const uint32_t t3 = x3(x3(L.t(s3 & 0xff), rotl32(L.t((s0 >> 8) & 0xff), 8), rotl32(L.t((s1 >> 16) & 0xff), 16)), rotl32(L.t(s2 >> 24), 24), rk[4 * r + 3]);
            s0 = t0; s1 = t1; s2 = t2; s3 = t3;
        }
        out[0] = (L.sb(s0 & 0xff) | (L.sb((s1 >> 8) & 0xff) << 8) | (L.sb((s2 >> 16) & 0xff) << 16) | (L.sb(s3 >> 24) << 24)) ^ rk[56];
        out[1] = (L.sb(s1 & 0xff) | (L.sb((s2 >> 8) & 0xff) << 8) | (L.sb((s3 >> 16) & 0xff) << 16) | (L.sb(s0 >> 24) << 24)) ^ rk[57];
        out[2] = (L.sb(s2 & 0xff) | (L.sb((s3 >> 8) & 0xff) << 8) | (L.sb((s0 >> 16) & 0xff) << 16) | (L.sb(s1 >> 24) << 24)) ^ rk[58];
        out[3] = (L.sb(s3 & 0xff) | (L.sb((s0 >> 8) & 0xff) << 8) | (L.sb((s1 >> 16) & 0xff) << 16) | (L.sb(s2 >> 24) << 24)) ^ rk[59];
    }
};

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
