// falcon_verify_dev.h - the device side of the batched Falcon verifier that is not shared with the signer: the signature's unpacking (as a role of the hash
// kernel's launch) and the finishing kernel - s1 = c - s2 h centred, the float norm test (csrc/falcon.hip fal_verify_core queues them; the product s2 h is
// the NTT kernel, the hash fal_hash2_body of falcon_sign_dev.h).  Replaces falcon_sig_verify (src/schemes/sig/falcon/falcon.c:955-1040).  Included by
// falcon.hip INSIDE its anonymous namespace behind falcon_sign_dev.h; a header so that a verification call can also be run lane by lane on the CPU
// (tests/test_emu_falcon_sign.py).

// ---- verify (falcon.c:988-1024) -----------------------------------------------------------------------------------------
// bits: field width (Falcon q_bits - 2; ENS / DLP signatures q_bits); centre == 0: the signed field is taken as it is (ens_dlp_sig.c:1617-1618)
__device__ __forceinline__ void fal_unpack_body(size_t block, uint32_t batch, const uint8_t *__restrict__ sigs, size_t sig_stride, const uint32_t *__restrict__ siglens, int32_t *__restrict__ s2,
                                                uint32_t *__restrict__ verdict, const FalP &P, int bits, int centre)
{
    // one lane per PAIR of coefficients; with the 12-bit coding of both parameter sets a pair is exactly three bytes
    const size_t tt = block * blockDim.x + threadIdx.x;
    const int hn = P.n / 2;
    const uint32_t b = (uint32_t)(tt >> sc::pow2_shift(hn)); const int e = (int)(tt & (size_t)(hn - 1)) * 2;     // n is a power of two
    if (b >= batch) return;
    int32_t *o = s2 + (size_t)b * P.n + e;
    if ((size_t)siglens[b] * 8 < (size_t)P.n * bits || siglens[b] > sig_stride) { if (e == 0) verdict[b] = 1; o[0] = o[1] = 0; return; }   // short, or longer than its row
    const uint8_t *s = sigs + (size_t)b * sig_stride;
    if (bits == 12 && centre) {
        const uint8_t *p = s + (e / 2) * 3;
        const uint32_t w = ((uint32_t)p[0] << 16) | ((uint32_t)p[1] << 8) | p[2];
        o[0] = centre_mod((int32_t)((w >> 12) << 20) >> 20, P.q);
        o[1] = centre_mod((int32_t)((w & 0xFFFu) << 20) >> 20, P.q);
        return;
    }
    for (int h = 0; h < 2; h++) {
        uint32_t val = 0;
        for (int k = 0; k < bits; k++) { const uint32_t bp = (uint32_t)(e + h) * bits + k; val = (val << 1) | ((s[bp >> 3] >> (7 - (bp & 7))) & 1u); }
        const int32_t sv = (int32_t)(val << (32 - bits)) >> (32 - bits);
        o[h] = centre ? centre_mod(sv, P.q) : sv;
    }
}
__global__ void fal_unpack_kernel(uint32_t batch, const uint8_t *__restrict__ sigs, size_t sig_stride, const uint32_t *__restrict__ siglens, int32_t *__restrict__ s2,
                                  uint32_t *__restrict__ verdict, FalP P, int bits, int centre)
{ fal_unpack_body(blockIdx.x, batch, sigs, sig_stride, siglens, s2, verdict, P, bits, centre); }

// The verifiers' hash to the ring (two lanes per message) and the signature's unpacking as ONE launch of 256-thread workgroups: the first workgroups hash
// (their first HT lanes: the chain of 25 permutations starts with the launch), the rest unpack under it.  The two are independent; the unpacking used to be
// a launch of its own in front of the chain (17 us of a 272 us call at batch 8192).  (A second stream for the product as well was measured and dropped:
// DESIGN 4.8 - the product's wavefronts slow the chain; the unpacking is light.)
__global__ __launch_bounds__(256) void fal_hash2_unpack_kernel(uint32_t batch, const uint8_t *__restrict__ msgs, size_t msg_stride, const uint32_t *__restrict__ mlens, uint32_t fixed_len,
                                                              int32_t *__restrict__ c, FalP P, int dlp, const uint8_t *__restrict__ sigs, size_t sig_stride,
                                                              const uint32_t *__restrict__ siglens, int32_t *__restrict__ s2, uint32_t *__restrict__ verdict, int bits, int centre)
{
    const uint32_t nbh = (2 * batch + HT - 1) / HT;
    if (blockIdx.x >= nbh) fal_unpack_body(blockIdx.x - nbh, batch, sigs, sig_stride, siglens, s2, verdict, P, bits, centre);
    else if (threadIdx.x < HT) fal_hash2_body(blockIdx.x, batch, msgs, msg_stride, mlens, fixed_len, c, P, dlp);
}

// s1 = centre(c - s2*h) ; norm test.  One wavefront per signature.
__global__ __launch_bounds__(64 * FW) void fal_vfinish_kernel(uint32_t batch, const int32_t *__restrict__ c, const int32_t *__restrict__ prod, const int32_t *__restrict__ s2,
                                                              uint32_t *__restrict__ verdict, FalP P)
{
    extern __shared__ __attribute__((aligned(16))) int32_t fin_lds[];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, n = P.n;
    const uint32_t b = blockIdx.x * FW + wv;
    if (b >= batch) return;
    int32_t *sq = fin_lds + (size_t)wv * n;
    const int32_t *cb = c + (size_t)b * n, *pb = prod + (size_t)b * n, *a2 = s2 + (size_t)b * n;
    for (int i = lane; i < n; i += 64) {
        const int32_t x = centre_mod(cb[i] - pb[i], P.q), y = a2[i];
        sq[i] = x * x + y * y;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    if (lane == 0 && norm_fails_lds(sq, n, P.bd)) verdict[b] = 1;
}
