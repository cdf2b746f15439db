// tesla_verify_dev.h - the device side of the batched ring-TESLA verifier that is not shared with the signer: z's unpacking with test_rejection, F from the
// signature's digest, w = a z - t c through the public key's shift tables (csrc/tesla.hip sc_hip_tesla_verify queues them; rounding and the oracle are
// ts_round_kernel / ts_oracle_kernel of tesla_sign_dev.h).  Replaces ring_tesla_verify (src/schemes/sig/ring_tesla/ring_tesla.c:922-1040).  Included by
// tesla.hip INSIDE its anonymous namespace behind tesla_sign_dev.h; a header so that a verification call can also be run lane by lane on the CPU
// (tests/test_emu_tesla_sign.py).

// ---- verify ---------------------------------------------------------------------------------------------------------------

// z from the signature (ring_tesla.c:984-996) + test_rejection; one wavefront per signature
__device__ __forceinline__ void ts_unpack_body(uint32_t block, uint32_t batch, const uint8_t *sigs, size_t sig_stride, const uint32_t *siglens, int32_t *z, uint32_t *verdict, const TsP &P)
{
    const int lane = threadIdx.x & 63, n = P.n;
    const uint32_t item = block * 4 + (threadIdx.x >> 6);
    if (item >= batch) return;
    const uint32_t zb = ((uint32_t)n * P.q_bits + 7) / 8;
    if (siglens[item] < zb + 64) { for (int j = lane; j < n; j += 64) z[(size_t)item * n + j] = 0; if (lane == 0) verdict[item] = 1; return; }
    const uint8_t *s = sigs + (size_t)item * sig_stride;
    const int32_t th = P.b - P.u;
    int32_t bad = 0;
    for (int j = lane; j < n; j += 64) {
        uint32_t val = 0;
        if (P.q_bits == 24) val = ((uint32_t)s[3 * j] << 16) | ((uint32_t)s[3 * j + 1] << 8) | s[3 * j + 2];
        else for (int b = 0; b < P.q_bits; b++) { const uint32_t bp = (uint32_t)j * P.q_bits + b; val = (val << 1) | ((s[bp >> 3] >> (7 - (bp & 7))) & 1u); }
        const int32_t zz = (int32_t)(val << (32 - P.q_bits)) >> (32 - P.q_bits);
        z[(size_t)item * n + j] = zz;
        bad |= (zz < -th || zz > th) ? 1 : 0;
    }
    bad = wave_or(bad);
    if (bad && lane == 0) verdict[item] = 1;
}
// The verifier's two readers of the signature as ONE launch: the first workgroups (their first HT lanes, a lane per signature) turn the digest into the
// challenge's positions (F: twelve Salsa20 blocks in a row), the rest unpack z under them.
__global__ __launch_bounds__(256) void ts_unpack_f_kernel(uint32_t batch, const uint8_t *sigs, size_t sig_stride, const uint32_t *siglens, int32_t *z, uint32_t *verdict, TsP P,
                                                         uint32_t md_off, int32_t *cidx)
{
    const uint32_t nbf = (batch + HT - 1) / HT;
    if (blockIdx.x >= nbf) ts_unpack_body(blockIdx.x - nbf, batch, sigs, sig_stride, siglens, z, verdict, P);
    else if (threadIdx.x < HT) ts_f_body(blockIdx.x, nullptr, batch, nullptr, sigs, sig_stride, md_off, 1, cidx, P, nullptr);
}

// w = normalize(prod - t*c)  (ring_tesla.c:1014-1026); one wavefront per signature, in place on the products.  The two public polynomials sit in LDS as
// SHIFT TABLES (sparse_shift above; built when the key is loaded, copied in 16-byte pieces): a term of a sparse product is one LDS read at an address the
// lane's eight coefficients share up to a constant, and one addition.  Round 3 had the plain polynomials in LDS (they were fetched from global memory
// before that: 173 us of a 330 us call) under persistent workgroups, seven vector instructions per term: 105 us per 16384 signatures, 42 % of a
// verification call; round 4: a wavefront per signature, its two rows requested up front: 105 -> 33 us (ring-TESLA-I verification 66 -> 88-91 M/s at batch 16384).
__global__ __launch_bounds__(256) void ts_vw_kernel(uint32_t batch, int32_t *w1, int32_t *w2, const int32_t *ttab, const int32_t *cidx, TsP P, sc::ModK mk)
{
    extern __shared__ __attribute__((aligned(16))) int32_t tl[];      // tables of t1 | t2: 2n entries each
    const int lane = threadIdx.x & 63, n = P.n, C = n / 64;
    for (int i = threadIdx.x; i < 4 * n / 4; i += blockDim.x) reinterpret_cast<uint4 *>(tl)[i] = reinterpret_cast<const uint4 *>(ttab)[i];
    __syncthreads();
    const uint32_t item = __builtin_amdgcn_readfirstlane(blockIdx.x * 4 + (threadIdx.x >> 6));
    if (item >= batch) return;
    const int pos = lane < P.omega ? cidx[(size_t)item * P.omega + lane] : 0;
    int32_t r1[TS_MAXC], r2[TS_MAXC], sp[TS_MAXC];
#pragma unroll
    for (int c = 0; c < TS_MAXC; c++) {
        const size_t k = (size_t)item * n + lane + 64 * (c < C ? c : 0);
        r1[c] = w1[k]; r2[c] = w2[k];
    }
    // any int32 -> [0, q) by a Barrett step (modarith.h) instead of the division normalize_32 compiles to
    sparse_shift(sp, tl, pos, P.omega, n, lane);
#pragma unroll
    for (int c = 0; c < TS_MAXC; c++) if (c < C) w1[(size_t)item * n + lane + 64 * c] = (int32_t)sc::csub(sc::reduce_i32(r1[c] - sp[c], mk), mk.q);
    sparse_shift(sp, tl + 2 * n, pos, P.omega, n, lane);
#pragma unroll
    for (int c = 0; c < TS_MAXC; c++) if (c < C) w2[(size_t)item * n + lane + 64 * c] = (int32_t)sc::csub(sc::reduce_i32(r2[c] - sp[c], mk), mk.q);
}
