// bliss_verify_dev.h - the device side of the batched BLISS-B verifier that is not shared with the signer (raw signature coding): the signature's
// unpacking, the norm tests, and the recovery of z from a t + c (csrc/bliss.hip bl_verify_core queues them; the product a t is the NTT kernel, the oracle
// bl_oracle1 / bl_oracle2_kernel of bliss_sign_dev.h).  Replaces bliss_b_verify (src/schemes/sig/bliss_b/bliss_b.c:1583-1740).  Included by bliss.hip INSIDE
// its anonymous namespace behind bliss_sign_dev.h; a header so that a verification call can also be run lane by lane on the CPU (tests/test_emu_bliss_sign.py).

// ---- verify -----------------------------------------------------------------------------------------------------------

// unpack t (z1_bits, signed), z (z2_bits, signed), c_idx (n_bits): bliss_b.c:1641-1657.  A workgroup per signature (walking the batch): the row is copied
// into LDS in whole words - the bytes a field touches were fetched from global memory one load each, up to thirteen per thread, and the kappa index fields had a
// workgroup of their own with 23 of 256 lanes at work: 47 us of a 254 us verification call at batch 16384 - and a thread cuts FOUR consecutive fields of t
// or of z (one 16-byte / 8-byte store) or one index out of it.  A field's width and bit offset are the same for every signature.
__global__ __launch_bounds__(256) void bl_unpack_kernel(const uint8_t *sigs, size_t sig_stride, const uint32_t *siglens, uint32_t count, int32_t *t, int16_t *z, int32_t *cidx,
                                                        uint32_t *verdict, BlissP P)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t row[];      // the signature's bytes (+ 16 of slack behind them)
    const int n4 = P.n / 4, per = 2 * n4 + P.kappa;
    const uint32_t need_bits = (uint32_t)(P.n * (P.z1_bits + P.z2_bits) + P.kappa * P.n_bits), need_bytes = (need_bits + 7) / 8;
    for (uint32_t item = blockIdx.x; item < count; item += gridDim.x) {
        if (siglens[item] * 8 < need_bits || need_bytes > sig_stride) {
            if (threadIdx.x == 0) verdict[item] = 1;
            for (int e = threadIdx.x; e < n4; e += blockDim.x) *reinterpret_cast<int4 *>(t + (size_t)item * P.n + 4 * e) = make_int4(0, 0, 0, 0);
            continue;
        }
        const uint8_t *s = sigs + (size_t)item * sig_stride;
        __syncthreads();                                               // (the previous signature's readers are done with the row)
        if ((reinterpret_cast<uintptr_t>(s) & 3) == 0) for (uint32_t i = threadIdx.x; i < (need_bytes + 3) / 4 && 4 * i + 4 <= sig_stride; i += blockDim.x) reinterpret_cast<uint32_t *>(row)[i] = reinterpret_cast<const uint32_t *>(s)[i];
        else for (uint32_t i = threadIdx.x; i < need_bytes; i += blockDim.x) row[i] = s[i];
        if ((reinterpret_cast<uintptr_t>(s) & 3) == 0) for (uint32_t i = (uint32_t)(sig_stride / 4) * 4 + threadIdx.x; i < need_bytes; i += blockDim.x) row[i] = s[i];   // a row whose pitch is not a multiple of four: its last bytes
        __syncthreads();
        for (int e = threadIdx.x; e < per; e += blockDim.x) {
            const int sect = e < n4 ? 0 : e < 2 * n4 ? 1 : 2;
            const int bits = sect == 0 ? P.z1_bits : sect == 1 ? P.z2_bits : P.n_bits, nf = sect == 2 ? 1 : 4;
            const int f0 = sect == 0 ? 4 * e : sect == 1 ? 4 * (e - n4) : e - 2 * n4;                 // first field of this thread within its section
            const uint32_t off = sect == 0 ? (uint32_t)f0 * bits : sect == 1 ? (uint32_t)(P.n * P.z1_bits + f0 * bits) : (uint32_t)(P.n * (P.z1_bits + P.z2_bits) + f0 * bits);
            const uint32_t first = off >> 3, last = (off + (uint32_t)(nf * bits) - 1) >> 3;              // 4 * bits + 7 <= 103 bits: up to 13 bytes
            // the window as two 64-bit halves, most significant byte first, left-aligned at the first field's first bit
            uint64_t hi = 0, lo = 0;
            for (uint32_t b = first; b <= last; b++) { const uint32_t k = b - first; if (k < 8) hi |= (uint64_t)row[b] << (56 - 8 * k); else lo |= (uint64_t)row[b] << (56 - 8 * (k - 8)); }
            const uint32_t skew = off & 7;
            if (skew) { hi = (hi << skew) | (lo >> (64 - skew)); lo <<= skew; }
            int32_t val[4];
#pragma unroll
            for (int f = 0; f < 4; f++) {
                const uint32_t raw = (uint32_t)(hi >> (64 - bits));
                val[f] = sect == 2 ? (int32_t)raw : ((int32_t)(raw << (32 - bits)) >> (32 - bits));
                hi = (hi << bits) | (lo >> (64 - bits)); lo <<= bits;
            }
            if (sect == 0) *reinterpret_cast<int4 *>(t + (size_t)item * P.n + f0) = make_int4(val[0], val[1], val[2], val[3]);
            else if (sect == 1) {
                const uint32_t w0 = ((uint32_t)val[0] & 0xFFFFu) | ((uint32_t)val[1] << 16), w1 = ((uint32_t)val[2] & 0xFFFFu) | ((uint32_t)val[3] << 16);
                *reinterpret_cast<uint2 *>(z + (size_t)item * P.n + f0) = make_uint2(w0, w1);
            } else cidx[(size_t)item * P.kappa + f0] = val[0];
        }
    }
}

// check_norms on the received (t, z): one wavefront per signature
__global__ __launch_bounds__(256) void bl_vnorm_kernel(const int32_t *t, const int16_t *z, uint32_t count, uint32_t *verdict, BlissP P)
{
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, n = P.n;
    const uint32_t item = blockIdx.x * 4 + wid;
    if (item >= count) return;
    int32_t maxt = 0, maxz = 0, nt = 0, nz = 0;
    for (int j = lane; j < n; j += 64) {
        const int32_t tt = t[(size_t)item * n + j], zz = z[(size_t)item * n + j];
        maxt = max(maxt, max(tt, -tt)); maxz = max(maxz, max(zz, -zz));
        nt += tt * tt; nz += zz * zz;
    }
    maxt = wave_max(maxt); maxz = wave_max(maxz); nt = wave_sum(nt); nz = wave_sum(nz);
    if (lane == 0 && (maxt > P.b_inf || (maxz << P.d) > P.b_inf || nt + (nz << (2 * P.d)) > P.b_l2)) verdict[item] = 1;
}

// v += (v & 1) q ; then the kappa single-coefficient corrections, in order (bliss_b.c:1690-1698)
__global__ void bl_vparity_kernel(int32_t *v, size_t len, int32_t q)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < len) { const int32_t x = v[i]; v[i] = x + (x & 1) * q; }
}
__global__ void bl_vcidx_kernel(int32_t *v, const int32_t *cidx, uint32_t count, BlissP P)
{
    const uint32_t item = blockIdx.x * blockDim.x + threadIdx.x;
    if (item >= count) return;
    const int32_t two_q = 2 * P.q;
    for (int k = 0; k < P.kappa; k++) {
        int32_t *p = v + (size_t)item * P.n + (cidx[(size_t)item * P.kappa + k] & (P.n - 1));     // (a signature too short to hold its indices leaves them unset)
        *p = barrett((int64_t)*p + P.q, (1 << 30) / two_q, two_q);
    }
}
// z <- ((v + 2^(d-1)) >> d) + z mod p  (bliss_b.c:1700-1706)
__global__ void bl_vz_kernel(const int32_t *v, int16_t *z, size_t len, BlissP P)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= len) return;
    int32_t tmp = ((v[i] + (1 << (P.d - 1))) >> P.d) + z[i];
    tmp -= (tmp >= P.p) * P.p;
    tmp += (tmp < 0) * P.p;
    z[i] = (int16_t)tmp;
}

// bl_vparity_kernel + bl_vcidx_kernel + bl_vz_kernel as one pass, a wavefront per signature (n = 64 * NT coefficients, NT of them per lane in registers):
// v += (v & 1) q, the kappa single-coefficient corrections in order (the lane that owns the position picks the value out of its registers, corrects it and
// puts it back), z <- ((v + 2^(d-1)) >> d) + z mod p.  The product is read once and not written back - nothing after this reads it - where the three kernels
// moved it four times (51 -> ~25 us of a 256 us verification call at batch 16384).
template <int NT>
__global__ __launch_bounds__(256) void bl_vfix_kernel(const int32_t *v, const int32_t *cidx, int16_t *z, uint32_t count, BlissP P)
{
    const int lane = threadIdx.x & 63;
    const uint32_t item = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (item >= count) return;
    const int32_t *vi = v + (size_t)item * P.n;
    int32_t x[NT];
#pragma unroll
    for (int t = 0; t < NT; t++) { const int32_t a = vi[lane + 64 * t]; x[t] = a + (a & 1) * P.q; }
    const int32_t two_q = 2 * P.q;
    const int64_t mu = (1 << 30) / two_q;
    // the kappa <= 64 positions in one vector register, read with v_readlane: a load per step inside the loop was a chain of kappa dependent round trips
    const int my_pos = lane < P.kappa ? cidx[(size_t)item * P.kappa + lane] : 0;
    for (int k = 0; k < P.kappa; k++) {
        const int pos = __builtin_amdgcn_readlane(my_pos, k) & (P.n - 1);  // (a signature too short to hold its indices leaves them unset)
        if (lane == (pos & 63)) {
            const int t0 = pos >> 6;
            int32_t sel = x[0];
#pragma unroll
            for (int t = 1; t < NT; t++) sel = t == t0 ? x[t] : sel;
            const int32_t fixed = barrett((int64_t)sel + P.q, mu, two_q);
#pragma unroll
            for (int t = 0; t < NT; t++) x[t] = t == t0 ? fixed : x[t];
        }
    }
    int16_t *zi = z + (size_t)item * P.n;
#pragma unroll
    for (int t = 0; t < NT; t++) {
        int32_t tmp = ((x[t] + (1 << (P.d - 1))) >> P.d) + zi[lane + 64 * t];
        tmp -= (tmp >= P.p) * P.p;
        tmp += (tmp < 0) * P.p;
        zi[lane + 64 * t] = (int16_t)tmp;
    }
}
