// mlwe_round.h - the two per-coefficient routines of the Dilithium tests (round_alpha, check_norm_inf).  Included by mlwe_kernels.h INSIDE namespace
// sc_mlwe; a file of its own so that the CPU wave emulation under tests/emu/ compiles the same text (it supplies __umulhi and the qualifiers).
// round_alpha, module_lwe.h:52-66,110-136, alpha = 2*gamma_2, Barrett m = floor(2^30/alpha), k = 30.  The reference's
// (a * m) >> 30 in 64 bits equals umulhi(a, 4m) for a < 2^32 and 4m < 2^32, so the whole routine stays in 32-bit instructions
// (a < q < 2^24, every intermediate below 2^32).
__device__ __forceinline__ uint32_t round_alpha(uint32_t a, int32_t &a0, int32_t alpha, int32_t am, int32_t q)
{
    const uint32_t am4 = (uint32_t)am << 2;
    const uint32_t t = __umulhi(a, am4);
    int32_t c = (int32_t)(a - t * (uint32_t)alpha);
    if (alpha <= c) c -= alpha;
    a0 = c;
    if (a0 > (alpha >> 1)) a0 -= alpha;
    uint32_t a1 = a - (uint32_t)a0;
    if (a1 == (uint32_t)(q - 1)) { a1 = 0; a0 -= 1; }
    else {
        uint32_t t2 = __umulhi(a1, am4);
        const int32_t c2 = (int32_t)(a1 - t2 * (uint32_t)alpha);
        if (alpha <= c2) t2++;
        a1 = t2;
    }
    return a1;
}

// check_norm_inf on one canonical value, dilithium.c:1185-1200
__device__ __forceinline__ bool norm_bad(uint32_t v, uint32_t q, uint32_t b)
{
    const uint32_t lower_half = (q - 1) >> 1;
    return (v <= lower_half) ? (v >= b) : (v <= q - b);
}
