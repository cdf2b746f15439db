// Guide table of the CDT sampler: where in the table the search for a uniform 64-bit word x starts.
//
// The reference's halving search (src/utils/sampling/gaussian_cdf.c:536-553) returns, for a monotone table, the largest index b
// with cdf[b] < x (0 if there is none), i.e. the number of entries b >= 1 below x.  Entry b "switches on" at x_b = cdf[b] + 1.
// A guide maps x to a bin, bins ordered like x; guide[bin] = the number of entries that switched on in EARLIER bins, and no bin
// holds more than S switch points.  Then
//        result = guide[bin(x)] + #{ j = 1..S : cdf[guide[bin(x)] + j] < x }
// - one 16-bit LDS read and S independent 64-bit LDS reads, the same for every lane (no data-dependent loop, no divergence),
// instead of log2(size) dependent probes.  The table is padded with S entries of 2^64 - 1 (never below x).  A table's closing run of
// equal entries (hundreds of them for sigma = 215) would put all its switch points into one bin: it is answered by one comparison.
//
// Bins: the body of the distribution is cut linearly (the top LB bits of x); where the top K bits of x are all ones (the tail, where
// a linear bin would hold hundreds of entries) the bin is the floating-point form of y = ~x: the count of leading zeros of y and
// the MB bits below its leading one - the tail probability halves per exponent, the table's entries thin out as x / sigma^2 per
// unit of log-probability, so a bin holds one to three of them at every depth.
#pragma once
#include <cstdint>

constexpr int CDT_PAD = 8;      // entries of 2^64 - 1 behind the table (>= the largest S)

struct CdtGuide {
    uint32_t lb_shift;      // 32 - LB
    uint32_t tail_thr;      // the high word of x at which the tail bins start: ~0 << (32 - K)
    uint32_t K, MB;
    uint32_t lin;           // 1 << LB: index of the first tail bin
    uint32_t total;         // bins
    uint32_t S;             // probes
    // the run of equal entries a table ends in (the long-double sum stops growing some way before 2^64): above sat the answer is jump - the last
    // index of the run; the probe table holds 2^64 - 1 behind the run's first entry.  No such run: sat = 2^64 - 1 (nothing is above it).
    uint32_t sat_hi, sat_lo, jump;
};

__host__ __device__ __forceinline__ uint32_t cdt_guide_bin(uint32_t hi, uint32_t lo, const CdtGuide &g)
{
    const uint32_t yh = ~hi, yl = ~lo;
    const uint32_t h0 = yh ? yh : yl, l0 = yh ? yl : 0u, base = yh ? 0u : 32u;
    const uint32_t lzw = h0 ? (uint32_t)__builtin_clz(h0) : 31u;                  // y = 0 shares the bin of y = 1
    const uint32_t norm = (h0 << lzw) | (lzw ? (l0 >> (32u - lzw)) : 0u);          // leading one at bit 31
    const uint32_t mask = (1u << g.MB) - 1u;
    const uint32_t m = (norm >> (31u - g.MB)) & mask;
    const uint32_t tail = g.lin + ((base + lzw - g.K) << g.MB) + (mask - m);
    return (hi < g.tail_thr) ? (hi >> g.lb_shift) : tail;
}

// tab: the table in LDS followed by >= S entries of ~0; guide: the bins in LDS
template <int S>
__device__ __forceinline__ uint32_t cdt_lookup(uint64_t x, const uint64_t *tab, const uint16_t *guide, const CdtGuide &g)
{
    const uint32_t b0 = guide[cdt_guide_bin((uint32_t)(x >> 32), (uint32_t)x, g)];
    uint32_t r = b0;
    if (S > 0) {
#pragma unroll
        for (int j = 1; j <= S; j++) r += (tab[b0 + j] < x) ? 1u : 0u;
    } else {
        for (uint32_t j = 1; j <= g.S; j++) r += (tab[b0 + j] < x) ? 1u : 0u;
    }
    const uint64_t sat = ((uint64_t)g.sat_hi << 32) | g.sat_lo;
    return (x > sat) ? g.jump : r;
}
