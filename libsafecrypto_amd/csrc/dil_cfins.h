// dil_cfins.h - Dilithium's finishing kernel without transforms.  Included by dilithium.hip (inside its unnamed namespace, after DilP / DilFin / DilNtt, the
// round-state indices and the helpers it uses) and, as the same text, by the CPU wave emulation tests/emu/emu_dil_cfins.cpp, which runs it lane by lane
// against the oracle's attempts.
// ---- the finishing kernel WITHOUT transforms (one key for the batch; round 4) -------------------------------------------------------------------
// c has `weight` coefficients +-1, the key vectors are small (|s1|, |s2| <= eta, |t0| <= 2^(d-1)): c s is `weight` shifted copies of s added up.  With s as a
// SHIFT TABLE in LDS - U[m] = -s[m] (m < n), s[m - n] (n <= m < 2n), -s[m - 2n] (2n <= m < 3n): (x^p s)[j] = U[n + j - p] and -(x^p s)[j] = U[2n + j - p],
// the wrap's sign and the term's sign both inside the table - a term of c is one LDS read and one 32-bit addition per THREE rows: a word holds three rows'
// entries as 10-bit fields, each biased by B = max |s| so that no field ever borrows from its neighbour (the sum of `weight` entries stays below 2^10:
// checked when the key is loaded; the bias comes off once at the end).  t0 (14 bits) is two rows a word, added up in 32 bits.  No NTT(c), no products, no
// inverse transforms, no exchange images: the r stage is weight x ceil(k / 3) x 4 reads and additions per lane, nothing in it waits for anything but LDS,
// the kernel needs half the registers and three wavefronts share a SIMD.  Same integers as dil_cfin_kernel computes modulo q, same tests in the same order,
// the same packing code: the same signatures.  A lane owns the coefficients j_c = 64 c + 16 (lane & 3) + (lane >> 2), c = 0..3 - word 64 c + lane of a
// row of w in its 16-byte-piece layout (DilNtt::w_chunked), so a row of w is four fully coalesced loads (tests/test_dilithium_fin_tables.py walks the
// tables with this mapping on the CPU).
struct DilFinS {
    const uint32_t *tabs; uint32_t bias;      // us2 [ceil(k/3)][3n] | us1 [ceil(l/3)][3n] (10-bit fields) | ut0 [ceil(k/2)][3n] (16-bit pairs)
    // PERKEY (a key per message): the keys in the coefficient domain as the caller holds them, s1 [item][l][n], s2 [item][k][n] (canonical or centred), t0
    // [item][k][n] (canonical: dil_t0_kernel); every wavefront builds its message's tables in LDS itself.  *refused is OR-ed with 1 when a key's
    // coefficients exceed `bias` (the fields could run into each other): the message is then neither signed nor queued, and the host runs the call again
    // with the transform kernel
    const int32_t *s1, *s2, *t0; uint32_t *refused;
};
constexpr int CSW = 2;                        // words of 10-bit fields per coefficient: k, l <= 6
constexpr int CSP = 3;                        // row pairs of t0

static inline size_t cfins_table_words(const DilP &P) { return (size_t)((P.k + 2) / 3 + (P.l + 2) / 3 + (P.k + 1) / 2) * 3 * N; }
static inline size_t cfins_wave_words(const DilP &P) { return 256u * (size_t)P.l + 128u + 64u + 64u * (size_t)P.k + 64u; }

// PERKEY: two wavefronts a workgroup (each holds its own 21 KB of tables: two workgroups a CU), launched as (messages + 1) / 2 workgroups of 128 threads
template <int K, int L, bool PERKEY = false>
__global__ __launch_bounds__(256) void dil_cfins_kernel(DilNtt a, DilFin f, DilFinS g)
{
    extern __shared__ __attribute__((aligned(16))) uint32_t fin_lds[];
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const DilP &P = f.P;
    constexpr int l = L, kk = K, kw = (K + 2) / 3, lw = (L + 2) / 3, kp = (K + 1) / 2;
    static_assert(kw <= CSW && lw <= CSW && kp <= CSP, "row counts");
    const int32_t q = P.q;
    constexpr int WPB = PERKEY ? 2 : 4;                                 // wavefronts per workgroup
    constexpr uint32_t TW = (uint32_t)(kw + lw + kp) * 3 * N, WW = 256u * l + 128u + 64u + 64u * kk + 64u;
    uint32_t *T2 = fin_lds + 4 + (PERKEY ? (size_t)wid * (TW + WW) : 0), *T1 = T2 + kw * 3 * N, *T0 = T1 + lw * 3 * N;
    uint32_t *wbase = PERKEY ? T2 + TW : T2 + TW + (size_t)wid * WW;
    uint32_t *zl = wbase;                                               // z of the current attempt, natural order [l][n]
    uint32_t *tw = zl + 256 * l;                                        // signature tail, big-endian words
    uint32_t *cb = tw + 128;                                            // c as n signed bytes
    uint8_t *hl = reinterpret_cast<uint8_t *>(cb + 64);                 // hints [k][n]
    uint32_t *cpos = cb + 64 + 64 * kk;                                 // c's terms: n - p for a +1 at p, 2n - p for a -1
    if constexpr (!PERKEY) {
        for (uint32_t i = threadIdx.x; i < TW / 4; i += 256) reinterpret_cast<uint4 *>(T2)[i] = reinterpret_cast<const uint4 *>(g.tabs)[i];
    }
    if (threadIdx.x == 0) fin_lds[0] = 0;
    __syncthreads();
    const uint32_t nact = f.rs[RS_NACT], S = f.rs[RS_S];
    const uint32_t aidx = blockIdx.x * WPB + wid;
    const bool live = aidx < nact;                                      // wave-uniform
    int found = -1;
    uint32_t item = 0;
    if (live) {
        item = f.active ? f.active[aidx] : aidx;
        const int jb = 16 * (lane & 3) + (lane >> 2);                   // the lane's coefficient inside every block of 64
        const uint32_t *c32 = reinterpret_cast<const uint32_t *>(f.c8) + (size_t)aidx * S * 64 + lane;
        uint32_t creg = c32[0];
        if constexpr (PERKEY) {
            // the message's key -> the wavefront's tables (what fin_tables in dilithium.hip builds on the host for the handle's key)
            const int32_t B = (int32_t)g.bias, hq = (q - 1) >> 1;
            bool wide = false;
            auto cen = [&](int32_t x) { x -= (x > hq) ? q : 0; x += (x < -hq) ? q : 0; return x; };
            auto build10 = [&](uint32_t *T, const int32_t *src, auto rows_) {
                constexpr int rows = decltype(rows_)::value;
#pragma unroll
                for (int r = 0; r < (rows + 2) / 3; r++)
#pragma unroll
                    for (int c = 0; c < 4; c++) {
                        const int j = lane + 64 * c;
                        uint32_t neg = 0, pos = 0;
#pragma unroll
                        for (int fi = 0; fi < 3; fi++)
                            if (3 * r + fi < rows) {
                                const int32_t v = cen(src[(3 * r + fi) * N + j]);
                                wide |= v > B || v < -B;
                                neg |= (uint32_t)(B - v) << (10 * fi); pos |= (uint32_t)(B + v) << (10 * fi);
                            }
                        T[r * 3 * N + j] = neg; T[r * 3 * N + N + j] = pos; T[r * 3 * N + 2 * N + j] = neg;
                    }
            };
            build10(T2, g.s2 + (size_t)item * K * N, std::integral_constant<int, K>{});
            build10(T1, g.s1 + (size_t)item * L * N, std::integral_constant<int, L>{});
            const int32_t *t0 = g.t0 + (size_t)item * K * N;
#pragma unroll
            for (int pp = 0; pp < kp; pp++)
#pragma unroll
                for (int c = 0; c < 4; c++) {
                    const int j = lane + 64 * c;
                    const int32_t v0 = cen(t0[2 * pp * N + j]), v1 = 2 * pp + 1 < K ? cen(t0[(2 * pp + 1) * N + j]) : 0;
                    wide |= v0 > 32767 || v0 < -32767 || v1 > 32767 || v1 < -32767;
                    const uint32_t neg = (uint32_t)(uint16_t)(int16_t)(-v0) | ((uint32_t)(uint16_t)(int16_t)(-v1) << 16);
                    const uint32_t pos = (uint32_t)(uint16_t)(int16_t)v0 | ((uint32_t)(uint16_t)(int16_t)v1 << 16);
                    T0[pp * 3 * N + j] = neg; T0[pp * 3 * N + N + j] = pos; T0[pp * 3 * N + 2 * N + j] = neg;
                }
            wave_lds_fence();
            if (__ballot(wide) != 0) { found = -2; if (lane == 0) atomicOr(g.refused, 1u); }
        }
        for (uint32_t at = 0; at < S && found == -1; at++) {
            const uint32_t slot = aidx * S + at;
            // the attempt's w (k rows, four words a lane) is requested before anything is computed
            int32_t wv[3 * CSW][4];
#pragma unroll
            for (int i = 0; i < 3 * CSW; i++)
#pragma unroll
                for (int c = 0; c < 4; c++) wv[i][c] = i < kk ? a.w[((size_t)slot * kk + i) * N + 64 * c + lane] : 0;
            const uint32_t cw = creg;
            cb[lane] = cw;
            if (at + 1 < S) creg = c32[(size_t)(at + 1) * 64];              // the next attempt's c, should this one fail
            // c's terms: the lane's word holds c[4 lane .. 4 lane + 3] as signed bytes
            uint32_t mine_t = 0;
#pragma unroll
            for (int e = 0; e < 4; e++) mine_t += (uint8_t)(cw >> (8 * e)) != 0;
            uint32_t incl_t = mine_t;
            for (int o = 1; o < 64; o <<= 1) { const uint32_t t_ = (uint32_t)__shfl_up((int)incl_t, o); if (lane >= o) incl_t += t_; }
            uint32_t nt = (uint32_t)__builtin_amdgcn_readlane((int)incl_t, 63);      // = weight (a scalar: the loops below are uniform)
            nt = nt < 64u ? nt : 64u;
            {
                uint32_t o_ = incl_t - mine_t;
#pragma unroll
                for (int e = 0; e < 4; e++) {
                    const int32_t b = (int32_t)(int8_t)(cw >> (8 * e));
                    if (b != 0 && o_ < 64u) cpos[o_++] = (uint32_t)((b > 0 ? N : 2 * N) - (4 * lane + e));
                }
            }
            wave_lds_fence();
            const int my_base = (uint32_t)lane < nt ? (int)cpos[lane] : N;       // weight <= 64: term t in lane t
            const uint32_t off = nt * g.bias;
            // sums of the shifted rows of a table of 10-bit fields: acc[r][c], three rows a word
            // (four terms' reads are issued before the first addition: the loop is a chain of LDS round trips otherwise; weight = 60 leaves no remainder)
            auto sparse = [&](const uint32_t *T, auto nw_, uint32_t (&acc)[CSW][4]) {
                constexpr int nw = decltype(nw_)::value;
#pragma unroll
                for (int r = 0; r < CSW; r++)
#pragma unroll
                    for (int c = 0; c < 4; c++) acc[r][c] = 0;
                uint32_t t = 0;
                for (; t + 4 <= nt; t += 4) {
                    uint32_t v[4][nw][4];
#pragma unroll
                    for (int u = 0; u < 4; u++) {
                        const uint32_t *row = T + (jb + __builtin_amdgcn_readlane(my_base, (int)(t + u)));
#pragma unroll
                        for (int r = 0; r < nw; r++)
#pragma unroll
                            for (int c = 0; c < 4; c++) v[u][r][c] = row[r * 3 * N + 64 * c];
                    }
#pragma unroll
                    for (int u = 0; u < 4; u++)
#pragma unroll
                        for (int r = 0; r < nw; r++)
#pragma unroll
                            for (int c = 0; c < 4; c++) acc[r][c] += v[u][r][c];
                }
                for (; t < nt; t++) {
                    const uint32_t *row = T + (jb + __builtin_amdgcn_readlane(my_base, (int)t));
#pragma unroll
                    for (int r = 0; r < nw; r++)
#pragma unroll
                        for (int c = 0; c < 4; c++) acc[r][c] += row[r * 3 * N + 64 * c];
                }
            };
            auto field = [&](const uint32_t (&acc)[CSW][4], int i, int c) -> int32_t {          // row i's sum with the bias taken off
                const uint32_t w = acc[i / 3][c];
                const uint32_t raw = (i % 3) == 0 ? (w & 1023u) : (i % 3) == 1 ? ((w >> 10) & 1023u) : (w >> 20);
                return (int32_t)raw - (int32_t)off;
            };
            uint32_t acc[CSW][4];
            bool bad = false;
            // ---- r = w - c s2; ||LowBits(r)||inf < gamma_2 - beta (dilithium.c:1671-1700)
            sparse(T2, std::integral_constant<int, kw>{}, acc);
            int32_t rr[3 * CSW][4];
#pragma unroll
            for (int i = 0; i < 3 * CSW; i++)
#pragma unroll
                for (int c = 0; c < 4; c++) {
                    int32_t v = wv[i][c] - field(acc, i, c);                  // w canonical, |c s2| small: one correction either way
                    v += (v >> 31) & q; v -= (v >= q) ? q : 0;
                    rr[i][c] = v;
                    int32_t a0;
                    round_alpha((uint32_t)v, a0, a.alpha, a.am, q);
                    const uint32_t lo = a0 < 0 ? (uint32_t)(a0 + q) : (uint32_t)a0;       // low_order_bits
                    bad |= (i < kk) && norm_bad(lo, (uint32_t)q, (uint32_t)a.gamma2_beta);
                }
            bool ok = __ballot(bad) == 0;
            if (ok) {
                // ---- z = y + c s1, normalised; ||z||inf < gamma_1 - beta (dilithium.c:1655-1669)
                int32_t yv[3 * CSW][4];
#pragma unroll
                for (int i = 0; i < 3 * CSW; i++)
#pragma unroll
                    for (int c = 0; c < 4; c++) yv[i][c] = i < l ? a.y[((size_t)slot * l + i) * N + 64 * c + jb] : 0;
                sparse(T1, std::integral_constant<int, lw>{}, acc);
#pragma unroll
                for (int i = 0; i < 3 * CSW; i++)
#pragma unroll
                    for (int c = 0; c < 4; c++) {
                        if (i < l) {
                            int32_t v = yv[i][c] - q + field(acc, i, c);    // y is stored as q + (a value in (-gamma_1, gamma_1)) (dil_expand_kernel)
                            v += (v >> 31) & q;
                            zl[i * N + 64 * c + jb] = (uint32_t)v;
                            bad |= norm_bad((uint32_t)v, (uint32_t)q, (uint32_t)a.gamma1_beta);
                        }
                    }
                ok = __ballot(bad) == 0;
            }
            uint32_t ones = 0;
            if (ok) {
                // ---- c t0; ||c t0||inf < gamma_2 - beta; hints (make_hint, dilithium.c:1258-1284, 1702-1725).  |c t0| can pass 2^15: the 16-bit table entries are
                // added up in 32 bits
                int32_t a0s[2 * CSP][4];
#pragma unroll
                for (int i = 0; i < 2 * CSP; i++)
#pragma unroll
                    for (int c = 0; c < 4; c++) a0s[i][c] = 0;
                uint32_t t = 0;
                for (; t + 2 <= nt; t += 2) {
                    uint32_t v[2][kp][4];
#pragma unroll
                    for (int u = 0; u < 2; u++) {
                        const uint32_t *row = T0 + (jb + __builtin_amdgcn_readlane(my_base, (int)(t + u)));
#pragma unroll
                        for (int pp = 0; pp < kp; pp++)
#pragma unroll
                            for (int c = 0; c < 4; c++) v[u][pp][c] = row[pp * 3 * N + 64 * c];
                    }
#pragma unroll
                    for (int u = 0; u < 2; u++)
#pragma unroll
                        for (int pp = 0; pp < kp; pp++)
#pragma unroll
                            for (int c = 0; c < 4; c++) { a0s[2 * pp][c] += (int32_t)(int16_t)v[u][pp][c]; a0s[2 * pp + 1][c] += (int32_t)v[u][pp][c] >> 16; }
                }
                for (; t < nt; t++) {
                    const uint32_t *row = T0 + (jb + __builtin_amdgcn_readlane(my_base, (int)t));
#pragma unroll
                    for (int pp = 0; pp < kp; pp++)
#pragma unroll
                        for (int c = 0; c < 4; c++) {
                            const uint32_t w = row[pp * 3 * N + 64 * c];
                            a0s[2 * pp][c] += (int32_t)(int16_t)w; a0s[2 * pp + 1][c] += (int32_t)w >> 16;
                        }
                }
                uint32_t cnt = 0;
#pragma unroll
                for (int i = 0; i < 2 * CSP; i++)
#pragma unroll
                    for (int c = 0; c < 4; c++) {
                        if (i < kk) {
                            int32_t x = a0s[i][c];
                            x += (x >> 31) & q;                                   // canonical c t0
                            const uint32_t sum = csub((uint32_t)rr[i][c] + (uint32_t)x, (uint32_t)q);        // (w - c s2) + c t0
                            int32_t a0;
                            const uint32_t hb = round_alpha(sum, a0, a.alpha, a.am, q) != round_alpha((uint32_t)rr[i][c], a0, a.alpha, a.am, q);
                            bad |= norm_bad((uint32_t)x, (uint32_t)q, (uint32_t)a.gamma2_beta);
                            hl[i * N + 64 * c + jb] = (uint8_t)hb; cnt += hb;
                        }
                    }
                for (int o_ = 32; o_ > 0; o_ >>= 1) cnt += (uint32_t)__shfl_xor((int)cnt, o_);
                ones = cnt;
                ok = __ballot(bad) == 0 && ones <= (uint32_t)P.omega;
            }
            if (!ok) { wave_lds_fence(); continue; }                            // (cpos is rewritten by the next attempt)
            // ---- accepted: the signature (as dil_cfin_kernel)
            found = (int)at;
            wave_lds_fence();
            uint8_t *out = f.sigs + (size_t)item * f.sig_stride;
            {
                const int32_t half = (q - 1) >> 1;
                const int pairs = l * N / 2, nbytes = 2 * P.z_bits / 8;
                const uint64_t m = (1ull << P.z_bits) - 1;
                for (int pr = lane; pr < pairs; pr += 64) {
                    int32_t v0 = (int32_t)zl[2 * pr], v1 = (int32_t)zl[2 * pr + 1];
                    if (v0 > half) v0 -= q;
                    if (v1 > half) v1 -= q;
                    const uint64_t bits = (((uint64_t)(uint32_t)v0 & m) << P.z_bits) | ((uint64_t)(uint32_t)v1 & m);     // 2 z_bits bits, MSB first
                    uint8_t *dst = out + (size_t)pr * nbytes;
                    for (int b = 0; b < nbytes; b++) dst[b] = (uint8_t)(bits >> (8 * (nbytes - 1 - b)));
                }
            }
            const int kn = kk * P.n, h_bits = 8 + ((kk + 1) >> 1);
            tw[lane] = 0; tw[lane + 64] = 0;
            wave_lds_fence();
            auto put = [&](uint32_t off_, uint32_t v, int bits) {               // bits <= 16: at most two words
                const uint32_t w = off_ >> 5; const int sh = 32 - (int)(off_ & 31) - bits;
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
    }
    const bool again = live && found == -1 && lane == 0;
    uint32_t rpos = 0;
    if (again) rpos = atomicAdd(&fin_lds[0], 1u);
    __syncthreads();
    if (threadIdx.x == 0 && fin_lds[0]) fin_lds[1] = atomicAdd(f.next_count, fin_lds[0]);
    __syncthreads();
    if (again) { f.kappa[item] += S; f.next_active[fin_lds[1] + rpos] = item; }
}
