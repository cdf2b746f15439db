"""Python model of the HIP NTT kernel's index math (development aid, not product code).

Mirrors libsafecrypto_amd/csrc/ntt_kernels.hip: 16 coefficients per lane, n/16 lanes per
polynomial, rounds of <=4 in-register merged-twiddle butterfly stages separated by an LDS
exchange.  Checked here against tests/golden/ntt_ref.txt before the kernel is written.
"""
import sys, os
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..'))


def brv(x, bits):
    r = 0
    for b in range(bits):
        r |= ((x >> b) & 1) << (bits - 1 - b)
    return r


def find_root(q, n):
    for m in range(2, q - 1):
        if pow(m, n, q) == q - 1:
            return m
    return 0


def zetas_fwd(q, n):
    """zeta[k] = g^brv_L(k): Kyber-style table consumed as k = 2^s + block."""
    L = n.bit_length() - 1
    g = find_root(q, n)
    return [pow(g, brv(k, L), q) for k in range(n)]


def rounds(L):
    """[(stages, regbits)] ; regbits = position bits held in registers, high..low"""
    out = []
    s = 0
    while s < L:
        ns = min(4, L - s)
        act = [L - 1 - (s + t) for t in range(ns)]
        regbits = act if ns == 4 else [3, 2, 1, 0]
        out.append((list(range(s, s + ns)), regbits))
        s += ns
    return out


def split(L, regbits):
    lanebits = [b for b in range(L) if b not in regbits]   # low -> high
    return lanebits


def pos_of(lane, reg, L, regbits):
    lanebits = split(L, regbits)
    j = 0
    for t, b in enumerate(lanebits):
        j |= ((lane >> t) & 1) << b
    for t, b in enumerate(reversed(regbits)):        # reg bit 0 = lowest regbit
        j |= ((reg >> t) & 1) << b
    return j


def coords_of(j, L, regbits):
    lanebits = split(L, regbits)
    lane = 0
    for t, b in enumerate(lanebits):
        lane |= ((j >> b) & 1) << t
    reg = 0
    for t, b in enumerate(reversed(regbits)):
        reg |= ((j >> b) & 1) << t
    return lane, reg


def swz(addr):
    return addr ^ (((addr >> 6) & 3) << 2)


def fwd_model(t, q, n):
    """returns V natural order canonical"""
    L = n.bit_length() - 1
    z = zetas_fwd(q, n)
    lanes = n // 16
    rs = rounds(L)
    # load: round 0 layout
    regbits = rs[0][1]
    x = np.zeros((lanes, 16), dtype=object)
    for l in range(lanes):
        for r in range(16):
            x[l, r] = int(t[pos_of(l, r, L, regbits)]) % q
    for ri, (stages, regbits) in enumerate(rs):
        if ri > 0:
            # exchange through "LDS": addr = lane'*16+reg' swizzled
            lds = [None] * n
            pregbits = rs[ri - 1][1]
            for l in range(lanes):
                for r in range(16):
                    j = pos_of(l, r, L, pregbits)
                    l2, r2 = coords_of(j, L, regbits)
                    lds[swz(l2 * 16 + r2)] = x[l, r]
            for l in range(lanes):
                for r in range(16):
                    x[l, r] = lds[swz(l * 16 + r)]
        for s in stages:
            p = L - 1 - s
            tbit = list(reversed(regbits)).index(p)
            for l in range(lanes):
                for r in range(16):
                    if (r >> tbit) & 1:
                        continue
                    j = pos_of(l, r, L, regbits)
                    k = (1 << s) + (j >> (L - s))
                    a, b = x[l, r], x[l, r | (1 << tbit)]
                    tt = b * z[k] % q
                    x[l, r], x[l, r | (1 << tbit)] = (a + tt) % q, (a - tt) % q
    out = [0] * n
    regbits = rs[-1][1]
    for l in range(lanes):
        for r in range(16):
            j = pos_of(l, r, L, regbits)
            out[brv(j, L)] = x[l, r]
    return out


def inv_model(V, q, n):
    L = n.bit_length() - 1
    z = zetas_fwd(q, n)
    zi = [pow(v, q - 2, q) for v in z]
    ninv = pow(n, q - 2, q)
    lanes = n // 16
    rs = rounds(L)
    regbits = rs[-1][1]
    x = np.zeros((lanes, 16), dtype=object)
    for l in range(lanes):
        for r in range(16):
            x[l, r] = int(V[brv(pos_of(l, r, L, regbits), L)]) % q
    for ri in range(len(rs) - 1, -1, -1):
        stages, regbits = rs[ri]
        if ri < len(rs) - 1:
            lds = [None] * n
            pregbits = rs[ri + 1][1]
            for l in range(lanes):
                for r in range(16):
                    j = pos_of(l, r, L, pregbits)
                    l2, r2 = coords_of(j, L, regbits)
                    lds[swz(l2 * 16 + r2)] = x[l, r]
            for l in range(lanes):
                for r in range(16):
                    x[l, r] = lds[swz(l * 16 + r)]
        for s in reversed(stages):
            p = L - 1 - s
            tbit = list(reversed(regbits)).index(p)
            for l in range(lanes):
                for r in range(16):
                    if (r >> tbit) & 1:
                        continue
                    j = pos_of(l, r, L, regbits)
                    k = (1 << s) + (j >> (L - s))
                    a, b = x[l, r], x[l, r | (1 << tbit)]
                    x[l, r], x[l, r | (1 << tbit)] = (a + b) % q, (a - b) * zi[k] % q
    out = [0] * n
    regbits = rs[0][1]
    for l in range(lanes):
        for r in range(16):
            out[pos_of(l, r, L, regbits)] = x[l, r] * ninv % q
    return out


if __name__ == '__main__':
    from tests.golden_io import load_golden
    G = load_golden('ntt_ref.txt')
    for q, n in [(7681, 256), (8380417, 256), (12289, 512), (12289, 1024), (8399873, 512)]:
        a = G[f'q{q}_n{n}_a'].reshape(4, n)
        f = G[f'q{q}_n{n}_fp_fwd'].reshape(4, n)
        assert find_root(q, n) == int(G[f'q{q}_n{n}_w'][1])
        for k in range(4):
            V = fwd_model(a[k], q, n)
            assert V == [int(v) for v in f[k]], (q, n, k)
            t = inv_model(V, q, n)
            assert t == [int(v) % q for v in a[k]]
        print('model ok', q, n)
