"""CPU: the coefficient-domain shift tables sc_hip_dilithium_set_privkey builds for the finishing kernel without transforms (dil_cfins_kernel,
csrc/dilithium.hip) - host logic of the product library, no device involved.  The test walks the tables exactly as the kernel's lanes do (lane owns the
coefficients 64 c + 16 (lane & 3) + (lane >> 2); a term of c at position p reads word n + j - p of each row pair; +1 terms add, -1 terms subtract; the two
16-bit halves of a word are two rows) and must find c s2, c s1, c t0 of the negacyclic ring - the integers dilithium.c:1655-1725 computes with NTTs."""
import ctypes as C

import numpy as np
import pytest

from tests.golden_io import load_golden
from tests import oracle_lib as O

N = 256


def _lib():
    from libsafecrypto_amd import engine
    L = C.CDLL(engine.LIB_PATH)
    L.sc_hip_dilithium_fin_tables_host.argtypes = [C.c_void_p] * 5 + [C.c_size_t, C.c_void_p]
    L.sc_hip_dilithium_fin_tables_host.restype = C.c_int
    return L


def _tables(L, par, s1, s2, t):
    k, l = int(par[3]), int(par[4])
    words = (2 * ((k + 1) // 2) + (l + 1) // 2) * 2 * N
    out = np.zeros(words, dtype=np.uint32)
    fits = C.c_int(-1)
    p = lambda a: a.ctypes.data_as(C.c_void_p)
    rc = L.sc_hip_dilithium_fin_tables_host(p(par), p(s1), p(s2), p(t), p(out), words, C.byref(fits))
    assert rc == 0
    return out, fits.value


def _negacyclic(c, s):
    full = np.zeros(2 * N, dtype=np.int64)
    for p in np.nonzero(c)[0]:
        full[p:p + N] += int(c[p]) * s
    return full[:N] - full[N:]


def _walk(T, rows, c):
    """what the kernel's lanes add up: [rows][n] integers, natural order"""
    lane = np.arange(64)
    jb = 16 * (lane & 3) + (lane >> 2)
    out = np.zeros((rows, N), dtype=np.int64)
    for pp in range((rows + 1) // 2):
        tab = T[pp * 2 * N:(pp + 1) * 2 * N]
        lo = tab.astype(np.uint16).astype(np.int16).astype(np.int64)
        hi = (tab >> 16).astype(np.uint16).astype(np.int16).astype(np.int64)
        for cc in range(4):
            j = 64 * cc + jb
            for p in np.nonzero(c)[0]:
                idx = N + j - p
                assert idx.min() >= 1 and idx.max() < 2 * N
                out[2 * pp, j] += int(c[p]) * lo[idx]
                if 2 * pp + 1 < rows:
                    out[2 * pp + 1, j] += int(c[p]) * hi[idx]
                else:
                    assert not hi[idx].any()                   # the unused half of the last pair stays zero
    return out


@pytest.mark.parametrize('tag', ['dil0', 'dil1', 'dil2', 'dil3'])
def test_tables_of_the_fixture_keys(tag):
    G = load_golden('dilithium_ref.txt')
    key = O.DilKey.from_golden(G, tag)
    L = _lib()
    T, fits = _tables(L, key.par, key.s1, key.s2, key.t)
    assert fits == 1
    k, l, q, d, weight = key.k, key.l, key.q, int(key.par[5]), int(key.par[6])
    kp, lp2 = (k + 1) // 2, (l + 1) // 2
    centre = lambda a: ((a.astype(np.int64) % q) + q) % q
    cen = lambda a: np.where(centre(a) > (q - 1) // 2, centre(a) - q, centre(a))
    s1, s2 = cen(key.s1).reshape(l, N), cen(key.s2).reshape(k, N)
    t0 = key.t.astype(np.int64) & ((1 << d) - 1)
    t0 = np.where(t0 > (1 << (d - 1)), t0 - (1 << d), t0).reshape(k, N)          # decompose, module_lwe.c:89-104
    rng = np.random.default_rng(k * 16 + l)
    for _ in range(3):
        c = np.zeros(N, dtype=np.int64)
        c[rng.choice(N, weight, replace=False)] = rng.choice([-1, 1], weight)
        got2 = _walk(T[:kp * 2 * N], k, c)
        got1 = _walk(T[kp * 2 * N:(kp + lp2) * 2 * N], l, c)
        got0 = _walk(T[(kp + lp2) * 2 * N:], k, c)
        for i in range(k):
            assert np.array_equal(got2[i], _negacyclic(c, s2[i])), (tag, 's2', i)
            assert np.array_equal(got0[i], _negacyclic(c, t0[i])), (tag, 't0', i)
        for i in range(l):
            assert np.array_equal(got1[i], _negacyclic(c, s1[i])), (tag, 's1', i)
        assert np.abs(got2).max() < 32768 and np.abs(got1).max() < 32768          # the packed 16-bit sums cannot overflow


def test_keys_that_do_not_fit_are_refused():
    G = load_golden('dilithium_ref.txt')
    key = O.DilKey.from_golden(G, 'dil3')
    L = _lib()
    s1 = key.s1.copy(); s1[5] = 600                          # 60 terms of 600 leave 16 bits
    assert _tables(L, key.par, s1, key.s2, key.t)[1] == 0
    s2 = key.s2.copy(); s2[-1] = key.q - 547                  # canonical form of -547: 60 x 547 = 32820
    assert _tables(L, key.par, key.s1, s2, key.t)[1] == 0
    s2[-1] = key.q - 546                                     # 60 x 546 = 32760 fits
    assert _tables(L, key.par, key.s1, s2, key.t)[1] == 1
    par = key.par.copy(); par[3] = 7                          # k = 7: more row pairs than the kernel holds
    big = np.zeros(7 * N, dtype=np.int32)
    assert _tables(L, par, key.s1, big, big)[1] == 0
