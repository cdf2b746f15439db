"""CPU: the oracle restatement (oracle/sc_oracle.c) against fixtures produced by the
reference itself (tests/golden/*.txt, generator oracle/ref_driver.c) and against the
hard-coded answers of the reference's unit tests."""
import os

import numpy as np
import pytest

from tests.golden_io import load_golden
from tests import oracle_lib as O

CFGS = [(7681, 256), (8380417, 256), (12289, 512), (12289, 1024), (8399873, 512)]
ROOTS = {(7681, 256): 62, (8380417, 256): 1753, (12289, 512): 49, (12289, 1024): 7}  # SURVEY 8(a) a2 [probe]


@pytest.fixture(scope='module')
def G():
    return load_golden('ntt_ref.txt')


@pytest.mark.parametrize('q,n', CFGS)
def test_tables_match_reference(G, q, n):
    w, r = O.tables(q, n)
    assert np.array_equal(w, G[f'q{q}_n{n}_w'])
    assert np.array_equal(r, G[f'q{q}_n{n}_r'])
    if (q, n) in ROOTS:
        assert O.lib().orc_find_root(q, n) == ROOTS[(q, n)]
    assert (int(r[0]) * n + 1) % q == 0          # r[0] == -n^{-1}


def test_scalar_known_answers(G):
    # src/unit/unit_ntt.c:166-320: 1234*5678 mod 12289 = 1922 etc.
    L = O.lib()
    assert L.orc_muln(1234, 5678, 12289) == 1922
    assert L.orc_muln(-1234, 5678, 12289) == -1922
    assert L.orc_muln(12288, 12288, 12289) == 1
    assert L.orc_modn(12289 * 12289 - 1, 12289) == 12288
    assert list(G['scalar_q12289_fp']) == [1922, -1922, 1, 12288]


@pytest.mark.parametrize('q,n', CFGS)
def test_fwd_inv_mul_center_vs_reference(G, q, n):
    w, r = O.tables(q, n)
    a = G[f'q{q}_n{n}_a'].astype(np.int32).reshape(4, n)
    b = G[f'q{q}_n{n}_b'].astype(np.int32).reshape(4, n)
    fwd = O.fwd_ntt(a, q, w)
    backends = ['fp', 'ref'] + (['sol'] if q in (7681, 8380417) else [])
    for be in backends:   # every reference backend agrees on the canonical value
        assert np.array_equal(fwd.reshape(-1), G[f'q{q}_n{n}_{be}_fwd']), be
        ref_inv = G[f'q{q}_n{n}_{be}_inv']
        if be == 'sol':   # the Solinas backends' weak reduction leaves inv_ntt non-canonical in the
            ref_inv = np.mod(ref_inv, q)   # reference itself (not the default backend): compare in Z_q
        assert np.array_equal(O.inv_ntt(fwd, q, w, r).reshape(-1), ref_inv), be
    assert np.array_equal(O.inv_ntt(fwd, q, w, r), np.mod(a, q))
    c = np.empty_like(a)
    for k in range(4):
        O.lib().orc_polymul(c[k], n, q, a[k], b[k], w, r)
    assert np.array_equal(c.reshape(-1), G[f'q{q}_n{n}_fp_mul'])
    # fused form used by the GPU config-2 kernel: INTT(NTT(a) o bhat)
    bhat = O.fwd_ntt(b, q, w)
    assert np.array_equal(O.polymul_ntt(a, bhat, q, w, r), c)
    cen = a.copy().reshape(-1)
    O.lib().orc_center(cen, cen.size, q)
    assert np.array_equal(cen, G[f'q{q}_n{n}_fp_center'])


def test_survey_sanity_vectors():
    # SURVEY.md 8(a) [probe] values for a[i] = i
    exp = {
        (7681, 256): ([3472, 1413, 1460, 7105, 2628, 1995, 3472, 7092], 5877, [7405, 5745, 4343, 3201, 2321, 1705]),
        (8380417, 256): ([8023823, 8368027, 4046506, 252176, 2269315, 2703904, 7301606, 1128875], 3279003,
                         [5584257, 5551873, 5519747, 5487881, 5456277, 5424937]),
        (12289, 512): ([3563, 6129, 5350, 5063, 2188, 5583, 2185, 11079], 0, [8733, 1319, 6708, 324, 6747, 1401]),
        (12289, 1024): ([11436, 1100, 67, 1250, 1639, 2573, 12210, 6092], 8705, [7907, 1293, 7994, 3434, 12193, 9695]),
    }
    for (q, n), (f8, flast, aa) in exp.items():
        w, r = O.tables(q, n)
        a = np.arange(n, dtype=np.int32)[None, :]
        f = O.fwd_ntt(a, q, w)[0]
        assert list(f[:8]) == f8 and f[-1] == flast
        c = np.empty(n, dtype=np.int32)
        O.lib().orc_polymul(c, n, q, a[0], a[0], w, r)
        assert list(c[:6]) == aa


def test_mixed_ntt32_7681_roundtrip():
    # config #1 = src/unit/unit_ntt.c:993-1036 test_mixed_ntt32_7681: small input survives fwd->normalize->inv
    q, n = 7681, 256
    w, r = O.tables(q, n)
    rng = np.random.default_rng(1)
    a = rng.integers(-12, 13, size=(1000, n)).astype(np.int32)
    back = O.inv_ntt(O.fwd_ntt(a, q, w), q, w, r)
    assert np.array_equal(back, np.mod(a, q))


def test_sparse_mult_matches_dense():
    q, n = 8399873, 512
    w, r = O.tables(q, n)
    rng = np.random.default_rng(2)
    t = rng.integers(0, q, size=n).astype(np.int32)
    pos = rng.choice(n, size=11, replace=False).astype(np.int32)
    v = np.empty(n, dtype=np.int32)
    O.lib().orc_mult_sparse(v, n, 11, t, pos)
    c = np.zeros(n, dtype=np.int32)
    c[pos] = 1
    d = np.empty(n, dtype=np.int32)
    O.lib().orc_polymul(d, n, q, t, c, w, r)
    # ntt.c:381-400 subtracts where the schoolbook product adds: v == -(t*c)
    assert np.array_equal(np.mod(-v.astype(np.int64), q), d)


@pytest.mark.parametrize('tag,tail,sigma', [('bliss1_s215_t13.42', 13.42, 215.0), ('unit_s100_t12', 12.0, 100.0), ('leaf_s1.7_t10', 10.0, 1.7)])
def test_cdt32_table_and_samples_vs_reference(tag, tail, sigma):
    """the sampler at 32-bit precision (create_sampler(..., SAMPLING_32BIT, ...): tests/golden/cdt32_ref.txt made by ref_driver cdt32): the table - float
    accumulators, so whatever the host's expl() does in its last bit is rounded away in all but rare entries - and (table, 32-bit word) -> sample"""
    G = load_golden('cdt32_ref.txt')
    want = np.asarray(G[f'{tag}_tab32']).astype(np.int64) & 0xFFFFFFFF
    cdf = O.cdt32_table(tail, sigma)
    assert np.abs(cdf.astype(np.int64) - want).max() <= 256          # one unit of a float's last place near 2^32
    ref_drv = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'oracle', '_ref', 'ref_driver')
    if os.path.exists(ref_drv):       # the reference itself, re-run on this host: exact
        import subprocess, tempfile
        with tempfile.TemporaryDirectory() as d:
            subprocess.check_call([ref_drv, 'cdt32', d])
            live = load_golden(os.path.join(d, 'cdt32_ref.txt'))
        assert np.array_equal(cdf.astype(np.int64), np.asarray(live[f'{tag}_tab32']).astype(np.int64) & 0xFFFFFFFF)
    tab = want.astype(np.uint32)
    assert tab[0] == 0 and tab[-1] == 0xFFFFFFFF and np.all(np.diff(tab.astype(np.int64)) >= 0)
    u = (np.asarray(G[f'{tag}_words32']).astype(np.int64) & 0xFFFFFFFF).astype(np.uint32)
    v = O.cdt32_sample(u, tab)
    assert np.array_equal(v, G[f'{tag}_samples'])
    assert np.abs(v).max() < tab.size


@pytest.mark.parametrize('tag,tail,sigma', [('bliss1_s215_t13.42', 13.42, 215.0), ('unit_s100_t12', 12.0, 100.0),
                                            ('leaf_s1.7_t10', 10.0, 1.7)])
def test_cdt_table_and_samples_vs_reference(tag, tail, sigma):
    G = load_golden('cdt_ref.txt')
    cdf = O.cdt_table(tail, sigma)
    # long-double expl(): bit-identical on the CPU family that made the fixture, and within a
    # unit or two in the last place on another vendor's x87 (f2xm1 is not specified bit-exactly)
    diff = np.abs(cdf.astype(object) - G[f'{tag}_cdf'].astype(object)).max()
    assert diff <= 2
    ref_drv = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'oracle', '_ref', 'ref_driver')
    if os.path.exists(ref_drv):       # the reference itself, re-run on this host: exact
        import subprocess, tempfile
        with tempfile.TemporaryDirectory() as d:
            subprocess.check_call([ref_drv, 'cdt', d])
            live = load_golden(os.path.join(d, 'cdt_ref.txt'))
        assert np.array_equal(cdf, live[f'{tag}_cdf'])
        assert np.array_equal(O.cdt_sample(live[f'{tag}_uniform'], cdf), live[f'{tag}_samples'])
    from libsafecrypto_amd import build_cdt_table      # the product's host-side table builder
    assert np.array_equal(build_cdt_table(tail, sigma), cdf)
    cdf = G[f'{tag}_cdf']             # samples: pure function of (table, word) -> exact everywhere
    assert cdf[0] == 0 and cdf[-1] == np.uint64(2**64 - 1) and np.all(np.diff(cdf.astype(object)) >= 0)
    v = O.cdt_sample(G[f'{tag}_uniform'], cdf)
    assert np.array_equal(v, G[f'{tag}_samples'])
    assert np.abs(v).max() < cdf.size          # unit_sampling.c:194-222 bound


@pytest.mark.parametrize('q,n', [(51750913, 512), (51750913, 1024), (4206593, 512), (4206593, 1024), (16813057, 512), (134348801, 1024)])
def test_large_modulus_variants_vs_reference(q, n):
    """fwd_ntt_32_32_large / inv_ntt_32_32_large of the reference (26-bit modulus, ENS / DLP / ring-TESLA-1) - the oracle's
    transforms are the same functions mod q, compared after normalize_32 (tests/golden/ntt_large_ref.txt)"""
    G = load_golden('ntt_large_ref.txt')
    w, r = O.tables(q, n)
    assert int(w[1]) == int(G[f'q{q}_n{n}_w1'][0])
    a = G[f'q{q}_n{n}_a'].astype(np.int32).reshape(2, n)
    b = G[f'q{q}_n{n}_b'].astype(np.int32).reshape(2, n)
    f = O.fwd_ntt(a, q, w)
    assert np.array_equal(f.reshape(-1), G[f'q{q}_n{n}_fwd'])
    assert np.array_equal(O.inv_ntt(f, q, w, r).reshape(-1), G[f'q{q}_n{n}_inv'])
    assert np.array_equal(O.polymul_ntt(a, O.fwd_ntt(b, q, w), q, w, r).reshape(-1), G[f'q{q}_n{n}_mul'])
