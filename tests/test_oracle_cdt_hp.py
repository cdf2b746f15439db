"""CPU: the oracle's 128- / 192-bit CDT sampler (oracle/sc_oracle.c orc_cdt_hp_sample: gaussian_cdf_sample_128 / _192, binary_search_128 / _192 and compare_ge_prec of
/root/reference/src/utils/sampling/gaussian_cdf.c:110-184,468-509, restated as written) against the reference's own get_vector_32 run over tables of known content
(tests/golden/cdt_hp_ref.txt, oracle/ref_driver.c gen_cdt_hp) - and what that search amounts to."""
import numpy as np
import pytest

from tests.golden_io import load_golden
from tests import oracle_lib as O

TAGS = ['hp128_bliss1_s215_t13.42', 'hp192_bliss1_s215_t13.42', 'hp128_unit_s3.33_t12', 'hp192_leaf_s1.7_t10']


@pytest.fixture(scope='module')
def G():
    return load_golden('cdt_hp_ref.txt')


def case(G, tag):
    prec, size, ns, ties = (int(v) for v in G[f'{tag}_par'])
    W = prec // 64
    tab = np.asarray(G[f'{tag}_table'], dtype=np.uint64).reshape(size, W)
    x = np.asarray(G[f'{tag}_words'], dtype=np.uint64).reshape(ns, W)
    return W, tab, x, np.asarray(G[f'{tag}_samples'], dtype=np.int32), ties


def halving(x, tab, ge):
    out = np.zeros(len(x), dtype=np.int32)
    size = len(tab)
    for i in range(len(x)):
        a, st = 0, size >> 1
        while st:
            b = a + st
            if b < size and ge(x[i], tab[b]):
                a = b
            st >>= 1
        out[i] = a if int(x[i][0]) & 1 else -a
    return out


@pytest.mark.parametrize('tag', TAGS)
def test_oracle_reproduces_the_references_samples(G, tag):
    W, tab, x, want, ties = case(G, tag)
    assert ties >= 16                                                       # the fixture does contain words that tie with an entry's top limb
    assert np.array_equal(O.cdt_hp_sample(x, tab, W), want)
    assert np.array_equal(O.cdt_hp_sample(x, tab, W, centre=7), want + 7)


@pytest.mark.parametrize('tag', TAGS)
def test_what_the_search_amounts_to(G, tag):
    """the top limb's ">=" alone decides; a full-width comparison and the 64-bit sampler's strict "<" both give other samples on this fixture"""
    W, tab, x, want, _ = case(G, tag)
    top = halving(x, tab, lambda xx, e: int(xx[W - 1]) >= int(e[W - 1]))
    assert np.array_equal(top, want)
    wide = lambda v: sum(int(v[j]) << (64 * j) for j in range(W))
    assert not np.array_equal(halving(x, tab, lambda xx, e: wide(xx) >= wide(e)), want)
    assert not np.array_equal(halving(x, tab, lambda xx, e: int(xx[W - 1]) > int(e[W - 1])), want)
    # the lower limbs of the table are noise and change nothing
    noise = tab.copy(); noise[:, :W - 1] ^= np.uint64(0x5555555555555555)
    assert np.array_equal(O.cdt_hp_sample(x, noise, W), want)


def test_extreme_words():
    tab = np.zeros((8, 2), dtype=np.uint64)
    tab[:, 1] = np.array([0, 5, 5, 9, 2**63, 2**64 - 2, 2**64 - 1, 2**64 - 1], dtype=np.uint64)
    x = np.array([[1, 0], [0, 0], [1, 5], [1, 4], [3, 2**64 - 1], [2, 2**64 - 1], [1, 2**64 - 2], [1, 9], [1, 2**63 - 1]], dtype=np.uint64)
    want = halving(x, tab, lambda xx, e: int(xx[1]) >= int(e[1]))
    assert np.array_equal(O.cdt_hp_sample(x, tab, 2), want)
    assert want.tolist() == [0, 0, 2, 0, 7, -7, 5, 3, 3]
