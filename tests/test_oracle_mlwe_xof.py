"""CPU: the oracle's XOF-driven module-LWE helpers (oracle/sc_oracle_mlwe_xof.c) against the reference's own functions called directly
(tests/golden/mlwe_xof_ref.txt, oracle/ref_driver.c gen_mlwe_xof: uniform_random_ring_q_xof, create_rand_product_32_xof / _16_xof in both loop orders,
uniform_rand_sample_xof / _small_xof, binomial_rand_sample_xof - /root/reference/src/utils/arith/module_lwe.c:330-367,449-512,550-560,751-912)."""
import numpy as np
import pytest

from tests.golden_io import load_golden
from tests import oracle_lib as O

CASES = [0, 1, 2, 3, 4]


@pytest.fixture(scope='module')
def G():
    return load_golden('mlwe_xof_ref.txt')


def case(G, c):
    q, q_bits, k, l, is16, shake, seedlen, n = (int(x) for x in G[f'xof{c}_params'])
    seed = G[f'xof{c}_seed'].astype(np.uint8)
    assert seed.size == seedlen
    return q, q_bits, k, l, shake, seed, n


@pytest.mark.parametrize('c', CASES)
def test_ring_entries_are_the_references(G, c):
    q, q_bits, k, l, shake, seed, n = case(G, c)
    got, pos = O.ring_q_xof(shake, seed, 0, k * l, q, q_bits, n)
    assert np.array_equal(got, G[f'xof{c}_ring'])
    assert pos == k * l * n * 4
    # the stream is the reference's SHAKE (its permutation is not the published one: tests/test_oracle.py) cut into 32-bit little-endian words, masked, minus q
    h = O.shake(shake, bytes(seed), k * l * n * 4)
    w = np.frombuffer(h, dtype='<u4').astype(np.int64) & ((1 << q_bits) - 1)
    w = np.where(w >= q, w - q, w)
    assert np.array_equal(w, G[f'xof{c}_ring'])
    # a start in the middle of the stream: polynomial 1 onwards
    part, pos2 = O.ring_q_xof(shake, seed, n * 4, k * l - 1, q, q_bits, n)
    assert np.array_equal(part, G[f'xof{c}_ring'][n:]) and pos2 == pos


@pytest.mark.parametrize('c', CASES)
@pytest.mark.parametrize('tr', [0, 1])
def test_rand_product_is_the_references(G, c, tr):
    q, q_bits, k, l, shake, seed, n = case(G, c)
    width = 16 if int(G[f'xof{c}_params'][4]) else 32                 # which of create_rand_product_16_xof / _32_xof ref_driver called
    t, pos = O.rand_product_xof(shake, seed, 0, q, q_bits, k, l, tr, G[f'xof{c}_y{tr}'], n, width)
    assert np.array_equal(t, G[f'xof{c}_t{tr}'])
    assert pos == k * l * n * 4
    # the product proper (what the _16 form computes and the engine serves for every modulus), from the matrix entries in the fixture
    w, r = O.tables(q, n)
    A = G[f'xof{c}_ring'].reshape(k * l, n)
    yh = O.fwd_ntt(G[f'xof{c}_y{tr}'].reshape(l, n), q, w).astype(np.int64)
    want = np.stack([O.inv_ntt((sum(yh[j] * A[j * k + i if tr else i * l + j] for j in range(l)) % q).astype(np.int32)[None], q, w, r)[0] for i in range(k)])
    proper, _ = O.rand_product_xof(shake, seed, 0, q, q_bits, k, l, tr, G[f'xof{c}_y{tr}'], n, 16)
    assert np.array_equal(proper, want.reshape(-1))
    if width == 32 and tr and l > 1:
        assert not np.array_equal(proper, t)                          # module_lwe.c:796 - the 32-bit form's transposed branch is not A^T y
    else:
        assert np.array_equal(proper, t)


@pytest.mark.parametrize('c', CASES)
def test_small_samplers_chain_on_one_stream(G, c):
    """the five calls ref_driver made one after the other on one xof context, the unwritten entries (marker 99) included"""
    q, q_bits, k, l, shake, seed, n = case(G, c)
    m, eta, bits = max(k, l), (4 if q == 7681 else 5), 3
    pos = 0
    got, pos = O.sample_xof('uniform', shake, seed, pos, eta, bits, n * m)
    assert np.array_equal(got, G[f'xof{c}_unif']) and pos == n * m * 4
    got, p2 = O.sample_xof('small', shake, seed, pos, eta, bits, n * 5)
    assert np.array_equal(got, G[f'xof{c}_small5']) and p2 == pos + 512           # one whole block of 1024 values: the last 256 stay as they were
    assert (got[1024:] == 99).all() and (np.abs(got[:1024]) <= eta).all()
    got, p3 = O.sample_xof('small', shake, seed, p2, eta, bits, n * 8)
    assert np.array_equal(got, G[f'xof{c}_small8']) and p3 == p2 + 1024
    got, p4 = O.sample_xof('binomial', shake, seed, p3, min(eta, 4), 0, n * 4)
    assert np.array_equal(got, G[f'xof{c}_binomial4']) and p4 == p3 + min(eta, 4) * 256
    got, p5 = O.sample_xof('binomial', shake, seed, p4, 3, 0, n * 8)
    assert np.array_equal(got, G[f'xof{c}_binomial8']) and p5 == p4 + 2 * 3 * 256   # two blocks drawn, both written at s[0..1024): the second one stays
    assert (got[1024:] == 99).all()
