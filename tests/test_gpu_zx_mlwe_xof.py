"""(Sorted behind the rest of the GPU suite on purpose: written in round 6, while the GPU pool was closed - these tests have run against the CPU build of the library only, and a
run that stops at the first failure should have been through everything that HAS run on hardware before it gets here.)

GPU: the XOF-driven module-LWE helpers through the C ABI (sc_hip_mlwe_expand_matrix_xof, sc_hip_mlwe_sample_xof, and the product as expansion + sc_hip_matvec)
against the reference's own functions (tests/golden/mlwe_xof_ref.txt: uniform_random_ring_q_xof, create_rand_product_32_xof / _16_xof, uniform_rand_sample_xof /
_small_xof, binomial_rand_sample_xof - /root/reference/src/utils/arith/module_lwe.c:330-367,449-512,550-560,751-912) and, on seeded batches, the oracle."""
import numpy as np
import pytest

from tests.golden_io import load_golden
from tests import oracle_lib as O

pytestmark = pytest.mark.gpu

CASES = [0, 1, 2, 3, 4]


@pytest.fixture(scope='module')
def eng():
    from libsafecrypto_amd import Engine
    e = Engine(0)
    yield e
    e.close()


@pytest.fixture(scope='module')
def G():
    return load_golden('mlwe_xof_ref.txt')


def dev(a, dt=np.int32):
    import torch
    return torch.from_numpy(np.ascontiguousarray(a, dtype=dt)).cuda()


def case(G, c):
    q, q_bits, k, l, is16, shake, seedlen, n = (int(x) for x in G[f'xof{c}_params'])
    return q, q_bits, k, l, is16, shake, G[f'xof{c}_seed'].astype(np.uint8), n


@pytest.mark.parametrize('c', CASES)
def test_matrix_entries_are_the_references(eng, G, c):
    q, q_bits, k, l, _, shake, seed, n = case(G, c)
    got = eng.expand_matrix_xof(dev(seed[None], np.uint8), k * l, q, q_bits, shake).cpu().numpy()
    assert np.array_equal(got.reshape(-1), G[f'xof{c}_ring'])
    # from the second polynomial on (a start inside the stream, not on a rate boundary: 1024 bytes = 6 blocks of 168 + 16)
    part = eng.expand_matrix_xof(dev(seed[None], np.uint8), k * l - 1, q, q_bits, shake, skip=4 * n).cpu().numpy()
    assert np.array_equal(part.reshape(-1), G[f'xof{c}_ring'][n:])


@pytest.mark.parametrize('c', CASES)
@pytest.mark.parametrize('tr', [0, 1])
def test_rand_product_is_the_references(eng, G, c, tr):
    """create_rand_product_*_xof = the expansion, then sc_hip_matvec.  The _32 form's transposed branch is the one place where the reference does not compute
    A^T y (module_lwe.c:796; oracle/sc_oracle_mlwe_xof.c): there the engine is held to the product proper, which the oracle restates too (width 16)"""
    q, q_bits, k, l, is16, shake, seed, n = case(G, c)
    y = G[f'xof{c}_y{tr}'].reshape(1, l, n)
    p = eng.plan(q, n)
    t, pos = p.rand_product_xof(dev(seed, np.uint8), dev(y), k, l, q_bits, bool(tr), shake)
    assert pos == k * l * n * 4
    if is16 or not tr or l == 1:
        assert np.array_equal(t.cpu().numpy().reshape(-1), G[f'xof{c}_t{tr}'])
    want, _ = O.rand_product_xof(shake, seed, 0, q, q_bits, k, l, tr, y, n, 16)
    assert np.array_equal(t.cpu().numpy().reshape(-1), want)
    p.close()


@pytest.mark.parametrize('c', CASES)
def test_small_samplers_chain_on_one_stream(eng, G, c):
    """the five calls ref_driver made one after the other on one xof context; what the reference's functions do not write (marker 99) stays unwritten"""
    q, q_bits, k, l, _, shake, seed, n = case(G, c)
    m, eta, bits = max(k, l), (4 if q == 7681 else 5), 3
    sd = dev(seed[None], np.uint8)
    got, pos = eng.sample_xof('uniform', sd, eta, bits, n * m, shake, 0, fill=99)
    assert np.array_equal(got.cpu().numpy().reshape(-1), G[f'xof{c}_unif']) and pos == n * m * 4
    got, p2 = eng.sample_xof('small', sd, eta, bits, n * 5, shake, pos, fill=99)
    assert np.array_equal(got.cpu().numpy().reshape(-1), G[f'xof{c}_small5']) and p2 == pos + 512
    got, p3 = eng.sample_xof('small', sd, eta, bits, n * 8, shake, p2, fill=99)
    assert np.array_equal(got.cpu().numpy().reshape(-1), G[f'xof{c}_small8']) and p3 == p2 + 1024
    got, p4 = eng.sample_xof('binomial', sd, min(eta, 4), 0, n * 4, shake, p3, fill=99)
    assert np.array_equal(got.cpu().numpy().reshape(-1), G[f'xof{c}_binomial4']) and p4 == p3 + min(eta, 4) * 256
    got, p5 = eng.sample_xof('binomial', sd, 3, 0, n * 8, shake, p4, fill=99)
    assert np.array_equal(got.cpu().numpy().reshape(-1), G[f'xof{c}_binomial8']) and p5 == p4 + 2 * 3 * 256


@pytest.mark.parametrize('shake,seedlen,nkeys', [(128, 32, 133), (256, 48, 70), (128, 1, 3), (128, 200, 65)])
def test_a_seed_per_key_vs_oracle(eng, shake, seedlen, nkeys):
    """a stream per lane: ragged last workgroup, seeds shorter and longer than a rate block, Kyber- and Dilithium-shaped matrices and all three samplers"""
    rng = np.random.default_rng(shake + seedlen + nkeys)
    seeds = rng.integers(0, 256, size=(nkeys, seedlen), dtype=np.uint8)
    sd = dev(seeds, np.uint8)
    for q, q_bits, npoly in ((8380417, 23, 7), (7681, 13, 4)):
        got = eng.expand_matrix_xof(sd, npoly, q, q_bits, shake).cpu().numpy()
        for i in range(nkeys):
            assert np.array_equal(got[i].reshape(-1), O.ring_q_xof(shake, seeds[i], 0, npoly, q, q_bits)[0]), (q, i)
    for kind, eta, bits, count, skip in (('uniform', 5, 3, 300, 0), ('small', 5, 3, 2048, 52), ('small', 2, 2, 1024, 7), ('binomial', 4, 0, 1024, 168), ('binomial', 2, 0, 3072, 1)):
        got, pos = eng.sample_xof(kind, sd, eta, bits, count, shake, skip, fill=-7)
        got = got.cpu().numpy()
        for i in range(nkeys):
            want, wpos = O.sample_xof(kind, shake, seeds[i], skip, eta, bits, count, fill=-7)
            assert np.array_equal(got[i], want) and pos == wpos, (kind, i)


def test_arguments_are_checked(eng):
    import torch
    from libsafecrypto_amd import EngineError
    sd = torch.zeros((2, 32), dtype=torch.uint8, device='cuda')
    with pytest.raises(EngineError):
        eng.expand_matrix_xof(sd, 4, 7681, 13, shake_bits=512)
    with pytest.raises(EngineError):
        eng.expand_matrix_xof(sd, 4, 7681, 13, skip=6)                    # a matrix entry is a 32-bit word of the stream
    with pytest.raises(EngineError):
        eng.sample_xof('binomial', sd, 6, 0, 1024)                        # the reference's block buffer holds eta <= 5
    with pytest.raises(EngineError):
        eng.sample_xof('small', sd, 5, 4, 1024)                           # two values a byte
    assert eng.expand_matrix_xof(sd[:0], 4, 7681, 13).shape[0] == 0       # empty batches are no-ops
