"""GPU parity for the device CTR-DRBG (C ABI: sc_hip_drbg_mem / sc_hip_drbg_words): the reference's own prng_mem / prng_32
streams (tests/golden/drbg_ref.txt) and the pinned oracle on a seeded batch."""
import numpy as np
import pytest

from tests.golden_io import load_golden
from tests import oracle_lib as O

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def eng():
    from libsafecrypto_amd import Engine
    e = Engine(0)
    yield e
    e.close()


@pytest.fixture(scope='module')
def G():
    return load_golden('drbg_ref.txt')


def stream_from_mem_calls(G, c):
    """the reference recorded prng_mem calls of assorted lengths; each consumed ceil(len/64)*64 stream bytes"""
    ref = bytes(np.asarray(G[f'drbg{c}_mem'], dtype=np.uint8))
    pieces, off, pos = [], 0, 0
    for n in (int(x) for x in G['drbg_mem_calls']):
        pieces.append((pos, ref[off:off + n])); off += n; pos += (n + 63) // 64 * 64
    return pieces, pos


@pytest.mark.parametrize('c', range(5))
def test_streams_match_reference(eng, G, c):
    import torch
    seed = np.asarray(G[f'drbg{c}_seed'], dtype=np.uint8)
    pieces, total = stream_from_mem_calls(G, c)
    out = eng.drbg_mem(torch.from_numpy(seed[None, :].copy()).cuda(), total).cpu().numpy()[0].tobytes()
    for pos, want in pieces:
        assert out[pos:pos + len(want)] == want
    if f'drbg{c}_words' in G:
        ref = (np.asarray(G[f'drbg{c}_words']).astype(np.int64) & 0xFFFFFFFF).astype(np.uint32)
        w = eng.drbg_words(torch.from_numpy(seed[None, :].copy()).cuda(), 4216).cpu().numpy()[0].view(np.uint32)
        assert np.array_equal(w, ref)


def test_batch_vs_oracle(eng):
    import torch
    rng = np.random.default_rng(31)
    B = 700
    seeds = rng.integers(0, 256, size=(B, 32), dtype=np.uint8)
    out = eng.drbg_mem(torch.from_numpy(seeds).cuda(), 1792).cpu().numpy()
    words = eng.drbg_words(torch.from_numpy(seeds).cuda(), 64).cpu().numpy().view(np.uint32)
    for i in (0, 1, 63, 64, 255, 256, 699):
        d = O.Drbg(seeds[i].tobytes()); assert d.mem(1792) == out[i].tobytes(); d.close()
        d = O.Drbg(seeds[i].tobytes()); assert [d.u32() for _ in range(64)] == words[i].tolist(); d.close()


def test_matrix_expansion_matches_reference(eng):
    """the public matrices of the Kyber and Dilithium fixtures, regenerated on the device from the 32-byte rho"""
    import torch
    K = load_golden('kyber_ref.txt')
    D = load_golden('dilithium_ref.txt')
    for tag in ('kyber1', 'kyber2'):
        par = K[f'{tag}_params']
        k = int(par[5])
        rho = torch.from_numpy(np.asarray(K[f'{tag}_rho'], dtype=np.uint8)[None, :].copy()).cuda()
        A = eng.expand_matrix(rho, k * k, int(par[1]), int(par[2])).cpu().numpy().reshape(-1)
        assert np.array_equal(A, np.asarray(K[f'{tag}_Agen'], dtype=np.int32)), tag
    tags = sorted({n.split('_')[0] for n in D if n.endswith('_Ahat')})
    assert tags
    rhos = np.stack([np.asarray(D[f'{t}_rho'], dtype=np.uint8) for t in tags])
    for i, t in enumerate(tags):
        par = D[f'{t}_params']
        npoly = int(par[3]) * int(par[4])
        A = eng.expand_matrix(torch.from_numpy(rhos).cuda(), npoly, int(par[1]), int(par[2])).cpu().numpy()
        assert np.array_equal(A[i].reshape(-1), np.asarray(D[f'{t}_Ahat'], dtype=np.int32)), t
