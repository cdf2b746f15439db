"""CPU: the oracle's Kyber CPA enc/dec (oracle/sc_oracle_kyber.c) against the reference's own outputs
(tests/golden/kyber_ref.txt, generator oracle/ref_driver.c), both parameter sets in the fixture."""
import numpy as np
import pytest

from tests.golden_io import load_golden
from tests import oracle_lib as O


@pytest.mark.parametrize('tag,nmsg', [('kyber1', 6), ('kyber2', 2), ('kyber0', 2)])
def test_enc_dec_vs_reference(tag, nmsg):
    G = load_golden('kyber_ref.txt')
    par = G[f'{tag}_params']
    for i in range(nmsg):
        u, v = O.kyber_enc(par, G[f'{tag}_t_ntt'], G[f'{tag}_Agen'], G[f'{tag}_m{i}'], G[f'{tag}_noise{i}'])
        assert np.array_equal(u, G[f'{tag}_u{i}']) and np.array_equal(v, G[f'{tag}_v{i}']), (tag, i)
        m = O.kyber_dec(par, G[f'{tag}_s_ntt'], u, v)
        assert np.array_equal(m, G[f'{tag}_dec{i}']) and np.array_equal(m, G[f'{tag}_m{i}'])


@pytest.mark.parametrize('tag,nmsg', [('kyber1', 6), ('kyber2', 2), ('kyber0', 2)])
def test_kem_vs_reference(tag, nmsg):
    """safecrypto_encapsulation / safecrypto_decapsulation: ciphertext bytes and shared keys for the same key pair and m"""
    G = load_golden('kyber_ref.txt')
    par = G[f'{tag}_params']
    u8 = lambda a: bytes(np.asarray(a, dtype=np.uint8))
    for i in range(nmsg):
        ct, key = O.kyber_kem_enc(par, G[f'{tag}_t'], G[f'{tag}_t_ntt'], G[f'{tag}_rho'], G[f'{tag}_Agen'], G[f'{tag}_kem_m{i}'])
        assert ct == u8(G[f'{tag}_kem_ct{i}']) and key == u8(G[f'{tag}_kem_key{i}']), (tag, i)
        args = (par, G[f'{tag}_s_ntt'], G[f'{tag}_t'], G[f'{tag}_t_ntt'], G[f'{tag}_rho'], G[f'{tag}_Agen'])
        assert O.kyber_kem_dec(*args, ct) == u8(G[f'{tag}_kem_dec{i}']) == key
        bad = bytearray(ct); bad[len(bad) // 2] ^= 0x20
        assert O.kyber_kem_dec(*args, bytes(bad)) == u8(G[f'{tag}_kem_decbad{i}'])


@pytest.mark.parametrize('tag', ['kyber1', 'kyber2', 'kyber0'])
def test_keygen_vs_reference(tag):
    """safecrypto_keygen with the context CSPRNG seeded: every component of the key pair; and the matrix expansion from rho"""
    G = load_golden('kyber_ref.txt')
    par = G[f'{tag}_params']
    for i in range(2):
        s_ntt, t, t_ntt, rho, z = O.kyber_keygen(par, G[f'{tag}_kg_seed{i}'])
        assert np.array_equal(rho, G[f'{tag}_kg_rho{i}']) and np.array_equal(z, G[f'{tag}_kg_z{i}'])
        assert np.array_equal(s_ntt, G[f'{tag}_kg_s_ntt{i}'])
        assert np.array_equal(t, G[f'{tag}_kg_t{i}']) and np.array_equal(t_ntt, G[f'{tag}_kg_t_ntt{i}'])
    k = int(par[5])
    assert np.array_equal(O.expand_matrix(G[f'{tag}_rho'], k * k, int(par[1]), int(par[2])), G[f'{tag}_Agen'])
