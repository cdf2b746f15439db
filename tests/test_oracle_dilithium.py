"""CPU: the oracle's Dilithium sign/verify (oracle/sc_oracle_dilithium.c) against signatures the
reference itself produced (tests/golden/dilithium_ref.txt, generator oracle/ref_driver.c)."""
import hashlib

import numpy as np
import pytest

from tests.golden_io import load_golden
from tests import oracle_lib as O


@pytest.fixture(scope='module')
def G():
    return load_golden('dilithium_ref.txt')


def msgs(G, tag):
    i = 0
    while f'{tag}_msg{i}' in G:
        yield i, bytes(G[f'{tag}_msg{i}'].astype(np.uint8)), bytes(G[f'{tag}_sig{i}'].astype(np.uint8))
        i += 1


def test_sponge_matches_hashlib_with_the_standard_chi():
    for n in (0, 1, 135, 136, 137, 167, 168, 169, 1000):
        d = bytes((i * 7 + 3) & 0xFF for i in range(n))
        assert O.shake(128, d, 300, standard=True) == hashlib.shake_128(d).digest(300)
        assert O.shake(256, d, 300, standard=True) == hashlib.shake_256(d).digest(300)


def test_reference_permutation_known_answer():
    """the reference's SHAKE256 (its unrolled chi quirk) of a 50-byte packed (rho, t1) record, captured
    from the reference build (oracle/_ref) - NOT the FIPS 202 value"""
    msg = bytes.fromhex('01080f161d242b323940474e555c636a71787f868d949ba2a9b0b7bec5ccd3da'
                        '02890866240a817ede7ec72775abce636fd6')
    out = O.shake(256, msg, 64)
    assert out[:48].hex() == ('ba1789c93f9d0640b0f5e41a6f59c1310f643847fc10afeb65ee638e67610bf9'
                              'dba6b1551cd5eac169615555dbf22a67')
    assert out[48:].hex() == 'c94eb5663bb46cb7cb7941a8565f8788'
    assert out != hashlib.shake_256(msg).digest(64)


@pytest.mark.parametrize('tag', ['dil3', 'dil1', 'dil0', 'dil2'])
def test_key_consistency(G, tag):
    """t1 of the public key is decompose(t) and tr = CRH(rho, t1) (dilithium.c keygen, module_lwe.c:140-180)"""
    key = O.DilKey.from_golden(G, tag)
    d, q = int(key.par[5]), key.q
    t1, t0 = np.empty_like(key.t), np.empty_like(key.t)
    O.lib().orc_dil_decompose(t1, t0, key.t, key.t.size, d, q)
    assert np.array_equal(t1, key.t1)
    assert np.array_equal(G[f'{tag}_rho'], G[f'{tag}_pub_rho'])
    tr = np.zeros(48, dtype=np.uint8)
    O.lib().orc_dilithium_crh_t1(key.rho, key.t1, key.t1.size, int(key.par[2]) - d, tr)
    assert np.array_equal(tr, key.tr)


@pytest.mark.parametrize('tag', ['dil3', 'dil1', 'dil0', 'dil2'])
def test_sign_reproduces_reference_signatures(G, tag):
    key = O.DilKey.from_golden(G, tag)
    for i, m, sig in msgs(G, tag):
        got, trials = O.dilithium_sign(key, m)
        assert got == sig, (tag, i)
        assert trials == int(G[f'{tag}_trials'][i])


@pytest.mark.parametrize('tag', ['dil3', 'dil1', 'dil0', 'dil2'])
def test_verify_matches_reference_verdicts(G, tag):
    key = O.DilKey.from_golden(G, tag)
    for i, m, sig in msgs(G, tag):
        assert O.dilithium_verify(key, m, sig) == int(G[f'{tag}_verify'][2 * i]) == 0
        bad = bytearray(sig)
        bad[len(bad) // 2] ^= 0x10
        assert O.dilithium_verify(key, m, bytes(bad)) == int(G[f'{tag}_verify'][2 * i + 1]) == 1
        assert O.dilithium_verify(key, m + b'x', sig) == 1


@pytest.mark.parametrize('tag', ['dil3', 'dil1', 'dil0', 'dil2'])
def test_keygen_vs_reference(tag):
    """safecrypto_keygen with the context CSPRNG seeded: s1, s2, t, rho, K, tr, t1 of the reference's key pair"""
    G = load_golden('dilithium_ref.txt')
    for i in range(2):
        out = O.dilithium_keygen(G[f'{tag}_params'], int(G[f'{tag}_eta_bits'][0]), G[f'{tag}_kg_seed{i}'])
        for got, name in zip(out, ('s1_', 's2_', 't', 'rho', 'K', 'tr', 't1_')):
            assert np.array_equal(got, G[f'{tag}_kg_{name}{i}']), (tag, i, name)
