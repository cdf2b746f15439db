"""CPU: the oracle's Ring-LWE encryption (oracle/sc_oracle_rlwe.c) against key pairs, ciphertexts and decryptions the reference
itself produced from the same random tapes (tests/golden/rlwe_ref.txt, generator oracle/ref_driver.c)."""
import numpy as np
import pytest

from tests.golden_io import load_golden
from tests import oracle_lib as O

CASES = [('rlwe0', 3, 6), ('rlwe1', 2, 4)]


@pytest.fixture(scope='module')
def G():
    return load_golden('rlwe_ref.txt')


@pytest.mark.parametrize('tag,nkeys,nmsg', CASES)
def test_sampler_table_matches_reference(G, tag, nkeys, nmsg):
    R = O.Rlwe(G[f'{tag}_params'])
    assert np.array_equal(R.cdf, np.asarray(G[f'{tag}_cdf'], dtype=np.uint64))


@pytest.mark.parametrize('tag,nkeys,nmsg', CASES)
def test_keygen_matches_reference(G, tag, nkeys, nmsg):
    R = O.Rlwe(G[f'{tag}_params'])
    for k in range(nkeys):
        tape = G[f'{tag}_kg_tape{k}']
        assert len(tape) == R.kg_words
        rc, a, p, r2 = R.keygen(tape)
        assert rc == 0
        assert np.array_equal(a, G[f'{tag}_kg_a{k}']) and np.array_equal(p, G[f'{tag}_kg_p{k}']) and np.array_equal(r2, G[f'{tag}_kg_r2_{k}'])
    assert R.keygen(G[f'{tag}_kg_tape0'][:-1])[0] == 2


@pytest.mark.parametrize('tag,nkeys,nmsg', CASES)
def test_encrypt_decrypt_match_reference(G, tag, nkeys, nmsg):
    R = O.Rlwe(G[f'{tag}_params'])
    k = nkeys - 1                                                 # the key pair the reference had loaded
    a, p, r2 = G[f'{tag}_kg_a{k}'], G[f'{tag}_kg_p{k}'], G[f'{tag}_kg_r2_{k}']
    for m in range(nmsg):
        msg = bytes(np.asarray(G[f'{tag}_msg{m}'], dtype=np.uint8))
        ref = bytes(np.asarray(G[f'{tag}_ct{m}'], dtype=np.uint8))
        tape = G[f'{tag}_tape{m}']
        assert len(tape) == R.enc_words and len(ref) == R.ct_bytes
        rc, ct = R.encrypt(a, p, msg, tape)
        assert rc == 0 and ct == ref, m
        rc, dec = R.decrypt(r2, ref)
        assert rc == 0 and dec == bytes(np.asarray(G[f'{tag}_dec{m}'], dtype=np.uint8)) == msg
        rnd = bytes(np.asarray(G[f'{tag}_randct{m}'], dtype=np.uint8))
        rc, dec = R.decrypt(r2, rnd)
        assert rc == 0 and dec == bytes(np.asarray(G[f'{tag}_randdec{m}'], dtype=np.uint8))
    assert R.encrypt(a, p, msg, tape[:-2])[0] == 2
    assert R.decrypt(r2, ref[:-1])[0] == 1
