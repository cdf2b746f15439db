"""CPU: the oracle's DLP-IBE extract (oracle/sc_oracle_falcon.c: orc_dlp_extract) against user secret keys the reference itself
extracted under its own master key from the same CSPRNG words (tests/golden/dlp_ref.txt, generator oracle/ref_driver.c)."""
import numpy as np
import pytest

from tests.golden_io import load_golden
from tests import oracle_lib as O


@pytest.fixture(scope='module')
def G():
    return load_golden('dlp_ref.txt')


@pytest.mark.parametrize('T', ['dlp0', 'dlp1'])
def test_extract_matches_reference(G, T):
    par = G[T + '_params']
    tree = np.asarray(G[T + '_tree'], dtype=np.int64).view(np.float64)
    for m in range(4):
        ident = bytes(np.asarray(G[f'{T}_id{m}'], dtype=np.uint8))
        rc, sk = O.dlp_extract(par, tree, ident, G[f'{T}_tape{m}'])
        assert rc == 0 and sk == bytes(np.asarray(G[f'{T}_sk{m}'], dtype=np.uint8)), m
    assert O.dlp_extract(par, tree, ident, G[T + '_tape3'][:100])[0] != 0


@pytest.mark.parametrize('T', ['dlp0', 'dlp1'])
def test_encrypt_decrypt_match_reference(G, T):
    par = G[T + '_params']
    for m in range(4):
        ident = bytes(np.asarray(G[f'{T}_id{m}'], dtype=np.uint8))
        msg = bytes(np.asarray(G[f'{T}_msg{m}'], dtype=np.uint8))
        ref = bytes(np.asarray(G[f'{T}_ct{m}'], dtype=np.uint8))
        rc, ct = O.dlp_encrypt(par, G[T + '_h_ntt'], ident, msg, G[f'{T}_etape{m}'])
        assert rc == 0 and ct == ref, m
        rc, dec = O.dlp_decrypt(par, G[f'{T}_uk_ntt{m}'], ref)
        assert rc == 0 and dec == bytes(np.asarray(G[f'{T}_dec{m}'], dtype=np.uint8)) == msg, m
    assert O.dlp_encrypt(par, G[T + '_h_ntt'], ident, msg, G[T + '_etape3'][:50])[0] == 2
    assert O.dlp_decrypt(par, G[T + '_uk_ntt3'], ref[:-1])[0] == 1
