"""CPU: the oracle's ENS-KEM encapsulation / decapsulation (oracle/sc_oracle_ens.c) against ciphertexts and keys the reference
itself produced under its own key pairs from the same CSPRNG words (tests/golden/ens_ref.txt, generator oracle/ref_driver.c)."""
import numpy as np
import pytest

from tests.golden_io import load_golden
from tests import oracle_lib as O

CASES = [('ens0', 5), ('ens2', 3), ('ens1', 3), ('ens3', 3)]


@pytest.fixture(scope='module')
def G():
    return load_golden('ens_ref.txt')


@pytest.mark.parametrize('tag,nmsg', CASES)
def test_encaps_decaps_match_reference(G, tag, nmsg):
    E = O.Ens(G[f'{tag}_params'])
    h, g, ig = G[f'{tag}_h'], G[f'{tag}_g'], G[f'{tag}_inv_g']
    for m in range(nmsg):
        tape = G[f'{tag}_tape{m}']
        ref_ct = bytes(np.asarray(G[f'{tag}_ct{m}'], dtype=np.uint8)); ref_key = bytes(np.asarray(G[f'{tag}_key{m}'], dtype=np.uint8))
        rc, ct, key, used = E.encaps(h, tape)
        assert rc == 0 and used == len(tape) and ct == ref_ct and key == ref_key, m
        rc, k2 = E.decaps(g, ig, ref_ct)
        assert rc == 0 and k2 == bytes(np.asarray(G[f'{tag}_dec{m}'], dtype=np.uint8)) == ref_key, m
        rc, k3 = E.decaps(g, ig, bytes(np.asarray(G[f'{tag}_randct{m}'], dtype=np.uint8)))
        assert rc == 0 and k3 == bytes(np.asarray(G[f'{tag}_randdec{m}'], dtype=np.uint8)), m
    assert E.encaps(h, tape[:300])[0] == 2
    assert E.decaps(g, ig, ref_ct[:-1])[0] == 1
