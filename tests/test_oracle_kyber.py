"""CPU: the oracle's Kyber CPA enc/dec (oracle/sc_oracle_kyber.c) against the reference's own outputs
(tests/golden/kyber_ref.txt, generator oracle/ref_driver.c), both parameter sets in the fixture."""
import numpy as np
import pytest

from tests.golden_io import load_golden
from tests import oracle_lib as O


@pytest.mark.parametrize('tag,nmsg', [('kyber1', 6), ('kyber2', 2)])
def test_enc_dec_vs_reference(tag, nmsg):
    G = load_golden('kyber_ref.txt')
    par = G[f'{tag}_params']
    for i in range(nmsg):
        u, v = O.kyber_enc(par, G[f'{tag}_t_ntt'], G[f'{tag}_Agen'], G[f'{tag}_m{i}'], G[f'{tag}_noise{i}'])
        assert np.array_equal(u, G[f'{tag}_u{i}']) and np.array_equal(v, G[f'{tag}_v{i}']), (tag, i)
        m = O.kyber_dec(par, G[f'{tag}_s_ntt'], u, v)
        assert np.array_equal(m, G[f'{tag}_dec{i}']) and np.array_equal(m, G[f'{tag}_m{i}'])
