"""CPU: the oracle's ENS / DLP signature verification (orc_ensdlp_verify) against the reference's verdicts on its own signatures and
on tampered copies (tests/golden/ensdlp_ref.txt, generator oracle/ref_driver.c)."""
import numpy as np
import pytest

from tests.golden_io import load_golden
from tests import oracle_lib as O

CASES = [('enssig0', 5), ('dlpsig1', 4)]


@pytest.fixture(scope='module')
def G():
    return load_golden('ensdlp_ref.txt')


@pytest.mark.parametrize('tag,nmsg', CASES)
def test_verdicts_match_reference(G, tag, nmsg):
    par, h = G[f'{tag}_params'], G[f'{tag}_h']
    for m in range(nmsg):
        msg = bytes(np.asarray(G[f'{tag}_msg{m}'], dtype=np.uint8))
        sig = bytearray(np.asarray(G[f'{tag}_sig{m}'], dtype=np.uint8))
        v = [int(x) for x in G[f'{tag}_verify'][3 * m:3 * m + 3]]
        assert O.ensdlp_verify(par, h, msg, bytes(sig)) == v[0] == 0, m
        sig[len(sig) // 2] ^= 0x01
        assert O.ensdlp_verify(par, h, msg, bytes(sig)) == v[1], m
        sig[len(sig) // 2] ^= 0x01; sig[3] ^= 0x80
        assert O.ensdlp_verify(par, h, msg, bytes(sig)) == v[2], m
        sig[3] ^= 0x80
        assert O.ensdlp_verify(par, h, msg + b'x', bytes(sig)) == 1
        assert O.ensdlp_verify(par, h, msg, bytes(sig[:-1])) == 1
