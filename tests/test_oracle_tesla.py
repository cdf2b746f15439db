"""CPU: the oracle's ring-TESLA sign/verify (oracle/sc_oracle_tesla.c) against signatures the reference itself produced from
the same (key, message, random tape) (tests/golden/tesla_ref.txt, generator oracle/ref_driver.c)."""
import numpy as np
import pytest

from tests.golden_io import load_golden
from tests import oracle_lib as O


@pytest.fixture(scope='module')
def G():
    return load_golden('tesla_ref.txt')


@pytest.mark.parametrize('tag,nmsg', [('tesla0', 8), ('tesla1', 5)])
def test_signatures_match_reference(G, tag, nmsg):
    key = O.TeslaKey.from_golden(G, tag)
    for m in range(nmsg):
        msg = bytes(np.asarray(G[f'{tag}_msg{m}'], dtype=np.uint8))
        ref = bytes(np.asarray(G[f'{tag}_sig{m}'], dtype=np.uint8))
        rc, sig, trials, used = O.tesla_sign(key, msg, G[f'{tag}_tape{m}'])
        assert rc == 0 and sig == ref, m
        assert trials == int(G[f'{tag}_trials'][m]) and used == len(G[f'{tag}_tape{m}']) == trials * key.n
        assert O.tesla_verify(key, msg, ref) == int(G[f'{tag}_verify'][2 * m]) == 0
        bad = bytearray(ref); bad[len(bad) // 3] ^= 0x08
        # the reference's verdict on this altered z is reproduced, whatever it is: its oracle input does not depend on w1, w2 (see
        # sc_oracle_tesla.c), so the change is only caught when it pushes the coefficient outside the rejection bound
        assert O.tesla_verify(key, msg, bytes(bad)) == int(G[f'{tag}_verify'][2 * m + 1])
        bad = bytearray(ref); bad[-1] ^= 1                       # the digest itself
        assert O.tesla_verify(key, msg, bytes(bad)) == 1
        bad = bytearray(ref); bad[0] ^= 0x40                      # z[0] far outside the rejection bound
        assert O.tesla_verify(key, msg, bytes(bad)) == 1
        assert O.tesla_verify(key, msg + b'x', ref) == 1


def test_tape_exhaustion_is_reported(G):
    key = O.TeslaKey.from_golden(G, 'tesla0')
    m = int(np.argmax(G['tesla0_trials']))
    msg = bytes(np.asarray(G[f'tesla0_msg{m}'], dtype=np.uint8))
    rc, _, trials, used = O.tesla_sign(key, msg, G[f'tesla0_tape{m}'][:key.n * 2])
    assert rc == 2 and trials == 2 and used == 2 * key.n
