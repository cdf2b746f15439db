"""CPU: the oracle's ENS / DLP signatures WITH MESSAGE RECOVERY (orc_ensdlp_sign_recovery / _verify_recovery) against the reference's own
(ens_dlp_sig_sign_recovery / ens_dlp_sig_verify_recovery, src/schemes/sig/ens_dlp/ens_dlp_sig.c:1319-1501, 1663-1826): signature and remainder m2
byte for byte given the same PRNG word tape, and the verifier's verdicts and rebuilt messages on the reference's signatures and on tampered ones
(tests/golden/ensdlp_rec_ref.txt, generator oracle/ref_driver.c: ensdlp_rec_case).  Message lengths lie on both sides of the bits that travel
inside the signature - including the length at which the reference loses the message's last bits and rejects its own signature."""
import numpy as np
import pytest

from tests.golden_io import load_golden
from tests import oracle_lib as O

CASES = ['dlprec0', 'ensrec0']


@pytest.fixture(scope='module')
def G():
    return load_golden('ensdlp_rec_ref.txt')


def _b(x):
    return bytes(np.asarray(x, dtype=np.uint8))


@pytest.mark.parametrize('tag', CASES)
def test_signatures_and_remainders_match_reference(G, tag):
    par = G[f'{tag}_params']
    n, q = int(par[0]), int(par[1])
    b, bgs, inv = O.ensdlp_basis(n, q, G[f'{tag}_fgFG'])
    for m in range(8):
        tape = np.asarray(G[f'{tag}_tape{m}'], dtype=np.int64).astype(np.uint32)
        rc, sig, m2 = O.ensdlp_sign_recovery(par, b, bgs, inv, _b(G[f'{tag}_msg{m}']), tape)
        assert rc == 0 and sig == _b(G[f'{tag}_sig{m}']), m
        assert m2 == _b(G[f'{tag}_m2_{m}']), m
        assert O.ensdlp_sign_recovery(par, b, bgs, inv, _b(G[f'{tag}_msg{m}']), tape[:-1])[0] == 2


@pytest.mark.parametrize('tag', CASES)
def test_verdicts_and_rebuilt_messages_match_reference(G, tag):
    par, h = G[f'{tag}_params'], G[f'{tag}_h']
    m1 = O.ensdlp_rec_m1_bytes(par)
    accepted = 0
    for m in range(8):
        msg, sig, m2 = _b(G[f'{tag}_msg{m}']), bytearray(_b(G[f'{tag}_sig{m}'])), bytearray(_b(G[f'{tag}_m2_{m}']))
        v = [int(x) for x in G[f'{tag}_verify'][3 * m:3 * m + 3]]
        rc, back = O.ensdlp_verify_recovery(par, h, m2, sig)
        assert rc == v[0], m
        assert back == _b(G[f'{tag}_back{m}']), m
        if rc == 0:
            accepted += 1
            assert back == (msg + bytes(m1))[:max(m1, len(msg))]          # the message, zero-extended to the bits the signature carries
        sig[len(sig) - 9] ^= 0x04
        assert O.ensdlp_verify_recovery(par, h, m2, sig)[0] == v[1], m
        sig[len(sig) - 9] ^= 0x04
        if m2:
            m2[0] ^= 0x80
        else:
            sig[2] ^= 0x40
        assert O.ensdlp_verify_recovery(par, h, m2, sig)[0] == v[2], m
    # how many the reference accepted is a property of the key it drew (a key can make it reject every one of its own signatures with recovery).  The fixtures are
    # seeded (tests/test_golden_reproducible.py) and their keys accept: a regenerated file whose key rejects everything would leave the accept + message-recovery
    # branch above untested - so the floor is asserted against the committed file (dlprec0 8 / 8, ensrec0 7 / 8)
    assert accepted == sum(1 for m in range(8) if int(G[f'{tag}_verify'][3 * m]) == 0)
    assert accepted >= 7, (tag, accepted)


def test_short_signature_is_refused(G):
    par, h = G['dlprec0_params'], G['dlprec0_h']
    assert O.ensdlp_verify_recovery(par, h, b'', _b(G['dlprec0_sig0'])[:-1])[0] == 1
