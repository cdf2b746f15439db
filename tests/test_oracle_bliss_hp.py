"""CPU: the oracle's BLISS-B signer with the sampler at 128- / 192-bit precision (oracle/sc_oracle_bliss.c orc_bliss_sign_hp) against the reference's own signatures
(tests/golden/bliss_hp_ref.txt, oracle/ref_driver.c bliss_hp_case: contexts created with SC_FLAG_0_SAMPLE_128BIT / _192BIT - what test/functional/func_alg_bliss_b.c:118
asks for - whose tables were replaced by ones of known content, recorded in the fixture)."""
import numpy as np
import pytest

from tests.golden_io import load_golden
from tests import oracle_lib as O

CASES = [('blisshp128_1', 7), ('blisshp192_1', 3), ('blisshp128_4', 3)]


@pytest.fixture(scope='module')
def G():
    return load_golden('bliss_hp_ref.txt')


def u32(a):
    return (np.asarray(a).astype(np.int64) & 0xFFFFFFFF).astype(np.uint32)


@pytest.mark.parametrize('tag,nmsg', CASES)
def test_signatures_match_reference(G, tag, nmsg):
    key = O.BlissKey.from_golden(G, tag)
    prec, size = (int(v) for v in G[f'{tag}_precision'])
    W = prec // 64
    table = np.asarray(G[f'{tag}_table'], dtype=np.uint64).reshape(size, W)
    for m in range(nmsg):
        msg = bytes(G[f'{tag}_msg{m}'].astype(np.uint8)); tape = u32(G[f'{tag}_tape{m}'])
        rc, sig, trials, used = O.bliss_sign_hp(key, msg, tape, table, W)
        assert rc == 0 and sig == bytes(G[f'{tag}_sig{m}'].astype(np.uint8)), (tag, m)
        assert trials == int(G[f'{tag}_trials'][m]) and used == tape.size
        assert O.bliss_verify(key, msg, sig) == int(G[f'{tag}_verify'][2 * m]) == 0
        bad = bytearray(sig); bad[len(sig) // 3] ^= 0x04
        assert O.bliss_verify(key, msg, bytes(bad)) == int(G[f'{tag}_verify'][2 * m + 1])
    # a fresh pair of vectors costs 2 n samples of W draws of two words each: the tape of a one-attempt signature
    for one in [m for m in range(nmsg) if int(G[f'{tag}_trials'][m]) == 1][:1]:
        assert G[f'{tag}_tape{one}'].size == 4 * key.n * W + 2
