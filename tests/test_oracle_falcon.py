"""Pins oracle/sc_oracle_falcon.c against the reference: tests/golden/falcon_ref.txt was written by
oracle/ref_driver.c (the reference's own gaussian_sample_with_tree / safecrypto_sign / safecrypto_verify on a
seeded CSPRNG, with the CSPRNG's word stream recorded as the tape)."""
import numpy as np
import pytest

from tests import oracle_lib as O
from tests.golden_io import load_golden

G = load_golden('falcon_ref.txt')
SETS = [(0, 2, 4), (2, 1, 2)]


@pytest.mark.parametrize('set_,nsamp,nmsg', SETS)
def test_sampler_matches_reference(set_, nsamp, nmsg):
    tag = f'falcon{set_}'
    key = O.FalconKey.from_golden(G, tag)
    for m in range(nsamp):
        rc, s1, s2 = O.falcon_sample(key, G[f'{tag}_c{m}'], G[f'{tag}_ctape{m}'])
        assert rc == 0
        assert np.array_equal(s1, np.asarray(G[f'{tag}_s1_{m}'], dtype=np.int32))
        assert np.array_equal(s2, np.asarray(G[f'{tag}_s2_{m}'], dtype=np.int32))


@pytest.mark.parametrize('set_,nsamp,nmsg', SETS)
def test_signatures_match_reference(set_, nsamp, nmsg):
    tag = f'falcon{set_}'
    key = O.FalconKey.from_golden(G, tag)
    for m in range(nmsg):
        msg = bytes(np.asarray(G[f'{tag}_msg{m}'], dtype=np.uint8))
        rc, sig = O.falcon_sign(key, msg, G[f'{tag}_tape{m}'])
        assert rc == int(G[f'{tag}_status'][m])
        ref = bytes(np.asarray(G[f'{tag}_sig{m}'], dtype=np.uint8))
        assert sig == ref
        assert len(G[f'{tag}_tape{m}']) == 4 * key.n           # two prng_64 per leaf
        assert O.falcon_verify(key, msg, ref) == int(G[f'{tag}_verify'][2 * m]) == 0
        bad = bytearray(ref)
        bad[len(bad) // 3] ^= 0x10
        assert O.falcon_verify(key, msg, bytes(bad)) == int(G[f'{tag}_verify'][2 * m + 1]) == 1


def test_tape_too_short_is_reported():
    key = O.FalconKey.from_golden(G, 'falcon0')
    rc, _, _ = O.falcon_sample(key, G['falcon0_c0'], G['falcon0_ctape0'][:100])
    assert rc == 2


def test_fft_round_trip_and_leaf_sigmas():
    rng = np.random.default_rng(5)
    for logn in (1, 2, 5, 9, 10):
        f = rng.integers(-200, 200, 1 << logn).astype(np.float64)
        g = f.copy()
        O.lib().orc_falcon_fft(g, logn)
        O.lib().orc_falcon_ifft(g, logn)
        assert np.max(np.abs(g - f)) < 1e-9
    key = O.FalconKey.from_golden(G, 'falcon0')
    sg = O.falcon_leaf_sigmas(key)
    assert sg.shape == (512,) and 1.2 < sg.min() and sg.max() < 2.0   # 1.55*sqrt(q)/||b~_i||: tables of <= 32 entries
