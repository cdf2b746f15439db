"""CPU: the oracle's restatement of the sampler's side-channel vector fillers (sampling.c:107-228: shuffle / blinding / sample discarding) against
vectors the reference itself filled from a seeded CSPRNG (tests/golden/sampling_ref.txt, generator oracle/ref_driver.c gen_sampling)."""
import numpy as np
import pytest

from tests.golden_io import load_golden
from tests import oracle_lib as O


@pytest.fixture(scope='module')
def G():
    return load_golden('sampling_ref.txt')


def test_every_variant_matches_the_reference(G):
    cdf_plain, cdf_blind = O.cdt_table(13.42, 215.0), O.cdt_table_blinding(13.42, 215.0)
    seen = set()
    for k in range(int(G['sca_cases'][0])):
        blinding, discard, n, centre = (int(x) for x in G[f'sca{k}_par'])
        v, used = O.sample_vector_sca(n, centre, blinding, discard, G[f'sca{k}_tape'], cdf_blind if blinding == 1 else cdf_plain)
        assert used > 0, (k, 'tape too short')
        assert np.array_equal(v, G[f'sca{k}_v']), (k, blinding, discard)
        seen.add((blinding, discard))
        lo = 2 * n + (0 if blinding == 0 else n - 1) + (2 * n if blinding == 1 else 0)
        assert used >= lo and (discard != 0 or blinding != 0 or used == 2 * n)
    assert len(seen) == 12                                # three fillers x four discard settings


def test_plain_filler_is_the_old_one(G):
    """NORMAL_SAMPLES without discarding is sample_vector_32 as test_oracle.py pins it (cdt_ref.txt)"""
    cdf = O.cdt_table(13.42, 215.0)
    tape = (np.asarray(G['sca0_tape']).astype(np.int64) & 0xFFFFFFFF).astype(np.uint64)
    n = int(G['sca0_par'][2])
    words = (tape[0:2 * n:2] << np.uint64(32)) | tape[1:2 * n:2]
    assert np.array_equal(O.cdt_sample(words, cdf), G['sca0_v'])
