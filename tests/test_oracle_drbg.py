"""CPU: the oracle's AES-256 CTR-DRBG (oracle/sc_oracle_drbg.c) against byte / word streams the reference's own prng produced
for the same seeds (tests/golden/drbg_ref.txt, generator oracle/ref_driver.c) and against the FIPS-197 AES-256 example."""
import numpy as np
import pytest

from tests.golden_io import load_golden
from tests import oracle_lib as O


@pytest.fixture(scope='module')
def G():
    return load_golden('drbg_ref.txt')


def test_aes256_fips197_example():
    key = np.arange(32, dtype=np.uint8)
    pt = np.frombuffer(bytes.fromhex('00112233445566778899aabbccddeeff'), dtype=np.uint8).copy()
    ct = np.zeros(16, dtype=np.uint8)
    O.lib().orc_aes256_ecb(key, pt, ct)
    assert bytes(ct).hex() == '8ea2b7ca516745bfeafc49904b496089'          # FIPS-197 appendix C.3


@pytest.mark.parametrize('c', range(5))
def test_prng_mem_stream(G, c):
    seed = bytes(np.asarray(G[f'drbg{c}_seed'], dtype=np.uint8))
    d = O.Drbg(seed)
    got = b''.join(d.mem(int(n)) for n in G['drbg_mem_calls'])
    d.close()
    assert got == bytes(np.asarray(G[f'drbg{c}_mem'], dtype=np.uint8))


@pytest.mark.parametrize('c', [1, 4])
def test_prng_32_64_stream(G, c):
    seed = bytes(np.asarray(G[f'drbg{c}_seed'], dtype=np.uint8))
    ref = (np.asarray(G[f'drbg{c}_words']).astype(np.int64) & 0xFFFFFFFF).astype(np.uint32)
    d = O.Drbg(seed)
    got = [d.u32() for _ in range(4200)]
    for _ in range(8):
        x = d.u64()
        got += [x >> 32, x & 0xFFFFFFFF]
    d.close()
    assert np.array_equal(np.array(got, dtype=np.uint32), ref)
