"""GPU: the CDT sampler at 128- / 192-bit precision through the C ABI (sc_hip_cdt_hp_*) against the reference's own get_vector_32 -> gaussian_cdf_sample_128 / _192
run over tables of known content (tests/golden/cdt_hp_ref.txt; /root/reference/src/utils/sampling/gaussian_cdf.c:110-184,468-509) and, on seeded tables and words, the oracle."""
import ctypes as C

import numpy as np
import pytest

from tests.golden_io import load_golden
from tests import oracle_lib as O
from tests.test_oracle_cdt_hp import TAGS, case

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def eng():
    from libsafecrypto_amd import Engine
    e = Engine(0)
    yield e
    e.close()


@pytest.fixture(scope='module')
def G():
    return load_golden('cdt_hp_ref.txt')


def dev64(a):
    import torch
    return torch.from_numpy(np.ascontiguousarray(a, dtype=np.uint64).view(np.int64)).cuda()


@pytest.mark.parametrize('tag', TAGS)
def test_reference_samples(eng, G, tag):
    W, tab, x, want, _ = case(G, tag)
    s = eng.sampler_hp(64 * W, tab)
    assert s.size == len(tab) and s.precision == 64 * W and np.array_equal(s.table(), tab)
    assert np.array_equal(s.sample(dev64(x)).cpu().numpy(), want)
    assert np.array_equal(s.sample(dev64(x), centre=-3).cpu().numpy(), want - 3)
    assert np.array_equal(s.sample(dev64(x[:5])).cpu().numpy(), want[:5])            # a ragged tail
    s.close()


@pytest.mark.parametrize('W,size,sigma', [(2, 4096, 215.0), (3, 2048, 100.0), (2, 64, 3.33), (3, 8, 1.2), (2, 2, 0.4)])
def test_seeded_tables_and_extreme_words_vs_oracle(eng, W, size, sigma):
    """every guide layout the table builder picks (wide body, thin tail, a closing run of equal entries, tiny tables: the halving search itself), words that tie with
    entries' top limbs, the smallest and the largest words"""
    rng = np.random.default_rng(size + W)
    steps = np.cumsum(rng.exponential(1.3 / size, size))                            # monotone, passes 1 some way before the end: a closing run of 2^64 - 1
    top = np.where(steps >= 1.0, np.uint64(2**64 - 1), (np.minimum(steps, 0.999999) * 2.0**64).astype(np.uint64))
    top[0] = 0
    tab = rng.integers(0, 2**64, size=(size, W), dtype=np.uint64)
    tab[:, W - 1] = np.sort(top)
    n = 3 * size + 2000
    x = rng.integers(0, 2**64, size=(n, W), dtype=np.uint64)
    x[:size, W - 1] = tab[:, W - 1]                                                 # ties with every entry
    x[size:2 * size, W - 1] = tab[:, W - 1] - np.uint64(1)
    x[2 * size:3 * size, W - 1] = tab[:, W - 1] + np.uint64(1)
    x[-4:, W - 1] = np.array([0, 1, 2**64 - 2, 2**64 - 1], dtype=np.uint64)
    s = eng.sampler_hp(64 * W, tab)
    got = s.sample(dev64(x)).cpu().numpy()
    assert np.array_equal(got, O.cdt_hp_sample(x, tab, W))
    out = np.zeros(n, dtype=np.int32)                                               # host buffers in and out
    assert eng.L.sc_hip_cdt_hp_sample_host(s.h, n, x.ctypes.data_as(C.c_void_p), 5, out.ctypes.data_as(C.c_void_p)) == 0
    assert np.array_equal(out, got + 5)
    s.close()


def test_what_is_refused(eng):
    from libsafecrypto_amd import EngineError
    tab = np.zeros((8, 4), dtype=np.uint64)
    with pytest.raises(EngineError, match='uninitialised'):
        eng.sampler_hp(256, tab)                                                    # gaussian_cdf_sample_256 has no defined result
    with pytest.raises(EngineError):
        eng.sampler_hp(160, np.zeros((8, 2), dtype=np.uint64))
    bad = np.zeros((8, 2), dtype=np.uint64); bad[:, 1] = np.arange(8, dtype=np.uint64)[::-1]
    with pytest.raises(EngineError, match='monotone'):
        eng.sampler_hp(128, bad)
    ok = np.zeros((8, 2), dtype=np.uint64); ok[:, 1] = np.arange(8, dtype=np.uint64); ok[:, 0] = np.arange(8, dtype=np.uint64)[::-1]      # the lower limbs need not be ordered
    s = eng.sampler_hp(128, ok)
    import torch
    with pytest.raises(EngineError):                                                # a 128-bit handle is not served by the 64-bit entry
        if eng.L.sc_hip_cdt_sample(s.h, 1, C.c_void_p(torch.zeros(2, dtype=torch.int64, device='cuda').data_ptr()), 0, C.c_void_p(torch.zeros(1, dtype=torch.int32, device='cuda').data_ptr()), None) != 0:
            raise EngineError(eng.L.sc_hip_last_error().decode())
    s.close()
