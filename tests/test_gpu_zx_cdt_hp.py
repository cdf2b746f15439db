"""(Sorted behind the rest of the GPU suite on purpose: written in round 6, while the GPU pool was closed - these tests have run against the CPU build of the library only, and a
run that stops at the first failure should have been through everything that HAS run on hardware before it gets here.)

GPU: the CDT sampler at 128- / 192-bit precision through the C ABI (sc_hip_cdt_hp_*) against the reference's own get_vector_32 -> gaussian_cdf_sample_128 / _192
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


@pytest.mark.parametrize('tag', ['hp128_bliss1_s215_t13.42', 'hp192_leaf_s1.7_t10'])
def test_sampler_table_at_128_and_192_bits(tag):
    """create_sampler(..., SAMPLING_128BIT / SAMPLING_192BIT, ...) behind the utils_sampling_t mirror (sc_hip_sampler_create_hp): the reference's table handed over, 2 / 3
    draws of the generator per sample in the reference's order - the vector the reference's own get_vector_32 filled over the same table (tests/golden/cdt_hp_ref.txt)"""
    from libsafecrypto_amd import load_library
    from tests.golden_io import load_golden
    L = load_library()
    G = load_golden('cdt_hp_ref.txt')
    prec, size, ns, _ = (int(v) for v in G[f'{tag}_par'])
    W = prec // 64
    table = np.asarray(G[f'{tag}_table'], dtype=np.uint64)
    words = np.asarray(G[f'{tag}_words'], dtype=np.uint64)
    want = np.asarray(G[f'{tag}_samples'], dtype=np.int32)

    class SAMP(C.Structure):
        _pack_ = 1
    SAMP._fields_ = [('create', C.c_void_p), ('destroy', C.CFUNCTYPE(C.c_int32, C.POINTER(C.c_void_p))), ('get_prng', C.c_void_p), ('sample', C.CFUNCTYPE(C.c_int32, C.c_void_p)),
                     ('vector_16', C.CFUNCTYPE(C.c_int32, C.POINTER(SAMP), C.POINTER(C.c_int16), C.c_size_t, C.c_int32)), ('vector_32', C.CFUNCTYPE(C.c_int32, C.POINTER(SAMP), C.POINTER(C.c_int32), C.c_size_t, C.c_int32)),
                     ('precision', C.c_int32), ('dimension', C.c_int32), ('bootstrapped', C.c_int32), ('tail', C.c_float), ('sigma', C.c_float), ('sigma2', C.c_float),
                     ('gauss', C.c_void_p), ('prng_ctx', C.c_void_p), ('discard', C.c_uint32), ('bootstrap', C.c_void_p)]
    state = {'i': 0}
    RNG = C.CFUNCTYPE(C.c_uint64, C.c_void_p)

    def draw(_ctx):
        v = int(words[state['i']]); state['i'] += 1
        return v
    cb = RNG(draw)
    h = C.c_void_p()
    L.sc_hip_sampler_create_hp.argtypes = [C.c_int, C.c_float, C.c_float, C.c_void_p, C.c_int, RNG, C.c_void_p, C.POINTER(C.c_void_p)]
    assert L.sc_hip_sampler_create_hp(prec, 13.42, 215.0, table.ctypes.data, size, cb, None, C.byref(h)) == 0, L.sc_hip_last_error()
    tab = SAMP()
    L.sc_hip_utils_sampling_fill.argtypes = [C.c_void_p, C.c_void_p]
    assert L.sc_hip_utils_sampling_fill(C.byref(tab), h) == 0 and tab.precision == prec
    n = min(ns - 1, 300)
    v = np.zeros(n, dtype=np.int32)
    assert tab.vector_32(C.byref(tab), v.ctypes.data_as(C.POINTER(C.c_int32)), n, 0) == 0 and state['i'] == n * W
    assert np.array_equal(v, want[:n])
    assert tab.sample(tab.gauss) == int(want[n]) and state['i'] == (n + 1) * W
    L.sc_hip_utils_sampling_fill_sca.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
    assert L.sc_hip_utils_sampling_fill_sca(C.byref(SAMP()), h, 2) != 0               # the shuffle / blinding fillers stay 64-bit
    g = C.c_void_p(tab.gauss)
    assert tab.destroy(C.byref(g)) == 0 and not g.value
