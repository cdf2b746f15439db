"""GPU parity for the CDT Gaussian sampler (gaussian_cdf.c:536-659 behaviour)."""
import numpy as np
import pytest

from tests.golden_io import load_golden
from tests import oracle_lib as O

pytestmark = pytest.mark.gpu

CASES = [('bliss1_s215_t13.42', 13.42, 215.0), ('unit_s100_t12', 12.0, 100.0), ('leaf_s1.7_t10', 10.0, 1.7)]


@pytest.fixture(scope='module')
def eng():
    from libsafecrypto_amd import Engine
    e = Engine(0)
    yield e
    e.close()


def _live_reference(tmp_path):
    """re-run the reference binary (oracle/_ref, built from /root/reference in the dev container,
    travels as a binary) on THIS host; None if it is not there"""
    import os, subprocess
    drv = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'oracle', '_ref', 'ref_driver')
    if not os.path.exists(drv):
        return None
    subprocess.check_call([drv, 'cdt', str(tmp_path)])
    return load_golden(os.path.join(str(tmp_path), 'cdt_ref.txt'))


@pytest.mark.parametrize('tag,tail,sigma', CASES)
def test_samples_vs_reference_fixture(eng, tag, tail, sigma):
    """(table, uniform word) -> sample is host independent: the reference's fixture table is
    uploaded as is and the reference's samples must come back bit for bit"""
    import torch
    G = load_golden('cdt_ref.txt')
    s = eng.sampler(table=G[f'{tag}_cdf'])
    u = G[f'{tag}_uniform']
    du = torch.from_numpy(u.view(np.int64)).cuda()
    assert np.array_equal(s.sample(du).cpu().numpy(), G[f'{tag}_samples'])
    assert np.array_equal(s.sample(du, centre=7).cpu().numpy(), G[f'{tag}_samples'] + 7)
    assert np.array_equal(s.sample_host(u), G[f'{tag}_samples'])
    s.close()


@pytest.mark.parametrize('tag,tail,sigma', CASES)
def test_table_vs_same_host_reference(eng, tag, tail, sigma, tmp_path):
    """the table is long-double expl() arithmetic, whose last bit follows the host CPU's x87
    (the committed fixture was made on another CPU): compare with the oracle on this host,
    with the reference binary re-run on this host when present, and - within 2 units - with
    the fixture"""
    G = load_golden('cdt_ref.txt')
    s = eng.sampler(tail, sigma)
    t = s.table()
    assert np.array_equal(t, O.cdt_table(tail, sigma))
    live = _live_reference(tmp_path)
    if live is not None:
        assert np.array_equal(t, live[f'{tag}_cdf'])
        import torch
        u = live[f'{tag}_uniform']
        assert np.array_equal(s.sample(torch.from_numpy(u.view(np.int64)).cuda()).cpu().numpy(), live[f'{tag}_samples'])
    diff = np.abs(t.astype(object) - G[f'{tag}_cdf'].astype(object))
    assert diff.max() <= 2
    s.close()


@pytest.mark.parametrize('tag,tail,sigma', CASES)
def test_32_bit_precision_sampler(eng, tag, tail, sigma):
    """create_sampler(..., SAMPLING_32BIT, ...): the reference's 32-bit table uploaded as is gives its samples bit for bit; the table the engine builds equals
    the oracle's on this host; a large ragged batch with the edge words against the oracle's search; the two precisions' handles do not mix"""
    import torch
    G = load_golden('cdt32_ref.txt')
    tab = (np.asarray(G[f'{tag}_tab32']).astype(np.int64) & 0xFFFFFFFF).astype(np.uint32)
    u = (np.asarray(G[f'{tag}_words32']).astype(np.int64) & 0xFFFFFFFF).astype(np.uint32)
    s = eng.sampler32(table=tab)
    du = torch.from_numpy(u.view(np.int32)).cuda()
    assert np.array_equal(s.sample(du).cpu().numpy(), G[f'{tag}_samples'])
    assert np.array_equal(s.sample(du, centre=-3).cpu().numpy(), G[f'{tag}_samples'] - 3)
    assert np.array_equal(s.table(), tab)
    rng = np.random.default_rng(32)
    big = rng.integers(0, 2**32, size=1_000_003, dtype=np.uint64).astype(np.uint32)
    edges = np.array([0, 1, 2**32 - 1, 2**32 - 2] + [int(tab[i]) + d for i in (1, 2, tab.size // 2, tab.size - 2) for d in (-1, 0, 1) if 0 <= int(tab[i]) + d < 2**32], dtype=np.uint32)
    big[:edges.size] = edges
    got = s.sample(torch.from_numpy(big.view(np.int32)).cuda()).cpu().numpy()
    assert np.array_equal(got, O.cdt32_sample(big, tab))
    w64 = torch.zeros(8, dtype=torch.int64, device='cuda'); o = torch.zeros(8, dtype=torch.int32, device='cuda')
    assert eng.L.sc_hip_cdt_sample(s.h, 8, w64.data_ptr(), 0, o.data_ptr(), None) != 0      # a 32-bit handle refuses the 64-bit entry
    s.close()
    s2 = eng.sampler32(tail, sigma)
    assert np.array_equal(s2.table(), O.cdt32_table(tail, sigma))
    s2.close()


def test_large_batch_vs_oracle_and_edges(eng):
    import torch
    tail, sigma = 13.42, 215.0
    s = eng.sampler(tail, sigma)
    cdf = O.cdt_table(tail, sigma)
    rng = np.random.default_rng(11)
    u = rng.integers(0, 2**64 - 1, size=16384 * 1024 // 8 + 5, dtype=np.uint64, endpoint=True)
    # edge words: 0, 1, all-ones, exact table entries and their neighbours (strict '<' matters)
    edges = [0, 1, 2**64 - 1, 2**64 - 2] + [int(cdf[i]) + d for i in (1, 2, 100, 4094) for d in (-1, 0, 1)]
    u[:len(edges)] = np.array(edges, dtype=np.uint64)
    got = s.sample(torch.from_numpy(u.view(np.int64)).cuda()).cpu().numpy()
    assert np.array_equal(got, O.cdt_sample(u, cdf))
    assert abs(float(got.std()) - sigma) < 1.0 and abs(float(got.mean())) < 0.5
    s.close()


@pytest.mark.parametrize('tail,sigma', [(13.42, 19.53), (13.42, 271.0), (13.42, 250.0), (10.0, 1.3), (10.0, 6.3), (12.0, 30.0), (13.0, 3.2)])
def test_guide_layouts_vs_oracle(eng, tail, sigma):
    """the sampler's guide (csrc/cdt_guide.h) picks a different bin layout and probe count per table - one probe for sigma ~ 20, three for the BLISS
    sigmas, the runtime-count loop for the narrow Falcon-leaf-like tables, the closing run of equal entries where the table has one: every table entry
    and its two neighbours, the extremes, the words just below / above each power of two of the tail probability, and a random batch, against the
    oracle's halving search on the table built on this host"""
    import torch
    s = eng.sampler(tail, sigma)
    cdf = O.cdt_table(tail, sigma)
    assert np.array_equal(s.table(), cdf)
    rng = np.random.default_rng(int(sigma * 100))
    words = [0, 1, 2, 2**64 - 1, 2**64 - 2, 2**63, 2**63 - 1, 2**63 + 1]
    for e in cdf.tolist():
        words += [max(e - 1, 0), e, min(e + 1, 2**64 - 1)]
    for k in range(1, 64):
        words += [2**64 - (1 << k), 2**64 - (1 << k) - 1, 2**64 - (1 << k) + 1, (1 << k) - 1, 1 << k]
    u = np.concatenate([np.array(words, dtype=np.uint64), rng.integers(0, 2**64 - 1, size=1 << 18, dtype=np.uint64, endpoint=True)])
    got = s.sample(torch.from_numpy(u.view(np.int64)).cuda()).cpu().numpy()
    assert np.array_equal(got, O.cdt_sample(u, cdf))
    assert np.array_equal(s.sample(torch.from_numpy(u[1:].copy().view(np.int64)).cuda(), centre=-3).cpu().numpy(), O.cdt_sample(u[1:], cdf) - 3)   # the unaligned path
    s.close()
