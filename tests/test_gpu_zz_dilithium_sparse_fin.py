"""The Dilithium finishing kernel WITHOUT transforms (dil_cfins_kernel, sc_hip_dilithium_set_tuning bit 11): c s2, c s1, c t0 as sums of shifted table
rows in the coefficient domain instead of NTT(c) and five-to-six inverse transforms per stage.  It was written while the GPU pool was closed to this
build and HAS NOT RUN ON HARDWARE (its text has run on the CPU lane by lane against the oracle: tests/test_emu_dil_cfins.py): it is not the product path (the tuning bit selects it) and this file is skipped unless SC_TEST_UNVALIDATED=1 -
`SC_TEST_UNVALIDATED=1 python -m pytest tests/test_gpu_zz_dilithium_sparse_fin.py -q`, then `python tools/bench_scheme.py dil_sign 9` against
`SC_BENCH_DIL_TUNING=2048 python tools/bench_scheme.py dil_sign 9`, is the first thing to do with a GPU.  Same bar as every other form of the
signer: the reference's signatures byte for byte, attempt counts included."""
import os

import numpy as np
import pytest

from tests.golden_io import load_golden
from tests import oracle_lib as O
from tests.test_gpu_dilithium import make, golden_msgs


def make_sparse(eng, key, tuning):
    """bit 11 is honoured at key load only (the tables are made from the raw key there and no host copy of the key is kept): set it, then load the key"""
    d = eng.dilithium(key.par)
    d.set_tuning(tuning)
    d.set_privkey(key.s1, key.s2, key.t, key.K, key.tr, key.Ahat)
    d.set_pubkey(key.t1, key.rho, key.Ahat)
    return d

pytestmark = [pytest.mark.gpu,
              pytest.mark.skipif(os.environ.get('SC_TEST_UNVALIDATED') != '1', reason='kernel not yet run on hardware: opt in with SC_TEST_UNVALIDATED=1')]

SPARSE = 1 << 11


@pytest.fixture(scope='module')
def eng():
    from libsafecrypto_amd import Engine
    e = Engine(0)
    yield e
    e.close()


@pytest.fixture(scope='module')
def G():
    return load_golden('dilithium_ref.txt')


@pytest.mark.parametrize('tag', ['dil3', 'dil1', 'dil0', 'dil2'])
def test_reference_signatures(eng, G, tag):
    key = O.DilKey.from_golden(G, tag)
    d = make_sparse(eng, key, SPARSE)
    gm = golden_msgs(G, tag)
    sigs, trials = d.sign_host([m for m, _, _ in gm])
    for i, (m, sig, tr) in enumerate(gm):
        assert sigs[i] == sig, (tag, i)
        assert trials[i] == tr
    d.close()


@pytest.mark.parametrize('B', [1, 257, 5000])
def test_same_bytes_as_the_transform_kernel(eng, G, B):
    key = O.DilKey.from_golden(G, 'dil3')
    d = make(eng, key)
    rng = np.random.default_rng(B)
    msgs = [bytes(rng.integers(0, 256, size=int(rng.integers(0, 201)), dtype=np.uint8)) for _ in range(B)]
    want, wt = d.sign_host(msgs)
    with pytest.raises(Exception):                  # a key loaded without the bit: the request is refused, not silently ignored
        d.set_tuning(SPARSE)
    d.close()
    d = make_sparse(eng, key, SPARSE)
    for tuning in (SPARSE, SPARSE | (1 << 9), SPARSE | (1 << 8)):
        d.set_tuning(tuning)
        got, gt = d.sign_host(msgs)
        assert got == want and gt == wt, tuning
    assert d.verify_host(msgs, got) == [0] * B
    d.set_tuning(0)
    d.close()


@pytest.mark.parametrize('tag,B', [('dil3', 1), ('dil3', 1500), ('dil1', 333)])
def test_a_key_per_message_same_bytes(eng, G, tag, B):
    """sc_hip_dilithium_sign_keys with the bit set: every wavefront builds its message's tables from the raw key rows (dil_cfins_kernel<K, L, true>)"""
    import torch
    par, eb = G[f'{tag}_params'], int(G[f'{tag}_eta_bits'][0])
    d = eng.dilithium(par)
    g = torch.Generator(device='cuda'); g.manual_seed(B)
    seeds = torch.randint(0, 256, (B, 32), dtype=torch.uint8, device='cuda', generator=g)
    msgs = torch.randint(0, 256, (B, 64), dtype=torch.uint8, device='cuda', generator=g)
    s1, s2, t, rho, K, tr, t1 = d.keygen(seeds, eb)
    want = d.sign_keys(s1, s2, t, rho, K, tr, msgs, fixed_len=64)
    d.set_tuning(SPARSE)
    got = d.sign_keys(s1, s2, t, rho, K, tr, msgs, fixed_len=64)
    assert all(torch.equal(a, b) for a, b in zip(want, got))
    # a key outside the range the fields are sized for: refused on the device, the call signed again through the transform kernel - same bytes as without the bit
    s2w = s2.clone(); s2w.view(-1)[5] = int(par[10]) + 1
    got_w = d.sign_keys(s1, s2w, t, rho, K, tr, msgs, fixed_len=64)
    d.set_tuning(0)
    want_w = d.sign_keys(s1, s2w, t, rho, K, tr, msgs, fixed_len=64)
    assert all(torch.equal(a, b) for a, b in zip(want_w, got_w))
    d.close()
