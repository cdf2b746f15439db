"""GPU: the N > 1 path of the REAL engine on a one-GPU box.
(1) the in-library multi-device driver (sc_hip_multi_*: one host thread per worker, contiguous slices, results written at the slice's offset)
    with two and three workers on device 0 against one-device results;
(2) two torch.distributed ranks of the real engine: tests/test_gpu_00_two_ranks.py (it has to run before this process opens the GPU)."""
import numpy as np
import pytest

from tests.golden_io import load_golden
from tests import oracle_lib as O
from libsafecrypto_amd.shard import shard_range

pytestmark = pytest.mark.gpu


def _dil_key(G, tag='dil3'):
    return {x: G[f'{tag}_{x}'] for x in ('params', 's1', 's2', 't', 'K', 'tr', 'Ahat', 't1', 'rho')}


@pytest.mark.parametrize('workers', [2, 3])
def test_multi_driver_matches_one_device(workers):
    from libsafecrypto_amd import Engine, Multi
    G = load_golden('dilithium_ref.txt')
    k = _dil_key(G)
    q, n, batch = 7681, 256, 1001                                   # not a multiple of the worker count
    rng = np.random.default_rng(workers)
    a = rng.integers(0, q, size=(batch, n)).astype(np.int32)
    bhat = rng.integers(0, q, size=(batch, n)).astype(np.int32)
    w, r = O.tables(q, n)
    m = Multi([0] * workers)
    assert m.workers == workers
    got = m.polymul_ntt_host(q, a, bhat)
    assert np.array_equal(got, O.polymul_ntt(a, bhat, q, w, r))
    f = m.ntt_host(q, a)
    assert np.array_equal(f, O.fwd_ntt(a, q, w)) and np.array_equal(m.ntt_host(q, f, inverse=True), a)
    # Dilithium-III: 301 messages over the workers == one handle on one device
    msgs = rng.integers(0, 256, size=(301, 64), dtype=np.uint8)
    m.dilithium(k['params'], k['s1'], k['s2'], k['t'], k['K'], k['tr'], k['Ahat'], k['t1'], k['rho'])
    sigs, sl, tr = m.dilithium_sign_host(msgs, 64)
    eng = Engine(0)
    d = eng.dilithium(k['params'])
    d.set_privkey(k['s1'], k['s2'], k['t'], k['K'], k['tr'], k['Ahat']); d.set_pubkey(k['t1'], k['rho'], k['Ahat'])
    one, one_tr = d.sign_host([bytes(x) for x in msgs])
    for i in range(301):
        assert bytes(sigs[i, :sl[i]]) == one[i] and int(tr[i]) == one_tr[i], i
    assert (m.dilithium_verify_host(msgs, 64, sigs, sl) == 0).all()
    bad = sigs.copy(); bad[::3, 100] ^= 1
    v = m.dilithium_verify_host(msgs, 64, bad, sl)
    assert (v[::3] == 1).all() and (np.delete(v, np.s_[::3]) == 0).all()
    d.close(); eng.close(); m.close()
