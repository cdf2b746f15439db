"""GPU: the N > 1 path of the REAL engine on a one-GPU box.
(1) the in-library multi-device driver (sc_hip_multi_*: one host thread per worker, contiguous slices, results written at the slice's offset)
    with two and three workers on device 0 against one-device results;
(2) two torch.distributed ranks (gloo) each with its own Engine(0) context signing / transforming its shard, concatenated by rank order and
    compared with the single-rank result - the shape bench.py --gpus N has, minus RCCL."""
import os
import socket

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


def _free_port():
    s = socket.socket(); s.bind(('127.0.0.1', 0)); p = s.getsockname()[1]; s.close()
    return p


def _rank(rank, world, port, outdir):
    import torch
    import torch.distributed as dist
    os.environ['MASTER_ADDR'] = '127.0.0.1'; os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from libsafecrypto_amd import Engine
    from libsafecrypto_amd.shard import max_over_ranks
    G = load_golden('dilithium_ref.txt')
    k = _dil_key(G)
    eng = Engine(0)                                                  # every rank its own context (bench.py: Engine(local_rank))
    d = eng.dilithium(k['params'])
    d.set_privkey(k['s1'], k['s2'], k['t'], k['K'], k['tr'], k['Ahat']); d.set_pubkey(k['t1'], k['rho'], k['Ahat'])
    rng = np.random.default_rng(7)                                   # the same global batch on every rank
    msgs = rng.integers(0, 256, size=(515, 64), dtype=np.uint8)
    a = rng.integers(0, 8380417, size=(333, 256)).astype(np.int32)
    lo, hi = shard_range(515, rank, world)
    plo, phi = shard_range(333, rank, world)
    dist.barrier()
    mine = torch.from_numpy(msgs[lo:hi]).cuda()
    sigs, siglens, trials = d.sign(mine, fixed_len=64)
    verd = d.verify(mine, sigs, siglens, fixed_len=64)
    p = eng.plan(8380417, 256)
    f = p.fwd(torch.from_numpy(a[plo:phi]).cuda())
    torch.cuda.synchronize()
    dist.barrier()
    assert max_over_ranks(1.0 + rank, dist) == float(world)
    np.savez(os.path.join(outdir, f'rank{rank}.npz'), sigs=sigs.cpu().numpy(), siglens=siglens.cpu().numpy(), verd=verd.cpu().numpy(), f=f.cpu().numpy())
    p.close(); d.close(); eng.close()
    dist.destroy_process_group()


def test_two_ranks_of_the_engine_concatenate_to_the_single_rank_result(tmp_path):
    import torch
    import torch.multiprocessing as mp
    world = 2
    mp.spawn(_rank, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    parts = [np.load(tmp_path / f'rank{r}.npz') for r in range(world)]
    from libsafecrypto_amd import Engine
    G = load_golden('dilithium_ref.txt')
    k = _dil_key(G)
    eng = Engine(0)
    d = eng.dilithium(k['params'])
    d.set_privkey(k['s1'], k['s2'], k['t'], k['K'], k['tr'], k['Ahat']); d.set_pubkey(k['t1'], k['rho'], k['Ahat'])
    rng = np.random.default_rng(7)
    msgs = rng.integers(0, 256, size=(515, 64), dtype=np.uint8)
    a = rng.integers(0, 8380417, size=(333, 256)).astype(np.int32)
    sigs, siglens, _ = d.sign(torch.from_numpy(msgs).cuda(), fixed_len=64)
    assert np.array_equal(np.concatenate([x['sigs'] for x in parts]), sigs.cpu().numpy())
    assert np.array_equal(np.concatenate([x['siglens'] for x in parts]), siglens.cpu().numpy())
    assert (np.concatenate([x['verd'] for x in parts]) == 0).all()
    w, r = O.tables(8380417, 256)
    assert np.array_equal(np.concatenate([x['f'] for x in parts]), O.fwd_ntt(a, 8380417, w))
    d.close(); eng.close()
