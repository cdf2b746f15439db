"""GPU: two torch.distributed ranks (gloo) each with its own Engine(0) context signing / transforming its shard, concatenated by rank order and compared
with the single-rank result - the shape `bench.py --gpus N` has, minus RCCL.

This file sorts first on purpose.  The ranks are started the way the driver starts bench.py - by a launcher process of their own (tests/two_rank_job.py:
one child per rank, RANK / WORLD_SIZE / MASTER_* in the environment) - and they run BEFORE this process has opened the GPU.  With the ranks started from
a process that already held a GPU context (and large pageable copies in flight later in the session) the session was lost to a GPU memory access fault
in about one run out of three on the GPU pool; run first, the ranks have the GPU to themselves, and this process opens it afterwards for the
single-rank reference.  (The fault itself is the runtime's in-place pinning of pageable copies - tests/conftest.py turns that off for the session.)"""
import os
import socket

import numpy as np
import pytest

from tests.golden_io import load_golden
from tests import oracle_lib as O

pytestmark = pytest.mark.gpu


def _dil_key(G, tag='dil3'):
    return {x: G[f'{tag}_{x}'] for x in ('params', 's1', 's2', 't', 'K', 'tr', 'Ahat', 't1', 'rho')}


def _free_port():
    s = socket.socket(); s.bind(('127.0.0.1', 0)); p = s.getsockname()[1]; s.close()
    return p


def test_two_ranks_of_the_engine_concatenate_to_the_single_rank_result(tmp_path):
    """the ranks are started the way the driver starts bench.py - by a launcher process of their own (tests/two_rank_job.py, one child per rank, RANK /
    WORLD_SIZE / MASTER_* in the environment) - not forked out of this process, which has the GPU open"""
    import subprocess, sys, torch
    world = 2
    job = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'two_rank_job.py')
    r = subprocess.run([sys.executable, job, 'launch', str(world), str(_free_port()), str(tmp_path)], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
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
