"""GPU: two torch.distributed ranks (gloo) each with its own Engine(0) context signing / transforming its shard, concatenated by rank order and compared
with the single-rank result - the shape `bench.py --gpus N` has, minus RCCL.

The ranks are started the way the driver starts bench.py - by a launcher process of their own (tests/two_rank_job.py: one child per rank, RANK /
WORLD_SIZE / MASTER_* in the environment) - and the single-rank reference is the same job with world = 1, so this process does not touch the GPU for the
comparison and the test does not depend on its place in the session.  (History: with the ranks forked out of a process that already held a GPU context the
session was lost to a GPU memory access fault in about one run out of three in round 2 - profiles/r02_pageable_copy_fault.txt, r03_host_pin_bisect.txt.)"""
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
    WORLD_SIZE / MASTER_* in the environment) - and so is the single-rank reference (world = 1, the same job): this process never opens the GPU for the
    comparison, so the test does not depend on where in the session it runs (the file-name ordering was an unenforced precondition)"""
    import subprocess, sys
    job = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'two_rank_job.py')
    outs = {}
    for world in (2, 1):
        out = tmp_path / f'w{world}'
        out.mkdir()
        r = subprocess.run([sys.executable, job, 'launch', str(world), str(_free_port()), str(out)], capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
        outs[world] = [np.load(out / f'rank{k}.npz') for k in range(world)]
    parts, one = outs[2], outs[1][0]
    assert np.array_equal(np.concatenate([x['sigs'] for x in parts]), one['sigs'])
    assert np.array_equal(np.concatenate([x['siglens'] for x in parts]), one['siglens'])
    assert (np.concatenate([x['verd'] for x in parts]) == 0).all() and (one['verd'] == 0).all()
    rng = np.random.default_rng(7)
    rng.integers(0, 256, size=(515, 64), dtype=np.uint8)                  # (the job draws the messages first)
    a = rng.integers(0, 8380417, size=(333, 256)).astype(np.int32)
    w, r_ = O.tables(8380417, 256)
    assert np.array_equal(np.concatenate([x['f'] for x in parts]), O.fwd_ntt(a, 8380417, w)) and np.array_equal(one['f'], O.fwd_ntt(a, 8380417, w))
