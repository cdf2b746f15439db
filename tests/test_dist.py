"""CPU, world_size 2, gloo: the N > 1 path of bench.py - contiguous sharding of independent
polynomials with no data-path collective, barrier + MAX-over-ranks timing."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from libsafecrypto_amd.shard import shard_range, max_over_ranks


def test_shard_range_tiles_exactly():
    for total in (0, 1, 7, 16384, 65536, 65537):
        for world in (1, 2, 3, 4, 8):
            cuts = [shard_range(total, r, world) for r in range(world)]
            assert cuts[0][0] == 0 and cuts[-1][1] == total
            assert all(cuts[i][1] == cuts[i + 1][0] for i in range(world - 1))
            sizes = [hi - lo for lo, hi in cuts]
            assert max(sizes) - min(sizes) <= 1


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q, n, batch, outdir):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from tests import oracle_lib as O
    w, r = O.tables(q, n)
    rng = np.random.default_rng(42)                      # every rank derives the same global batch
    a = rng.integers(0, q, size=(batch, n)).astype(np.int32)
    bhat = rng.integers(0, q, size=(batch, n)).astype(np.int32)
    lo, hi = shard_range(batch, rank, world)
    dist.barrier()
    mine = O.polymul_ntt(a[lo:hi], bhat[lo:hi], q, w, r)   # the unit of work (the oracle stands in for the GPU here)
    dist.barrier()
    t = max_over_ranks(0.5 + rank, dist)
    assert t == 0.5 + (world - 1)
    np.save(os.path.join(outdir, f'part{rank}.npy'), mine)
    np.save(os.path.join(outdir, f'range{rank}.npy'), np.array([lo, hi]))
    dist.destroy_process_group()


def test_two_rank_sharding_matches_single_process(tmp_path):
    q, n, batch, world = 7681, 256, 203, 2
    port = _free_port()
    mp.spawn(_worker, args=(world, port, q, n, batch, str(tmp_path)), nprocs=world, join=True)
    from tests import oracle_lib as O
    w, r = O.tables(q, n)
    rng = np.random.default_rng(42)
    a = rng.integers(0, q, size=(batch, n)).astype(np.int32)
    bhat = rng.integers(0, q, size=(batch, n)).astype(np.int32)
    full = O.polymul_ntt(a, bhat, q, w, r)
    parts = [np.load(tmp_path / f'part{r_}.npy') for r_ in range(world)]
    assert np.array_equal(np.concatenate(parts), full)     # host-side concatenation, no collective
    rng_ = [tuple(np.load(tmp_path / f'range{r_}.npy')) for r_ in range(world)]
    assert rng_ == [shard_range(batch, r_, world) for r_ in range(world)]


def test_library_shard_range_is_the_same_rule():
    """sc_hip_shard_range (the C driver's cut, host-only) tiles a batch exactly as shard.py's rule does"""
    from libsafecrypto_amd import engine
    for total in (0, 1, 7, 203, 16384, 65537):
        for world in (1, 2, 3, 8):
            for r in range(world):
                lo, hi = shard_range(total, r, world)
                assert engine.shard_range(total, world, r) == (lo, hi - lo)
    assert engine.shard_range(10, 0, 0) == (0, 0) and engine.shard_range(10, 2, 5) == (0, 0)     # nonsense arguments: an empty slice


def _share_worker(rank, world, port, outdir):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    import bench
    rows = []
    for total in (65536, 16384, 8192, 203):
        for shard in (False, True):
            mine = bench.batch_share(total, world, rank, shard)
            t = torch.tensor([mine], dtype=torch.int64)
            dist.all_reduce(t, op=dist.ReduceOp.SUM)               # what bench.py's `value` counts: the units all ranks processed
            rows.append((total, int(shard), mine, int(t.item())))
    np.save(os.path.join(outdir, f'share{rank}.npy'), np.array(rows))
    dist.destroy_process_group()


def test_bench_batch_shares_weak_and_strong(tmp_path):
    """bench.py --gpus N [--shard]: weak scaling repeats the BASELINE batch on every rank, --shard cuts it - the per-rank batches the timed legs run on add
    up to N x batch and to batch, and the slices differ by at most one unit"""
    world = 2
    mp.spawn(_share_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    per = [np.load(tmp_path / f'share{r_}.npy') for r_ in range(world)]
    for row0, row1 in zip(*per):
        total, shard, m0, s0 = (int(x) for x in row0)
        m1, s1 = int(row1[2]), int(row1[3])
        assert s0 == s1 == (total if shard else total * world)
        assert m0 + m1 == s0 and abs(m0 - m1) <= 1
