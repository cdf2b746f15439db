"""Two (or more) torch.distributed ranks (gloo) of the real engine on device 0: every rank signs / transforms its shard of one global batch with its
own Engine(0) context and writes rank<r>.npz.  `launch` starts one child process per rank with RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT set - what
`python -m torch.distributed.run` does for bench.py - and exits with the worst of their return codes; the launcher itself never touches the GPU.
usage: python tests/two_rank_job.py launch <world> <port> <outdir>      (tests/test_gpu_multi.py)"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def rank_main(outdir):
    import numpy as np
    import torch
    import torch.distributed as dist
    from tests.golden_io import load_golden
    from libsafecrypto_amd import Engine
    from libsafecrypto_amd.shard import shard_range, max_over_ranks
    rank, world = int(os.environ['RANK']), int(os.environ['WORLD_SIZE'])
    dist.init_process_group('gloo', rank=rank, world_size=world)
    G = load_golden('dilithium_ref.txt')
    k = {x: G[f'dil3_{x}'] for x in ('params', 's1', 's2', 't', 'K', 'tr', 'Ahat', 't1', 'rho')}
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


if __name__ == '__main__':
    if sys.argv[1] == 'launch':
        world, port, outdir = int(sys.argv[2]), sys.argv[3], sys.argv[4]
        procs = []
        for r in range(world):
            env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(world), MASTER_ADDR='127.0.0.1', MASTER_PORT=port)
            procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__), 'rank', outdir], env=env))
        sys.exit(max(abs(p.wait()) for p in procs))
    rank_main(sys.argv[2])
