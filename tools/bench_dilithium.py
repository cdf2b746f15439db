"""Dilithium-III batched sign / verify throughput on the GPU box (development aid + bench extra)."""
import sys, os, time, json
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import numpy as np
import torch
from libsafecrypto_amd import Engine
from tests.golden_io import load_golden


def run(batch=16384, reps=3, tag='dil3'):
    G = load_golden('dilithium_ref.txt')
    eng = Engine(0)
    d = eng.dilithium(G[f'{tag}_params'])
    d.set_privkey(*(G[f'{tag}_{x}'] for x in ('s1', 's2', 't', 'K', 'tr', 'Ahat')))
    d.set_pubkey(G[f'{tag}_t1'], G[f'{tag}_rho'], G[f'{tag}_Ahat'])
    g = torch.Generator(device='cuda'); g.manual_seed(3)
    msgs = torch.randint(0, 256, (batch, 64), dtype=torch.uint8, device='cuda', generator=g)
    sigs, siglens, trials = d.sign(msgs, fixed_len=64)       # warm-up (allocates the workspace)
    torch.cuda.synchronize()
    ts, tv = [], []
    for _ in range(reps):
        t0 = time.perf_counter(); sigs, siglens, trials = d.sign(msgs, fixed_len=64); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
        t0 = time.perf_counter(); v = d.verify(msgs, sigs, siglens, fixed_len=64); torch.cuda.synchronize(); tv.append(time.perf_counter() - t0)
    assert int(v.sum()) == 0
    res = {'batch': batch, 'sign_ops_per_s': batch / min(ts), 'verify_ops_per_s': batch / min(tv), 'sign_ms': min(ts) * 1e3,
           'verify_ms': min(tv) * 1e3, 'mean_trials': float(trials.float().mean()), 'max_trials': int(trials.max())}
    d.close(); eng.close()
    return res


if __name__ == '__main__':
    if len(sys.argv) > 1:
        print(json.dumps(run(int(sys.argv[1]))))
        sys.exit(0)
    for b in (1024, 16384, 65536):
        print(json.dumps(run(b)))
