"""Falcon batched tree sampler / sign / verify throughput on the GPU box."""
import sys, os, time, json
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import numpy as np
import torch
from libsafecrypto_amd import Engine
from tests.golden_io import load_golden


def run(set_=0, batch=16384, reps=3, lanes=0, variant=None):
    G = load_golden('falcon_ref.txt')
    tag = f'falcon{set_}'
    eng = Engine(0)
    f = eng.falcon(G[f'{tag}_params'])
    f.set_privkey(np.asarray(G[f'{tag}_sk'], dtype=np.int64).view(np.float64)); f.set_pubkey(G[f'{tag}_h'])
    n = f.n
    f.set_mapping(lanes)
    if variant:
        f.set_variant(*variant)
    g = torch.Generator(device='cuda'); g.manual_seed(13)
    msgs = torch.randint(0, 256, (batch, 64), dtype=torch.uint8, device='cuda', generator=g)
    tapes = torch.randint(-2**31, 2**31 - 1, (batch, 4 * n), dtype=torch.int32, device='cuda', generator=g)
    sigs, siglens, status = f.sign(msgs, tapes, fixed_len=64); torch.cuda.synchronize()
    ts, tv = [], []
    for _ in range(reps):
        t0 = time.perf_counter(); sigs, siglens, status = f.sign(msgs, tapes, fixed_len=64); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
        t0 = time.perf_counter(); v = f.verify(msgs, sigs, siglens, fixed_len=64); torch.cuda.synchronize(); tv.append(time.perf_counter() - t0)
    ok = status == 1
    assert int(v[ok].sum()) == 0
    res = {'set': set_, 'n': n, 'batch': batch, 'lanes': lanes, 'variant': variant, 'sign_ops_per_s': batch / min(ts), 'verify_ops_per_s': batch / min(tv), 'sign_ms': min(ts) * 1e3,
           'verify_ms': min(tv) * 1e3, 'signed': int(ok.sum())}
    f.close(); eng.close()
    return res


if __name__ == '__main__':
    if len(sys.argv) > 1 and sys.argv[1] == 'one':
        print(json.dumps(run(0, 131072, reps=1)), flush=True)
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == 'cfg4':
        print(json.dumps(run(2, 8192, reps=5, lanes=int(sys.argv[2]) if len(sys.argv) > 2 else 0)), flush=True)
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == 'variants':       # the default mapping's four variants and the launcher's choice
        for s, bsz in ((2, 8192), (2, 16384), (2, 32768), (2, 131072), (0, 8192), (0, 16384), (0, 131072)):
            for v in ((8 | 1 << 8, 7), (8 | 2 << 8, 7), (8, 6), (0, 0)):
                print(json.dumps(run(s, bsz, lanes=64, variant=v)), flush=True)
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == 'sweep':          # every mapping at the configuration's batch and at larger ones
        for s, bsz in ((2, 8192), (2, 16384), (2, 32768), (2, 131072), (0, 8192), (0, 131072)):
            for lanes in (64, 8, 1):
                print(json.dumps(run(s, bsz, lanes=lanes)), flush=True)
        sys.exit(0)
    for s, bsz in ((0, 4096), (0, 32768), (0, 131072), (2, 32768), (2, 131072)):
        print(json.dumps(run(s, bsz)), flush=True)
