import sys, os, time, json, statistics
sys.path.insert(0, '/root/repo')
import numpy as np, torch
from libsafecrypto_amd import Engine
from tests.golden_io import load_golden
eng = Engine(0)
G = load_golden('falcon_ref.txt'); tag = 'falcon2'; B = 8192
f = eng.falcon(G[f'{tag}_params'])
f.set_privkey(np.asarray(G[f'{tag}_sk'], dtype=np.int64).view(np.float64)); f.set_pubkey(G[f'{tag}_h'])
g = torch.Generator(device='cuda'); g.manual_seed(3)
msgs = torch.randint(0, 256, (B, 64), dtype=torch.uint8, device='cuda', generator=g)
tapes = torch.randint(-2**31, 2**31 - 1, (B, 4 * f.n), dtype=torch.int32, device='cuda', generator=g)
s, l, st = f.sign(msgs, tapes, fixed_len=64); torch.cuda.synchronize()
for rep in range(3):
    ts = []
    for _ in range(41):
        t0 = time.perf_counter(); f.verify(msgs, s, l, fixed_len=64); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    ts.sort()
    print(json.dumps({'min_ms': ts[0] * 1e3, 'median_ms': ts[20] * 1e3, 'p90_ms': ts[36] * 1e3}))
