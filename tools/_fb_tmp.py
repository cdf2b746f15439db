import sys, json, os, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import numpy as np, torch
import libsafecrypto_amd.engine as E
if len(sys.argv) > 1:
    E.load_library(sys.argv[1])
from libsafecrypto_amd import Engine
from tests.golden_io import load_golden
G = load_golden('falcon_ref.txt')
eng = Engine(0)
for tag in ('falcon2', 'falcon0'):
    f = eng.falcon(G[f'{tag}_params'])
    f.set_privkey(np.asarray(G[f'{tag}_sk'], dtype=np.int64).view(np.float64))
    n = f.n
    for batch in (64, 8192):
        g = torch.Generator(device='cuda'); g.manual_seed(13)
        c = torch.randint(0, 12289, (batch, n), dtype=torch.int32, device='cuda', generator=g)
        tapes = torch.randint(-2**31, 2**31 - 1, (batch, 4 * n), dtype=torch.int32, device='cuda', generator=g)
        f.sample(c, tapes); torch.cuda.synchronize()
        ts = []
        for _ in range(3):
            t0 = time.perf_counter(); f.sample(c, tapes); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
        print(tag, batch, 'sample_ms', round(min(ts) * 1e3, 3), flush=True)
