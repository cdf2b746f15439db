"""DLP-IBE batched extract throughput on the GPU box."""
import sys, os, time, json
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import numpy as np
import torch
from libsafecrypto_amd import Engine
from tests.golden_io import load_golden

G = load_golden('dlp_ref.txt')
par = np.asarray(G['dlp0_params'])
n, q, q_bits, n_bits, set_, ubits = (int(x) for x in par[:6])
eng = Engine(0)
f = eng.falcon(np.array([n, q, q_bits, n_bits, 0, 0, 0, 0], dtype=np.int32))
f.set_privkey(np.asarray(G['dlp0_tree'], dtype=np.int64).view(np.float64))
g = torch.Generator(device='cuda'); g.manual_seed(21)
for B in ([int(sys.argv[1])] if len(sys.argv) > 1 else (8192, 32768, 131072)):
    ids = torch.randint(0, 256, (B, 16), dtype=torch.uint8, device='cuda', generator=g)
    tapes = torch.randint(-2**31, 2**31 - 1, (B, 4 * n), dtype=torch.int32, device='cuda', generator=g)
    f.dlp_extract(ubits, ids, tapes, fixed_len=16); torch.cuda.synchronize()
    ts = []
    for _ in range(3):
        t0 = time.perf_counter(); sk, st = f.dlp_extract(ubits, ids, tapes, fixed_len=16); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    assert int((st == 1).sum()) == B
    print(json.dumps({'set': set_, 'n': n, 'q': q, 'batch': B, 'extract_ops_per_s': B / min(ts), 'extract_ms': min(ts) * 1e3}), flush=True)
    del ids, tapes, sk
