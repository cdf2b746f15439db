import sys, os, time, json, statistics
sys.path.insert(0, '/root/repo')
import numpy as np, torch
from libsafecrypto_amd import Engine
from tests.golden_io import load_golden
eng = Engine(0)
G = load_golden('bliss_ref.txt'); B = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
g = torch.Generator(device='cuda'); g.manual_seed(3)
msgs = torch.randint(0, 256, (B, 64), dtype=torch.uint8, device='cuda', generator=g)
tapes = torch.randint(-2**31, 2**31 - 1, (B, 4 * 2048 + 64), dtype=torch.int32, device='cuda', generator=g)
seeds = torch.randint(0, 256, (B, 32), dtype=torch.uint8, device='cuda', generator=g)
for spec in (1, 0, 2, 3):
    b = eng.bliss(G['bliss1_params']); b.set_key(G['bliss1_f'], G['bliss1_g'], G['bliss1_a']); b.set_speculation(spec)
    def med(f, n=15):
        f(); torch.cuda.synchronize(); ts = []
        for _ in range(n):
            t0 = time.perf_counter(); f(); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
        return statistics.median(ts) * 1e3
    print(json.dumps({'batch': B, 'speculation': spec, 'tape_ms': med(lambda: b.sign(msgs, tapes, fixed_len=64)), 'seeded_ms': med(lambda: b.sign_seeded(msgs, seeds, fixed_len=64))}))
    b.close()
