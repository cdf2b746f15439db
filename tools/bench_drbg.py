"""the device CTR-DRBG alone: python3 tools/bench_drbg.py [batch] [words]  (default: BLISS-B-I's first pair of attempts, 16384 x 2052 words)"""
import sys, os, time, json, statistics
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import torch
from libsafecrypto_amd import Engine
B = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
W = int(sys.argv[2]) if len(sys.argv) > 2 else 2052
eng = Engine(0)
g = torch.Generator(device='cuda'); g.manual_seed(1)
seeds = torch.randint(0, 256, (B, 32), dtype=torch.uint8, device='cuda', generator=g)
eng.drbg_words(seeds, W); torch.cuda.synchronize()
ts = []
for _ in range(9):
    t0 = time.perf_counter(); eng.drbg_words(seeds, W); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
t = statistics.median(ts)
print(json.dumps({'batch': B, 'words': W, 'us': t * 1e6, 'tape_GB_per_s': B * W * 4 / t / 1e9, 'blocks_per_s': B * W / 4 / t}))
