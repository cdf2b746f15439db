"""end-to-end (seeded) against tape-resident signing, one scheme: python3 tools/bench_seeded.py falcon|bliss|tesla [batch]"""
import sys, os, time, json, statistics
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import numpy as np
import torch
from libsafecrypto_amd import Engine
from tests.golden_io import load_golden
which = sys.argv[1]
eng = Engine(0)
g = torch.Generator(device='cuda'); g.manual_seed(5)
if which == 'falcon':
    G = load_golden('falcon_ref.txt'); B = int(sys.argv[2]) if len(sys.argv) > 2 else 8192
    h = eng.falcon(G['falcon2_params']); h.set_privkey(np.asarray(G['falcon2_sk'], dtype=np.int64).view(np.float64)); W = 4 * h.n
elif which == 'bliss':
    G = load_golden('bliss_ref.txt'); B = int(sys.argv[2]) if len(sys.argv) > 2 else 16384
    h = eng.bliss(G['bliss1_params']); h.set_key(G['bliss1_f'], G['bliss1_g'], G['bliss1_a']); W = 4 * 2048 + 64
else:
    G = load_golden('tesla_ref.txt'); B = int(sys.argv[2]) if len(sys.argv) > 2 else 16384
    h = eng.tesla(G['tesla0_params']); h.set_key(*(G[f'tesla0_{x}'] for x in ('a1', 'a2', 's', 'e1', 'e2', 't1', 't2'))); W = 24 * 512
msgs = torch.randint(0, 256, (B, 64), dtype=torch.uint8, device='cuda', generator=g)
seeds = torch.randint(0, 256, (B, 32), dtype=torch.uint8, device='cuda', generator=g)
tapes = eng.drbg_words(seeds, W)
def med(f, n=9):
    f(); torch.cuda.synchronize(); ts = []
    for _ in range(n):
        t0 = time.perf_counter(); f(); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    return statistics.median(ts)
tt = med(lambda: h.sign(msgs, tapes, fixed_len=64)); ts = med(lambda: h.sign_seeded(msgs, seeds, fixed_len=64)); td = med(lambda: eng.drbg_words(seeds, W))
print(json.dumps({'scheme': which, 'batch': B, 'tape_resident_ms': tt * 1e3, 'seeded_ms': ts * 1e3, 'drbg_full_tape_ms': td * 1e3, 'tape_resident_ops_per_s': B / tt, 'seeded_ops_per_s': B / ts}))
