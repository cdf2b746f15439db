"""CDT sampler kernel alone: GB/s of algorithmic traffic (12 B per sample) for a few sigmas / counts.  usage: python tools/bench_cdt.py"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from libsafecrypto_amd import Engine

eng = Engine(0)
g = torch.Generator(device='cuda'); g.manual_seed(5)
for tail, sigma in ((13.42, 215.0), (13.42, 271.0), (12.0, 100.0), (13.42, 19.53), (10.0, 1.7)):
    s = eng.sampler(tail, sigma)
    for cnt in (1 << 24, 1 << 27):
        u = torch.randint(-2**63, 2**63 - 1, (cnt,), dtype=torch.int64, device='cuda', generator=g)
        o = s.sample(u); torch.cuda.synchronize()
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ev0.record()
        for _ in range(10):
            s.sample(u)
        ev1.record(); torch.cuda.synchronize()
        ms = ev0.elapsed_time(ev1) / 10
        print(f'tail {tail} sigma {sigma} count 2^{cnt.bit_length() - 1}: {ms * 1e3:8.1f} us  {cnt / ms / 1e6:7.1f} G samples/s  {cnt * 12 / ms / 1e6:7.0f} GB/s', flush=True)
        del u, o
    s.close()
