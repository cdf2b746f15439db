"""Per-kernel throughput table on the GPU box (development aid): every op x (q,n) x arithmetic
mode, HIP-event timed on the launch stream; plus a device-to-device copy for the HBM yardstick."""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import torch
from libsafecrypto_amd import Engine

eng = Engine(0)
gen = torch.Generator(device='cuda'); gen.manual_seed(7)
x = torch.empty(64 * 1024 * 1024, dtype=torch.int32, device='cuda'); y = torch.empty_like(x)
for _ in range(3): y.copy_(x)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20): y.copy_(x)
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 20
print(f'copy 256MB: {2 * x.numel() * 4 / ms / 1e6:.0f} GB/s')
del x, y
cfgs = [(7681, 256, 65536), (8380417, 256, 65536), (8380417, 256, 16384 * 11), (12289, 512, 32768), (12289, 1024, 16384), (8399873, 512, 32768)]
if len(sys.argv) > 1 and sys.argv[1] == 'short':
    cfgs = cfgs[:2]
for q, n, batch in cfgs:
    p = eng.plan(q, n)
    a = torch.randint(0, q, (batch, n), dtype=torch.int32, device='cuda', generator=gen)
    b = torch.randint(0, q, (batch, n), dtype=torch.int32, device='cuda', generator=gen)
    o = torch.empty_like(a)
    for mode in ([1, 2] if q < (1 << 23) else [2]):
        p.set_arith(mode)
        for op, name, nbytes in [(0, 'fwd', 8), (1, 'inv', 8), (2, 'mulntt', 12), (3, 'mul', 12)]:
            p.time_op(op, a, b, o, 5)
            ms = p.time_op(op, a, b, o, 30)
            gbs = batch * n * nbytes / ms / 1e6
            print(f'q={q:8d} n={n:4d} batch={batch:6d} mode={mode} {name:7s} {ms*1e3:8.1f} us  {gbs:7.0f} GB/s  {batch/ms/1e3:8.1f} Mpoly/s')
    p.close()
s = eng.sampler(13.42, 215.0)
cnt = 16384 * 1024
u = torch.randint(-2**63, 2**63 - 1, (cnt,), dtype=torch.int64, device='cuda', generator=gen)
o = s.sample(u)
e0.record()
for _ in range(10): s.sample(u, out=o)
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 10
print(f'cdt sigma=215 count={cnt}: {ms*1e3:.1f} us  {cnt*12/ms/1e6:.0f} GB/s  {cnt/ms/1e6:.1f} Gsamples/s')
