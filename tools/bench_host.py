"""PCIe-inclusive rate of the host-buffer entry points (sc_hip_polymul_ntt_host / sc_hip_ntt_fwd_host): numpy arrays in, numpy arrays out;
development aid.  usage: python tools/bench_host.py [batch=262144]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from libsafecrypto_amd import Engine

batch = int(sys.argv[1]) if len(sys.argv) > 1 else 262144
q, n = 7681, 256
eng = Engine(0)
p = eng.plan(q, n)
rng = np.random.default_rng(1)
a = rng.integers(0, q, size=(batch, n), dtype=np.int32)
b = rng.integers(0, q, size=(batch, n), dtype=np.int32)
bh = p.fwd_host(b)
for name, fn, nbytes in (('fused product', lambda: p.polymul_ntt_host(a, bh), 12 * n), ('forward NTT', lambda: p.fwd_host(a), 8 * n)):
    out = fn()
    ts = []
    for _ in range(3):
        t0 = time.perf_counter(); out = fn(); ts.append(time.perf_counter() - t0)
    print(f'{name}, host buffers, batch {batch}: {min(ts) * 1e3:.1f} ms, {batch / min(ts) / 1e6:.2f} M polynomials/s, {batch * nbytes / min(ts) / 1e9:.1f} GB/s over PCIe (both directions)')
import torch
import ctypes as C
# the same entry point on buffers the caller already pinned (here: torch pinned tensors) and reuses - no registration, no first-touch faults
ap, bp, op = (torch.empty((batch, n), dtype=torch.int32).pin_memory() for _ in range(3))
ap.numpy()[:] = a; bp.numpy()[:] = bh
ptr = lambda t: C.c_void_p(t.data_ptr())
for name, call, nbytes in (('fused product', lambda: p.L.sc_hip_polymul_ntt_host(p.h, batch, ptr(ap), ptr(bp), ptr(op)), 12 * n),
                           ('forward NTT', lambda: p.L.sc_hip_ntt_fwd_host(p.h, batch, ptr(ap), ptr(op)), 8 * n)):
    assert call() == 0
    ts = []
    for _ in range(5):
        t0 = time.perf_counter(); assert call() == 0; ts.append(time.perf_counter() - t0)
    print(f'{name}, caller-pinned host buffers, batch {batch}: {min(ts) * 1e3:.1f} ms, {batch / min(ts) / 1e6:.2f} M polynomials/s, {batch * nbytes / min(ts) / 1e9:.1f} GB/s over PCIe (both directions)')
assert np.array_equal(op.numpy(), p.fwd_host(a))
c = p.polymul_ntt(torch.from_numpy(a[:1000]).cuda(), torch.from_numpy(bh[:1000]).cuda()).cpu().numpy()
assert np.array_equal(c, p.polymul_ntt_host(a, bh)[:1000])
print('host path == device path on a sample')
