"""Dilithium-III verification with a distinct public key per item (sc_hip_dilithium_verify_keys) against the one-key path; development aid.
usage: python tools/bench_dil_keys.py [batch=16384]"""
import os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from libsafecrypto_amd import Engine
from tests.golden_io import load_golden

batch = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
G = load_golden('dilithium_ref.txt')
par, eb = G['dil3_params'], int(G['dil3_eta_bits'][0])
eng = Engine(0)
d = eng.dilithium(par)
g = torch.Generator(device='cuda'); g.manual_seed(1)
seeds = torch.randint(0, 256, (batch, 32), dtype=torch.uint8, device='cuda', generator=g)
t0 = time.perf_counter(); s1, s2, t, rho, K, tr, t1 = d.keygen(seeds, eb); torch.cuda.synchronize(); tk = time.perf_counter() - t0
k, l = int(par[3]), int(par[4])
A = eng.expand_matrix(rho[:1], k * l, int(par[1]), int(par[2]))
h = [x[0].cpu().numpy().reshape(-1) for x in (s1, s2, t, K, tr, t1, rho)]
d.set_privkey(h[0], h[1], h[2], h[3], h[4], A.cpu().numpy().reshape(-1)); d.set_pubkey(h[5], h[6], A.cpu().numpy().reshape(-1))
msgs = torch.randint(0, 256, (batch, 64), dtype=torch.uint8, device='cuda', generator=g)
sigs, siglens, _ = d.sign(msgs, fixed_len=64)
# every item is given key 0's public key in its own slot: the distinct-key path does not know (or care) that they coincide
t1i = t1[:1].reshape(1, -1).repeat(batch, 1).contiguous(); rhoi = rho[:1].repeat(batch, 1).contiguous()
for name, fn in (('one key', lambda: d.verify(msgs, sigs, siglens, fixed_len=64)), ('key per item', lambda: d.verify_keys(t1i, rhoi, msgs, sigs, siglens, fixed_len=64))):
    v = fn(); torch.cuda.synchronize(); assert int(v.sum()) == 0
    ts = []
    for _ in range(5):
        t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    print(f'dilithium-III verify, {name}: batch {batch}, {min(ts) * 1e3:.2f} ms, {batch / min(ts):,.0f} verify/s')
for name, fn in (('one key', lambda: d.sign(msgs, fixed_len=64)), ('key per item', lambda: d.sign_keys(s1, s2, t, rho, K, tr, msgs, fixed_len=64))):
    fn(); torch.cuda.synchronize()
    ts = []
    for _ in range(3):
        t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    print(f'dilithium-III sign, {name}: batch {batch}, {min(ts) * 1e3:.2f} ms, {batch / min(ts):,.0f} sign/s')
print(f'(keygen of {batch} keys: {tk * 1e3:.1f} ms)')
