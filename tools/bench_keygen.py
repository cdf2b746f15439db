"""Batched key generation throughput (Kyber set 1, Dilithium-III) on the GPU box."""
import sys, os, time, json
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import torch
from libsafecrypto_amd import Engine
from tests.golden_io import load_golden

eng = Engine(0)
g = torch.Generator(device='cuda'); g.manual_seed(23)
K = load_golden('kyber_ref.txt'); D = load_golden('dilithium_ref.txt')
ky = eng.kyber(K['kyber1_params']); dl = eng.dilithium(D['dil3_params'])
for B in (1024, 16384):
    seeds = torch.randint(0, 256, (B, 32), dtype=torch.uint8, device='cuda', generator=g)
    ky.keygen(seeds); dl.keygen(seeds, int(D['dil3_eta_bits'][0])); torch.cuda.synchronize()
    tk, td = [], []
    for _ in range(3):
        t0 = time.perf_counter(); ky.keygen(seeds); torch.cuda.synchronize(); tk.append(time.perf_counter() - t0)
        t0 = time.perf_counter(); dl.keygen(seeds, int(D['dil3_eta_bits'][0])); torch.cuda.synchronize(); td.append(time.perf_counter() - t0)
    print(json.dumps({'batch': B, 'kyber1_keygen_per_s': B / min(tk), 'dilithium3_keygen_per_s': B / min(td)}), flush=True)
