"""Batched key generation throughput (Kyber set 1, Dilithium-III, BLISS-B-I, RLWE-Enc) on the GPU box."""
import sys, os, time, json
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import torch
from libsafecrypto_amd import Engine
from tests.golden_io import load_golden

eng = Engine(0)
g = torch.Generator(device='cuda'); g.manual_seed(23)
K = load_golden('kyber_ref.txt'); D = load_golden('dilithium_ref.txt')
ky = eng.kyber(K['kyber1_params']); dl = eng.dilithium(D['dil3_params'])
BL = load_golden('bliss_ref.txt'); bl = eng.bliss(BL['bliss1_params'])
RL = load_golden('rlwe_ref.txt'); rl = eng.rlwe(RL['rlwe0_params'])
for B in (1024, 16384):
    seeds = torch.randint(0, 256, (B, 32), dtype=torch.uint8, device='cuda', generator=g)
    ky.keygen(seeds); dl.keygen(seeds, int(D['dil3_eta_bits'][0])); torch.cuda.synchronize()
    tk, td = [], []
    for _ in range(3):
        t0 = time.perf_counter(); ky.keygen(seeds); torch.cuda.synchronize(); tk.append(time.perf_counter() - t0)
        t0 = time.perf_counter(); dl.keygen(seeds, int(D['dil3_eta_bits'][0])); torch.cuda.synchronize(); td.append(time.perf_counter() - t0)
    btapes = torch.randint(-2**31, 2**31 - 1, (B, 1536), dtype=torch.int32, device='cuda', generator=g)
    rtapes = torch.randint(-2**31, 2**31 - 1, (B, rl.keygen_words), dtype=torch.int32, device='cuda', generator=g)
    bl.keygen(BL['bliss1_kg_nz'], btapes); rl.keygen(rtapes); torch.cuda.synchronize()
    tb, tr = [], []
    for _ in range(3):
        t0 = time.perf_counter(); f, gg, a, att, st = bl.keygen(BL['bliss1_kg_nz'], btapes); torch.cuda.synchronize(); tb.append(time.perf_counter() - t0)
        t0 = time.perf_counter(); rl.keygen(rtapes); torch.cuda.synchronize(); tr.append(time.perf_counter() - t0)
    assert int((st == 1).sum()) == B
    print(json.dumps({'batch': B, 'kyber1_keygen_per_s': B / min(tk), 'dilithium3_keygen_per_s': B / min(td), 'bliss1_keygen_per_s': B / min(tb),
                      'bliss1_mean_f_candidates': float(att.float().mean()), 'rlwe0_keygen_per_s': B / min(tr)}), flush=True)
