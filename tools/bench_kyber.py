"""Kyber set 1 CPA / KEM throughput on the GPU box."""
import sys, os, time, json
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import torch
from libsafecrypto_amd import Engine
from tests.golden_io import load_golden


def run(batch=65536, reps=3):
    G = load_golden('kyber_ref.txt')
    eng = Engine(0)
    ky = eng.kyber(G['kyber1_params']); ky.set_pubkey(G['kyber1_t_ntt'], G['kyber1_Agen']); ky.set_privkey(G['kyber1_s_ntt'])
    ky.set_kem_pubkey(G['kyber1_t'], G['kyber1_rho'])
    g = torch.Generator(device='cuda'); g.manual_seed(19)
    m = torch.randint(0, 256, (batch, 32), dtype=torch.uint8, device='cuda', generator=g)
    ct, keys = ky.encaps(m); torch.cuda.synchronize()
    te, td = [], []
    for _ in range(reps):
        t0 = time.perf_counter(); ct, keys = ky.encaps(m); torch.cuda.synchronize(); te.append(time.perf_counter() - t0)
        t0 = time.perf_counter(); k2 = ky.decaps(ct); torch.cuda.synchronize(); td.append(time.perf_counter() - t0)
    assert torch.equal(k2, keys)
    res = {'batch': batch, 'encaps_ops_per_s': batch / min(te), 'decaps_ops_per_s': batch / min(td), 'encaps_ms': min(te) * 1e3, 'decaps_ms': min(td) * 1e3}
    ky.close(); eng.close()
    return res


if __name__ == '__main__':
    for bsz in ([int(sys.argv[1])] if len(sys.argv) > 1 else (4096, 65536, 262144)):
        print(json.dumps(run(bsz)), flush=True)
