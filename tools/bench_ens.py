"""ENS-KEM batched encapsulation / decapsulation throughput on the GPU box."""
import sys, os, time, json
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import torch
from libsafecrypto_amd import Engine
from tests.golden_io import load_golden


def run(tag='ens0', batch=65536, reps=3):
    G = load_golden('ens_ref.txt')
    eng = Engine(0)
    k = eng.ens(G[f'{tag}_params'])
    k.set_pubkey(G[f'{tag}_h']); k.set_privkey(G[f'{tag}_g'], G[f'{tag}_inv_g'])
    g = torch.Generator(device='cuda'); g.manual_seed(17)
    W = 1024 if k.n == 512 else 2048
    tapes = torch.randint(-2**31, 2**31 - 1, (batch, W), dtype=torch.int32, device='cuda', generator=g)
    k.encaps(tapes); torch.cuda.synchronize()
    te, td = [], []
    for _ in range(reps):
        t0 = time.perf_counter(); ct, keys, used, st = k.encaps(tapes); torch.cuda.synchronize(); te.append(time.perf_counter() - t0)
        t0 = time.perf_counter(); k2 = k.decaps(ct); torch.cuda.synchronize(); td.append(time.perf_counter() - t0)
    agree = int((k2 == keys).all(dim=1).sum())
    res = {'set': tag, 'n': k.n, 'batch': batch, 'encaps_ops_per_s': batch / min(te), 'decaps_ops_per_s': batch / min(td), 'encaps_ms': min(te) * 1e3, 'decaps_ms': min(td) * 1e3,
           'mean_words': float(used.float().mean()), 'keys_agree': agree}
    k.close(); eng.close()
    return res


if __name__ == '__main__':
    for tag in ('ens0', 'ens2'):
        for bsz in ([int(sys.argv[1])] if len(sys.argv) > 1 else (16384, 65536)):
            print(json.dumps(run(tag, bsz)), flush=True)
