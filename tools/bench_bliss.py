"""BLISS-B-I batched sign / verify throughput on the GPU box."""
import sys, os, time, json
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import torch
from libsafecrypto_amd import Engine
from tests.golden_io import load_golden


def run(batch=16384, reps=3):
    G = load_golden('bliss_ref.txt')
    eng = Engine(0)
    b = eng.bliss(G['bliss1_params']); b.set_key(G['bliss1_f'], G['bliss1_g'], G['bliss1_a'])
    g = torch.Generator(device='cuda'); g.manual_seed(11)
    W = 4 * 2048 + 64
    msgs = torch.randint(0, 256, (batch, 64), dtype=torch.uint8, device='cuda', generator=g)
    tapes = torch.randint(-2**31, 2**31 - 1, (batch, W), dtype=torch.int32, device='cuda', generator=g)
    sigs, siglens, trials, status = b.sign(msgs, tapes, fixed_len=64); torch.cuda.synchronize()
    ts, tv = [], []
    for _ in range(reps):
        t0 = time.perf_counter(); sigs, siglens, trials, status = b.sign(msgs, tapes, fixed_len=64); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
        t0 = time.perf_counter(); v = b.verify(msgs, sigs, siglens, fixed_len=64); torch.cuda.synchronize(); tv.append(time.perf_counter() - t0)
    ok = status == 1
    assert int(v[ok].sum()) == 0
    res = {'batch': batch, 'sign_ops_per_s': batch / min(ts), 'verify_ops_per_s': batch / min(tv), 'sign_ms': min(ts) * 1e3, 'verify_ms': min(tv) * 1e3,
           'mean_trials': float(trials[ok].float().mean()), 'max_trials': int(trials.max()), 'signed': int(ok.sum())}
    b.close(); eng.close()
    return res


if __name__ == '__main__':
    if len(sys.argv) > 1:
        print(json.dumps(run(int(sys.argv[1]))))
        sys.exit(0)
    for bsz in (1024, 16384, 65536):
        print(json.dumps(run(bsz)))
