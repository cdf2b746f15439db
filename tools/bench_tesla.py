"""ring-TESLA batched sign / verify throughput on the GPU box."""
import sys, os, time, json
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import torch
from libsafecrypto_amd import Engine
from tests.golden_io import load_golden


def run(batch=16384, reps=3):
    G = load_golden('tesla_ref.txt')
    eng = Engine(0)
    ts = eng.tesla(G['tesla0_params'])
    ts.set_key(*(G[f'tesla0_{x}'] for x in ('a1', 'a2', 's', 'e1', 'e2', 't1', 't2')))
    g = torch.Generator(device='cuda'); g.manual_seed(17)
    W = 40 * 512
    msgs = torch.randint(0, 256, (batch, 64), dtype=torch.uint8, device='cuda', generator=g)
    tapes = torch.randint(-2**31, 2**31 - 1, (batch, W), dtype=torch.int32, device='cuda', generator=g)
    ts.sign(msgs, tapes, fixed_len=64); torch.cuda.synchronize()
    tsn, tvf = [], []
    for _ in range(reps):
        t0 = time.perf_counter(); sigs, siglens, trials, status = ts.sign(msgs, tapes, fixed_len=64); torch.cuda.synchronize(); tsn.append(time.perf_counter() - t0)
        t0 = time.perf_counter(); v = ts.verify(msgs, sigs, siglens, fixed_len=64); torch.cuda.synchronize(); tvf.append(time.perf_counter() - t0)
    ok = status == 1
    assert int(v[ok].sum()) == 0
    res = {'batch': batch, 'sign_ops_per_s': batch / min(tsn), 'verify_ops_per_s': batch / min(tvf), 'sign_ms': min(tsn) * 1e3, 'verify_ms': min(tvf) * 1e3,
           'mean_trials': float(trials[ok].float().mean()), 'max_trials': int(trials.max()), 'signed': int(ok.sum())}
    ts.close(); eng.close()
    return res


if __name__ == '__main__':
    for bsz in ([int(sys.argv[1])] if len(sys.argv) > 1 else (1024, 16384, 65536)):
        print(json.dumps(run(bsz)), flush=True)
