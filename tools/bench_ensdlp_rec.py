"""ENS / DLP signatures with message recovery: signing and verification throughput on the device for the keys in tests/golden/ensdlp_rec_ref.txt; development aid.
usage: python3 tools/bench_ensdlp_rec.py [tag=dlprec0|ensrec0] [batch=16384]"""
import sys, os, time, json, statistics
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import numpy as np, torch
from libsafecrypto_amd import Engine
from tests.golden_io import load_golden
tag = sys.argv[1] if len(sys.argv) > 1 else 'ensrec0'
B = int(sys.argv[2]) if len(sys.argv) > 2 else 16384
G = load_golden('ensdlp_rec_ref.txt')
eng = Engine(0)
par = np.asarray(G[f'{tag}_params']); hp = par.copy(); hp[4] = 0
n = int(par[0]); m1 = ((n - int(par[6])) * (int(par[2]) - 1) + 7) // 8
s = eng.ens_dlp_signer(hp); s.set_privkey(G[f'{tag}_fgFG'])
f = eng.falcon(hp); f.set_pubkey(G[f'{tag}_h'])
g = torch.Generator(device='cuda'); g.manual_seed(1)
L = m1 + 64                                                        # 64 bytes travel beside the signature
msgs = torch.randint(0, 256, (B, L), dtype=torch.uint8, device='cuda', generator=g)
tapes = torch.randint(-2**31, 2**31 - 1, (B, 4 * n), dtype=torch.int32, device='cuda', generator=g)
def med(fn, reps=7):
    fn(); torch.cuda.synchronize(); ts = []
    for _ in range(reps):
        t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    return statistics.median(ts)
sigs, m2, m2l, st = s.sign_recovery(msgs, tapes, fixed_len=L)
sl = torch.full((B,), sigs.shape[1], dtype=torch.int32, device='cuda')
ts = med(lambda: s.sign_recovery(msgs, tapes, fixed_len=L))
tv = med(lambda: f.ens_dlp_verify_recovery(sigs, sl, m2, m2l))
v, out, ol = f.ens_dlp_verify_recovery(sigs, sl, m2, m2l)
ok = int((v == 0).sum())
same = bool(torch.equal(out[v == 0][:, :L], msgs[v == 0]))
print(json.dumps({'tag': tag, 'batch': B, 'message_bytes': L, 'carried_in_signature': m1, 'sign_ms': ts * 1e3, 'sign_ops_per_s': B / ts, 'verify_ms': tv * 1e3, 'verify_ops_per_s': B / tv,
                  'verified': ok, 'messages_rebuilt_equal': same}))
