"""ENS / DLP signing throughput on the GPU (sc_hip_ens_dlp_sig_sign) for the key in tests/golden/ensdlp_ref.txt; development aid.
usage: python tools/bench_ensdlp.py [tag=enssig0] [batch=16384] [reps=3]"""
import os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from libsafecrypto_amd import Engine
from tests.golden_io import load_golden
from tests import oracle_lib as O

tag = sys.argv[1] if len(sys.argv) > 1 else 'enssig0'
batch = int(sys.argv[2]) if len(sys.argv) > 2 else 16384
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
G = load_golden('ensdlp_ref.txt')
par = np.asarray(G[f'{tag}_params'])
n, q = int(par[0]), int(par[1])
b, bgs, inv = O.ensdlp_basis(n, q, G[f'{tag}_fgFG'])
eng = Engine(0)
s = eng.ens_dlp_signer(par)
s.set_key(b, bgs, inv)
if len(sys.argv) > 4:
    s.set_mapping(int(sys.argv[4]))
g = torch.Generator(device='cuda'); g.manual_seed(3)
msgs = torch.randint(0, 256, (batch, 64), dtype=torch.uint8, device='cuda', generator=g)
tapes = torch.randint(-2**31, 2**31 - 1, (batch, 4 * n), dtype=torch.int32, device='cuda', generator=g)
sigs, st = s.sign(msgs, tapes, fixed_len=64)
torch.cuda.synchronize()
assert int((st == 1).sum()) == batch, st.unique(return_counts=True)
t0 = time.perf_counter()
for _ in range(reps):
    s.sign(msgs, tapes, fixed_len=64)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / reps
# spot check against the oracle
for i in (0, batch - 1):
    rc, want = O.ensdlp_sign(par, b, bgs, inv, bytes(msgs[i].cpu().numpy()), tapes[i].cpu().numpy().view(np.uint32))
    assert rc == 0 and bytes(sigs[i].cpu().numpy()) == want, i
state_bytes = batch * 2 * n * 4
print(f'{tag} n={n} batch={batch}: {dt * 1e3:.1f} ms per batch, {batch / dt:,.0f} sign/s; sampler state {state_bytes / 2**20:.0f} MiB, '
      f'swept {2 * n} times = {2 * state_bytes * 2 * n / dt / 1e12:.2f} TB/s of state traffic')
