"""DLP-IBE batched extract throughput on the GPU box."""
import sys, os, time, json
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import numpy as np
import torch
from libsafecrypto_amd import Engine
from tests.golden_io import load_golden

G = load_golden('dlp_ref.txt')
par = np.asarray(G['dlp0_params'])
n, q, q_bits, n_bits, set_, ubits = (int(x) for x in par[:6])
eng = Engine(0)
f = eng.falcon(np.array([n, q, q_bits, n_bits, 0, 0, 0, 0], dtype=np.int32))
f.set_privkey(np.asarray(G['dlp0_tree'], dtype=np.int64).view(np.float64))
g = torch.Generator(device='cuda'); g.manual_seed(21)
for B in ([int(sys.argv[1])] if len(sys.argv) > 1 else (8192, 32768, 131072)):
    ids = torch.randint(0, 256, (B, 16), dtype=torch.uint8, device='cuda', generator=g)
    tapes = torch.randint(-2**31, 2**31 - 1, (B, 4 * n), dtype=torch.int32, device='cuda', generator=g)
    f.dlp_extract(ubits, ids, tapes, fixed_len=16); torch.cuda.synchronize()
    ts = []
    for _ in range(3):
        t0 = time.perf_counter(); sk, st = f.dlp_extract(ubits, ids, tapes, fixed_len=16); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    assert int((st == 1).sum()) == B
    print(json.dumps({'set': set_, 'n': n, 'q': q, 'batch': B, 'extract_ops_per_s': B / min(ts), 'extract_ms': min(ts) * 1e3}), flush=True)
    del ids, tapes, sk

# encrypt / decrypt (every message to identity 2, whose user key the fixture holds)
from libsafecrypto_amd import Engine as _E
d = eng.dlp(par)
d.set_pubkey(G['dlp0_h_ntt']); d.set_user_key(G['dlp0_uk_ntt2'])
ident = np.asarray(G['dlp0_id2'], dtype=np.uint8)
for B in ([int(sys.argv[1])] if len(sys.argv) > 1 else (16384, 65536)):
    ids = torch.from_numpy(np.tile(ident, (B, 1))).cuda()
    msgs = torch.randint(0, 256, (B, d.msg_bytes), dtype=torch.uint8, device='cuda', generator=g)
    tapes = torch.randint(-2**31, 2**31 - 1, (B, d.enc_words + 1), dtype=torch.int32, device='cuda', generator=g)
    d.encrypt(ids, msgs, tapes, fixed_len=len(ident)); torch.cuda.synchronize()
    te, td = [], []
    for _ in range(3):
        t0 = time.perf_counter(); ct, st = d.encrypt(ids, msgs, tapes, fixed_len=len(ident)); torch.cuda.synchronize(); te.append(time.perf_counter() - t0)
        t0 = time.perf_counter(); mm = d.decrypt(ct); torch.cuda.synchronize(); td.append(time.perf_counter() - t0)
    wrong = int((mm != msgs).any(dim=1).sum())
    print(json.dumps({'batch': B, 'encrypt_ops_per_s': B / min(te), 'decrypt_ops_per_s': B / min(td), 'decryption_failures': wrong}), flush=True)
