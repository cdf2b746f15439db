"""experiment (round 4): Dilithium-III signing of one batch as P part-batches on P handles / streams / host threads side by side, staggered, against
one call - do one part's hash-chain kernels (ExpandMask, the challenge: latency-bound) run under the other part's transform / finishing kernels?
usage: python3 tools/exp_dil_overlap.py [batch]"""
import sys, os, time, json, threading, statistics
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import torch
from libsafecrypto_amd import Engine
from tests.golden_io import load_golden

G = load_golden('dilithium_ref.txt'); tag = 'dil3'
eng = Engine(0)
def mk(slots=0):
    d = eng.dilithium(G[f'{tag}_params'])
    d.set_privkey(*(G[f'{tag}_{x}'] for x in ('s1', 's2', 't', 'K', 'tr', 'Ahat')))
    if slots: d.set_tuning(slots)
    return d
B = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
g = torch.Generator(device='cuda'); g.manual_seed(11)
msgs = torch.randint(0, 256, (B, 64), dtype=torch.uint8, device='cuda', generator=g)
d0 = mk()
ref = d0.sign(msgs, fixed_len=64); torch.cuda.synchronize()
ts = []
for _ in range(7):
    t0 = time.perf_counter(); ref = d0.sign(msgs, fixed_len=64); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
print(json.dumps({'batch': B, 'one_call_ms': statistics.median(ts) * 1e3}), flush=True)
for parts in (2, 3, 4):
  for slots in (3, 4, 6):
    hs = [mk(slots) for _ in range(parts)]
    streams = [torch.cuda.Stream() for _ in range(parts)]
    cut = [B * i // parts for i in range(parts + 1)]
    cm = [msgs[cut[i]:cut[i + 1]].contiguous() for i in range(parts)]
    outs = [None] * parts
    def work(i, delay):
        if delay:
            t_end = time.perf_counter() + delay * i
            while time.perf_counter() < t_end:
                pass
        with torch.cuda.stream(streams[i]):
            outs[i] = hs[i].sign(cm[i], fixed_len=64)
        streams[i].synchronize()
    for i in range(parts): work(i, 0)
    for delay_us in (0, 150, 300, 500):
        tt = []
        for rep in range(7):
            th = [threading.Thread(target=work, args=(i, delay_us * 1e-6)) for i in range(parts)]
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for t in th: t.start()
            for t in th: t.join()
            torch.cuda.synchronize()
            tt.append(time.perf_counter() - t0)
        ok = torch.equal(torch.cat([o[0] for o in outs]), ref[0])
        print(json.dumps({'parts': parts, 'slots': slots, 'delay_us': delay_us, 'ms_median': statistics.median(tt) * 1e3, 'ms_min': min(tt) * 1e3, 'same_signatures': bool(ok)}), flush=True)
    for h in hs: h.close()
