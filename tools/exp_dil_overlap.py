"""experiment: two half-batch Dilithium-III signing calls on two handles / streams / host threads, the second started `delay` us after the first,
against one call on the whole batch.  usage: python3 tools/exp_dil_overlap.py"""
import sys, os, time, json, threading
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import torch
from libsafecrypto_amd import Engine
from tests.golden_io import load_golden

G = load_golden('dilithium_ref.txt'); tag = 'dil3'
eng = Engine(0)
def mk():
    d = eng.dilithium(G[f'{tag}_params'])
    d.set_privkey(*(G[f'{tag}_{x}'] for x in ('s1', 's2', 't', 'K', 'tr', 'Ahat')))
    return d
B = 16384
g = torch.Generator(device='cuda'); g.manual_seed(3)
msgs = torch.randint(0, 256, (B, 64), dtype=torch.uint8, device='cuda', generator=g)
d0 = mk()
d0.sign(msgs, fixed_len=64); torch.cuda.synchronize()
ts = []
for _ in range(5):
    t0 = time.perf_counter(); ref = d0.sign(msgs, fixed_len=64); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
print(json.dumps({'one_call_ms': min(ts) * 1e3}), flush=True)
for parts in (2, 3, 4):
    hs = [mk() for _ in range(parts)]
    for h in hs:
        h.set_tuning(3, 0)                      # the whole-batch call's 3 attempts per message per round, not the per-handle slot budget
    streams = [torch.cuda.Stream() for _ in range(parts)]
    cut = [B * i // parts for i in range(parts + 1)]
    chunks = [msgs[cut[i]:cut[i + 1]].contiguous() for i in range(parts)]
    outs = [None] * parts
    def work(i, delay):
        if delay:
            t_end = time.perf_counter() + delay * i
            while time.perf_counter() < t_end:
                pass
        with torch.cuda.stream(streams[i]):
            outs[i] = hs[i].sign(chunks[i], fixed_len=64)
        streams[i].synchronize()
    for delay_us in (0, 100, 200, 300, 450, 600):
        best = 1e9
        for rep in range(4):
            th = [threading.Thread(target=work, args=(i, delay_us * 1e-6)) for i in range(parts)]
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for t in th: t.start()
            for t in th: t.join()
            torch.cuda.synchronize()
            best = min(best, time.perf_counter() - t0)
        ok = torch.equal(torch.cat([o[0] for o in outs]), ref[0])
        print(json.dumps({'parts': parts, 'delay_us': delay_us, 'ms': best * 1e3, 'same_signatures': bool(ok)}), flush=True)
    for h in hs: h.close()
