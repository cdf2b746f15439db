"""experiment: where the time of bench.py's 20-step timed region goes (round 4)"""
import sys, os, time, statistics, ctypes
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import torch
for name in ('libamdhip64.so.7', 'libamdhip64.so.6', 'libamdhip64.so'):
    try:
        hip = ctypes.CDLL(name); break
    except OSError:
        continue
hip.hipSetDeviceFlags(ctypes.c_uint(1))
from libsafecrypto_amd import Engine
eng = Engine(0)
Q, N, B = 7681, 256, 65536
plan = eng.plan(Q, N)
g = torch.Generator(device='cuda'); g.manual_seed(1)
a = torch.randint(0, Q, (B, N), dtype=torch.int32, device='cuda', generator=g)
b = torch.randint(0, Q, (B, N), dtype=torch.int32, device='cuda', generator=g)
o = torch.empty_like(a)
side = torch.cuda.Stream()
K = 20
L = plan.L
def med(f, n=15):
    ts = []
    for _ in range(n):
        torch.cuda.synchronize(); t0 = time.perf_counter(); f(); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    return statistics.median(ts) * 1e6
with torch.cuda.stream(side):
    for _ in range(5): plan.polymul_ntt(a, b, o)
    plan.polymul_ntt_steps(a, b, o, K)
torch.cuda.synchronize()
def r_now():
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    with torch.cuda.stream(side):
        ev0.record(side); plan.polymul_ntt_steps(a, b, o, K); ev1.record(side)
def r_noev():
    with torch.cuda.stream(side):
        plan.polymul_ntt_steps(a, b, o, K)
sh = ctypes.c_void_p(side.cuda_stream)
pa, pb, po = ctypes.c_void_p(a.data_ptr()), ctypes.c_void_p(b.data_ptr()), ctypes.c_void_p(o.data_ptr())
def r_raw():
    L.sc_hip_polymul_ntt_steps(plan.h, B, pa, pb, po, K, sh)
ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
def r_raw_ev():
    ev0.record(side); L.sc_hip_polymul_ntt_steps(plan.h, B, pa, pb, po, K, sh); ev1.record(side)
nul = ctypes.c_void_p()
def r_raw_null():
    L.sc_hip_polymul_ntt_steps(plan.h, B, pa, pb, po, K, nul)
for name, f in (('as bench.py (events, stream context)', r_now), ('no events', r_noev), ('raw ctypes call, side stream', r_raw), ('raw ctypes call + events', r_raw_ev), ('raw ctypes call, null stream', r_raw_null)):
    print(f'{name:40s} {med(f):8.1f} us per {K}-step region = {med(f) / K:6.2f} us/step', flush=True)
print('time_op 20 launches (events in C):', plan.time_op(2, a, b, o, 20) * 1e3, 'us/step')
