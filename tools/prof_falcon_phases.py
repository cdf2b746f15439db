"""Per-phase shader-cycle breakdown of the default Falcon sampler mapping (sc_hip_falcon_sample_profile, include/sc_hip_bench.h)."""
import sys, os, json, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import numpy as np
import torch
from libsafecrypto_amd import Engine
from tests.golden_io import load_golden

NAMES = ['fft+targets', 'tree levels in HBM', 'level-128 node', 'tree levels in LDS', 'register sub-trees', 'lattice+ifft+round']


def run(set_, batch, variant=(0, 0)):
    G = load_golden('falcon_ref.txt')
    tag = f'falcon{set_}'
    eng = Engine(0)
    f = eng.falcon(G[f'{tag}_params'])
    f.set_privkey(np.asarray(G[f'{tag}_sk'], dtype=np.int64).view(np.float64))
    f.set_variant(*variant)
    g = torch.Generator(device='cuda'); g.manual_seed(13)
    c = torch.randint(0, 12289, (batch, f.n), dtype=torch.int32, device='cuda', generator=g)
    tapes = torch.randint(-2**31, 2**31 - 1, (batch, 4 * f.n), dtype=torch.int32, device='cuda', generator=g)
    f.sample(c, tapes); torch.cuda.synchronize()
    t0 = time.perf_counter(); f.sample(c, tapes); torch.cuda.synchronize(); plain = time.perf_counter() - t0
    f.sample_profile(c, tapes)
    t0 = time.perf_counter(); s1, s2, cyc = f.sample_profile(c, tapes); torch.cuda.synchronize(); prof = time.perf_counter() - t0
    spw = (variant[0] & 0xFF) if variant[0] else 8
    waves = (batch + spw - 1) // spw
    per = cyc.astype(np.float64) / waves
    out = {'set': set_, 'n': f.n, 'batch': batch, 'variant': list(variant), 'sample_ms': plain * 1e3, 'sample_ms_instrumented': prof * 1e3, 'wavefronts': waves,
           'cycles_per_wavefront': {k: round(float(v)) for k, v in zip(NAMES, per)}, 'total_cycles_per_wavefront': round(float(per.sum())),
           'share': {k: round(float(v / per.sum()), 3) for k, v in zip(NAMES, per)}}
    f.close(); eng.close()
    return out


if __name__ == '__main__':
    for s, b in ((2, 8192), (2, 65536)):
        for v in ((8 | 1 << 8, 7), (8 | 2 << 8, 7)):
            print(json.dumps(run(s, b, v)), flush=True)
