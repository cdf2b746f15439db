#!/usr/bin/env python3
"""bench.py - headline measurement of the MI355X lattice-arithmetic engine.

Workload (BASELINE.json configs[1], with the reference's real Kyber modulus - SURVEY.md section 0):
n = 256, q = 7681, batch = 65536 polynomials per GPU; one "step" = one pass of the fused hot
path  c = INTT( NTT(a) o bhat )  over the batch (sc_hip_polymul_ntt: forward NTT, pointwise
multiply by an NTT-domain operand, inverse NTT, canonical output), inputs resident in HBM.
Metric: NTT/s, counting the two transforms of each triple (BASELINE.md counts the same way).

  python bench.py --gpus N --steps K --warmup W
  (N > 1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...)

Independent polynomials / signatures shard across ranks with no data-path collective (weak scaling: every
rank runs its own batch); the only collectives are the timing barrier, the MAX over ranks of the elapsed
time and the gather of the per-rank scheme rates.  The scheme-level legs (BASELINE configs[2..4]: Dilithium-III
sign / verify, BLISS-B-I sign, Falcon-1024 sign) run on every rank as well and are reported per GPU and summed.

`roofline` blocks: the headline's is HBM (algorithmic bytes / HIP-event kernel time / 8 TB/s, at the BASELINE size and at a working
set far past the 256 MiB Infinity Cache); the scheme legs' are VALU: wave-level VALU instructions and HBM bytes per operation counted by
rocprofv3 --pmc over the SAME leg in a child process of this run (tools/pmc_leg.py; the verify legs load signatures made by a separate
process, so no signing kernel is in their counts), x the operations per second measured here, against the chip's issue ceiling of one
wave64 instruction per 2 cycles per SIMD (1228.8 G wave-instr/s at 2.4 GHz - the guide's 157.3 TF vector peak; tools/exp_valu_cycles.hip
measured which instructions reach it and which take 4 cycles: profiles/r03_valu_issue_cycles.txt).  No `frac` may exceed 1 (asserted).
The north-star figures (Dilithium-III, BLISS-B-I, Falcon-1024) are repeated compactly under `north_star`, the LAST key of the line.
"""
import argparse
import json
import os
import statistics
import subprocess
import sys
import time

# pageable host<->device copies of this process are staged by the runtime, not pinned in place (tests/conftest.py, profiles/r02_pageable_copy_fault.txt);
# read when the HIP runtime initialises, so set before torch is imported; an explicit setting in the environment wins
os.environ.setdefault('GPU_PINNED_XFER_SIZE', '4096')
os.environ.setdefault('GPU_PINNED_MIN_XFER_SIZE', '4194304')

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

Q, N, BATCH = 7681, 256, 65536
HBM_PEAK_GBS = 8000.0            # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured copy)
BYTES_PER_TRIPLE = 12 * N        # read a (4n) + read bhat (4n) + write c (4n)  
# VALU issue ceiling, chip-wide, in G wave-instructions / s (profiles/r03_valu_issue_cycles.txt, measured with s_memtime on this part):
#   2 cycles per wave64 instruction per SIMD for v_add / v_sub / v_xor / v_and / v_bitop3 / v_fma_f32 ... (2.24-2.5 measured at >= 2 waves per SIMD),
#   4 cycles for multiplies, v_alignbit, v_min / v_max, v_mad, 64-bit and packed operations (4.2 measured); one wave alone: 5-6 cycles either way.
# `peak` is the 2-cycle ceiling (what MI355X_MICROARCH.md's 157.3 TF vector peak implies), so every VALU `frac` is a lower bound of the issue
# utilisation of a kernel that mixes both classes; `frac_4cycle_ceiling` is the same figure against the other class's ceiling.
VALU_PEAK_GWI = 256 * 4 * 2.4 / 2.0
VALU_PEAK_GWI_4CYC = 256 * 4 * 2.4 / 4.0
PAST_L3_BATCH = 1 << 20          # 3 x 1 GiB of operands: 12x the Infinity Cache

DRV = os.path.join(ROOT, 'oracle', '_ref', 'ref_driver')
DRV_AVX2 = os.path.join(ROOT, 'oracle', '_ref', 'ref_driver_avx2')


def _cpu_model():
    try:
        for line in open('/proc/cpuinfo'):
            if line.startswith('model name'):
                return line.split(':', 1)[1].strip()
    except OSError:
        pass
    return 'unknown'


def _has_avx2():
    try:
        return any(' avx2' in l for l in open('/proc/cpuinfo') if l.startswith('flags'))
    except OSError:
        return False


def _drv_json(drv, args, timeout):
    out = subprocess.check_output([drv] + [str(a) for a in args], timeout=timeout, stderr=subprocess.DEVNULL)
    return json.loads(out.decode().strip().splitlines()[-1])


def cpu_baseline(seconds=10.0):
    """SURVEY 8(d)'s CPU protocol for the headline: the reference itself (oracle/_ref, built from the reference sources in the dev
    container; plain gcc -O2 build and the -mavx2 -DHAVE_AVX2 build), one pinned thread, median of 5 runs, plus an nproc-way figure for
    context.  Falls back to our C port when the reference binaries did not travel."""
    runs, per = 5, max(0.5, seconds / 5.0)
    sample = (f'median of {runs} runs of {per:.1f} s of fwd_ntt->mul_32_pointwise->inv_ntt on n={N}, q={Q} polynomials (pool of 1024, uniform [0,q)), 1 thread'
              f' pinned to core 0; host CPU: {_cpu_model()}')
    if os.path.exists(DRV):
        try:
            pin = ['taskset', '-c', '0'] if os.path.exists('/usr/bin/taskset') else []

            def med(drv):
                v = []
                for _ in range(runs):
                    out = subprocess.check_output(pin + [drv, 'bench', str(Q), str(N), str(per)], timeout=per * 3 + 30)
                    v.append(2.0 * json.loads(out.decode().strip().splitlines()[-1])['triples_per_sec'])
                return statistics.median(v), v
            plain, plain_runs = med(DRV)
            res = {'value': plain, 'unit': 'NTT/s', 'cores': 1, 'kind': 'reference',
                   'sample': sample + ' (reference SC_NTT_FLOATING_POINT backend, gcc -O2)', 'runs': plain_runs, 'cpu_model': _cpu_model()}
            if os.path.exists(DRV_AVX2) and _has_avx2():
                try:
                    avx, avx_runs = med(DRV_AVX2)
                    res['avx2_build'] = {'value': avx, 'unit': 'NTT/s', 'cores': 1, 'runs': avx_runs,
                                         'sample': 'the same with the reference built -mavx2 -DHAVE_AVX2 -DHAVE_AVX (SC_NTT_AVX backend)'}
                except Exception as e:
                    res['avx2_build'] = {'error': str(e)}
            try:                                     # all host cores, one process each (the reference has no threaded NTT path)
                ncpu = os.cpu_count() or 1
                ps = [subprocess.Popen([DRV, 'bench', str(Q), str(N), str(per)], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL) for _ in range(ncpu)]
                tot = sum(2.0 * json.loads(p.communicate(timeout=per * 3 + 30)[0].decode().strip().splitlines()[-1])['triples_per_sec'] for p in ps)
                res['all_cores'] = {'value': tot, 'unit': 'NTT/s', 'cores': ncpu, 'sample': f'{ncpu} concurrent single-thread processes, {per:.1f} s'}
            except Exception as e:
                res['all_cores'] = {'error': str(e)}
            return res
        except Exception as e:  # fall through to the port
            sys.stderr.write(f'cpu_baseline: reference binary failed ({e}); using the port\n')
    import numpy as np
    from tests import oracle_lib as O
    w, r = O.tables(Q, N)
    rng = np.random.default_rng(1)
    a = rng.integers(0, Q, size=(1024, N)).astype(np.int32)
    b = rng.integers(0, Q, size=(1024, N)).astype(np.int32)
    done, t0 = 0, time.perf_counter()
    while time.perf_counter() - t0 < seconds:
        O.polymul_ntt(a, b, Q, w, r)
        done += 1024
    dt = time.perf_counter() - t0
    return {'value': 2.0 * done / dt, 'unit': 'NTT/s', 'cores': 1, 'kind': 'port', 'sample': sample + ' (oracle/sc_oracle.c, gcc -O2)', 'cpu_model': _cpu_model()}


def _cpu_ref(mode, set_, sec, keys, sample):
    """the reference's own safecrypto_* loop for one scheme, single thread; None when the binary is absent or fails"""
    if sec <= 0 or not os.path.exists(DRV):
        return None
    try:
        r = _drv_json(DRV, [mode, set_, sec], sec * 3 + 240)
        out = {k: r[v] for k, v in keys.items()}
        out.update({'cores': 1, 'kind': 'reference', 'sample': sample, 'cpu_model': _cpu_model()})
        return out
    except Exception as e:
        sys.stderr.write(f'{mode} cpu reference failed: {e}\n')
        return None


_PMC_COMMITTED = None
PMC_LIVE = True                  # --no-live-pmc: use the committed counts (profiles/r03_scheme_pmc.json) instead of profiling the legs in this run
PMC_BUDGET_S = 240.0             # the counter passes (3 child processes per leg) stop being started once this much has been spent on them
_pmc_spent = 0.0


def _pmc(leg):
    """counters of one leg: live (a child process of this run under rocprofv3 --pmc), else the committed pass of the same tool"""
    global _PMC_COMMITTED, _pmc_spent
    if PMC_LIVE and _pmc_spent < PMC_BUDGET_S:
        try:
            sys.path.insert(0, os.path.join(ROOT, 'tools'))
            import pmc_leg
            t0 = time.perf_counter()
            r = pmc_leg.profile_leg(leg, calls=2)
            _pmc_spent += time.perf_counter() - t0
            if r.get('ok'):
                r['source'] = 'live: ' + r.pop('method')
                return r
            sys.stderr.write(f'live PMC pass of {leg} failed: {r.get("error")}\n')
        except Exception as e:
            sys.stderr.write(f'live PMC pass of {leg} failed: {e}\n')
    if _PMC_COMMITTED is None:
        try:
            _PMC_COMMITTED = json.load(open(os.path.join(ROOT, 'profiles', 'r03_scheme_pmc.json')))
        except Exception:
            _PMC_COMMITTED = {}
    r = _PMC_COMMITTED.get(leg)
    if r:
        r = dict(r)
        r['source'] = 'committed: profiles/r03_scheme_pmc.json (tools/pmc_leg.py run by tools/prof_schemes_r3.sh on an earlier box) - NOT this run'
    return r


def valu_roofline(leg, ops_per_s, algorithmic_bytes_per_op, note, top=4):
    """VALU roofline of a scheme leg: instructions and HBM bytes per operation from a counter pass over the same leg x the rate measured now"""
    p = _pmc(leg)
    if not p:
        return {'bound': 'valu', 'achieved': None, 'peak': VALU_PEAK_GWI, 'unit': 'G wave-instr/s', 'frac': None, 'traffic': None,
                'note': 'no counter pass for this leg (rocprofv3 unavailable and no committed profile)'}
    wi = p['valu_wave_instructions_per_op']
    ach = wi * ops_per_s / 1e9
    kern = {}
    for name, k in list(p.get('kernels', {}).items())[:top]:
        kern[name] = {'share_of_kernel_cycles': k['shader_cycles_per_call'] / p['shader_cycles_per_call_sum_of_kernels'],
                      'valu_instr_per_simd_cycle': k['valu_instr_per_simd_cycle'], 'wave_cycles_waiting_frac': k['wave_cycles_waiting_frac'],
                      'launches_per_call': k['launches_per_call']}
    return {'bound': 'valu', 'achieved': ach, 'peak': VALU_PEAK_GWI, 'unit': 'G wave-instr/s', 'frac': ach / VALU_PEAK_GWI,
            'frac_4cycle_ceiling': ach / VALU_PEAK_GWI_4CYC,
            'traffic': p['hbm_bytes_per_op_2xfetch_plus_write'], 'traffic_unit': 'HBM bytes per operation (rocprofv3 FETCH_SIZE x 2 + WRITE_SIZE)',
            'algorithmic_bytes_per_op': algorithmic_bytes_per_op, 'valu_wave_instructions_per_op': wi,
            'valu_instr_per_simd_cycle_in_kernels': p.get('valu_instr_per_simd_cycle'),
            'valu_instr_per_simd_cycle_ceilings': {'2_cycle_class': 0.5, '4_cycle_class': 0.25, 'measured': 'profiles/r03_valu_issue_cycles.txt: 0.40-0.43 / 0.235'},
            'hbm_GBs_at_this_rate': p['hbm_bytes_per_op_2xfetch_plus_write'] * ops_per_s / 1e9,
            'dominant_kernels': kern, 'source': p.get('source'), 'note': note}


def _timed(torch, fn, reps):
    """(best host seconds, best HIP-event milliseconds on the current stream) of fn() + synchronize"""
    th, te = [], []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter(); e0.record(); out = fn(); e1.record(); torch.cuda.synchronize(); th.append(time.perf_counter() - t0)
        te.append(e0.elapsed_time(e1))
    return min(th), min(te), out


def dilithium_extra(eng, torch, cpu_seconds, distinct=True):
    """North-star companion figure: Dilithium-III (reference set 3: k=6, l=5) full sign / verify of 16384 64-byte
    messages under one key, messages and signatures resident in HBM (BASELINE configs[2])."""
    from tests.golden_io import load_golden
    G = load_golden('dilithium_ref.txt')       # key material produced by the reference's keygen (data, not code)
    tag, batch = 'dil3', 16384
    d = eng.dilithium(G[f'{tag}_params'])
    d.set_privkey(*(G[f'{tag}_{x}'] for x in ('s1', 's2', 't', 'K', 'tr', 'Ahat')))
    d.set_pubkey(G[f'{tag}_t1'], G[f'{tag}_rho'], G[f'{tag}_Ahat'])
    g = torch.Generator(device='cuda')
    g.manual_seed(3)
    msgs = torch.randint(0, 256, (batch, 64), dtype=torch.uint8, device='cuda', generator=g)
    sigs, siglens, trials = d.sign(msgs, fixed_len=64)
    torch.cuda.synchronize()
    ts, ts_ev, (sigs, siglens, trials) = _timed(torch, lambda: d.sign(msgs, fixed_len=64), 5)
    tv, tv_ev, v = _timed(torch, lambda: d.verify(msgs, sigs, siglens, fixed_len=64), 5)
    assert int(v.sum()) == 0, 'bench signatures failed verification'
    out = {'workload': 'BASELINE configs[2]: Dilithium-III (n=256, q=8380417, k=6, l=5) sign + verify, batch 16384, 64-byte messages, one key',
           'sign_ops_per_s': batch / ts, 'verify_ops_per_s': batch / tv, 'sign_ms_per_batch': ts * 1e3, 'verify_ms_per_batch': tv * 1e3,
           'sign_ms_hip_events': ts_ev, 'verify_ms_hip_events': tv_ev,
           'mean_attempts_per_signature': float(trials.float().mean()), 'dtype': 'u32'}
    out['roofline'] = valu_roofline('dil_sign', out['sign_ops_per_s'], 64 + 3370,
                                    'sign: 64 B message in + 3370 B signature out per operation; the work is ~3.2 attempts of 35 Keccak permutations, 11 transforms and 30 '
                                    'pointwise products each - instruction-bound, not HBM-bound')
    out['roofline_verify'] = valu_roofline('dil_verify', out['verify_ops_per_s'], 64 + 3370 + 1, 'verify: message + signature in, one verdict out')
    # the configuration's second leg: 16384 DISTINCT key pairs (generated on the device from 32-byte seeds), one message under each
    if distinct:
        try:
            seeds = torch.randint(0, 256, (batch, 32), dtype=torch.uint8, device='cuda', generator=g)
            s1, s2, t, rho, K, tr, t1 = d.keygen(seeds, int(G[f'{tag}_eta_bits'][0]))
            t1f = t1.reshape(batch, -1).contiguous()
            ks, kt = d.sign_keys(s1, s2, t, rho, K, tr, msgs, fixed_len=64)[:2]; torch.cuda.synchronize()
            ts2, tv2 = [], []
            for _ in range(3):
                t0 = time.perf_counter(); ks, kl, ktr = d.sign_keys(s1, s2, t, rho, K, tr, msgs, fixed_len=64); torch.cuda.synchronize(); ts2.append(time.perf_counter() - t0)
                t0 = time.perf_counter(); kv = d.verify_keys(t1f, rho, msgs, ks, kl, fixed_len=64); torch.cuda.synchronize(); tv2.append(time.perf_counter() - t0)
            assert int(kv.sum()) == 0, 'distinct-key signatures failed verification'
            out['distinct_keys'] = {'workload': f'the same with {batch} distinct key pairs, one message under each: per-item matrix from rho, NTT of s1 / s2 / t0 / t1 and tr derived '
                                                'on the device inside the timed call', 'sign_ops_per_s': batch / min(ts2), 'verify_ops_per_s': batch / min(tv2),
                                    'algorithmic_bytes_per_key': {'sign': (5 + 2 * 6) * 256 * 4 + 112, 'verify': 6 * 256 * 4 + 32}}
            del s1, s2, t, rho, K, tr, t1, t1f, ks
        except Exception as e:
            out['distinct_keys'] = {'error': str(e)}
    ref = _cpu_ref('bench_dil', 3, cpu_seconds, {'sign_ops_per_s': 'sign_per_sec', 'verify_ops_per_s': 'verify_per_sec'},
                   f'{cpu_seconds:.0f} s of safecrypto_sign + {cpu_seconds / 4:.0f} s of safecrypto_verify, set 3, 64-byte messages, one key, 1 thread')
    if ref:
        out['cpu_reference'] = ref
        out['sign_speedup_vs_cpu_thread'] = out['sign_ops_per_s'] / ref['sign_ops_per_s']
        if os.path.exists(DRV_AVX2) and _has_avx2():
            try:
                r = _drv_json(DRV_AVX2, ['bench_dil', 3, max(2.0, cpu_seconds / 3)], 240)
                out['cpu_reference_avx2_build'] = {'sign_ops_per_s': r['sign_per_sec'], 'verify_ops_per_s': r['verify_per_sec'], 'cores': 1}
            except Exception as e:
                out['cpu_reference_avx2_build'] = {'error': str(e)}
    d.close()
    return out


def falcon_extra(eng, torch, cpu_seconds, sets=(('falcon2', 2, (8192, 131072)), ('falcon0', 0, (131072,)))):
    """BASELINE configs[4]: Falcon-1024 (reference set 2) signing - SHAKE hash to the ring, fast Fourier sampling over the LDL
    tree in binary64, norm test, packing - at the configuration's batch 8192 per GPU and at a throughput-saturating batch;
    Falcon-512 (set 0) beside it.  One key per set (the reference keygen's, from the fixture)."""
    import numpy as np
    from tests.golden_io import load_golden
    G = load_golden('falcon_ref.txt')
    out = {}
    g = torch.Generator(device='cuda'); g.manual_seed(13)
    for tag, set_, batches in sets:
        f = eng.falcon(G[f'{tag}_params'])
        f.set_privkey(np.asarray(G[f'{tag}_sk'], dtype=np.int64).view(np.float64)); f.set_pubkey(G[f'{tag}_h'])
        res = {'workload': f'Falcon-{f.n} (reference set {set_}, q=12289) sign with the FFT tree sampler on the GPU (f64) + verify, 64-byte messages, '
                           'per-message CSPRNG word tapes resident in HBM', 'dtype': 'f64'}
        for B in batches:
            msgs = torch.randint(0, 256, (B, 64), dtype=torch.uint8, device='cuda', generator=g)
            tapes = torch.randint(-2**31, 2**31 - 1, (B, 4 * f.n), dtype=torch.int32, device='cuda', generator=g)
            f.sign(msgs, tapes, fixed_len=64); torch.cuda.synchronize()
            ts, ts_ev, (sigs, siglens, status) = _timed(torch, lambda: f.sign(msgs, tapes, fixed_len=64), 3)
            tv, tv_ev, v = _timed(torch, lambda: f.verify(msgs, sigs, siglens, fixed_len=64), 3)
            ok = status == 1
            assert int(v[ok].sum()) == 0 and int(ok.sum()) > B - B // 100
            res[f'batch_{B}'] = {'sign_ops_per_s': B / ts, 'verify_ops_per_s': B / tv, 'sign_ms_per_batch': ts * 1e3, 'sign_ms_hip_events': ts_ev, 'verify_ms_hip_events': tv_ev}
            del msgs, tapes, sigs
        ref = _cpu_ref('bench_falcon', set_, min(cpu_seconds, 4.0), {'sign_ops_per_s': 'sign_per_sec', 'verify_ops_per_s': 'verify_per_sec'},
                       f'{min(cpu_seconds, 4.0):.0f} s of safecrypto_sign + safecrypto_verify, set {set_}, 64-byte messages, one key, 1 thread')
        if ref:
            res['cpu_reference'] = ref
            res['sign_speedup_vs_cpu_thread'] = res[f'batch_{batches[-1]}']['sign_ops_per_s'] / ref['sign_ops_per_s']
        if set_ == 2:
            res['workload'] = 'BASELINE configs[4]: ' + res['workload'] + ' (batch_8192 is the configuration as stated)'
            r = res['batch_8192']['sign_ops_per_s']
            res['roofline'] = valu_roofline('falcon_sign', r, 64 + 1536 + 16384,
                                            'batch 8192: 64 B message + 16 KB tape in, 1536 B signature out per operation (sampler alone, SURVEY 8(d) row 5: c 4 KB in, s1 + s2 8 KB out); '
                                            'the sampler is a 2n-leaf sequential walk per signature: latency- and f64-issue-bound, its HBM traffic is the spilled working set')
        if set_ == 2:
            # SURVEY 8(d) distinct keys: a resident key set, a key index per message (the fixture key loaded 64 times: distinct memory, same values)
            NK, B = 64, 8192
            sk = np.asarray(G[f'{tag}_sk'], dtype=np.int64).view(np.float64)
            f.set_privkeys(np.tile(sk, (NK, 1))); f.set_pubkeys(np.tile(np.asarray(G[f'{tag}_h'], dtype=np.int16), (NK, 1)))
            msgs = torch.randint(0, 256, (B, 64), dtype=torch.uint8, device='cuda', generator=g)
            tapes = torch.randint(-2**31, 2**31 - 1, (B, 4 * f.n), dtype=torch.int32, device='cuda', generator=g)
            ko = (torch.arange(B, device='cuda') // (B // NK)).to(torch.int32)        # grouped by key, as a multi-tenant signer would submit them
            f.sign_keys(ko, msgs, tapes, fixed_len=64); torch.cuda.synchronize()
            ts, tv = [], []
            for _ in range(3):
                t0 = time.perf_counter(); sigs, siglens, status = f.sign_keys(ko, msgs, tapes, fixed_len=64); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
                t0 = time.perf_counter(); v = f.verify_keys(ko, msgs, sigs, siglens, fixed_len=64); torch.cuda.synchronize(); tv.append(time.perf_counter() - t0)
            assert int(v[status == 1].sum()) == 0
            ki = (torch.arange(B, device='cuda') % NK).to(torch.int32)                  # the same keys interleaved: no two neighbours share one
            f.sign_keys(ki, msgs, tapes, fixed_len=64); torch.cuda.synchronize()
            t0 = time.perf_counter(); f.sign_keys(ki, msgs, tapes, fixed_len=64); torch.cuda.synchronize(); t_il = time.perf_counter() - t0
            res['distinct_keys'] = {'workload': f'batch {B}, a key set of {NK} resident keys, key index per message; messages grouped by key (runs of {B // NK}): the wavefront-per-8 form, '
                                                'keys interleaved (no two neighbours share one): the call regroups the batch by key, signs it on the same kernel and scatters the signatures back', 'sign_ops_per_s_interleaved': B / t_il,
                                    'sign_ops_per_s': B / min(ts), 'verify_ops_per_s': B / min(tv),
                                    'key_bytes_per_key': int((f.n * 15 + (f.n << 5)) * 8 + 2 * f.n)}
            del msgs, tapes, sigs
        out[f'falcon_{f.n}'] = res
        f.close()
    return out


def bliss_extra(eng, torch, cpu_seconds):
    """BASELINE configs[3]: BLISS-B-I sign with the CDT sampler on the GPU, batch 16384; key from the reference's keygen (fixture)"""
    import numpy as np
    from tests.golden_io import load_golden
    G = load_golden('bliss_ref.txt')
    b = eng.bliss(G['bliss1_params']); b.set_key(G['bliss1_f'], G['bliss1_g'], G['bliss1_a'])
    g = torch.Generator(device='cuda'); g.manual_seed(11)
    B, W = 16384, 4 * 2048 + 64
    msgs = torch.randint(0, 256, (B, 64), dtype=torch.uint8, device='cuda', generator=g)
    tapes = torch.randint(-2**31, 2**31 - 1, (B, W), dtype=torch.int32, device='cuda', generator=g)
    b.sign(msgs, tapes, fixed_len=64); torch.cuda.synchronize()
    ts, ts_ev, (sigs, siglens, trials, status) = _timed(torch, lambda: b.sign(msgs, tapes, fixed_len=64), 5)
    tv, tv_ev, v = _timed(torch, lambda: b.verify(msgs, sigs, siglens, fixed_len=64), 5)
    ok = status == 1
    assert int(v[ok].sum()) == 0 and int(ok.sum()) > B - 20
    attempts = float(trials[ok].float().mean())
    out = {'workload': 'BASELINE configs[3]: BLISS-B-I (n=512, q=12289) sign with the CDT sampler on the GPU, batch 16384, 64-byte messages, '
                       'per-message CSPRNG word tapes resident in HBM', 'sign_ops_per_s': B / ts, 'verify_ops_per_s': B / tv,
           'sign_ms_hip_events': ts_ev, 'verify_ms_hip_events': tv_ev,
           'mean_attempts_per_signature': attempts, 'undecided_near_ties': int((status == 5).sum()), 'dtype': 'u32'}
    out['roofline'] = valu_roofline('bliss_sign', out['sign_ops_per_s'], 64 + 986 + 8 * 1024 * (attempts + 1) / 2,
                                    'per signature: 64 B message + 986 B signature + 8 KB of uniform words per fresh pair of Gaussian vectors (12 B per sample in the sampler kernel)')
    out['roofline_verify'] = valu_roofline('bliss_verify', out['verify_ops_per_s'], 64 + 986 + 1, 'verify: message + signature in, one verdict out')
    ref = _cpu_ref('bench_bliss', 1, min(cpu_seconds, 3.0), {'sign_ops_per_s': 'sign_per_sec', 'verify_ops_per_s': 'verify_per_sec'},
                   f'{min(cpu_seconds, 3.0):.0f} s of safecrypto_sign + safecrypto_verify, set 1 (64-bit CDT), 64-byte messages, one key, 1 thread')
    if ref:
        out['cpu_reference'] = ref
    # SURVEY 8(d) distinct keys: a key pair per message (3 KB of key material per signature instead of one cache-resident key)
    kt = torch.randint(-2**31, 2**31 - 1, (B, 2048), dtype=torch.int32, device='cuda', generator=g)
    f, gk, a, _, kst = b.keygen(G['bliss1_kg_nz'], kt)
    if int((kst == 1).sum()) == B:
        b.sign_keys(f, gk, a, msgs, tapes, fixed_len=64); torch.cuda.synchronize()
        tk, tkv = [], []
        for _ in range(5):
            t0 = time.perf_counter(); s2, l2, _, st2 = b.sign_keys(f, gk, a, msgs, tapes, fixed_len=64); torch.cuda.synchronize(); tk.append(time.perf_counter() - t0)
            t0 = time.perf_counter(); v2 = b.verify_keys(a, msgs, s2, l2, fixed_len=64); torch.cuda.synchronize(); tkv.append(time.perf_counter() - t0)
        assert int(v2[st2 == 1].sum()) == 0
        out['distinct_keys'] = {'workload': 'the same batch with a key pair per message (keys from the batched keygen, resident in HBM)',
                                'sign_ops_per_s': B / min(tk), 'verify_ops_per_s': B / min(tkv), 'key_bytes_per_signature': 3 * 512 * 2}
    # the signatures in the reference's SC_ENTROPY_HUFFMAN_STATIC coding (SC_FLAG_0_ENTROPY_HUFFMAN; the code tables are the reference's data, from the fixture)
    H = load_golden('bliss_huff_ref.txt')
    pick = lambda k: ((H[f'huff{k}_code'].astype(np.int64) & 0xFFFFFFFF).astype(np.uint32), H[f'huff{k}_bits'].astype(np.uint16))
    b.set_entropy((pick(6), pick(2)))
    b.sign(msgs, tapes, fixed_len=64); torch.cuda.synchronize()
    th, thv = [], []
    for _ in range(5):
        t0 = time.perf_counter(); s3, l3, _, st3 = b.sign(msgs, tapes, fixed_len=64); torch.cuda.synchronize(); th.append(time.perf_counter() - t0)
        t0 = time.perf_counter(); v3 = b.verify(msgs, s3, l3, fixed_len=64); torch.cuda.synchronize(); thv.append(time.perf_counter() - t0)
    assert int(v3[st3 == 1].sum()) == 0 and torch.equal(st3, status)
    out['huffman_coded'] = {'workload': 'the same batch with the signatures entropy-coded as the reference codes them (static Huffman tables for t and z)',
                            'sign_ops_per_s': B / min(th), 'verify_ops_per_s': B / min(thv), 'mean_signature_bytes': float(l3[st3 == 1].float().mean()), 'raw_signature_bytes': 986}
    # ... and under the arithmetic coder (SC_FLAG_0_ENTROPY_BAC; the coder's two distributions are long double arithmetic of the reference's host: from the fixture)
    A = load_golden('bliss_bac_ref.txt')
    dist = lambda k: np.ascontiguousarray(np.ascontiguousarray(A[k], dtype=np.int32).view(np.uint64))
    b.set_entropy_bac(dist('blissb1_dist_t'), dist('blissb1_dist_z'))
    b.sign(msgs, tapes, fixed_len=64); torch.cuda.synchronize()
    ta, tav = [], []
    for _ in range(3):
        t0 = time.perf_counter(); s4, l4, _, st4 = b.sign(msgs, tapes, fixed_len=64); torch.cuda.synchronize(); ta.append(time.perf_counter() - t0)
        t0 = time.perf_counter(); v4 = b.verify(msgs, s4, l4, fixed_len=64); torch.cuda.synchronize(); tav.append(time.perf_counter() - t0)
    assert int(v4[st4 == 1].sum()) == 0 and torch.equal(st4, status)
    out['arithmetic_coded'] = {'workload': 'the same batch with t and z through the reference\'s binary arithmetic coder (a serial range coder: one lane per signature)',
                               'sign_ops_per_s': B / min(ta), 'verify_ops_per_s': B / min(tav), 'mean_signature_bytes': float(l4[st4 == 1].float().mean()), 'raw_signature_bytes': 986}
    b.close()
    return {'bliss_b_1': out}


def kyber_tesla_more_extra(eng, torch, cpu_on):
    """the scheme-level form of configs[1] (Kyber set 1 CPA / KEM, batch 65536), ring-TESLA (the fourth scheme the north star names) and the remaining
    NTT clients of SURVEY 8(f)3 that are served; keys produced by the reference's keygen (fixtures)."""
    from tests.golden_io import load_golden
    out = {}
    g = torch.Generator(device='cuda'); g.manual_seed(12)
    G = load_golden('kyber_ref.txt')
    ky = eng.kyber(G['kyber1_params']); ky.set_pubkey(G['kyber1_t_ntt'], G['kyber1_Agen']); ky.set_privkey(G['kyber1_s_ntt'])
    B = 65536
    m = torch.randint(0, 256, (B, 32), dtype=torch.uint8, device='cuda', generator=g)
    noise = torch.randint(0, 256, (B, ky.noise_bytes), dtype=torch.uint8, device='cuda', generator=g)
    u, v = ky.enc(m, noise); torch.cuda.synchronize()
    te, td = [], []
    for _ in range(5):
        t0 = time.perf_counter(); u, v = ky.enc(m, noise); torch.cuda.synchronize(); te.append(time.perf_counter() - t0)
        t0 = time.perf_counter(); mm = ky.dec(u, v); torch.cuda.synchronize(); td.append(time.perf_counter() - t0)
    assert torch.equal(mm, m)
    out['kyber_1_cpa'] = {'workload': 'Kyber set 1 (k=3, q=7681) CPA encryption / decryption, batch 65536, one key', 'enc_ops_per_s': B / min(te),
                          'dec_ops_per_s': B / min(td)}
    # KEM level: safecrypto_encapsulation / safecrypto_decapsulation entirely on the device (hashes, DRBG, CPA, packing)
    ky.set_kem_pubkey(G['kyber1_t'], G['kyber1_rho'])
    ct, keys = ky.encaps(m); torch.cuda.synchronize()
    te, td = [], []
    for _ in range(5):
        t0 = time.perf_counter(); ct, keys = ky.encaps(m); torch.cuda.synchronize(); te.append(time.perf_counter() - t0)
        t0 = time.perf_counter(); k2 = ky.decaps(ct); torch.cuda.synchronize(); td.append(time.perf_counter() - t0)
    assert torch.equal(k2, keys)
    kem = {'workload': 'Kyber set 1 KEM encapsulation / decapsulation (reference wire format), batch 65536, one key, 32-byte m per operation resident in HBM',
           'encaps_ops_per_s': B / min(te), 'decaps_ops_per_s': B / min(td)}
    ref = _cpu_ref('bench_kyber', 1, 3 if cpu_on else 0, {'encaps_ops_per_s': 'encaps_per_sec', 'decaps_ops_per_s': 'decaps_per_sec'},
                   '3 s of safecrypto_encapsulation + 1.5 s of safecrypto_decapsulation, set 1, one key, 1 thread')
    if ref:
        kem['cpu_reference'] = ref
    # SURVEY 8(d) distinct keys: a key pair per value, generated by the batched keygen (matrix expansion from each rho is part of the timed call)
    seeds = torch.randint(0, 256, (B, 32), dtype=torch.uint8, device='cuda', generator=g)
    s_ntt, tk, t_ntt, rho, _ = ky.keygen(seeds)
    ctk, kk = ky.encaps_keys(tk, t_ntt, rho, m); torch.cuda.synchronize()
    te, td = [], []
    for _ in range(5):
        t0 = time.perf_counter(); ctk, kk = ky.encaps_keys(tk, t_ntt, rho, m); torch.cuda.synchronize(); te.append(time.perf_counter() - t0)
        t0 = time.perf_counter(); k3 = ky.decaps_keys(s_ntt, tk, t_ntt, rho, ctk); torch.cuda.synchronize(); td.append(time.perf_counter() - t0)
    assert torch.equal(k3, kk)
    kem['distinct_keys'] = {'workload': 'the same batch with a key pair per value (keys from the batched keygen, resident in HBM; each matrix expanded from its rho inside the call)',
                            'encaps_ops_per_s': B / min(te), 'decaps_ops_per_s': B / min(td), 'key_bytes_per_value': 2 * 3 * 256 * 4 + 32}
    del s_ntt, tk, t_ntt, rho, ctk, kk, k3
    out['kyber_1_kem'] = kem
    ky.close()
    # ring-TESLA set 0 (n=512, q=8399873)
    G = load_golden('tesla_ref.txt')
    ts = eng.tesla(G['tesla0_params'])
    ts.set_key(*(G[f'tesla0_{x}'] for x in ('a1', 'a2', 's', 'e1', 'e2', 't1', 't2')))
    B, W = 16384, 24 * 512
    msgs = torch.randint(0, 256, (B, 64), dtype=torch.uint8, device='cuda', generator=g)
    tapes = torch.randint(-2**31, 2**31 - 1, (B, W), dtype=torch.int32, device='cuda', generator=g)
    ts.sign(msgs, tapes, fixed_len=64); torch.cuda.synchronize()
    tsn, tvf = [], []
    for _ in range(3):
        t0 = time.perf_counter(); sigs, siglens, trials, status = ts.sign(msgs, tapes, fixed_len=64); torch.cuda.synchronize(); tsn.append(time.perf_counter() - t0)
        t0 = time.perf_counter(); v = ts.verify(msgs, sigs, siglens, fixed_len=64); torch.cuda.synchronize(); tvf.append(time.perf_counter() - t0)
    ok = status == 1
    assert int(v[ok].sum()) == 0 and int(ok.sum()) > B - 20
    rt = {'workload': 'ring-TESLA set 0 (n=512, q=8399873) sign / verify, batch 16384, 64-byte messages, per-message CSPRNG word tapes resident in HBM',
          'sign_ops_per_s': B / min(tsn), 'verify_ops_per_s': B / min(tvf), 'mean_attempts_per_signature': float(trials[ok].float().mean())}
    ref = _cpu_ref('bench_tesla', 0, 3 if cpu_on else 0, {'sign_ops_per_s': 'sign_per_sec', 'verify_ops_per_s': 'verify_per_sec'},
                   '3 s of safecrypto_sign + 0.75 s of safecrypto_verify, set 0, 64-byte messages, one key, 1 thread')
    if ref:
        rt['cpu_reference'] = ref
    out['ring_tesla_0'] = rt
    ts.close()
    del msgs, tapes, sigs
    # RLWE-Enc (sets 0 and 1): fused encrypt / decrypt kernels, 6n CSPRNG words per encryption resident in HBM
    G = load_golden('rlwe_ref.txt')
    for tag, set_ in (('rlwe0', 0), ('rlwe1', 1)):
        r = eng.rlwe(G[f'{tag}_params'])
        r.set_pubkey(G[f'{tag}_kg_a0'], G[f'{tag}_kg_p0']); r.set_privkey(G[f'{tag}_kg_r2_0'])
        B = 262144
        msgs = torch.randint(0, 256, (B, r.msg_bytes), dtype=torch.uint8, device='cuda', generator=g)
        tapes = torch.randint(-2**31, 2**31 - 1, (B, r.enc_words), dtype=torch.int32, device='cuda', generator=g)
        ct = r.encrypt(msgs, tapes); torch.cuda.synchronize()
        te, td = [], []
        for _ in range(5):
            t0 = time.perf_counter(); ct = r.encrypt(msgs, tapes); torch.cuda.synchronize(); te.append(time.perf_counter() - t0)
            t0 = time.perf_counter(); mm = r.decrypt(ct); torch.cuda.synchronize(); td.append(time.perf_counter() - t0)
        wrong = int((mm != msgs).any(dim=1).sum())          # the scheme's own decryption-failure rate (same blocks fail in the reference)
        assert wrong < B // 10000
        tape_gbs = B * r.enc_words * 4 / min(te) / 1e9
        alg_bytes = r.enc_words * 4 + r.msg_bytes + r.ct_bytes        # per encryption: the tape and the block in, the ciphertext out
        alg_gbs = B * alg_bytes / min(te) / 1e9
        pm = _pmc('rlwe_encrypt') if (set_ == 0 and cpu_on) else None
        re = {'workload': f'RLWE-Enc set {set_} (n={r.n}) encrypt / decrypt, batch {B}, {r.msg_bytes}-byte blocks, one key pair, per-message CSPRNG word tapes resident in HBM',
              'encrypt_ops_per_s': B / min(te), 'decrypt_ops_per_s': B / min(td), 'tape_read_GB_per_s': tape_gbs, 'decryption_failures': wrong,
              'roofline': {'bound': 'hbm', 'achieved': alg_gbs, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s', 'frac': alg_gbs / HBM_PEAK_GBS,
                           'traffic': pm['hbm_bytes_per_op_2xfetch_plus_write'] if pm else None, 'algorithmic_bytes_per_op': alg_bytes,
                           'valu_instr_per_simd_cycle': pm.get('valu_instr_per_simd_cycle') if pm else None, 'counters': pm.get('source') if pm else None,
                           'note': 'encrypt: algorithmic bytes per block = the 24n-byte tape + n/8 B in + 2 n q_bits / 8 B out (rounds 1-2 priced the tape alone: '
                                   'tape_read_GB_per_s); the kernel is VALU-bound (3n Gaussian samples, two forward and one inverse transform per block): '
                                   'valu_instr_per_simd_cycle against ceilings of 0.25-0.5'}}
        ref = _cpu_ref('bench_rlwe', set_, 2 if cpu_on else 0, {'encrypt_ops_per_s': 'encrypt_per_sec', 'decrypt_ops_per_s': 'decrypt_per_sec'},
                       f'2 s of safecrypto_public_encrypt + 1 s of safecrypto_private_decrypt, set {set_}, one key pair, 1 thread')
        if ref:
            re['cpu_reference'] = ref
        out[f'rlwe_enc_{set_}'] = re
        del msgs, tapes, ct
        r.close()
    # ENS-KEM set 0 and DLP-IBE set 0 (extract / encrypt / decrypt)
    import numpy as np
    G = load_golden('ens_ref.txt')
    k = eng.ens(G['ens0_params']); k.set_pubkey(G['ens0_h']); k.set_privkey(G['ens0_g'], G['ens0_inv_g'])
    B = 65536
    tapes = torch.randint(-2**31, 2**31 - 1, (B, 1024), dtype=torch.int32, device='cuda', generator=g)
    k.encaps(tapes); torch.cuda.synchronize()
    te, td = [], []
    for _ in range(3):
        t0 = time.perf_counter(); ct, keys, used, st = k.encaps(tapes); torch.cuda.synchronize(); te.append(time.perf_counter() - t0)
        t0 = time.perf_counter(); k2 = k.decaps(ct); torch.cuda.synchronize(); td.append(time.perf_counter() - t0)
    assert int((st == 1).sum()) == B and int((k2 == keys).all(dim=1).sum()) >= B - B // 1000
    ek = {'workload': f'ENS-KEM set 0 (n=512, q=12289) encapsulation / decapsulation, batch {B}, one key pair, per-operation CSPRNG word tapes resident in HBM',
          'encaps_ops_per_s': B / min(te), 'decaps_ops_per_s': B / min(td)}
    ref = _cpu_ref('bench_ens', 0, 2 if cpu_on else 0, {'encaps_ops_per_s': 'encaps_per_sec', 'decaps_ops_per_s': 'decaps_per_sec'},
                   '2 s of safecrypto_encapsulation + 1 s of safecrypto_decapsulation, set 0, one key pair, 1 thread')
    if ref:
        ek['cpu_reference'] = ref
    out['ens_kem_0'] = ek
    del tapes, ct
    k.close()
    G = load_golden('dlp_ref.txt')
    par = np.asarray(G['dlp0_params'])
    n, q, q_bits, n_bits, set_, ubits = (int(x) for x in par[:6])
    f = eng.falcon(np.array([n, q, q_bits, n_bits, 0, 0, 0, 0], dtype=np.int32))
    f.set_privkey(np.asarray(G['dlp0_tree'], dtype=np.int64).view(np.float64))
    d = eng.dlp(par); d.set_pubkey(G['dlp0_h_ntt']); d.set_user_key(G['dlp0_uk_ntt2'])
    ident = np.asarray(G['dlp0_id2'], dtype=np.uint8)
    B = 65536
    ids = torch.from_numpy(np.tile(ident, (B, 1))).cuda()
    xt = torch.randint(-2**31, 2**31 - 1, (B, 4 * n), dtype=torch.int32, device='cuda', generator=g)
    msgs = torch.randint(0, 256, (B, d.msg_bytes), dtype=torch.uint8, device='cuda', generator=g)
    f.dlp_extract(ubits, ids, xt, fixed_len=len(ident)); d.encrypt(ids, msgs, xt, fixed_len=len(ident)); torch.cuda.synchronize()
    tx, te, td = [], [], []
    for _ in range(3):
        t0 = time.perf_counter(); sk, st = f.dlp_extract(ubits, ids, xt, fixed_len=len(ident)); torch.cuda.synchronize(); tx.append(time.perf_counter() - t0)
        t0 = time.perf_counter(); ct, es = d.encrypt(ids, msgs, xt, fixed_len=len(ident)); torch.cuda.synchronize(); te.append(time.perf_counter() - t0)
        t0 = time.perf_counter(); mm = d.decrypt(ct); torch.cuda.synchronize(); td.append(time.perf_counter() - t0)
    assert int((st == 1).sum()) == B and int((mm != msgs).any(dim=1).sum()) < B // 10000
    di = {'workload': f'DLP-IBE set 0 (n=512, q=4206593) extract (f64 tree sampler) / encrypt / decrypt, batch {B}, one master key, one identity, tapes resident in HBM',
          'extract_ops_per_s': B / min(tx), 'encrypt_ops_per_s': B / min(te), 'decrypt_ops_per_s': B / min(td)}
    ref = _cpu_ref('bench_dlp', 0, 3 if cpu_on else 0, {'extract_ops_per_s': 'extract_per_sec', 'encrypt_ops_per_s': 'encrypt_per_sec', 'decrypt_ops_per_s': 'decrypt_per_sec'},
                   '3 s of safecrypto_ibe_extract + 1.5 s each of encrypt / decrypt, set 0, 1 thread')
    if ref:
        di['cpu_reference'] = ref
    out['dlp_ibe_0'] = di
    f.close(); d.close()
    # ENS signature set 0 (n=512): Klein's sampler over the 1024 x 1024 basis, keyed from f | g | F | G
    G = load_golden('ensdlp_ref.txt')
    par = np.asarray(G['enssig0_params'])
    n = int(par[0])
    sg = eng.ens_dlp_signer(par); sg.set_privkey(G['enssig0_fgFG'])
    hp = par.copy(); hp[4] = 0
    vf = eng.falcon(hp); vf.set_pubkey(G['enssig0_h'])
    B = 16384
    msgs = torch.randint(0, 256, (B, 64), dtype=torch.uint8, device='cuda', generator=g)
    xt = torch.randint(-2**31, 2**31 - 1, (B, 4 * n), dtype=torch.int32, device='cuda', generator=g)
    sl = torch.full((B,), sg.sig_bytes, dtype=torch.int32, device='cuda')
    sigs, st = sg.sign(msgs, xt, fixed_len=64); torch.cuda.synchronize()
    ts_, tv = [], []
    for _ in range(3):
        t0 = time.perf_counter(); sigs, st = sg.sign(msgs, xt, fixed_len=64); torch.cuda.synchronize(); ts_.append(time.perf_counter() - t0)
        t0 = time.perf_counter(); vd = vf.ens_dlp_verify(msgs, sigs, sl, fixed_len=64); torch.cuda.synchronize(); tv.append(time.perf_counter() - t0)
    assert int((st == 1).sum()) == B and int((vd == 0).sum()) == B          # every signature made verifies
    es = {'workload': f'ENS signature set 0 (n=512, q=12289) sign (Klein sampler, 2n = 1024 sequential basis rows) / verify, batch {B}, one key pair, 64-byte messages, tapes resident in HBM',
          'sign_ops_per_s': B / min(ts_), 'verify_ops_per_s': B / min(tv)}
    if cpu_on:
        es['roofline'] = valu_roofline('ens_sign', es['sign_ops_per_s'], 64 + 16 * n + n * 14 // 8,
                                       'kleinr_kernel: 2n dependent basis rows per signature, each a 2n-term float dot product whose additions are sequential (the reference\'s order); '
                                       '64 B message + 4n CSPRNG words in, n q_bits-bit fields out')
    ref = _cpu_ref('bench_enssig', 0, 3 if cpu_on else 0, {'sign_ops_per_s': 'sign_per_sec', 'verify_ops_per_s': 'verify_per_sec'},
                   '3 s of safecrypto_sign + 0.75 s of safecrypto_verify, set 0, one key pair, 1 thread')
    if ref:
        es['cpu_reference'] = ref
    out['ens_sig_0'] = es
    sg.close(); vf.close()
    del msgs, xt, sigs
    # DLP signature set 0 (n=256): the same sampler over a 512 x 512 basis
    par = np.asarray(G['dlpsig0_params'])
    n = int(par[0])
    sg = eng.ens_dlp_signer(par); sg.set_privkey(G['dlpsig0_fgFG'])
    msgs = torch.randint(0, 256, (B, 64), dtype=torch.uint8, device='cuda', generator=g)
    xt = torch.randint(-2**31, 2**31 - 1, (B, 4 * n), dtype=torch.int32, device='cuda', generator=g)
    sigs, st = sg.sign(msgs, xt, fixed_len=64); torch.cuda.synchronize()
    ts_ = []
    for _ in range(3):
        t0 = time.perf_counter(); sigs, st = sg.sign(msgs, xt, fixed_len=64); torch.cuda.synchronize(); ts_.append(time.perf_counter() - t0)
    assert int((st == 1).sum()) == B
    out['dlp_sig_0'] = {'workload': f'DLP signature set 0 (n=256) sign (Klein sampler, 512 sequential basis rows), batch {B}, one key pair, 64-byte messages, tapes resident in HBM',
                        'sign_ops_per_s': B / min(ts_)}
    sg.close()
    return out


def cdt_extra(eng, torch):
    """the CDT sampler kernel alone: 12 algorithmic bytes per sample (8 in, 4 out), sigma = 215 (BLISS-B-I)"""
    s = eng.sampler(13.42, 215.0)
    g = torch.Generator(device='cuda'); g.manual_seed(5)
    out = {}
    for cnt in (1 << 24, 1 << 27):
        u = torch.randint(-2**63, 2**63 - 1, (cnt,), dtype=torch.int64, device='cuda', generator=g)
        o = s.sample(u); torch.cuda.synchronize()
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ev0.record()
        for _ in range(10):
            s.sample(u, out=o) if 'out' in s.sample.__code__.co_varnames else s.sample(u)
        ev1.record(); torch.cuda.synchronize()
        ms = ev0.elapsed_time(ev1) / 10
        gbs = cnt * 12 / (ms * 1e-3) / 1e9
        out[f'samples_{cnt}'] = {'samples_per_s': cnt / (ms * 1e-3), 'GB/s': gbs, 'frac': gbs / HBM_PEAK_GBS, 'working_set_MiB': cnt * 12 / 2**20}
        del u, o
    s.close()
    return {'workload': 'cdt_kernel: 64-bit CDT sampler (table and guide in LDS), uniform words resident in HBM', 'bound': 'hbm', 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
            'algorithmic_bytes_per_sample': 12, **out}


def north_star_summary(extra):
    """BASELINE configs[2..4] in a few hundred bytes each (the full blocks are under `extra`)"""
    def rl(b):
        if not b:
            return None
        return {k: b.get(k) for k in ('bound', 'achieved', 'peak', 'unit', 'frac', 'frac_4cycle_ceiling', 'traffic', 'algorithmic_bytes_per_op', 'valu_wave_instructions_per_op')} | \
               {'counters': 'live' if str(b.get('source', '')).startswith('live') else 'committed'}
    out = {}
    d = extra.get('dilithium3')
    if d and 'sign_ops_per_s' in d:
        out['dilithium3_b16384'] = {'sign_ops_per_s': d['sign_ops_per_s'], 'verify_ops_per_s': d['verify_ops_per_s'], 'sign_ms_hip_events': d.get('sign_ms_hip_events'),
                                    'verify_ms_hip_events': d.get('verify_ms_hip_events'),
                                    'cpu_thread_sign_ops_per_s': (d.get('cpu_reference') or {}).get('sign_ops_per_s'), 'cpu_thread_verify_ops_per_s': (d.get('cpu_reference') or {}).get('verify_ops_per_s'),
                                    'sign_speedup_vs_cpu_thread': d.get('sign_speedup_vs_cpu_thread'), 'distinct_keys_sign_ops_per_s': (d.get('distinct_keys') or {}).get('sign_ops_per_s'),
                                    'distinct_keys_verify_ops_per_s': (d.get('distinct_keys') or {}).get('verify_ops_per_s'),
                                    'roofline_sign': rl(d.get('roofline')), 'roofline_verify': rl(d.get('roofline_verify'))}
    b = extra.get('bliss_b_1')
    if b and 'sign_ops_per_s' in b:
        out['bliss_b_1_b16384'] = {'sign_ops_per_s': b['sign_ops_per_s'], 'verify_ops_per_s': b['verify_ops_per_s'], 'sign_ms_hip_events': b.get('sign_ms_hip_events'),
                                   'cpu_thread_sign_ops_per_s': (b.get('cpu_reference') or {}).get('sign_ops_per_s'), 'roofline_sign': rl(b.get('roofline'))}
    f = extra.get('falcon_1024')
    if f and 'batch_8192' in f:
        out['falcon_1024_b8192'] = {'sign_ops_per_s': f['batch_8192']['sign_ops_per_s'], 'verify_ops_per_s': f['batch_8192']['verify_ops_per_s'],
                                    'sign_ms_hip_events': f['batch_8192'].get('sign_ms_hip_events'),
                                    'cpu_thread_sign_ops_per_s': (f.get('cpu_reference') or {}).get('sign_ops_per_s'), 'roofline_sign': rl(f.get('roofline'))}
    return out


def check_fracs(node, path='line'):
    """no roofline fraction anywhere in the line may exceed 1 (a 1.23 went out unnoticed)"""
    if isinstance(node, dict):
        for k, v in node.items():
            if k.startswith('frac') and isinstance(v, (int, float)) and k != 'frac_4cycle_ceiling':
                assert 0.0 <= v <= 1.0, f'{path}.{k} = {v}: a roofline fraction above 1 means the numerator or the peak is wrong'
            check_fracs(v, path + '.' + k)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=200)
    ap.add_argument('--warmup', type=int, default=20)
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-extras', action='store_true', help='only the contract line (profiling runs)')
    ap.add_argument('--cpu-seconds', type=float, default=10.0)
    ap.add_argument('--no-spin', action='store_true', help='leave the runtime\'s host-thread scheduling at its default (the bench asks it to spin in synchronisations)')
    ap.add_argument('--no-live-pmc', action='store_true', help='scheme rooflines from the committed counter pass instead of rocprofv3 child processes of this run')
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    global PMC_LIVE
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    PMC_LIVE = not args.no_live_pmc and world == 1
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit(f'--gpus {args.gpus} needs torch.distributed.run with {args.gpus} ranks (WORLD_SIZE={world})')
    if not torch.cuda.is_available():
        raise SystemExit('bench.py needs a HIP device: the engine has no CPU fallback')
    torch.cuda.set_device(local_rank)
    if not args.no_spin:
        # the timed region is K ~35 us kernels (0.7 ms at the driver's K = 20) bracketed by synchronisations: a host thread that sleeps in them adds its wake-up
        # latency to every figure (a box with a slow host gave 2.9 G NTT/s from the same 34 us kernel); ask the runtime to spin instead
        try:
            import ctypes
            hip = None
            for name in ('libamdhip64.so.7', 'libamdhip64.so.6', 'libamdhip64.so'):
                try:
                    hip = ctypes.CDLL(name); break
                except OSError:
                    continue
            if hip is not None:
                hip.hipSetDeviceFlags(ctypes.c_uint(1))        # hipDeviceScheduleSpin
        except Exception:
            pass
    if world > 1:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        dist.init_process_group('nccl', rank=rank, world_size=world, device_id=torch.device('cuda', local_rank))

    from libsafecrypto_amd import Engine
    eng = Engine(local_rank)
    plan = eng.plan(Q, N)

    gen = torch.Generator(device='cuda')
    gen.manual_seed(1 + rank)
    a = torch.randint(0, Q, (BATCH, N), dtype=torch.int32, device='cuda', generator=gen)
    bhat = torch.randint(0, Q, (BATCH, N), dtype=torch.int32, device='cuda', generator=gen)
    out = torch.empty_like(a)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # the K timed steps are K kernel launches captured in one HIP graph (sc_hip_polymul_ntt_steps; it runs the plain loop if the capture is refused):
    # the loop is launch-bound - 37 us of kernel against ~2 us of host work per plain call.  Captured (and checked) before the timed region.
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        for _ in range(args.warmup):
            plan.polymul_ntt(a, bhat, out)
        plan.polymul_ntt_steps(a, bhat, out, args.steps)            # capture + one untimed replay
    barrier()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    with torch.cuda.stream(side):                                    # HIP events on the stream the kernels run on, over the timed region
        ev0.record(side)
        plan.polymul_ntt_steps(a, bhat, out, args.steps)
        ev1.record(side)
    torch.cuda.synchronize()
    barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device='cuda')
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # size-independent self check of the timed output (never timed): NTT(c) == NTT(a) o bhat
    assert torch.equal(plan.fwd(out), plan.pointwise(plan.fwd(a), bhat)), 'bench output failed the convolution check'

    cpu_on = (not args.no_cpu_baseline) and world == 1 and rank == 0
    cpu_sec = args.cpu_seconds if cpu_on else 0.0

    # ---- the scheme-level legs, on every rank (independent batches: weak scaling), gathered afterwards
    extra = {}
    if not args.no_extras:
        for name, fn in (('dilithium3', lambda: {'dilithium3': dilithium_extra(eng, torch, cpu_sec, distinct=(world == 1))}),
                         ('bliss_b_1', lambda: bliss_extra(eng, torch, cpu_sec)),
                         ('falcon', lambda: falcon_extra(eng, torch, cpu_sec, sets=(('falcon2', 2, (8192, 131072)), ('falcon0', 0, (131072,))) if world == 1
                                                         else (('falcon2', 2, (8192,)),)))):
            try:
                barrier()
                extra.update(fn())
            except Exception as e:        # an extra must never break the contract line
                extra[name] = {'error': str(e)}
        if world > 1:
            # per-GPU and aggregate ops/s of the BASELINE legs: every rank signed / verified its own batch of the stated size
            keys = [('dilithium3', 'sign_ops_per_s'), ('dilithium3', 'verify_ops_per_s'), ('bliss_b_1', 'sign_ops_per_s'), ('bliss_b_1', 'verify_ops_per_s')]
            mine = [float(extra.get(k, {}).get(f, 0.0) or 0.0) for k, f in keys]
            mine.append(float(extra.get('falcon_1024', {}).get('batch_8192', {}).get('sign_ops_per_s', 0.0) or 0.0))
            tt = torch.tensor(mine, dtype=torch.float64, device='cuda')
            allr = [torch.zeros_like(tt) for _ in range(world)]
            dist.all_gather(allr, tt)
            if rank == 0:
                per = torch.stack(allr).cpu().tolist()
                names = [f'{k}.{f}' for k, f in keys] + ['falcon_1024.batch_8192.sign_ops_per_s']
                extra['multi_gpu'] = {'scaling': 'weak (each rank runs the stated batch on its own GPU; no data-path collective)',
                                      'per_gpu': {nm: [row[i] for row in per] for i, nm in enumerate(names)},
                                      'aggregate': {nm: sum(row[i] for row in per) for i, nm in enumerate(names)}}

    if rank == 0:
        # dominant kernel's duration, HIP events on the stream it is launched on
        kern_ms = ev0.elapsed_time(ev1) / args.steps                 # average over the K launches of the timed region
        kern_ms_plain = plan.time_op(2, a, bhat, out, 50)            # the same kernel as 50 plain launches between two events (C loop)
        achieved = BATCH * BYTES_PER_TRIPLE / (kern_ms * 1e-3) / 1e9
        # per-launch HBM bytes of this kernel (2 x FETCH_SIZE + WRITE_SIZE): counters collected over a child process of THIS run (tools/pmc_leg.py: the same
        # fused product at the same batch under rocprofv3 --pmc, one pass per counter as the guide prescribes), else the committed passes of tools/profile_r03.sh
        traffic = traffic_past = None
        traffic_source = None
        pm = _pmc('polymul') if (cpu_on and PMC_LIVE) else None
        if pm and str(pm.get('source', '')).startswith('live'):
            traffic = pm['hbm_bytes_per_op_2xfetch_plus_write'] * pm['batch']
            traffic_source = 'live: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE over `python3 tools/bench_scheme.py polymul` as a child of this run'
        for name in ('r03_pmc_traffic.json', 'pmc_traffic.json'):
            tp = os.path.join(ROOT, 'profiles', name)
            if not os.path.exists(tp):
                continue
            try:
                tj = json.load(open(tp))
                if traffic is None:
                    traffic = tj.get('polymul_ntt_n256_b65536_bytes'); traffic_source = 'committed: profiles/' + name
                traffic_past = tj.get('polymul_ntt_n256_b1048576_bytes')
                break
            except Exception:
                pass
        # the same kernel over a working set far past the Infinity Cache (3 x 1 GiB): what the HBM itself sustains for this access pattern
        past = {}
        try:
            a2 = torch.randint(0, Q, (PAST_L3_BATCH, N), dtype=torch.int32, device='cuda', generator=gen)
            b2 = torch.randint(0, Q, (PAST_L3_BATCH, N), dtype=torch.int32, device='cuda', generator=gen)
            o2 = torch.empty_like(a2)
            plan.polymul_ntt(a2, b2, o2); torch.cuda.synchronize()
            ms2 = plan.time_op(2, a2, b2, o2, 10)
            g2 = PAST_L3_BATCH * BYTES_PER_TRIPLE / (ms2 * 1e-3) / 1e9
            past = {'achieved_past_l3': g2, 'frac_past_l3': g2 / HBM_PEAK_GBS, 'past_l3_batch': PAST_L3_BATCH, 'past_l3_working_set_GiB': 3 * PAST_L3_BATCH * N * 4 / 2**30,
                    'past_l3_kernel_ms': ms2, 'past_l3_NTT_per_s': 2.0 * PAST_L3_BATCH / (ms2 * 1e-3), 'traffic_past_l3': traffic_past}
            del a2, b2, o2
        except Exception as e:
            past = {'achieved_past_l3': None, 'past_l3_error': str(e)}
        fwd_ms = {}
        for (q2, n2, b2) in [(7681, 256, 65536), (8380417, 256, 65536), (12289, 512, 32768), (12289, 1024, 16384), (7681, 256, 1 << 20), (12289, 1024, 1 << 18)]:
            p2 = plan if (q2, n2) == (Q, N) else eng.plan(q2, n2)
            x = torch.randint(0, q2, (b2, n2), dtype=torch.int32, device='cuda', generator=gen)
            y = torch.empty_like(x)
            p2.fwd(x, y)
            ms = p2.time_op(0, x, None, y, 50 if b2 <= 65536 else 10)
            gbs = b2 * 8 * n2 / (ms * 1e-3) / 1e9
            fwd_ms[f'q{q2}_n{n2}_b{b2}'] = {'NTT/s': b2 / (ms * 1e-3), 'GB/s': gbs, 'frac': gbs / HBM_PEAK_GBS, 'working_set_MiB': b2 * 8 * n2 / 2**20}
            del x, y
        value = 2.0 * BATCH * world * args.steps / elapsed
        res = {
            'metric': 'NTT/s (n=256, q=7681; fused fwd NTT + pointwise + inv NTT over batch 65536 per GPU, 2 NTTs per polynomial)',
            'value': value, 'unit': 'NTT/s', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': elapsed / args.steps * 1e3, 'higher_is_better': True, 'scaling': 'weak',
            'vs_baseline': None, 'dtype': 'u32', 'data': 'synthetic',
            'config': {'workload': 'BASELINE configs[1]: Kyber set 1 (n=256, q=7681 - the reference has no q=3329) '
                                   'batched NTT + pointwise-mul + INTT, batch=65536 per GPU', 'n': N, 'q': Q,
                       'batch_per_gpu': BATCH, 'parallelism': f'shard{world} (independent polynomials, no collective)'},
            'roofline': {'bound': 'hbm', 'achieved': achieved, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                         'frac': achieved / HBM_PEAK_GBS, 'traffic': traffic, 'traffic_source': traffic_source,
                         'kernel': 'ntt_kernel<8,*,*,OP_MULNTT>', 'kernel_ms': kern_ms, 'kernel_ms_plain_launches': kern_ms_plain,
                         'algorithmic_bytes_per_launch': BATCH * BYTES_PER_TRIPLE,
                         'note': 'achieved / frac are at the BASELINE batch, whose 192 MiB of operands fit the 256 MiB Infinity Cache when replayed back to back '
                                 '(FETCH_SIZE counts its hits); achieved_past_l3 / frac_past_l3 are the same kernel over 3 GiB of operands, i.e. what HBM itself '
                                 'sustains.  The kernel is integer-VALU-issue-bound at either size (DESIGN 4.1), so the two differ little.', **past},
            'extra': {'fwd_ntt': fwd_ms, **extra},
        }
        if world == 1 and not args.no_extras:
            try:
                res['extra']['cdt_sampler'] = cdt_extra(eng, torch)
            except Exception as e:
                res['extra']['cdt_sampler'] = {'error': str(e)}
            try:
                res['extra'].update(kyber_tesla_more_extra(eng, torch, cpu_on))
            except Exception as e:
                res['extra']['kyber_tesla_more'] = {'error': str(e)}
        if cpu_on:
            res['cpu_baseline'] = cpu_baseline(args.cpu_seconds)
        res['north_star'] = north_star_summary(res['extra'])          # LAST key: the driver keeps the tail of the line
        check_fracs(res)
        print(json.dumps(res), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
