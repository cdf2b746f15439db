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

Output: the LAST stdout line is the contract line - compact (< 6 KB, asserted; the driver keeps the last 8 KB of stdout), medians with their spread,
`roofline`, `cpu_baseline`, `north_star`; the full tree of every leg goes to bench_extra.json beside this script and only its path is printed (round 3's
single 27.7 KB line was not parsed by the driver).  --shard cuts the BASELINE batches across the ranks (strong scaling) instead of repeating them (weak, the default).
Sign figures of the schemes that draw randomness (BLISS-B, Falcon, ring-TESLA) are END TO END: a 32-byte seed per message, the reference's AES-256
CTR-DRBG on the device inside the timed call (sc_hip_*_sign_seeded); the tape-resident figure of rounds 1-3 is reported beside it, labelled.

`roofline` blocks: the headline's is HBM (algorithmic bytes / HIP-event kernel time / 8 TB/s, at the BASELINE size and at a working
set far past the 256 MiB Infinity Cache); the scheme legs' are VALU: wave-level VALU instructions and HBM bytes per operation counted by
rocprofv3 --pmc over the SAME leg in a child process of this run (tools/pmc_leg.py; the verify legs load signatures made by a separate
process, so no signing kernel is in their counts), x the operations per second measured here, against the chip's issue ceiling of one
wave64 instruction per 2 cycles per SIMD (1228.8 G wave-instr/s at 2.4 GHz - the guide's 157.3 TF vector peak; tools/exp_valu_cycles.hip
measured which instructions reach it and which take 4 cycles: profiles/r03_valu_issue_cycles.txt).  A `frac` above 1 anywhere is listed under `frac_violations` and the exit code is 3.
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
            _PMC_COMMITTED = json.load(open(os.path.join(ROOT, 'profiles', 'r04_scheme_pmc.json' if os.path.exists(os.path.join(ROOT, 'profiles', 'r04_scheme_pmc.json')) else 'r03_scheme_pmc.json')))
        except Exception:
            _PMC_COMMITTED = {}
    r = _PMC_COMMITTED.get(leg)
    if r:
        r = dict(r)
        r['source'] = 'committed: profiles/r04_scheme_pmc.json (tools/pmc_leg.py run by tools/profile_r04.sh on an earlier box) - NOT this run'
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


REPS = 7                         # repetitions of every timed leg; figures are MEDIANS, the spread is reported beside them


def _spread(ts):
    """median / min / max of a list of seconds, as milliseconds"""
    return {'median_ms': statistics.median(ts) * 1e3, 'min_ms': min(ts) * 1e3, 'max_ms': max(ts) * 1e3, 'reps': len(ts)}


def _med(ts):
    return statistics.median(ts)


def _timed(torch, fn, reps=REPS):
    """(median host seconds, median HIP-event milliseconds on the current stream, last result, list of host seconds) of fn() + synchronize"""
    th, te = [], []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter(); e0.record(); out = fn(); e1.record(); torch.cuda.synchronize(); th.append(time.perf_counter() - t0)
        te.append(e0.elapsed_time(e1))
    return statistics.median(th), statistics.median(te), out, th


def _timed_pair(torch, f1, f2, reps=REPS):
    """two dependent legs (sign then verify ...) timed alternately: lists of host seconds, and the last results"""
    t1, t2 = [], []
    o1 = o2 = None
    for _ in range(reps):
        t0 = time.perf_counter(); o1 = f1(); torch.cuda.synchronize(); t1.append(time.perf_counter() - t0)
        t0 = time.perf_counter(); o2 = f2(o1); torch.cuda.synchronize(); t2.append(time.perf_counter() - t0)
    return t1, t2, o1, o2


def _sweep(torch, make, batches):
    """per-leg batch sweep: {batch: {ops_per_s, ms}} from the median of REPS calls"""
    out = {}
    for B in batches:
        try:
            fn = make(B)
            fn(); torch.cuda.synchronize()
            t, _, _, ts = _timed(torch, fn, REPS)
            out[str(B)] = {'ops_per_s': B / t, **_spread(ts)}
        except Exception as e:
            out[str(B)] = {'error': str(e)}
    return out


SWEEP = (1024, 2048, 4096, 8192, 16384)


def dilithium_extra(eng, torch, cpu_seconds, batch=16384, distinct=True, sweep=True):
    """North-star companion figure: Dilithium-III (reference set 3: k=6, l=5) full sign / verify of `batch` 64-byte
    messages under one key, messages and signatures resident in HBM (BASELINE configs[2]).  Signing is deterministic
    (USE_DETERMINISTIC_DILITHIUM): no CSPRNG is drawn, the figure is end to end."""
    from tests.golden_io import load_golden
    G = load_golden('dilithium_ref.txt')       # key material produced by the reference's keygen (data, not code)
    tag = 'dil3'
    d = eng.dilithium(G[f'{tag}_params'])
    d.set_privkey(*(G[f'{tag}_{x}'] for x in ('s1', 's2', 't', 'K', 'tr', 'Ahat')))
    d.set_pubkey(G[f'{tag}_t1'], G[f'{tag}_rho'], G[f'{tag}_Ahat'])
    g = torch.Generator(device='cuda')
    g.manual_seed(3)
    msgs = torch.randint(0, 256, (batch, 64), dtype=torch.uint8, device='cuda', generator=g)
    sigs, siglens, trials = d.sign(msgs, fixed_len=64)
    torch.cuda.synchronize()
    ts, ts_ev, (sigs, siglens, trials), ts_all = _timed(torch, lambda: d.sign(msgs, fixed_len=64))
    tv, tv_ev, v, tv_all = _timed(torch, lambda: d.verify(msgs, sigs, siglens, fixed_len=64))
    assert int(v.sum()) == 0, 'bench signatures failed verification'
    out = {'workload': f'BASELINE configs[2]: Dilithium-III (n=256, q=8380417, k=6, l=5) sign + verify, batch {batch}, 64-byte messages, one key',
           'batch': batch, 'sign_ops_per_s': batch / ts, 'verify_ops_per_s': batch / tv, 'sign': _spread(ts_all), 'verify': _spread(tv_all),
           'sign_ms_hip_events': ts_ev, 'verify_ms_hip_events': tv_ev,
           'mean_attempts_per_signature': float(trials.float().mean()), 'dtype': 'u32',
           'timing': f'median of {REPS} calls (host clock around call + synchronize); deterministic signing: no CSPRNG on either side'}
    out['roofline'] = valu_roofline('dil_sign', out['sign_ops_per_s'], 64 + 3370,
                                    'sign: 64 B message in + 3370 B signature out per operation; the work is ~3.2 attempts of 35 Keccak permutations, 11 transforms and 30 '
                                    'pointwise products each - instruction-bound, not HBM-bound')
    out['roofline_verify'] = valu_roofline('dil_verify', out['verify_ops_per_s'], 64 + 3370 + 1, 'verify: message + signature in, one verdict out')
    if sweep:
        def mk_sign(B):
            m = msgs[:B].contiguous()
            return lambda: d.sign(m, fixed_len=64)

        def mk_verify(B):
            m = msgs[:B].contiguous(); sg, sl, _ = d.sign(m, fixed_len=64)
            return lambda: d.verify(m, sg, sl, fixed_len=64)
        bs = [B for B in SWEEP if B <= batch]
        out['batch_sweep'] = {'sign': _sweep(torch, mk_sign, bs), 'verify': _sweep(torch, mk_verify, bs)}
    # the configuration's second leg: `batch` DISTINCT key pairs (generated on the device from 32-byte seeds), one message under each
    if distinct:
        try:
            seeds = torch.randint(0, 256, (batch, 32), dtype=torch.uint8, device='cuda', generator=g)
            s1, s2, t, rho, K, tr, t1 = d.keygen(seeds, int(G[f'{tag}_eta_bits'][0]))
            t1f = t1.reshape(batch, -1).contiguous()
            d.sign_keys(s1, s2, t, rho, K, tr, msgs, fixed_len=64); torch.cuda.synchronize()
            ts2, tv2, (ks, kl, ktr), kv = _timed_pair(torch, lambda: d.sign_keys(s1, s2, t, rho, K, tr, msgs, fixed_len=64),
                                                      lambda o: d.verify_keys(t1f, rho, msgs, o[0], o[1], fixed_len=64), 5)
            assert int(kv.sum()) == 0, 'distinct-key signatures failed verification'
            out['distinct_keys'] = {'workload': f'the same with {batch} distinct key pairs, one message under each: per-item matrix from rho, NTT of s1 / s2 / t0 / t1 and tr derived '
                                                'on the device inside the timed call', 'sign_ops_per_s': batch / _med(ts2), 'verify_ops_per_s': batch / _med(tv2),
                                    'sign': _spread(ts2), 'verify': _spread(tv2),
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


def falcon_extra(eng, torch, cpu_seconds, sets=(('falcon2', 2, (8192, 131072)), ('falcon0', 0, (131072,))), cfg_batch=8192, sweep=True):
    """BASELINE configs[4]: Falcon-1024 (reference set 2) signing - SHAKE hash to the ring, fast Fourier sampling over the LDL
    tree in binary64, norm test, packing - at the configuration's batch per GPU and at a throughput-saturating batch;
    Falcon-512 (set 0) beside it.  One key per set (the reference keygen's, from the fixture).  Two sign figures per batch: END TO END
    (sign_seeded: 32-byte seed per message -> the reference's AES-256 CTR-DRBG on the device -> 16 KB of pool words -> sign, one call) and
    with the word tapes already resident in HBM (what rounds 1-3 reported)."""
    import numpy as np
    from tests.golden_io import load_golden
    G = load_golden('falcon_ref.txt')
    out = {}
    g = torch.Generator(device='cuda'); g.manual_seed(13)
    for tag, set_, batches in sets:
        f = eng.falcon(G[f'{tag}_params'])
        f.set_privkey(np.asarray(G[f'{tag}_sk'], dtype=np.int64).view(np.float64)); f.set_pubkey(G[f'{tag}_h'])
        res = {'workload': f'Falcon-{f.n} (reference set {set_}, q=12289) sign with the FFT tree sampler on the GPU (f64) + verify, 64-byte messages', 'dtype': 'f64',
               'timing': f'median of {REPS} calls; sign_ops_per_s = end to end with the CSPRNG inside the call (seeds in HBM), tape_resident_sign_ops_per_s = word tapes in HBM'}
        for B in batches:
            msgs = torch.randint(0, 256, (B, 64), dtype=torch.uint8, device='cuda', generator=g)
            seeds = torch.randint(0, 256, (B, 32), dtype=torch.uint8, device='cuda', generator=g)
            tapes = eng.drbg_words(seeds, 4 * f.n)
            f.sign(msgs, tapes, fixed_len=64); f.sign_seeded(msgs, seeds, fixed_len=64); torch.cuda.synchronize()
            tt, tt_ev, (sigs_t, _, _), tt_all = _timed(torch, lambda: f.sign(msgs, tapes, fixed_len=64))
            ts, ts_ev, (sigs, siglens, status), ts_all = _timed(torch, lambda: f.sign_seeded(msgs, seeds, fixed_len=64))
            assert torch.equal(sigs, sigs_t), 'seeded and tape-fed signatures differ'
            tv, tv_ev, v, tv_all = _timed(torch, lambda: f.verify(msgs, sigs, siglens, fixed_len=64))
            td, td_ev, _, td_all = _timed(torch, lambda: eng.drbg_words(seeds, 4 * f.n))
            ok = status == 1
            assert int(v[ok].sum()) == 0 and int(ok.sum()) > B - B // 100
            res[f'batch_{B}'] = {'sign_ops_per_s': B / ts, 'tape_resident_sign_ops_per_s': B / tt, 'verify_ops_per_s': B / tv,
                                 'sign': _spread(ts_all), 'tape_resident_sign': _spread(tt_all), 'verify': _spread(tv_all),
                                 'sign_ms_hip_events': ts_ev, 'tape_resident_sign_ms_hip_events': tt_ev, 'verify_ms_hip_events': tv_ev,
                                 'drbg_alone': {'ms': td * 1e3, 'tape_GB_per_s': B * 16 * f.n / td / 1e9, 'bytes_per_signature': 16 * f.n}}
            del msgs, tapes, sigs, sigs_t, seeds
        ref = _cpu_ref('bench_falcon', set_, min(cpu_seconds, 4.0), {'sign_ops_per_s': 'sign_per_sec', 'verify_ops_per_s': 'verify_per_sec'},
                       f'{min(cpu_seconds, 4.0):.0f} s of safecrypto_sign + safecrypto_verify, set {set_}, 64-byte messages, one key, 1 thread (its CSPRNG inside)')
        if ref:
            res['cpu_reference'] = ref
            res['sign_speedup_vs_cpu_thread'] = res[f'batch_{batches[-1]}']['sign_ops_per_s'] / ref['sign_ops_per_s']
        if set_ == 2:
            res['workload'] = 'BASELINE configs[4]: ' + res['workload'] + f' (batch_{cfg_batch} is the configuration as stated)'
            r = res[f'batch_{cfg_batch}']['tape_resident_sign_ops_per_s']
            res['roofline'] = valu_roofline('falcon_sign', r, 64 + 1536 + 16384,
                                            f'batch {cfg_batch}, tape-resident leg (the counter pass profiles that call): 64 B message + 16 KB tape in, 1536 B signature out per operation '
                                            '(sampler alone, SURVEY 8(d) row 5: c 4 KB in, s1 + s2 8 KB out); the sampler is a 2n-leaf sequential walk per signature: latency- and '
                                            'f64-issue-bound, its HBM traffic is the spilled working set')
            if sweep:
                def mk(B):
                    m = torch.randint(0, 256, (B, 64), dtype=torch.uint8, device='cuda', generator=g)
                    sd = torch.randint(0, 256, (B, 32), dtype=torch.uint8, device='cuda', generator=g)
                    return lambda: f.sign_seeded(m, sd, fixed_len=64)
                res['batch_sweep'] = {'sign': _sweep(torch, mk, [B for B in SWEEP if B <= cfg_batch])}
        if set_ == 2 and cfg_batch == 8192:
            # SURVEY 8(d) distinct keys: a resident key set, a key index per message (the fixture key loaded 64 times: distinct memory, same values)
            NK, B = 64, 8192
            sk = np.asarray(G[f'{tag}_sk'], dtype=np.int64).view(np.float64)
            f.set_privkeys(np.tile(sk, (NK, 1))); f.set_pubkeys(np.tile(np.asarray(G[f'{tag}_h'], dtype=np.int16), (NK, 1)))
            msgs = torch.randint(0, 256, (B, 64), dtype=torch.uint8, device='cuda', generator=g)
            tapes = torch.randint(-2**31, 2**31 - 1, (B, 4 * f.n), dtype=torch.int32, device='cuda', generator=g)
            ko = (torch.arange(B, device='cuda') // (B // NK)).to(torch.int32)        # grouped by key, as a multi-tenant signer would submit them
            f.sign_keys(ko, msgs, tapes, fixed_len=64); torch.cuda.synchronize()
            ts, tv, (sigs, siglens, status), v = _timed_pair(torch, lambda: f.sign_keys(ko, msgs, tapes, fixed_len=64),
                                                             lambda o: f.verify_keys(ko, msgs, o[0], o[1], fixed_len=64), 5)
            assert int(v[status == 1].sum()) == 0
            ki = (torch.arange(B, device='cuda') % NK).to(torch.int32)                  # the same keys interleaved: no two neighbours share one
            f.sign_keys(ki, msgs, tapes, fixed_len=64); torch.cuda.synchronize()
            t_il, _, _, _ = _timed(torch, lambda: f.sign_keys(ki, msgs, tapes, fixed_len=64), 3)
            res['distinct_keys'] = {'workload': f'batch {B}, a key set of {NK} resident keys, key index per message, tapes resident; messages grouped by key (runs of {B // NK}): the wavefront-per-8 form, '
                                                'keys interleaved (no two neighbours share one): the call regroups the batch by key, signs it on the same kernel and scatters the signatures back', 'sign_ops_per_s_interleaved': B / t_il,
                                    'sign_ops_per_s': B / _med(ts), 'verify_ops_per_s': B / _med(tv),
                                    'key_bytes_per_key': int((f.n * 15 + (f.n << 5)) * 8 + 2 * f.n)}
            del msgs, tapes, sigs
        out[f'falcon_{f.n}'] = res
        f.close()
    return out


def bliss_extra(eng, torch, cpu_seconds, batch=16384, sweep=True, variants=True):
    """BASELINE configs[3]: BLISS-B-I sign with the CDT sampler on the GPU, batch 16384; key from the reference's keygen (fixture).  Two sign figures:
    END TO END (sign_seeded: 32-byte seed per message, the reference's AES-256 CTR-DRBG on the device expands the words an attempt draws, inside the call)
    and with the word tapes already resident in HBM (what rounds 1-3 reported)."""
    import numpy as np
    from tests.golden_io import load_golden
    G = load_golden('bliss_ref.txt')
    b = eng.bliss(G['bliss1_params']); b.set_key(G['bliss1_f'], G['bliss1_g'], G['bliss1_a'])
    g = torch.Generator(device='cuda'); g.manual_seed(11)
    B, W = batch, 4 * 2048 + 64
    msgs = torch.randint(0, 256, (B, 64), dtype=torch.uint8, device='cuda', generator=g)
    seeds = torch.randint(0, 256, (B, 32), dtype=torch.uint8, device='cuda', generator=g)
    tapes = eng.drbg_words(seeds, W)
    b.sign(msgs, tapes, fixed_len=64); b.sign_seeded(msgs, seeds, fixed_len=64); torch.cuda.synchronize()
    tt, tt_ev, (sigs_t, _, _, status_t), tt_all = _timed(torch, lambda: b.sign(msgs, tapes, fixed_len=64))
    ts, ts_ev, (sigs, siglens, trials, status), ts_all = _timed(torch, lambda: b.sign_seeded(msgs, seeds, fixed_len=64))
    same = (status_t == 1) & (status == 1)
    assert torch.equal(sigs[same], sigs_t[same]) and int(same.sum()) > B - 40, 'seeded and tape-fed signatures differ'
    tv, tv_ev, v, tv_all = _timed(torch, lambda: b.verify(msgs, sigs, siglens, fixed_len=64))
    w1 = (int(b.L.sc_hip_bliss_tape_words(512, 2)) + 3) & ~3
    td, _, _, _ = _timed(torch, lambda: eng.drbg_words(seeds, w1))
    ok = status == 1
    assert int(v[ok].sum()) == 0 and int(ok.sum()) > B - 20
    attempts = float(trials[ok].float().mean())
    out = {'workload': f'BASELINE configs[3]: BLISS-B-I (n=512, q=12289) sign with the CDT sampler on the GPU, batch {B}, 64-byte messages', 'batch': B,
           'sign_ops_per_s': B / ts, 'tape_resident_sign_ops_per_s': B / tt, 'verify_ops_per_s': B / tv,
           'sign': _spread(ts_all), 'tape_resident_sign': _spread(tt_all), 'verify': _spread(tv_all),
           'sign_ms_hip_events': ts_ev, 'tape_resident_sign_ms_hip_events': tt_ev, 'verify_ms_hip_events': tv_ev,
           'drbg_alone': {'ms': td * 1e3, 'tape_GB_per_s': B * w1 * 4 / td / 1e9, 'bytes_per_message_first_pair': w1 * 4},
           'mean_attempts_per_signature': attempts, 'undecided_near_ties': int((status == 5).sum()), 'dtype': 'u32',
           'timing': f'median of {REPS} calls; sign_ops_per_s = end to end with the CSPRNG inside the call (seeds in HBM), tape_resident_sign_ops_per_s = word tapes in HBM'}
    out['drbg_alone']['roofline'] = valu_roofline('drbg', B / td, 32 + w1 * 4,
                                                  'drbg_fill_kernel: a 32-byte seed in, the 8208 bytes of a first pair of attempts out per message (513 AES-256 blocks); vector-ALU-bound on the byte '
                                                  'permutes and rotates of the T-table rounds (4-cycle class: the ceiling that applies is frac_4cycle_ceiling)', top=1)
    out['roofline'] = valu_roofline('bliss_sign', out['tape_resident_sign_ops_per_s'], 64 + 986 + 8 * 1024 * (attempts + 1) / 2,
                                    'tape-resident leg (the counter pass profiles that call); per signature: 64 B message + 986 B signature + 8 KB of uniform words per fresh pair of '
                                    'Gaussian vectors (12 B per sample in the sampler kernel)')
    out['roofline_verify'] = valu_roofline('bliss_verify', out['verify_ops_per_s'], 64 + 986 + 1, 'verify: message + signature in, one verdict out')
    ref = _cpu_ref('bench_bliss', 1, min(cpu_seconds, 3.0), {'sign_ops_per_s': 'sign_per_sec', 'verify_ops_per_s': 'verify_per_sec'},
                   f'{min(cpu_seconds, 3.0):.0f} s of safecrypto_sign + safecrypto_verify, set 1 (64-bit CDT), 64-byte messages, one key, 1 thread (its CSPRNG inside)')
    if ref:
        out['cpu_reference'] = ref
    if sweep:
        def mk(Bs):
            m, sd = msgs[:Bs].contiguous(), seeds[:Bs].contiguous()
            return lambda: b.sign_seeded(m, sd, fixed_len=64)
        out['batch_sweep'] = {'sign': _sweep(torch, mk, [x for x in SWEEP if x <= B])}
    if not variants:
        b.close()
        return {'bliss_b_1': out}
    # SURVEY 8(d) distinct keys: a key pair per message (3 KB of key material per signature instead of one cache-resident key)
    kt = torch.randint(-2**31, 2**31 - 1, (B, 2048), dtype=torch.int32, device='cuda', generator=g)
    f, gk, a, _, kst = b.keygen(G['bliss1_kg_nz'], kt)
    if int((kst == 1).sum()) == B:
        b.sign_keys(f, gk, a, msgs, tapes, fixed_len=64); torch.cuda.synchronize()
        tk, tkv, (s2, l2, _, st2), v2 = _timed_pair(torch, lambda: b.sign_keys(f, gk, a, msgs, tapes, fixed_len=64), lambda o: b.verify_keys(a, msgs, o[0], o[1], fixed_len=64), 5)
        assert int(v2[st2 == 1].sum()) == 0
        out['distinct_keys'] = {'workload': 'the same batch with a key pair per message (keys from the batched keygen, resident in HBM; tapes resident)',
                                'sign_ops_per_s': B / _med(tk), 'verify_ops_per_s': B / _med(tkv), 'key_bytes_per_signature': 3 * 512 * 2}
    # the signatures in the reference's SC_ENTROPY_HUFFMAN_STATIC coding (SC_FLAG_0_ENTROPY_HUFFMAN; the code tables are the reference's data, from the fixture)
    H = load_golden('bliss_huff_ref.txt')
    pick = lambda k: ((H[f'huff{k}_code'].astype(np.int64) & 0xFFFFFFFF).astype(np.uint32), H[f'huff{k}_bits'].astype(np.uint16))
    b.set_entropy((pick(6), pick(2)))
    b.sign(msgs, tapes, fixed_len=64); torch.cuda.synchronize()
    th, thv, (s3, l3, _, st3), v3 = _timed_pair(torch, lambda: b.sign(msgs, tapes, fixed_len=64), lambda o: b.verify(msgs, o[0], o[1], fixed_len=64), 5)
    assert int(v3[st3 == 1].sum()) == 0 and torch.equal(st3, status_t)
    out['huffman_coded'] = {'workload': 'the same batch (tapes resident) with the signatures entropy-coded as the reference codes them (static Huffman tables for t and z)',
                            'sign_ops_per_s': B / _med(th), 'verify_ops_per_s': B / _med(thv), 'mean_signature_bytes': float(l3[st3 == 1].float().mean()), 'raw_signature_bytes': 986}
    # ... and under the arithmetic coder (SC_FLAG_0_ENTROPY_BAC; the coder's two distributions are long double arithmetic of the reference's host: from the fixture)
    A = load_golden('bliss_bac_ref.txt')
    dist = lambda k: np.ascontiguousarray(np.ascontiguousarray(A[k], dtype=np.int32).view(np.uint64))
    b.set_entropy_bac(dist('blissb1_dist_t'), dist('blissb1_dist_z'))
    b.sign(msgs, tapes, fixed_len=64); torch.cuda.synchronize()
    ta, tav, (s4, l4, _, st4), v4 = _timed_pair(torch, lambda: b.sign(msgs, tapes, fixed_len=64), lambda o: b.verify(msgs, o[0], o[1], fixed_len=64), 3)
    assert int(v4[st4 == 1].sum()) == 0 and torch.equal(st4, status_t)
    out['arithmetic_coded'] = {'workload': 'the same batch (tapes resident) with t and z through the reference\'s binary arithmetic coder (a serial range coder: one lane per signature)',
                               'sign_ops_per_s': B / _med(ta), 'verify_ops_per_s': B / _med(tav), 'mean_signature_bytes': float(l4[st4 == 1].float().mean()), 'raw_signature_bytes': 986}
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
    out['kyber_1_cpa'] = {'workload': 'Kyber set 1 (k=3, q=7681) CPA encryption / decryption, batch 65536, one key', 'enc_ops_per_s': B / _med(te),
                          'dec_ops_per_s': B / _med(td)}
    # KEM level: safecrypto_encapsulation / safecrypto_decapsulation entirely on the device (hashes, DRBG, CPA, packing)
    ky.set_kem_pubkey(G['kyber1_t'], G['kyber1_rho'])
    ct, keys = ky.encaps(m); torch.cuda.synchronize()
    te, td = [], []
    for _ in range(5):
        t0 = time.perf_counter(); ct, keys = ky.encaps(m); torch.cuda.synchronize(); te.append(time.perf_counter() - t0)
        t0 = time.perf_counter(); k2 = ky.decaps(ct); torch.cuda.synchronize(); td.append(time.perf_counter() - t0)
    assert torch.equal(k2, keys)
    kem = {'workload': 'Kyber set 1 KEM encapsulation / decapsulation (reference wire format), batch 65536, one key, 32-byte m per operation resident in HBM',
           'encaps_ops_per_s': B / _med(te), 'decaps_ops_per_s': B / _med(td)}
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
                            'encaps_ops_per_s': B / _med(te), 'decaps_ops_per_s': B / _med(td), 'key_bytes_per_value': 2 * 3 * 256 * 4 + 32}
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
    for _ in range(5):
        t0 = time.perf_counter(); sigs, siglens, trials, status = ts.sign(msgs, tapes, fixed_len=64); torch.cuda.synchronize(); tsn.append(time.perf_counter() - t0)
        t0 = time.perf_counter(); v = ts.verify(msgs, sigs, siglens, fixed_len=64); torch.cuda.synchronize(); tvf.append(time.perf_counter() - t0)
    ok = status == 1
    assert int(v[ok].sum()) == 0 and int(ok.sum()) > B - 20
    seeds = torch.randint(0, 256, (B, 32), dtype=torch.uint8, device='cuda', generator=g)
    ts.sign_seeded(msgs, seeds, fixed_len=64); torch.cuda.synchronize()
    tse, _, (sg2, sl2, tr2, st2), tse_all = _timed(torch, lambda: ts.sign_seeded(msgs, seeds, fixed_len=64))
    assert int((st2 == 1).sum()) == B and int(ts.verify(msgs, sg2, sl2, fixed_len=64).sum()) == 0
    rt = {'workload': 'ring-TESLA set 0 (n=512, q=8399873) sign / verify, batch 16384, 64-byte messages; sign_ops_per_s = end to end (seed per message, the masks come straight out of '
                      'the device CTR-DRBG inside the call), tape_resident_sign_ops_per_s = per-message CSPRNG word tapes resident in HBM',
          'sign_ops_per_s': B / tse, 'sign': _spread(tse_all), 'tape_resident_sign_ops_per_s': B / _med(tsn), 'verify_ops_per_s': B / _med(tvf),
          'mean_attempts_per_signature': float(trials[ok].float().mean())}
    ref = _cpu_ref('bench_tesla', 0, 3 if cpu_on else 0, {'sign_ops_per_s': 'sign_per_sec', 'verify_ops_per_s': 'verify_per_sec'},
                   '3 s of safecrypto_sign + 0.75 s of safecrypto_verify, set 0, 64-byte messages, one key, 1 thread')
    if ref:
        rt['cpu_reference'] = ref
    rt['roofline'] = valu_roofline('tesla_sign', rt['tape_resident_sign_ops_per_s'], 64 + 1600 + 2048 * rt['mean_attempts_per_signature'],
                                   'tape-resident signing; algorithmic bytes: the message, the signature, 2 KB of tape per attempt the reference\'s loop makes', top=3)
    rt['roofline_verify'] = valu_roofline('tesla_verify', rt['verify_ops_per_s'], 64 + 1600 + 4, 'message + signature in, verdict out', top=3)
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
        tape_gbs = B * r.enc_words * 4 / _med(te) / 1e9
        alg_bytes = r.enc_words * 4 + r.msg_bytes + r.ct_bytes        # per encryption: the tape and the block in, the ciphertext out
        alg_gbs = B * alg_bytes / _med(te) / 1e9
        pm = _pmc('rlwe_encrypt') if (set_ == 0 and cpu_on) else None
        re = {'workload': f'RLWE-Enc set {set_} (n={r.n}) encrypt / decrypt, batch {B}, {r.msg_bytes}-byte blocks, one key pair, per-message CSPRNG word tapes resident in HBM',
              'encrypt_ops_per_s': B / _med(te), 'decrypt_ops_per_s': B / _med(td), 'tape_read_GB_per_s': tape_gbs, 'decryption_failures': wrong,
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
    for _ in range(5):
        t0 = time.perf_counter(); ct, keys, used, st = k.encaps(tapes); torch.cuda.synchronize(); te.append(time.perf_counter() - t0)
        t0 = time.perf_counter(); k2 = k.decaps(ct); torch.cuda.synchronize(); td.append(time.perf_counter() - t0)
    assert int((st == 1).sum()) == B and int((k2 == keys).all(dim=1).sum()) >= B - B // 1000
    ek = {'workload': f'ENS-KEM set 0 (n=512, q=12289) encapsulation / decapsulation, batch {B}, one key pair, per-operation CSPRNG word tapes resident in HBM',
          'encaps_ops_per_s': B / _med(te), 'decaps_ops_per_s': B / _med(td)}
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
    for _ in range(5):
        t0 = time.perf_counter(); sk, st = f.dlp_extract(ubits, ids, xt, fixed_len=len(ident)); torch.cuda.synchronize(); tx.append(time.perf_counter() - t0)
        t0 = time.perf_counter(); ct, es = d.encrypt(ids, msgs, xt, fixed_len=len(ident)); torch.cuda.synchronize(); te.append(time.perf_counter() - t0)
        t0 = time.perf_counter(); mm = d.decrypt(ct); torch.cuda.synchronize(); td.append(time.perf_counter() - t0)
    assert int((st == 1).sum()) == B and int((mm != msgs).any(dim=1).sum()) < B // 10000
    di = {'workload': f'DLP-IBE set 0 (n=512, q=4206593) extract (f64 tree sampler) / encrypt / decrypt, batch {B}, one master key, one identity, tapes resident in HBM',
          'extract_ops_per_s': B / _med(tx), 'encrypt_ops_per_s': B / _med(te), 'decrypt_ops_per_s': B / _med(td)}
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
    for _ in range(5):
        t0 = time.perf_counter(); sigs, st = sg.sign(msgs, xt, fixed_len=64); torch.cuda.synchronize(); ts_.append(time.perf_counter() - t0)
        t0 = time.perf_counter(); vd = vf.ens_dlp_verify(msgs, sigs, sl, fixed_len=64); torch.cuda.synchronize(); tv.append(time.perf_counter() - t0)
    assert int((st == 1).sum()) == B and int((vd == 0).sum()) == B          # every signature made verifies
    es = {'workload': f'ENS signature set 0 (n=512, q=12289) sign (Klein sampler, 2n = 1024 sequential basis rows) / verify, batch {B}, one key pair, 64-byte messages, tapes resident in HBM',
          'sign_ops_per_s': B / _med(ts_), 'verify_ops_per_s': B / _med(tv)}
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
    for _ in range(5):
        t0 = time.perf_counter(); sigs, st = sg.sign(msgs, xt, fixed_len=64); torch.cuda.synchronize(); ts_.append(time.perf_counter() - t0)
    assert int((st == 1).sum()) == B
    out['dlp_sig_0'] = {'workload': f'DLP signature set 0 (n=256) sign (Klein sampler, 512 sequential basis rows), batch {B}, one key pair, 64-byte messages, tapes resident in HBM',
                        'sign_ops_per_s': B / _med(ts_)}
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


def batch_share(total, world, rank, shard):
    """units of a BASELINE batch this rank works on: all of them (weak scaling, the default: every rank runs the stated batch) or, with --shard,
    its contiguous slice of them (strong scaling: sc_hip_shard_range, the cut the C driver csrc/multi.cpp makes; sizes differ by at most one)"""
    if not shard:
        return total
    from libsafecrypto_amd import shard_range
    return shard_range(total, world, rank)[1]


def north_star_summary(extra, cfg):
    """BASELINE configs[2..4] in a few hundred bytes each (the full blocks go to bench_extra.json)"""
    def rl(b):
        if not b:
            return None
        r = {k: b.get(k) for k in ('bound', 'frac', 'frac_4cycle_ceiling', 'traffic', 'algorithmic_bytes_per_op', 'valu_wave_instructions_per_op')}
        r['counters'] = 'live' if str(b.get('source', '')).startswith('live') else 'committed'
        return _round(r)

    def cpu(b, key='sign_ops_per_s'):
        return (b.get('cpu_reference') or {}).get(key)
    out = {}
    d = extra.get('dilithium3')
    if d and 'sign_ops_per_s' in d:
        out[f'dilithium3_b{d["batch"]}'] = _round({
            'sign_ops_per_s': d['sign_ops_per_s'], 'verify_ops_per_s': d['verify_ops_per_s'], 'sign_ms': d['sign'], 'verify_ms': d['verify'],
            'csprng': 'none (deterministic signing)', 'cpu_thread_sign_ops_per_s': cpu(d), 'cpu_thread_verify_ops_per_s': cpu(d, 'verify_ops_per_s'),
            'sign_speedup_vs_cpu_thread': d.get('sign_speedup_vs_cpu_thread'), 'distinct_keys_sign_ops_per_s': (d.get('distinct_keys') or {}).get('sign_ops_per_s'),
            'distinct_keys_verify_ops_per_s': (d.get('distinct_keys') or {}).get('verify_ops_per_s'),
            'roofline_sign': rl(d.get('roofline')), 'roofline_verify': rl(d.get('roofline_verify'))})
    b = extra.get('bliss_b_1')
    if b and 'sign_ops_per_s' in b:
        out[f'bliss_b_1_b{b["batch"]}'] = _round({
            'sign_ops_per_s': b['sign_ops_per_s'], 'csprng': 'inside the call (seed -> device CTR-DRBG -> sign)', 'tape_resident_sign_ops_per_s': b['tape_resident_sign_ops_per_s'],
            'verify_ops_per_s': b['verify_ops_per_s'], 'sign_ms': b['sign'], 'drbg_tape_GB_per_s': b['drbg_alone']['tape_GB_per_s'],
            'cpu_thread_sign_ops_per_s': cpu(b), 'roofline_sign_tape_resident': rl(b.get('roofline'))})
    f = extra.get('falcon_1024')
    key = f'batch_{cfg["falcon_batch"]}'
    if f and key in f:
        fb = f[key]
        out[f'falcon_1024_b{cfg["falcon_batch"]}'] = _round({
            'sign_ops_per_s': fb['sign_ops_per_s'], 'csprng': 'inside the call (seed -> device CTR-DRBG -> sign)', 'tape_resident_sign_ops_per_s': fb['tape_resident_sign_ops_per_s'],
            'verify_ops_per_s': fb['verify_ops_per_s'], 'sign_ms': fb['sign'], 'drbg_tape_GB_per_s': fb['drbg_alone']['tape_GB_per_s'],
            'cpu_thread_sign_ops_per_s': cpu(f), 'roofline_sign_tape_resident': rl(f.get('roofline'))})
    t = extra.get('ring_tesla_0')
    if t and 'sign_ops_per_s' in t:
        out['ring_tesla_0_b16384'] = _round({'sign_ops_per_s': t['sign_ops_per_s'], 'csprng': 'inside the call', 'tape_resident_sign_ops_per_s': t['tape_resident_sign_ops_per_s'],
                                             'verify_ops_per_s': t['verify_ops_per_s'], 'cpu_thread_sign_ops_per_s': cpu(t), 'roofline_sign_tape_resident': rl(t.get('roofline')),
                                             'roofline_verify': rl(t.get('roofline_verify'))})
    return out


def _round(x, sig=6):
    """floats to 6 significant digits (the line is parsed by a driver that keeps a bounded tail of stdout: every byte counts)"""
    if isinstance(x, float):
        return float(f'{x:.{sig}g}')
    if isinstance(x, dict):
        return {k: _round(v, sig) for k, v in x.items()}
    if isinstance(x, (list, tuple)):
        return [_round(v, sig) for v in x]
    return x


def frac_violations(node, path='line'):
    """roofline fractions outside [0, 1] anywhere in the tree (a 1.23 went out unnoticed): reported in the line and by the exit code - not
    asserted, a counter glitch in one leg must not discard the whole run's line"""
    bad = []
    if isinstance(node, dict):
        for k, v in node.items():
            if k.startswith('frac') and isinstance(v, (int, float)) and k != 'frac_4cycle_ceiling' and not 0.0 <= v <= 1.0:
                bad.append(f'{path}.{k}={v:.4g}')
            bad += frac_violations(v, path + '.' + k)
    return bad


REQUIRED_KEYS = ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better', 'scaling', 'vs_baseline', 'dtype', 'data', 'config', 'roofline', 'cpu_baseline')


def contract_line(res, limit=None):
    """The contract line as a string below `limit` bytes WHATEVER the legs produced: optional blocks are dropped largest first, then the north-star blocks are cut
    to their figures, then everything that is not a required key goes - a run never ends without its line because an optional block grew (the full tree is in
    bench_extra.json either way).  `res` is not modified."""
    import copy
    limit = LINE_LIMIT if limit is None else limit
    r = copy.deepcopy(res)
    steps = [
        lambda: [r.pop(k, None) for k in ('multi_gpu', 'frac_violations')],
        lambda: [v.pop(k, None) for v in r.get('north_star', {}).values() if isinstance(v, dict) for k in ('roofline_verify', 'sign_ms', 'verify_ms', 'roofline_sign_tape_resident')],
        lambda: [r.get('roofline', {}).pop('note', None), r.get('config', {}).pop('timing', None)],
        lambda: [v.pop(k, None) for v in r.get('north_star', {}).values() if isinstance(v, dict) for k in list(v) if isinstance(v.get(k), (dict, list))],
        lambda: [r.pop(k, None) for k in list(r) if k not in REQUIRED_KEYS and k != 'north_star'],
        lambda: r.pop('north_star', None),
        lambda: [r.__setitem__(k, {kk: vv for kk, vv in r[k].items() if not isinstance(vv, (dict, list, str)) or kk in ('workload', 'bound', 'unit', 'kind')}) for k in ('config', 'roofline', 'cpu_baseline') if isinstance(r.get(k), dict)],
    ]
    line = json.dumps(r)
    dropped = 0
    for step in steps:
        if len(line) < limit:
            break
        step()
        dropped += 1
        if dropped and 'line_trimmed' not in r:
            r['line_trimmed'] = True                                  # (says so: the reader goes to bench_extra.json for the rest)
        line = json.dumps(r)
    assert json.loads(line)['value'] == res['value'] and all(k in json.loads(line) for k in REQUIRED_KEYS if k in res)
    return line


LINE_LIMIT = 6000                # bytes: the driver parsed a 15.5 KB line in round 2 and failed on 27.7 KB in round 3; the contract line stays far below either


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=200)
    ap.add_argument('--warmup', type=int, default=20)
    ap.add_argument('--regions', type=int, default=7, help='timed regions of --steps steps each; value is their MEDIAN (min / max beside it)')
    ap.add_argument('--shard', action='store_true', help='strong scaling: the BASELINE batches are cut across the ranks (shard_range) instead of repeated on each')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-extras', action='store_true', help='only the headline (profiling runs)')
    ap.add_argument('--cpu-seconds', type=float, default=10.0)
    ap.add_argument('--no-spin', action='store_true', help='leave the runtime\'s host-thread scheduling at its default (the bench asks it to spin in synchronisations)')
    ap.add_argument('--no-live-pmc', action='store_true', help='scheme rooflines from the committed counter pass instead of rocprofv3 child processes of this run')
    ap.add_argument('--test-one-device', action='store_true', help='testing only: every rank on device 0 over gloo (exercises the N > 1 path on a one-GPU box; figures mean nothing)')
    ap.add_argument('--extra-file', default=os.path.join(ROOT, 'bench_extra.json'), help='where the full tree of legs goes ')
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
    if args.test_one_device:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    spin = not args.no_spin
    if spin:
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
            spin = False
    if world > 1:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        if args.test_one_device:
            dist.init_process_group('gloo', rank=rank, world_size=world)
        else:
            dist.init_process_group('nccl', rank=rank, world_size=world, device_id=torch.device('cuda', local_rank))

    from libsafecrypto_amd import Engine, shard_range
    eng = Engine(local_rank)
    plan = eng.plan(Q, N)

    # weak scaling (default): every rank runs the stated batch; --shard: the stated batch is cut into contiguous slices, one per rank
    def my_share(total):
        return batch_share(total, world, rank, args.shard)
    batch = my_share(BATCH)
    total_batch = BATCH if args.shard else BATCH * world

    gen = torch.Generator(device='cuda')
    gen.manual_seed(1 + rank)
    a = torch.randint(0, Q, (batch, N), dtype=torch.int32, device='cuda', generator=gen)
    bhat = torch.randint(0, Q, (batch, N), dtype=torch.int32, device='cuda', generator=gen)
    out = torch.empty_like(a)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # the K timed steps are K kernel launches issued by ONE C call (sc_hip_polymul_ntt_steps): plain launches back to back for K < 64, a captured HIP graph
    # replayed for longer loops (a replay has a fixed start-up that only pays off over many steps: 762 us against 664 us at K = 20, equal at K = 200).
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        for _ in range(args.warmup):
            plan.polymul_ntt(a, bhat, out)
        plan.polymul_ntt_steps(a, bhat, out, args.steps)            # capture + one untimed replay

    # everything the timed region needs is made before it: the two events exist (torch creates the HIP event lazily, at its first record: ~30 us) and the
    # step loop is ONE ctypes call with ready-made arguments (tools/exp_region.py: 709 us per 20-step region the naive way, 644 us like this)
    import ctypes
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ev0.record(side); ev1.record(side)
    c_steps = plan.L.sc_hip_polymul_ntt_steps
    c_args = (plan.h, batch, ctypes.c_void_p(a.data_ptr()), ctypes.c_void_p(bhat.data_ptr()), ctypes.c_void_p(out.data_ptr()))
    c_stream = ctypes.c_void_p(side.cuda_stream)

    def region(steps):
        """one timed region of exactly `steps` steps: barrier + synchronize on both sides, MAX over ranks; HIP events on the stream the kernels run on"""
        barrier()
        t0 = time.perf_counter()
        ev0.record(side)
        rc_ = c_steps(*c_args, steps, c_stream)
        ev1.record(side)
        torch.cuda.synchronize()
        barrier()
        el = time.perf_counter() - t0
        assert rc_ == 0, 'sc_hip_polymul_ntt_steps failed'
        if world > 1:
            t = torch.tensor([el], dtype=torch.float64, device='cuda')
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            el = float(t.item())
        return el, ev0.elapsed_time(ev1) / steps
    regs = [region(args.steps) for _ in range(max(1, args.regions))]
    order = sorted(range(len(regs)), key=lambda i: regs[i][0])
    elapsed, kern_ms = regs[order[len(order) // 2]]                # the MEDIAN region (its own HIP-event kernel time with it)
    long_steps = 200
    long_reg = None
    if args.steps != long_steps:
        with torch.cuda.stream(side):
            plan.polymul_ntt_steps(a, bhat, out, long_steps)        # capture for the other step count, untimed
        lr = sorted(region(long_steps)[0] for _ in range(3))
        long_reg = lr[1]
        with torch.cuda.stream(side):
            plan.polymul_ntt_steps(a, bhat, out, args.steps)

    # size-independent self check of the timed output (never timed): NTT(c) == NTT(a) o bhat
    assert torch.equal(plan.fwd(out), plan.pointwise(plan.fwd(a), bhat)), 'bench output failed the convolution check'

    cpu_on = (not args.no_cpu_baseline) and world == 1 and rank == 0
    cpu_sec = args.cpu_seconds if cpu_on else 0.0
    cfg = {'dil_batch': my_share(16384), 'bliss_batch': my_share(16384), 'falcon_batch': my_share(8192)}

    # ---- the scheme-level legs, on every rank (independent batches), gathered afterwards
    extra = {}
    if not args.no_extras:
        single = world == 1
        fsets = (('falcon2', 2, (cfg['falcon_batch'], 131072)), ('falcon0', 0, (131072,))) if single else (('falcon2', 2, (cfg['falcon_batch'],)),)
        for name, fn in (('dilithium3', lambda: {'dilithium3': dilithium_extra(eng, torch, cpu_sec, batch=cfg['dil_batch'], distinct=single, sweep=single)}),
                         ('bliss_b_1', lambda: bliss_extra(eng, torch, cpu_sec, batch=cfg['bliss_batch'], sweep=single, variants=single)),
                         ('falcon', lambda: falcon_extra(eng, torch, cpu_sec, sets=fsets, cfg_batch=cfg['falcon_batch'], sweep=single))):
            try:
                barrier()
                extra.update(fn())
            except Exception as e:        # an extra must never break the contract line
                extra[name] = {'error': str(e)}
        if world > 1:
            # per-GPU and aggregate ops/s of the BASELINE legs: every rank signed / verified its own batch (the stated size, or its slice of it with --shard)
            keys = [('dilithium3', 'sign_ops_per_s'), ('dilithium3', 'verify_ops_per_s'), ('bliss_b_1', 'sign_ops_per_s'), ('bliss_b_1', 'verify_ops_per_s')]
            mine = [float(extra.get(k, {}).get(f, 0.0) or 0.0) for k, f in keys]
            mine.append(float(extra.get('falcon_1024', {}).get(f'batch_{cfg["falcon_batch"]}', {}).get('sign_ops_per_s', 0.0) or 0.0))
            tt = torch.tensor(mine, dtype=torch.float64, device='cuda')
            allr = [torch.zeros_like(tt) for _ in range(world)]
            dist.all_gather(allr, tt)
            if rank == 0:
                per = torch.stack(allr).cpu().tolist()
                names = [f'{k}.{f}' for k, f in keys] + ['falcon_1024.sign_ops_per_s']
                extra['multi_gpu'] = {'scaling': ('strong (--shard: the stated batch cut into one contiguous slice per rank' if args.shard else 'weak (each rank runs the stated batch on its own GPU') +
                                                 '; no data-path collective; the legs of the ranks run concurrently, so the aggregate is the sum of the per-GPU rates)',
                                      'batch_per_gpu': cfg, 'per_gpu': {nm: [row[i] for row in per] for i, nm in enumerate(names)},
                                      'aggregate': {nm: sum(row[i] for row in per) for i, nm in enumerate(names)}}

    rc = 0
    if rank == 0:
        kern_ms_plain = plan.time_op(2, a, bhat, out, 50)            # the same kernel as 50 plain launches between two events (C loop)
        achieved = batch * BYTES_PER_TRIPLE / (kern_ms * 1e-3) / 1e9
        # per-launch HBM bytes of this kernel (2 x FETCH_SIZE + WRITE_SIZE): counters collected over a child process of THIS run (tools/pmc_leg.py: the same
        # fused product at the same batch under rocprofv3 --pmc, one pass per counter as the guide prescribes), else the committed passes of tools/profile_r03.sh
        traffic = traffic_past = None
        traffic_source = None
        pm = _pmc('polymul') if (cpu_on and PMC_LIVE and batch == BATCH) else None
        if pm and str(pm.get('source', '')).startswith('live'):
            traffic = pm['hbm_bytes_per_op_2xfetch_plus_write'] * pm['batch']
            traffic_source = 'live'
        for name in ('r04_pmc_traffic.json', 'r03_pmc_traffic.json', 'pmc_traffic.json'):
            tp = os.path.join(ROOT, 'profiles', name)
            if not os.path.exists(tp):
                continue
            try:
                tj = json.load(open(tp))
                if traffic is None and batch == BATCH:
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
            ms2 = statistics.median(plan.time_op(2, a2, b2, o2, 10) for _ in range(3))
            g2 = PAST_L3_BATCH * BYTES_PER_TRIPLE / (ms2 * 1e-3) / 1e9
            past = {'achieved_past_l3': g2, 'frac_past_l3': g2 / HBM_PEAK_GBS, 'past_l3_batch': PAST_L3_BATCH, 'past_l3_kernel_ms': ms2, 'traffic_past_l3': traffic_past}
            del a2, b2, o2
        except Exception as e:
            past = {'achieved_past_l3': None, 'past_l3_error': str(e)}
        fwd_ms = {}
        for (q2, n2, b2) in [(7681, 256, 65536), (8380417, 256, 65536), (12289, 512, 32768), (12289, 1024, 16384), (7681, 256, 1 << 20), (12289, 1024, 1 << 18)]:
            p2 = plan if (q2, n2) == (Q, N) else eng.plan(q2, n2)
            x = torch.randint(0, q2, (b2, n2), dtype=torch.int32, device='cuda', generator=gen)
            y = torch.empty_like(x)
            p2.fwd(x, y)
            ms = statistics.median(p2.time_op(0, x, None, y, 50 if b2 <= 65536 else 10) for _ in range(3))
            gbs = b2 * 8 * n2 / (ms * 1e-3) / 1e9
            fwd_ms[f'q{q2}_n{n2}_b{b2}'] = {'NTT/s': b2 / (ms * 1e-3), 'GB/s': gbs, 'frac': gbs / HBM_PEAK_GBS, 'working_set_MiB': b2 * 8 * n2 / 2**20}
            del x, y
        value = 2.0 * total_batch * args.steps / elapsed
        all_el = sorted(r[0] for r in regs)
        extra = {'fwd_ntt': fwd_ms, **extra}
        if world == 1 and not args.no_extras:
            try:
                extra['cdt_sampler'] = cdt_extra(eng, torch)
            except Exception as e:
                extra['cdt_sampler'] = {'error': str(e)}
            try:
                extra.update(kyber_tesla_more_extra(eng, torch, cpu_on))
            except Exception as e:
                extra['kyber_tesla_more'] = {'error': str(e)}
        cpu_b = cpu_baseline(args.cpu_seconds) if cpu_on else None
        res = {
            'metric': 'NTT/s (n=256, q=7681; fused fwd NTT + pointwise + inv NTT, 2 NTTs per polynomial)',
            'value': value, 'unit': 'NTT/s', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': elapsed / args.steps * 1e3, 'higher_is_better': True, 'scaling': 'strong' if args.shard else 'weak',
            'vs_baseline': None, 'dtype': 'u32', 'data': 'synthetic',
            'config': {'workload': 'BASELINE configs[1]: Kyber set 1 (n=256, q=7681 - the reference has no q=3329) batched NTT + pointwise-mul + INTT, batch=65536'
                                   + (' cut across the ranks' if args.shard else ' per GPU'), 'n': N, 'q': Q,
                       'batch_per_gpu': batch, 'parallelism': f'shard{world} (independent polynomials, no collective)',
                       'timing': f'value = MEDIAN of {len(regs)} timed regions of {args.steps} steps each ({"one HIP-graph replay" if args.steps >= 64 else "plain launches from one C call"}; barrier + synchronize both sides); '
                                 + ('host thread spins in synchronisations (hipDeviceScheduleSpin)' if spin else 'default host scheduling'),
                       'ms_per_step_min_max': [all_el[0] / args.steps * 1e3, all_el[-1] / args.steps * 1e3],
                       'ms_per_step_200_step_region': (long_reg / long_steps * 1e3) if long_reg else elapsed / args.steps * 1e3},
            'roofline': {'bound': 'hbm', 'achieved': achieved, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                         'frac': achieved / HBM_PEAK_GBS, 'traffic': traffic, 'traffic_source': traffic_source,
                         'kernel': 'ntt_kernel<8,*,*,OP_MULNTT>', 'kernel_ms': kern_ms, 'kernel_ms_plain_launches': kern_ms_plain,
                         'algorithmic_bytes_per_launch': batch * BYTES_PER_TRIPLE,
                         'note': 'at the BASELINE batch the 192 MiB of operands replay inside the 256 MiB Infinity Cache (FETCH_SIZE counts its hits); *_past_l3 = same kernel over 3 GiB', **past},
        }
        if cpu_b:
            res['cpu_baseline'] = {k: cpu_b[k] for k in ('value', 'unit', 'cores', 'kind', 'sample', 'cpu_model') if k in cpu_b}
            for k in ('avx2_build', 'all_cores'):
                if isinstance(cpu_b.get(k), dict) and 'value' in cpu_b[k]:
                    res['cpu_baseline'][k] = {'value': cpu_b[k]['value'], 'cores': cpu_b[k]['cores']}
            extra['cpu_baseline_full'] = cpu_b
        res['north_star'] = north_star_summary(extra, cfg)
        if 'multi_gpu' in extra:
            res['multi_gpu'] = _round({'scaling': 'strong' if args.shard else 'weak', 'batch_per_gpu': cfg, 'aggregate': extra['multi_gpu']['aggregate']})
        bad = frac_violations({'roofline': res['roofline'], 'extra': extra})
        if bad:
            res['frac_violations'] = bad[:8]
            rc = 3
        res = _round(res)
        # ---- output: the full tree to a file and to an EARLIER, prefixed stdout line; the contract line - compact, self-validated - is the LAST line
        full = json.dumps({'contract': res, 'extra': extra})
        try:
            with open(args.extra_file, 'w') as fh:
                fh.write(full + '\n')
            res['extra_file'] = os.path.relpath(args.extra_file, ROOT)
        except OSError as e:
            res['extra_file'] = f'not written: {e}'
        print(f'# full tree of every leg: {res["extra_file"]} ({len(full)} bytes)', flush=True)
        line = contract_line(res)
        print(line, flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    sys.exit(rc)


if __name__ == '__main__':
    main()
