"""Hardware counters of ONE scheme leg, collected by running tools/bench_scheme.py <leg> as a child process under rocprofv3 --pmc (the profiler is
never in the process that is being timed).  bench.py calls profile_leg() right after it has timed the same leg on the same code, so the instruction
and byte counts of a roofline block and its ops/s come from one run of one build (committed counts x live rate).

Passes (rocprofv3 refuses TCC FETCH_SIZE and WRITE_SIZE in one pass; MI355X_MICROARCH.md, rocprofv3 PMC slots):
  1. SQ_INSTS_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAVES GRBM_GUI_ACTIVE
  2. FETCH_SIZE      3. WRITE_SIZE
What the SQ counters read on gfx950 was calibrated on kernels of known issue rate (tools/exp_valu_cycles.hip, profiles/r03_valu_issue_cycles.txt):
SQ_BUSY_CYCLES is summed over the 32 shader engines, so a kernel's duration in shader cycles is SQ_BUSY_CYCLES / 32 and
    VALU instructions per SIMD-cycle = SQ_INSTS_VALU / (SQ_BUSY_CYCLES / 32 x 1024 SIMDs)
whose ceiling is 0.5 (2 cycles per wave64 instruction: v_add_u32, v_xor_b32, v_bitop3_b32, v_fma_f32 ...; 0.40-0.43 measured) or 0.25 (4 cycles:
multiplies, v_alignbit_b32, v_min_u32, 64-bit and packed operations; 0.235 measured).  (SQ_ACTIVE_INST_VALU equals SQ_INSTS_VALU on this part: no use.)
HBM bytes = 2 x FETCH_SIZE + WRITE_SIZE (KB), the gfx950 correction of the guide's HBM section.

CLI: python3 tools/pmc_leg.py <leg> [calls]   prints the JSON."""
import collections
import csv
import glob
import json
import os
import shutil
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SQ_GROUP = ['SQ_INSTS_VALU', 'SQ_BUSY_CYCLES', 'SQ_WAVE_CYCLES', 'SQ_WAIT_ANY', 'SQ_ACTIVE_INST_ANY', 'SQ_WAVES', 'GRBM_GUI_ACTIVE']
N_SIMD = 1024
SHADER_ENGINES = 32


def _short(name):
    return name.replace('(anonymous namespace)::', '').replace('void ', '').replace('sc_mlwe::', '').split('(')[0][:60]


def _is_ours(name):
    return not ('at::native' in name or 'elementwise' in name.lower() or name.startswith('__amd_rocclr'))


def _run_pass(counters, leg, calls, sigfile, out_dir, timeout):
    env = dict(os.environ, TMPDIR='/tmp')
    cmd = ['rocprofv3', '--pmc'] + counters + ['--output-format', 'csv', '-d', out_dir, '--', sys.executable, os.path.join(ROOT, 'tools', 'bench_scheme.py'), leg, str(calls)]
    if sigfile:
        cmd.append(sigfile)
    r = subprocess.run(cmd, cwd='/tmp', env=env, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, timeout=timeout)
    meta = None
    for line in r.stdout.decode(errors='replace').splitlines():
        if line.startswith('{'):
            meta = json.loads(line)
    acc = collections.defaultdict(lambda: collections.defaultdict(float))
    launches = collections.defaultdict(int)
    for f in glob.glob(os.path.join(out_dir, '*', '*counter_collection.csv')):
        for row in csv.DictReader(open(f)):
            acc[row['Kernel_Name']][row['Counter_Name']] += float(row['Counter_Value'])
            if row['Counter_Name'] == counters[0]:
                launches[row['Kernel_Name']] += 1
    return meta, acc, launches


def profile_leg(leg, calls=2, timeout=120):
    """{'ok': True, 'batch', 'calls', per-operation totals, 'kernels': {name: {...}}} or {'ok': False, 'error': ...}"""
    if shutil.which('rocprofv3') is None:
        return {'ok': False, 'error': 'rocprofv3 not found'}
    t_start = time.perf_counter()
    work = tempfile.mkdtemp(prefix='sc_pmc_', dir='/tmp')
    try:
        sigfile = None
        if leg.endswith('_verify'):
            # the signatures come from an unprofiled run of their own: the counter passes below then contain the verifier's kernels only
            sigfile = os.path.join(work, 'sigs.npz')
            subprocess.run([sys.executable, os.path.join(ROOT, 'tools', 'bench_scheme.py'), leg, '1', sigfile], cwd='/tmp', stdout=subprocess.DEVNULL,
                           stderr=subprocess.DEVNULL, timeout=timeout, check=True)
        meta, sq, launches = _run_pass(SQ_GROUP, leg, calls, sigfile, os.path.join(work, 'sq'), timeout)
        _, fe, _ = _run_pass(['FETCH_SIZE'], leg, calls, sigfile, os.path.join(work, 'fetch'), timeout)
        _, wr, _ = _run_pass(['WRITE_SIZE'], leg, calls, sigfile, os.path.join(work, 'write'), timeout)
        if not meta or not sq:
            return {'ok': False, 'error': 'no counter output (rocprofv3 pass failed)'}
        ncalls, batch = meta['calls_of_this_leg'], meta['batch']
        kern, tot = {}, collections.defaultdict(float)
        for name, c in sq.items():
            if not _is_ours(name):
                continue
            cyc = c.get('SQ_BUSY_CYCLES', 0.0) / SHADER_ENGINES
            k = {'launches_per_call': launches[name] / ncalls, 'valu_wave_instructions_per_call': c.get('SQ_INSTS_VALU', 0.0) / ncalls,
                 'shader_cycles_per_call': cyc / ncalls, 'waves_per_call': c.get('SQ_WAVES', 0.0) / ncalls,
                 'valu_instr_per_simd_cycle': c.get('SQ_INSTS_VALU', 0.0) / (cyc * N_SIMD) if cyc else None,
                 'wave_cycles_waiting_frac': c.get('SQ_WAIT_ANY', 0.0) / c['SQ_WAVE_CYCLES'] if c.get('SQ_WAVE_CYCLES') else None,
                 'wave_cycles_issuing_frac': c.get('SQ_ACTIVE_INST_ANY', 0.0) / c['SQ_WAVE_CYCLES'] if c.get('SQ_WAVE_CYCLES') else None,
                 'fetch_kb_per_call': fe.get(name, {}).get('FETCH_SIZE', 0.0) / ncalls, 'write_kb_per_call': wr.get(name, {}).get('WRITE_SIZE', 0.0) / ncalls}
            kern[_short(name)] = k
            tot['instr'] += c.get('SQ_INSTS_VALU', 0.0); tot['cyc'] += cyc
            tot['fetch'] += fe.get(name, {}).get('FETCH_SIZE', 0.0); tot['write'] += wr.get(name, {}).get('WRITE_SIZE', 0.0)
        kern = dict(sorted(kern.items(), key=lambda kv: -kv[1]['shader_cycles_per_call']))
        return {'ok': True, 'leg': leg, 'batch': batch, 'calls_profiled': ncalls, 'seconds_spent': time.perf_counter() - t_start,
                'valu_wave_instructions_per_op': tot['instr'] / ncalls / batch,
                'shader_cycles_per_call_sum_of_kernels': tot['cyc'] / ncalls,
                'valu_instr_per_simd_cycle': tot['instr'] / (tot['cyc'] * N_SIMD) if tot['cyc'] else None,
                'hbm_bytes_per_op_2xfetch_plus_write': (2 * tot['fetch'] + tot['write']) * 1024 / ncalls / batch,
                'fetch_kb_per_call': tot['fetch'] / ncalls, 'write_kb_per_call': tot['write'] / ncalls, 'kernels': kern,
                'method': 'rocprofv3 --pmc over `python3 tools/bench_scheme.py %s %d` as a child of this run: pass 1 %s; pass 2 FETCH_SIZE; pass 3 WRITE_SIZE' % (leg, calls, ' '.join(SQ_GROUP))}
    except Exception as e:                                  # a profiler failure must never break the bench line
        return {'ok': False, 'error': repr(e)[:300]}
    finally:
        shutil.rmtree(work, ignore_errors=True)


if __name__ == '__main__':
    print(json.dumps(profile_leg(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 2), indent=1))
