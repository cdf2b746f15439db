"""Folds the rocprofv3 output of tools/prof_schemes_r2.sh into one JSON: per leg, per kernel (torch's own kernels excluded): launches, average ns,
SQ_INSTS_VALU / SQ_WAVES / FETCH_SIZE (KB) / WRITE_SIZE (KB) per call of the leg, and the leg's totals per operation."""
import csv, glob, json, os, sys, collections

out_dir = sys.argv[1]
res = {}
for tr in sorted(glob.glob(os.path.join(out_dir, 'trace_*'))):
    leg = os.path.basename(tr)[len('trace_'):]
    try:
        meta = json.loads([l for l in open(os.path.join(out_dir, f'out_{leg}.txt')) if l.startswith('{')][-1])
    except Exception:
        continue
    calls = meta['calls_of_this_leg']
    kern = collections.OrderedDict()
    for f in glob.glob(os.path.join(tr, '*', '*kernel_stats.csv')):
        for r in csv.DictReader(open(f)):
            name = r['Name']
            if 'at::native' in name or 'elementwise' in name.lower():
                continue
            kern[name] = {'launches': int(r['Calls']), 'avg_ns': float(r['AverageNs']), 'total_ns': float(r['TotalDurationNs'])}
    for c in ('SQ_INSTS_VALU', 'SQ_WAVES', 'FETCH_SIZE', 'WRITE_SIZE'):
        acc = collections.defaultdict(float)
        for f in glob.glob(os.path.join(out_dir, f'pmc_{leg}_{c}', '*', '*counter_collection.csv')):
            for r in csv.DictReader(open(f)):
                acc[r['Kernel_Name']] += float(r['Counter_Value'])
        for name, v in acc.items():
            if name in kern:
                kern[name][c] = v
    short = lambda s: s.replace('(anonymous namespace)::', '').replace('void ', '').split('(')[0][:70]
    tot = collections.defaultdict(float)
    per_kernel = {}
    for name, k in kern.items():
        for c in ('SQ_INSTS_VALU', 'SQ_WAVES', 'FETCH_SIZE', 'WRITE_SIZE', 'total_ns'):
            tot[c] += k.get(c, 0.0)
        per_kernel[short(name)] = {x: (k[x] / calls if x != 'avg_ns' else k[x]) for x in k}
    B = meta['batch']
    res[leg] = {'batch': B, 'calls_profiled': calls, 'ms_per_call_unprofiled_min': meta['ms_per_call_min'],
                'kernel_ns_per_call': tot['total_ns'] / calls,
                'valu_wave_instructions_per_call': tot['SQ_INSTS_VALU'] / calls, 'waves_per_call': tot['SQ_WAVES'] / calls,
                'fetch_kb_per_call': tot['FETCH_SIZE'] / calls, 'write_kb_per_call': tot['WRITE_SIZE'] / calls,
                'valu_wave_instructions_per_op': tot['SQ_INSTS_VALU'] / calls / B,
                'hbm_bytes_per_op_2xfetch_plus_write': (2 * tot['FETCH_SIZE'] + tot['WRITE_SIZE']) * 1024 / calls / B,
                'kernels': per_kernel}
print(json.dumps(res, indent=1))
