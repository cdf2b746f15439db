"""instruction mix per kernel from hipcc -S output (development aid)"""
import re, sys
from collections import Counter
s = open(sys.argv[1] if len(sys.argv) > 1 else '/tmp/sc_hip_gfx950.s').read()
pat = sys.argv[2] if len(sys.argv) > 2 else 'ntt_kernelILi8'
for m in re.finditer(r'^(_Z\w+):.*?\n(.*?)\.Lfunc_end', s, flags=re.S | re.M):
    name, body = m.group(1), m.group(2)
    if pat not in name: continue
    ins = [l.strip().split()[0] for l in body.split('\n') if l.startswith('\t') and not l.strip().startswith(('.', ';'))]
    c = Counter(ins)
    g = lambda f: sum(v for k, v in c.items() if f(k))
    print(name[:48], 'total', len(ins), 'valu', g(lambda k: k.startswith('v_')),
          'mul32', g(lambda k: k.startswith(('v_mul_lo', 'v_mul_hi_u32_e', 'v_mul_hi_i32', 'v_mad_u64')) and 'u24' not in k),
          'u24', g(lambda k: 'u24' in k), 'ds', g(lambda k: k.startswith('ds_')), 'gld', g(lambda k: k.startswith('global_load')),
          'gst', g(lambda k: k.startswith('global_store')), 'sld', g(lambda k: k.startswith('s_load')), 'wait', c['s_waitcnt'],
          'scratch', g(lambda k: k.startswith('scratch')))
