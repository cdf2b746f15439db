#!/usr/bin/env python3
"""Are two device-code listings the same kernels?  tools/isa_cmp.py before.s after.s

For refactors that only MOVE kernel text (e.g. out of a .hip file into a header so that the CPU emulation under tests/emu can compile it too): the listings
(hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC --cuda-device-only -S x.hip -o x.s) are cut into functions and compared name by name, so a changed
order of definitions or the compilation unit's id hash is not a difference, and a changed instruction is."""
import re
import sys


def functions(path):
    out, name, body = {}, None, []
    for line in open(path):
        m = re.match(r'^([A-Za-z_$][\w$.]*):', line)
        if m and not line.startswith('.L') and name is None:
            name, body = m.group(1), []
            continue
        if name is not None:
            if line.startswith('.Lfunc_end'):
                out[name] = ''.join(body)
                name = None
            else:
                line = re.sub(r'\s*;.*$', '', line.rstrip('\n')) + '\n'          # (comments name blocks by the function's number too)
                body.append(re.sub(r'\.L(BB|tmp|func_begin|JTI)\d+_', r'.L\1_', re.sub(r'__hip_cuid_[0-9a-f]+', '__hip_cuid_X', line)))    # (labels carry the function's number in the module)
    return out


def main():
    a, b = functions(sys.argv[1]), functions(sys.argv[2])
    names_a = {re.sub(r'_GLOBAL__N__[0-9a-f_]+', '_GLOBAL__N_', n): n for n in a}
    names_b = {re.sub(r'_GLOBAL__N__[0-9a-f_]+', '_GLOBAL__N_', n): n for n in b}
    only_a, only_b = sorted(set(names_a) - set(names_b)), sorted(set(names_b) - set(names_a))
    norm = lambda s: re.sub(r'_GLOBAL__N__[0-9a-f_]+', '_GLOBAL__N_', s)
    differ = [n for n in sorted(set(names_a) & set(names_b)) if norm(a[names_a[n]]) != norm(b[names_b[n]])]
    print(f'{len(a)} / {len(b)} functions; only in first: {len(only_a)}; only in second: {len(only_b)}; bodies that differ: {len(differ)}')
    for n in (only_a + only_b + differ)[:20]:
        print('  ', n[:160])
    return 1 if (only_a or only_b or differ) else 0


if __name__ == '__main__':
    sys.exit(main())
