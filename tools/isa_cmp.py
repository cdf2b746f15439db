#!/usr/bin/env python3
"""Are two device-code listings the same kernels?  tools/isa_cmp.py [--allow-additions] before.s after.s

For refactors that only MOVE kernel text (e.g. out of a .hip file into a header so that the CPU emulation under tests/emu can compile it too): the listings
(hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC --cuda-device-only -S x.hip -o x.s) are cut into their parts and compared name by name, so a changed
order of definitions or the compilation unit's id hash is not a difference, and a changed instruction is.  Compared:
  * function bodies (label .. .Lfunc_end),
  * kernel descriptors (.amdhsa_kernel .. .end_amdhsa_kernel: LDS size, register counts, the enabled SGPR inputs, ...),
  * data objects (.type X,@object .. .size X: __constant__ tables, .rodata, .data, zero-filled .bss objects by size),
  * each kernel's entry in the .amdgpu_metadata note (argument layout, group / private segment sizes, wavefront size, workgroup limits).
With --allow-additions, parts that exist only in the second listing are listed but do not fail the comparison (new kernels beside unchanged ones).
A listing without functions is an error (a failed compile leaves an empty or missing file), not "0 differences"."""
import re
import sys

ANON = re.compile(r'_GLOBAL__N__[0-9a-f_]+')
CUID = re.compile(r'__hip_cuid_[0-9a-f]+')


def norm(s):
    return CUID.sub('__hip_cuid_X', ANON.sub('_GLOBAL__N_', s))


def strip(line):
    line = re.sub(r'\s*;.*$', '', line.rstrip('\n'))                                    # (comments name blocks by the function's number too)
    return re.sub(r'\.L(BB|tmp|func_begin|func_end|JTI|__unnamed_)\d+(_?)', r'.L\1\2', line)  # (labels carry the function's number in the module)


def parts(path):
    """-> {'function': {name: text}, 'descriptor': {...}, 'object': {...}, 'metadata': {...}}"""
    fn, desc, obj, meta = {}, {}, {}, {}
    lines = open(path).read().split('\n')
    objects = set()
    for line in lines:
        m = re.match(r'^\s*\.type\s+([^\s,]+),@object', line)
        if m:
            objects.add(m.group(1))
    name = kind = None
    body = []
    dname, dbody = None, []                                                             # (a kernel's descriptor sits between its last instruction and .Lfunc_end)
    in_meta = False
    meta_entry, meta_name = [], None

    def close_meta():
        nonlocal meta_entry, meta_name
        if meta_entry and meta_name:
            meta[norm(meta_name)] = norm('\n'.join(meta_entry))
        meta_entry, meta_name = [], None

    for line in lines:
        if in_meta:
            if line.strip() == '.end_amdgpu_metadata':
                close_meta(); in_meta = False
                continue
            if re.match(r'^  - \.', line):                                              # a new entry of amdhsa.kernels
                close_meta()
            m = re.match(r'^\s+\.name:\s+(\S+)\s*$', line)
            if m and re.match(r'^    \.name:', line):                                   # the kernel's own name (arguments' names sit deeper)
                meta_name = m.group(1)
            if re.match(r'^\s+\.symbol:', line) or line.startswith('amdhsa.') or line.startswith('---') or line.startswith('...'):
                if line.startswith('amdhsa.') and not line.startswith('amdhsa.kernels'):
                    close_meta()
                continue
            meta_entry.append(line.rstrip())
            continue
        if line.strip() == '.amdgpu_metadata':
            in_meta = True
            continue
        if dname is not None:
            if line.strip() == '.end_amdhsa_kernel':
                desc[norm(dname)] = norm('\n'.join(dbody)); dname = None
            else:
                dbody.append(strip(line))
            continue
        m = re.match(r'^\s*\.amdhsa_kernel\s+(\S+)', line)
        if m:
            dname, dbody = m.group(1), []
            continue
        if kind is None:
            m = re.match(r'^([A-Za-z_$][\w$.]*):', line)
            if m and not line.startswith('.L'):
                name, body = m.group(1), []
                kind = 'object' if name in objects else 'function'
                continue
        else:
            if kind == 'function' and line.startswith('.Lfunc_end'):
                fn[norm(name)] = norm('\n'.join(body)); kind = None
            elif kind == 'object' and re.match(r'^\s*\.size\s+' + re.escape(name) + r'\s*,', line):
                body.append(strip(line).replace(name, 'SELF'))
                if not name.startswith('__hip_cuid_'):
                    obj[norm(name)] = norm('\n'.join(body))
                kind = None
            else:
                body.append(strip(line))
    return {'function': fn, 'descriptor': desc, 'object': obj, 'metadata': meta}


def main():
    allow_add = '--allow-additions' in sys.argv
    sys.argv = [x for x in sys.argv if x != '--allow-additions']
    a, b = parts(sys.argv[1]), parts(sys.argv[2])
    bad = False
    for side, p in (('first', a), ('second', b)):
        if not p['function']:
            print(f'ERROR: no functions in the {side} listing ({sys.argv[1 if side == "first" else 2]}) - a failed compile?')
            bad = True
        kernels = set(p['descriptor'])
        if kernels - set(p['function']) or kernels - set(p['metadata']):
            print(f'ERROR: {side} listing: kernels with a descriptor but no body / no metadata entry: {sorted((kernels - set(p["function"])) | (kernels - set(p["metadata"])))[:5]}')
            bad = True
    summary = []
    for kind in ('function', 'descriptor', 'object', 'metadata'):
        x, y = a[kind], b[kind]
        only_a, only_b = sorted(set(x) - set(y)), sorted(set(y) - set(x))
        differ = [n for n in sorted(set(x) & set(y)) if x[n] != y[n]]
        summary.append(f'{kind}s {len(x)} / {len(y)} (only first {len(only_a)}, only second {len(only_b)}, differ {len(differ)})')
        for tag, names in (('only in first', only_a), ('only in second', only_b), ('DIFFERS', differ)):
            for n in names[:12]:
                print(f'   {kind} {tag}: {n[:150]}')
        if only_a or differ or (only_b and not allow_add):
            bad = True
    print('; '.join(summary))
    return 1 if bad else 0


if __name__ == '__main__':
    sys.exit(main())
