"""Builds and runs the lane-by-lane CPU emulation of the engine's kernels (tests/emu/; test infrastructure only).

A harness tests/emu/<name>.cpp includes kernel headers of libsafecrypto_amd/csrc/ as HOST code over tests/emu/wave_emu.h.  The headers are compiled from
a copy (tests/emu/_build/src*/) in which ONE construct is rewritten mechanically - the declaration of a workgroup's dynamic LDS,
`extern __shared__ [aligned] T name[];`, becomes a pointer to the emulation's LDS array - everything else is the product's text; csrc/scalar_hint.h and
<hip/hip_runtime.h> come from tests/emu/shim.  `rewrites` = further (header, old text, new text, occurrences) substitutions a harness needs and documents (a register read across lanes inside a
branch only one lane takes, which the device answers from the idle lanes' registers and a thread per lane cannot).  `mutate` removes the k-th `wave_lds_fence();` of one header: the harness's own test - a build with a fence
missing must come back from ThreadSanitizer as a data race."""
import glob
import hashlib
import os
import re
import shutil
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EMU = os.path.join(ROOT, 'tests', 'emu')
CSRC = os.path.join(ROOT, 'libsafecrypto_amd', 'csrc')
CLANG = '/opt/rocm/lib/llvm/bin/clang++'          # host mode: the kernel text uses clang vector types and builtins (ext_vector_type, __builtin_nontemporal_*)
DYN_LDS = re.compile(r'extern __shared__ (?:__attribute__\(\(aligned\(16\)\)\) )?(\w+) (\w+)\[\];')


def _rewrite(name, text, rewrites):
    """(file, old text or compiled pattern, new text, occurrences) substitutions of one file"""
    for fname, old, new, count in rewrites:
        if fname != name:
            continue
        if isinstance(old, re.Pattern):
            text, k = old.subn(new, text)
            assert k == count, (fname, old.pattern, k)
        else:
            assert text.count(old) == count, (fname, old, text.count(old))
            text = text.replace(old, new)
    return text


def _sources(dst, mutate=None, rewrites=()):
    shutil.rmtree(dst, ignore_errors=True)
    os.makedirs(dst)
    for h in glob.glob(os.path.join(CSRC, '*.h')):
        base = os.path.basename(h)
        if os.path.exists(os.path.join(EMU, 'shim', base)):
            continue
        text = DYN_LDS.sub(lambda m: f'{m.group(1)} *const {m.group(2)} = reinterpret_cast<{m.group(1)} *>(emu::dyn_lds);', open(h).read())
        if mutate and mutate[0] == base:
            at = [m.start() for m in re.finditer(r'wave_lds_fence\(\);', text)][mutate[1]]
            text = text[:at] + text[at + len('wave_lds_fence();'):]
        text = _rewrite(base, text, rewrites)
        with open(os.path.join(dst, base), 'w') as f:
            f.write(text)


class _BuildLock:
    """one builder per target at a time (the CPU suite runs its files on several workers: two of them asking for the same library must not compile into the same directory)"""

    def __init__(self, target):
        self.path = target + '.lock'

    def __enter__(self):
        import fcntl
        self.fh = open(self.path, 'w')
        fcntl.flock(self.fh, fcntl.LOCK_EX)
        return self

    def __exit__(self, *a):
        import fcntl
        fcntl.flock(self.fh, fcntl.LOCK_UN)
        self.fh.close()


def build(name, sanitize, mutate=None, out=None, rewrites=()):
    """tests/emu/<name>.cpp -> lib<name>_<sanitize>.so under tests/emu/_build (or `out`), rebuilt when a source is newer.  -O0: the run time is the threads'
    barriers either way, and the sanitizer builds of the larger harnesses take a minute at -O1"""
    out = out or os.path.join(EMU, '_build')
    os.makedirs(out, exist_ok=True)
    tag = (sanitize or 'plain').replace(',', '_') + (f'_mut{mutate[1]}' if mutate else '') + (('_' + hashlib.sha1(repr(rewrites).encode()).hexdigest()[:8]) if rewrites else '')
    so = os.path.join(out, f'lib{name}_{tag}.so')
    deps = [os.path.join(EMU, f'{name}.cpp'), os.path.join(EMU, 'wave_emu.h'), __file__] + glob.glob(os.path.join(EMU, 'shim', '*.h')) + \
           glob.glob(os.path.join(EMU, 'shim', 'hip', '*.h')) + glob.glob(os.path.join(CSRC, '*.h'))
    with _BuildLock(so):
        if not os.path.exists(so) or any(os.path.getmtime(d) > os.path.getmtime(so) for d in deps):
            src = os.path.join(out, f'src_{name}_{tag}')
            _sources(src, mutate, rewrites)
            flags = ['-O0', '-g', '-std=c++17', '-fPIC', '-shared', '-pthread', '-Wall', '-Wno-unknown-pragmas', '-Wno-keyword-macro', '-Wno-unused-function',
                     '-Wno-unused-variable', '-Wno-shift-count-negative', '-I' + os.path.join(EMU, 'shim'), '-I' + EMU, '-I' + src]
            if sanitize:
                flags += ['-fsanitize=' + sanitize, '-fno-omit-frame-pointer', '-fno-sanitize-recover=all', '-shared-libsan']
            tmp = so + f'.tmp{os.getpid()}'
            subprocess.check_call([CLANG] + flags + ['-o', tmp, deps[0]])
            os.replace(tmp, so)                                  # (a library in use elsewhere is replaced, never rewritten in place)
    return so


def sanitizer_env(sanitize):
    """a sanitized .so loaded into an unsanitized python needs its runtime preloaded"""
    env = dict(os.environ)
    if sanitize:
        rt = 'tsan' if 'thread' in sanitize else 'asan'
        env['LD_PRELOAD'] = subprocess.check_output([CLANG, f'-print-file-name=libclang_rt.{rt}-x86_64.so']).decode().strip()
        env['ASAN_OPTIONS'] = 'detect_leaks=0:abort_on_error=1'
        env['UBSAN_OPTIONS'] = 'halt_on_error=1:print_stacktrace=1'
        env['TSAN_OPTIONS'] = 'halt_on_error=1:report_signal_unsafe=0'
    return env


def drive(driver, args, sanitize, timeout=1500):
    """runs `driver` (python source) in a child with the sanitizer's runtime preloaded; returns the CompletedProcess"""
    return subprocess.run(['python3', '-c', driver, ROOT] + [str(a) for a in args], env=sanitizer_env(sanitize), capture_output=True, text=True, timeout=timeout)


def drive_all(cases, workers=4):
    """cases: {key: (driver, args, sanitize[, timeout])} -> {key: CompletedProcess}, `workers` children at a time (a child is a few hundred threads that
    mostly wait on each other: several of them side by side is what keeps the CPU suite in minutes)"""
    from concurrent.futures import ThreadPoolExecutor
    keys = list(cases)
    with ThreadPoolExecutor(max_workers=workers) as ex:
        futs = [ex.submit(drive, *cases[k][:3], **({'timeout': cases[k][3]} if len(cases[k]) > 3 else {})) for k in keys]
        return {k: f.result() for k, f in zip(keys, futs)}


# What the CPU build of the whole library rewrites besides the dynamic LDS declarations: the constructs that only mean something on the device - inline
# assembly (the ENS / DLP signer's v_add_f32_dpp chain -> the same additions in the same order through emu_row_ror; RLWE-Enc's borrow-chain compare-count ->
# the two 64-bit comparisons it computes; s_nop wait states and register-class hints -> nothing) - plus the lockstep statements of the single-scheme
# harnesses (tests/test_emu_*.py REWRITES), which apply to the same headers here.
LIB_REWRITES = (
    ('falcon.hip', re.compile(r'    if constexpr \(G == 4\)\n        asm\(.*?\n    return acc; ', re.S),
     '    for (int i = G - 1; i >= 1; i--) acc = emu_row_ror(p, (16 / G) * i) + acc;\n    acc = acc + p;\n    return acc; ', 1),
    ('falcon.hip', re.compile(r'asm volatile\("s_nop \d+"\);'), '', 2),
    ('rlwe.hip', re.compile(r'asm\("v_sub_co_u32 %1, vcc, %3, %7.*?: "vcc"\);', re.S),
     'scratch = 0; (void)scratch; { const uint64_t u_ = ((uint64_t)w[c + e].x << 32) | w[c + e].y; a = lo + ((((uint64_t)t1.y << 32) | t1.x) < u_ ? 1u : 0u) + ((((uint64_t)t2.y << 32) | t2.x) < u_ ? 1u : 0u); }', 1),
    ('rlwe.hip', 'asm volatile("" : "+v"(off));', '', 2),
    # the lockstep statements (each explained where its single-scheme harness lists it: tests/test_emu_bliss_sign.py, test_emu_dil_sign.py, test_emu_falcon_sign.py)
    ('bliss_sign_dev.h', 'bw.put((uint32_t)__builtin_amdgcn_readlane(pos_v, k), P.n_bits);', 'bw.put((uint32_t)cidx[(size_t)aidx * P.kappa + k], P.n_bits);', 2),
    ('keccak_dev.h', 'for (int w = 0; w < 25; w++) st[w * NT + h] = x[w];', 'for (int w = 0; w < 25; w++) st[w * NT + h] = x[w];\n        emu_pair_sync();', 2),
    ('bliss_sign_dev.h', 'wv = myfl[(idx >> 5) * (HT / 2)];', 'wv = myfl[(idx >> 5) * (HT / 2)]; emu_pair_sync();', 1),
    ('dil_sign_dev.h', 'if (neq) verdict[item] = 1;', 'if (neq) __atomic_store_n(&verdict[item], 1u, __ATOMIC_RELAXED);', 1),
    ('dil_verify_dev.h', 'if (bad) verdict[item] = 1;', 'if (bad) __atomic_store_n(&verdict[item], 1u, __ATOMIC_RELAXED);', 1),
    ('dil_verify_dev.h', 'fresh > (uint32_t)P.omega)) verdict[item] = 1;', 'fresh > (uint32_t)P.omega)) __atomic_store_n(&verdict[item], 1u, __ATOMIC_RELAXED);', 1),
    ('falcon_sign_dev.h', 'const double *LL = stree + (toff - toffLM);', 'const double *LL = l <= LM ? stree + (toff - toffLM) : stree;', 1),
    # the lanes of a signature all test its status word before lane 0 clears it - one load instruction, then one store, on the device
    ('falcon_sign_dev.h', 'if (status[b] != 7) return;', '{ const bool deferred_ = status[b] == 7; emu::wave_sync(); if (!deferred_) return; }', 1),
)


# ---- the whole product library for the CPU: libsafecrypto_amd/csrc/*.hip and *.cpp - host code and kernels - over tests/emu/hip_host_emu.h --------------------
LIB_SOURCES = ['sc_hip.hip', 'dilithium.hip', 'kyber.hip', 'bliss.hip', 'falcon.hip', 'ops.hip', 'drbg.hip', 'tesla.hip', 'rlwe.hip', 'ens.hip', 'dlp.hip', 'entropy.hip',
               'mlwe_xof.hip', 'sc_vtable.cpp', 'multi.cpp']


def build_library(sanitize, rewrites=LIB_REWRITES, sources=LIB_SOURCES, jobs=4):
    """tests/emu/_build/libsc_hip_emu_<sanitize>.so: every source of csrc/Makefile compiled as host C++ (same rewrites as `build`), linked into one library with
    the product's C ABI.  Rebuilt when a source is newer; one builder at a time per library, and a library in use elsewhere is replaced, never rewritten in place."""
    out = os.path.join(EMU, '_build')
    os.makedirs(out, exist_ok=True)
    tag = (sanitize or 'plain').replace(',', '_') + (('_' + hashlib.sha1(repr(rewrites).encode()).hexdigest()[:8]) if rewrites else '')
    if sanitize and os.environ.get('EMU_UBSAN_RECOVER'):          # triage: report every finding of a run instead of stopping at the first
        tag += '_recover'
    so = os.path.join(out, f'libsc_hip_emu_{tag}.so')
    deps = [os.path.join(EMU, 'wave_emu.h'), os.path.join(EMU, 'hip_host_emu.h'), __file__] + glob.glob(os.path.join(EMU, 'shim', '*.h')) + \
           glob.glob(os.path.join(EMU, 'shim', 'hip', '*.h')) + glob.glob(os.path.join(CSRC, '*.h')) + [os.path.join(CSRC, f) for f in sources] + \
           glob.glob(os.path.join(ROOT, 'include', '*.h'))
    with _BuildLock(so):
        if os.path.exists(so) and not any(os.path.getmtime(d) > os.path.getmtime(so) for d in deps):
            return so
        _build_library(so, tag, sanitize, rewrites, sources, jobs, out)
    return so


def _build_library(so, tag, sanitize, rewrites, sources, jobs, out):
    from concurrent.futures import ThreadPoolExecutor
    src = os.path.join(out, f'src_lib_{tag}')
    _sources(src, None, rewrites)
    os.makedirs(os.path.join(src, 'inc'), exist_ok=True)
    for h in glob.glob(os.path.join(ROOT, 'include', '*.h')):          # the sources include "../../include/sc_hip.h" relative to csrc/
        shutil.copy(h, os.path.join(src, 'inc'))
    flags = ['-O0', '-g', '-std=c++17', '-fPIC', '-pthread', '-Wall', '-Wno-unknown-pragmas', '-Wno-keyword-macro', '-Wno-unused-function', '-Wno-unused-variable',
             '-Wno-shift-count-negative', '-Wno-unused-but-set-variable', '-DEMU_SHARED=static', '-DEMU_HOST_API', '-DEMU_HAVE_NTT_CORE', '-D__HIPCC__',
             '-I' + os.path.join(EMU, 'shim'), '-I' + EMU, '-I' + src]
    if sanitize:
        # (float-cast-overflow off: the CDT table builder converts a long double that has reached 2^64 to a 64-bit integer and tests for 0, as the reference's does
        # (gaussian_cdf.c:598-602) - undefined as C, 0 from both compilers on x86-64, and the table has to be the reference's)
        flags += ['-fsanitize=' + sanitize, '-fno-omit-frame-pointer', '-fno-sanitize-recover=all', '-shared-libsan'] + (['-fno-sanitize=float-cast-overflow'] if 'undefined' in sanitize else [])
        if os.environ.get('EMU_UBSAN_RECOVER'):
            flags += ['-fsanitize-recover=undefined']

    def one(f):
        text = DYN_LDS.sub(lambda m: f'{m.group(1)} *const {m.group(2)} = reinterpret_cast<{m.group(1)} *>(emu::dyn_lds);', open(os.path.join(CSRC, f)).read())
        text = text.replace('#include "../../include/', '#include "inc/')
        text = _rewrite(f, text, rewrites)
        cpp = os.path.join(src, os.path.splitext(f)[0] + '_emu.cpp')
        with open(cpp, 'w') as fh:
            fh.write(text)
        obj = cpp[:-4] + '.o'
        extra = ['-ffp-contract=off'] if f == 'falcon.hip' else []
        subprocess.check_call([CLANG] + flags + extra + ['-c', '-o', obj, cpp])
        return obj

    with ThreadPoolExecutor(max_workers=jobs) as ex:
        objs = list(ex.map(one, sources))
    tmp = so + f'.tmp{os.getpid()}'
    subprocess.check_call([CLANG, '-shared', '-pthread'] + (['-fsanitize=' + sanitize, '-shared-libsan'] if sanitize else []) + ['-o', tmp] + objs)
    os.replace(tmp, so)
