"""CPU: Dilithium's finishing kernel without transforms (libsafecrypto_amd/csrc/dil_cfins.h; sc_hip_dilithium_set_tuning bit 11) run lane by lane.  The
kernel's own text is compiled with g++ over tests/emu/wave_emu.h (a host thread per lane, every cross-lane operation a barrier) with AddressSanitizer
and UBSan, fed the product's own tables (sc_hip_dilithium_fin_tables_host) and, per attempt, the mask y, w = A y and the challenge c the oracle computes;
the rounds of the signer are replayed around it (slots, active lists, kappa) as sign_core queues them.  Every signature, length and attempt count must be
the oracle's - whose signatures are pinned against the reference's (tests/test_oracle_dilithium.py).  This checks the kernel's logic, not its code
generation: the GPU run of tests/test_gpu_zz_dilithium_sparse_fin.py is still owed."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

from tests.golden_io import load_golden
from tests import oracle_lib as O
from tests.test_dilithium_fin_tables import _lib, _tables

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EMU = os.path.join(ROOT, 'tests', 'emu')


def _build(sanitize):
    out = os.path.join(EMU, '_build')
    os.makedirs(out, exist_ok=True)
    so = os.path.join(out, f'libemu_dil_cfins_{sanitize or "plain"}.so')
    csrc = os.path.join(ROOT, 'libsafecrypto_amd', 'csrc')
    deps = [os.path.join(EMU, 'emu_dil_cfins.cpp'), os.path.join(EMU, 'wave_emu.h'), os.path.join(csrc, 'dil_cfins.h'), os.path.join(csrc, 'mlwe_round.h')]
    if not os.path.exists(so) or any(os.path.getmtime(d) > os.path.getmtime(so) for d in deps):
        flags = ['-O1', '-g', '-std=c++17', '-fPIC', '-shared', '-pthread', '-Wall', '-Wno-unknown-pragmas', '-I' + csrc]
        if sanitize:
            flags += ['-fsanitize=' + sanitize, '-fno-omit-frame-pointer', '-fno-sanitize-recover=all']
        subprocess.check_call(['g++'] + flags + ['-o', so, deps[0]])
    return so


DRIVER = r'''
import ctypes as C, sys, numpy as np
sys.path.insert(0, sys.argv[1])
from tests.golden_io import load_golden
from tests import oracle_lib as O
from tests.test_dilithium_fin_tables import _lib, _tables
so, tag, B, S0, seed = sys.argv[2], sys.argv[3], int(sys.argv[4]), int(sys.argv[5]), int(sys.argv[6])
E = C.CDLL(so)
E.emu_dil_cfins_round.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32] + [C.c_void_p] * 6 + [C.c_size_t] + [C.c_void_p] * 4
G = load_golden('dilithium_ref.txt')
key = O.DilKey.from_golden(G, tag)
T, fits, bias = _tables(_lib(), key.par, key.s1, key.s2, key.t)
assert fits == 1
rng = np.random.default_rng(seed)
msgs = [bytes(rng.integers(0, 256, size=int(rng.integers(0, 100)), dtype=np.uint8)) for _ in range(B)]
want = [O.dilithium_sign(key, m) for m in msgs]
stride = 4096
sigs = np.zeros((B, stride), dtype=np.uint8); siglens = np.zeros(B, dtype=np.uint32); trials = np.zeros(B, dtype=np.uint32)
kappa = np.zeros(B, dtype=np.uint32)
p = lambda a: a.ctypes.data_as(C.c_void_p) if a is not None else None
active, S, rounds = None, S0, 0
while True:
    items = list(range(B)) if active is None else [int(i) for i in active]
    nact = len(items)
    y = np.zeros((nact * S, key.l, key.n), dtype=np.int32); w = np.zeros((nact * S, key.k, key.n), dtype=np.int32); c8 = np.zeros((nact * S, key.n), dtype=np.int8)
    for a, it in enumerate(items):
        y[a * S:(a + 1) * S], w[a * S:(a + 1) * S], c8[a * S:(a + 1) * S] = O.dilithium_attempts(key, msgs[it], int(kappa[it]), S)
    nxt = np.full(B + 4, 0xFFFFFFFF, dtype=np.uint32); cnt = np.zeros(1, dtype=np.uint32)
    act = None if active is None else np.ascontiguousarray(active, dtype=np.uint32)
    before = kappa.copy()
    rc = E.emu_dil_cfins_round(p(key.par), p(T), bias, nact, S, p(act), p(y), p(w), p(c8), p(kappa), p(sigs), stride, p(siglens), p(trials), p(nxt), p(cnt))
    assert rc == 0
    left = sorted(int(i) for i in nxt[:int(cnt[0])])
    expect_left = sorted(it for it in items if want[it][1] > int(before[it]) + S)
    assert left == expect_left, (rounds, left, expect_left)
    for it in items:
        if it in left:
            assert int(kappa[it]) == int(before[it]) + S
        else:
            es, et = want[it]
            assert int(trials[it]) == et and int(siglens[it]) == len(es), (it, int(trials[it]), et)
            assert bytes(sigs[it, :len(es)]) == es, it
            assert not sigs[it, len(es):].any()
    rounds += 1
    if not left:
        break
    active, S = np.array(left, dtype=np.uint32), max(1, S - 1)
    assert rounds < 40
print('OK', tag, B, rounds, max(t for _, t in want))
'''


def _run(so, tag, B, S, seed, preload=None):
    env = dict(os.environ)
    if preload:
        env['LD_PRELOAD'] = preload
        env['ASAN_OPTIONS'] = 'detect_leaks=0'
    r = subprocess.run(['python3', '-c', DRIVER, ROOT, so, tag, str(B), str(S), str(seed)], env=env, capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0 and r.stdout.strip().startswith('OK'), (r.stdout[-2000:], r.stderr[-4000:])
    return r.stdout.strip().split()


def _runtime(name):
    out = subprocess.run(['g++', '-print-file-name=' + name], capture_output=True, text=True).stdout.strip()
    return out if os.path.isabs(out) and os.path.exists(out) else None


@pytest.mark.parametrize('tag,B,S', [('dil3', 10, 3), ('dil2', 5, 2), ('dil1', 5, 3), ('dil0', 6, 1), ('dil3', 3, 40)])
def test_rounds_of_the_signer_give_the_oracles_signatures(tag, B, S):
    """ragged last workgroup (B not a multiple of 4), several rounds (the slot count shrinks as sign_core's does), every parameter set"""
    out = _run(_build(None), tag, B, S, 7 * B + S)
    assert int(out[3]) >= 1


def test_under_address_and_undefined_behaviour_sanitizers():
    asan = _runtime('libasan.so')
    if not asan:
        pytest.skip('no libasan beside g++')
    out = _run(_build('address,undefined'), 'dil3', 6, 3, 99, preload=asan)
    assert out[0] == 'OK'


KEYS_DRIVER = r'''
import ctypes as C, sys, numpy as np
sys.path.insert(0, sys.argv[1])
from tests.golden_io import load_golden
from tests import oracle_lib as O
so, tag, B, S0, seed, spoil = sys.argv[2], sys.argv[3], int(sys.argv[4]), int(sys.argv[5]), int(sys.argv[6]), int(sys.argv[7])
E = C.CDLL(so)
E.emu_dil_cfins_keys_round.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32] + [C.c_void_p] * 6 + [C.c_size_t] + [C.c_void_p] * 8
G = load_golden('dilithium_ref.txt')
par, eb = np.ascontiguousarray(G[f'{tag}_params'], dtype=np.int32), int(G[f'{tag}_eta_bits'][0])
n, q, qb, k, l, d, eta = (int(par[i]) for i in (0, 1, 2, 3, 4, 5, 10))
rng = np.random.default_rng(seed)
keys = []
for i in range(B):
    s1, s2, t, rho, K, tr, t1 = O.dilithium_keygen(par, eb, rng.integers(0, 256, size=32, dtype=np.uint8))
    keys.append(O.DilKey(par, s1, s2, t, rho, K, tr, t1, O.expand_matrix(rho, k * l, q, qb)))
msgs = [bytes(rng.integers(0, 256, size=int(rng.integers(0, 100)), dtype=np.uint8)) for _ in range(B)]
want = [O.dilithium_sign(keys[i], msgs[i]) for i in range(B)]
S1 = np.ascontiguousarray(np.stack([kk.s1 for kk in keys]), dtype=np.int32); S2 = np.ascontiguousarray(np.stack([kk.s2 for kk in keys]), dtype=np.int32)
T = np.stack([kk.t for kk in keys]).astype(np.int64)
t0 = T & ((1 << d) - 1); t0 = np.where(t0 > (1 << (d - 1)), t0 - (1 << d), t0); T0 = np.ascontiguousarray(np.where(t0 < 0, t0 + q, t0), dtype=np.int32)     # dil_t0_kernel
if spoil >= 0:
    S2[spoil, 3] = q - (eta + 1)                     # one key outside the range the fields are sized for
    if spoil + 1 < B: S1[spoil + 1, 0] = (S1[spoil + 1, 0] + q) % q if S1[spoil + 1, 0] < 0 else S1[spoil + 1, 0]       # (a canonical entry among centred ones is fine)
stride = 4096
sigs = np.zeros((B, stride), dtype=np.uint8); siglens = np.zeros(B, dtype=np.uint32); trials = np.zeros(B, dtype=np.uint32)
kappa = np.zeros(B, dtype=np.uint32); refused = np.zeros(1, dtype=np.uint32)
p = lambda a: a.ctypes.data_as(C.c_void_p) if a is not None else None
active, S, rounds = None, S0, 0
while True:
    items = list(range(B)) if active is None else [int(i) for i in active]
    nact = len(items)
    y = np.zeros((nact * S, l, n), dtype=np.int32); w = np.zeros((nact * S, k, n), dtype=np.int32); c8 = np.zeros((nact * S, n), dtype=np.int8)
    for a, it in enumerate(items):
        y[a * S:(a + 1) * S], w[a * S:(a + 1) * S], c8[a * S:(a + 1) * S] = O.dilithium_attempts(keys[it], msgs[it], int(kappa[it]), S)
    nxt = np.full(B + 4, 0xFFFFFFFF, dtype=np.uint32); cnt = np.zeros(1, dtype=np.uint32)
    act = None if active is None else np.ascontiguousarray(active, dtype=np.uint32)
    before = kappa.copy()
    rc = E.emu_dil_cfins_keys_round(p(par), eta, nact, S, p(act), p(y), p(w), p(c8), p(kappa), p(sigs), stride, p(siglens), p(trials), p(nxt), p(cnt), p(S1), p(S2), p(T0), p(refused))
    assert rc == 0
    left = sorted(int(i) for i in nxt[:int(cnt[0])])
    expect_left = sorted(it for it in items if it != spoil and want[it][1] > int(before[it]) + S)
    assert left == expect_left, (rounds, left, expect_left)
    for it in items:
        if it == spoil:
            assert int(siglens[it]) == 0 and not sigs[it].any() and int(kappa[it]) == 0      # neither signed nor queued
        elif it in left:
            assert int(kappa[it]) == int(before[it]) + S
        else:
            es, et = want[it]
            assert int(trials[it]) == et and int(siglens[it]) == len(es), (it, int(trials[it]), et)
            assert bytes(sigs[it, :len(es)]) == es, it
    rounds += 1
    if not left:
        break
    active, S = np.array(left, dtype=np.uint32), max(1, S - 1)
    assert rounds < 40
assert int(refused[0]) == (1 if spoil >= 0 else 0)
print('OK', tag, B, rounds)
'''


def _run_keys(so, tag, B, S, seed, spoil, preload=None):
    env = dict(os.environ)
    if preload:
        env['LD_PRELOAD'] = preload
        env['ASAN_OPTIONS'] = 'detect_leaks=0'
    r = subprocess.run(['python3', '-c', KEYS_DRIVER, ROOT, so, tag, str(B), str(S), str(seed), str(spoil)], env=env, capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0 and r.stdout.strip().startswith('OK'), (r.stdout[-2000:], r.stderr[-4000:])


@pytest.mark.parametrize('tag,B,S,spoil', [('dil3', 5, 3, -1), ('dil2', 3, 2, -1), ('dil1', 4, 3, 1), ('dil0', 3, 2, -1)])
def test_a_key_per_message(tag, B, S, spoil):
    """the PERKEY form of the kernel: every wavefront builds its message's tables from the raw key rows (oracle-generated key pairs, a ragged last
    workgroup, a key outside the supported range refused without touching its neighbours)"""
    _run_keys(_build(None), tag, B, S, 31 * B + S, spoil)


def test_a_key_per_message_under_sanitizers():
    asan = _runtime('libasan.so')
    if not asan:
        pytest.skip('no libasan beside g++')
    _run_keys(_build('address,undefined'), 'dil3', 3, 3, 5, 2, preload=asan)
