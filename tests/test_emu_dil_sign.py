"""CPU: a whole Dilithium signing call of the engine, one key for the batch, run lane by lane - every kernel sc_hip_dilithium_set_privkey and sign_core
queue (csrc/dilithium.hip), in their order and with their grids: key preparation through the transform kernel, the matrix into plain + Shoup form, mu, and
per round the round state, ExpandMask (the two-lane Keccak with its DPP exchanges), the fused NTT(y) -> A NTT(y) -> INTT kernel with the matrix in LDS, both
forms of the challenge hash, the finishing kernel - the transform one, which is the product's default, or the table one behind tuning bit 11 - and the wipe.

The kernels' own text (csrc/dil_sign_dev.h, mlwe_dev.h, keccak_dev.h, ntt_core.h, dil_cfins.h) is compiled as host code over tests/emu/wave_emu.h
(tests/emu_harness.py) with AddressSanitizer + UBSan, and with ThreadSanitizer (a missing wavefront fence is a data race: checked by taking one out).
The GPU pool offers neither.  Signatures, lengths and attempt counts must be the oracle's, byte for byte, under all four parameter sets, over several
rounds of the speculative rejection loop (few slots per message, so that messages go round again), ragged and empty messages; the per-call secrets must
be zero behind the call.  The round loop in tests/emu/emu_dil_sign.cpp restates sign_core's (HIP API calls around these launches); this checks the
kernels' logic and their interplay, not code generation or speed - tests/test_gpu_dilithium.py does that on the GPU."""
import pytest

from tests.emu_harness import build, drive, drive_all

DRIVER = r'''
import ctypes as C, sys, time, numpy as np
sys.path.insert(0, sys.argv[1])
from tests.golden_io import load_golden
from tests import oracle_lib as O
from tests.test_emu_ntt import PlanConsts, OP_INV, S32, U32
from tests.test_dilithium_fin_tables import _lib, _tables
so, tag, B, cap, s_max, flags, seed = sys.argv[2], sys.argv[3], int(sys.argv[4]), int(sys.argv[5]), int(sys.argv[6]), int(sys.argv[7]), int(sys.argv[8])
E = C.CDLL(so)
vp = C.c_void_p
E.emu_dil_sign.argtypes = [vp, vp, vp, vp, C.c_uint32, C.c_uint32, C.c_int, vp, vp, vp, vp, vp, vp, vp, C.c_uint32, C.c_uint32, vp, C.c_size_t, vp, C.c_uint32, C.c_uint32, C.c_uint32, C.c_int, vp, C.c_size_t, vp, vp, vp, vp]
G = load_golden('dilithium_ref.txt')
key = O.DilKey.from_golden(G, tag)
q, n = key.q, key.n
P = PlanConsts(q, n)
lm = max(key.k, key.l)
sg = 1 if (P.s32_ok and 32 * lm * q < (1 << 31)) else 0
tab, modk, cinv, _ = P.args(S32 if sg else U32, OP_INV)
_, modk32, _, _ = P.args(U32, OP_INV)
r2 = P.R; r2p = (P.R << 32) // q
T, fits, bias = _tables(_lib(), key.par, key.s1, key.s2, key.t)
rng = np.random.default_rng(seed)
msgs = [bytes(rng.integers(0, 256, size=int(rng.integers(0, 100)), dtype=np.uint8)) for _ in range(B)]
stride = 104
mb = np.zeros((B, stride), dtype=np.uint8); ml = np.zeros(B, dtype=np.uint32)
for i, m in enumerate(msgs):
    mb[i, :len(m)] = np.frombuffer(m, dtype=np.uint8); ml[i] = len(m)
sst = 4096
sigs = np.zeros((B, sst), dtype=np.uint8); siglens = np.zeros(B, dtype=np.uint32); trials = np.zeros(B, dtype=np.uint32)
rounds = C.c_uint32(0); residue = C.c_uint32(1)
p = lambda a: a.ctypes.data_as(vp)
t0 = time.time()
rc = E.emu_dil_sign(p(key.par), p(tab), p(modk32), p(cinv), r2, r2p, sg, p(key.s1), p(key.s2), p(key.t), p(key.K), p(key.tr), p(key.Ahat), p(T), bias,
                    B, p(mb), stride, p(ml), 0, cap, s_max, flags, p(sigs), sst, p(siglens), p(trials), C.byref(rounds), C.byref(residue))
dt = time.time() - t0
assert rc == 0, rc
for i, m in enumerate(msgs):
    es, et = O.dilithium_sign(key, m)
    assert int(trials[i]) == et, (i, int(trials[i]), et)
    assert int(siglens[i]) == len(es) and bytes(sigs[i, :len(es)]) == es, i
    assert not sigs[i, len(es):].any()
assert residue.value == 0
print('EMU_OK', tag, B, 'rounds', rounds.value, 'max trials', int(trials.max()), f'{dt:.1f}s')
'''


# the verifier's "this signature is invalid" flag is stored by every lane that finds a reason - the same value to the same word, one store instruction on the
# device - which two threads cannot do without a report: an atomic store in the copy that is compiled
REWRITES = (('dil_sign_dev.h', 'if (neq) verdict[item] = 1;', 'if (neq) __atomic_store_n(&verdict[item], 1u, __ATOMIC_RELAXED);', 1),
            ('dil_verify_dev.h', 'if (bad) verdict[item] = 1;', 'if (bad) __atomic_store_n(&verdict[item], 1u, __ATOMIC_RELAXED);', 1),
            ('dil_verify_dev.h', 'fresh > (uint32_t)P.omega)) verdict[item] = 1;', 'fresh > (uint32_t)P.omega)) __atomic_store_n(&verdict[item], 1u, __ATOMIC_RELAXED);', 1))


def _build(sanitize, **kw):
    return build('emu_dil_sign', sanitize, rewrites=REWRITES, **kw)


def _sign(sanitize, tag, B, cap, s_max, flags, seed, so=None):
    so = so or _build(sanitize)
    return drive(DRIVER, [so, tag, B, cap, s_max, flags, seed], sanitize)


# (parameter set, messages, attempt slots per round, most slots per message, flags: 1 = the table finishing kernel, 2 = one lane per hash, 4 = persistent fused kernel)
SIGN_CASES = [('dil3', 9, 18, 3, 0), ('dil3', 6, 12, 3, 1), ('dil3', 7, 20, 32, 2), ('dil3', 13, 32, 4, 4), ('dil0', 6, 12, 32, 0), ('dil1', 6, 12, 32, 1), ('dil2', 4, 8, 2, 3)]
# (parameter set, signatures - a sixth of them genuine, the rest tampered five ways -, flags: 2 = one lane per challenge hash)
VERIFY_CASES = [('dil3', 13, 0), ('dil3', 7, 2), ('dil0', 7, 0), ('dil1', 7, 0), ('dil2', 7, 0)]


@pytest.fixture(scope='module')
def runs():
    """every case of this module, a few children at a time"""
    asan, tsan = _build('address,undefined'), _build('thread')
    cases = {('sign', 'asan') + c: (DRIVER, [asan, c[0], c[1], c[2], c[3], c[4], 11 * c[1] + c[4]], 'address,undefined') for c in SIGN_CASES}
    cases.update({('sign', 'tsan', f): (DRIVER, [tsan, 'dil3', 5, 10, 3, f, 2], 'thread') for f in (0, 1)})
    cases.update({('verify', 'asan') + c: (VERIFY_DRIVER, [asan, c[0], c[1], c[2], 3 * c[1] + c[2]], 'address,undefined') for c in VERIFY_CASES})
    cases[('verify', 'tsan')] = (VERIFY_DRIVER, [tsan, 'dil3', 7, 0, 5], 'thread')
    return drive_all(cases)


def _ok(r):
    assert r.returncode == 0 and 'EMU_OK' in r.stdout, (r.stdout[-2000:], r.stderr[-4000:])


@pytest.mark.parametrize('tag,B,cap,s_max,flags', SIGN_CASES)
def test_signing_call_equals_the_oracle_asan_ubsan(runs, tag, B, cap, s_max, flags):
    _ok(runs[('sign', 'asan', tag, B, cap, s_max, flags)])


@pytest.mark.parametrize('flags', [0, 1])
def test_signing_call_is_race_free_tsan(runs, flags):
    _ok(runs[('sign', 'tsan', flags)])


def test_the_harness_reports_a_removed_fence(tmp_path):
    """the finishing kernel without the fence between a wavefront's write of c into LDS and the lanes' reads of it: a data race in dil_cfin_kernel"""
    so = _build('thread', mutate=('dil_sign_dev.h', 2), out=str(tmp_path))
    r = _sign('thread', 'dil3', 5, 10, 3, 0, 2, so=so)
    assert r.returncode != 0 and 'ThreadSanitizer: data race' in r.stderr and 'dil_cfin_kernel' in r.stderr, (r.stdout[-500:], r.stderr[-2000:])


VERIFY_DRIVER = r'''
import ctypes as C, sys, time, numpy as np
sys.path.insert(0, sys.argv[1])
from tests.golden_io import load_golden
from tests import oracle_lib as O
from tests.test_emu_ntt import PlanConsts, OP_INV, S32, U32
so, tag, B, flags, seed = sys.argv[2], sys.argv[3], int(sys.argv[4]), int(sys.argv[5]), int(sys.argv[6])
E = C.CDLL(so)
vp = C.c_void_p
E.emu_dil_verify.argtypes = [vp, vp, vp, vp, C.c_uint32, C.c_uint32, C.c_int, vp, vp, vp, C.c_uint32, vp, C.c_size_t, vp, C.c_uint32, vp, C.c_size_t, vp, C.c_int, vp]
G = load_golden('dilithium_ref.txt')
key = O.DilKey.from_golden(G, tag)
q, n = key.q, key.n
P = PlanConsts(q, n)
sg = 1 if (P.s32_ok and 32 * max(key.k, key.l) * q < (1 << 31)) else 0
tab, modk, cinv, _ = P.args(S32 if sg else U32, OP_INV)
r2 = P.R; r2p = (P.R << 32) // q
rng = np.random.default_rng(seed)
msgs, sigs = [], []
for i in range(B):
    m = bytes(rng.integers(0, 256, size=int(rng.integers(0, 100)), dtype=np.uint8))
    sg_, _ = O.dilithium_sign(key, m)
    kind = i % 6
    s = bytearray(sg_)
    if kind == 1: s[int(rng.integers(0, len(s)))] ^= 1 << int(rng.integers(0, 8))            # a flipped bit anywhere
    elif kind == 2: s[len(s) - 70 - int(rng.integers(0, 40))] ^= 0x10                           # in the tail: hint count / positions / c
    elif kind == 3: m = m + b'x'                                                               # another message
    elif kind == 4: s = s[:len(s) - 1 - int(rng.integers(0, 60))]                               # truncated
    elif kind == 5: s[key.l * n * int(key.par[11]) // 8] = 0xFF                                 # a hint count above omega
    msgs.append(m); sigs.append(bytes(s))
stride, sst = 104, 4096
mb = np.zeros((B, stride), dtype=np.uint8); ml = np.zeros(B, dtype=np.uint32)
sb = np.zeros((B, sst), dtype=np.uint8); sl = np.zeros(B, dtype=np.uint32)
for i in range(B):
    mb[i, :len(msgs[i])] = np.frombuffer(msgs[i], dtype=np.uint8); ml[i] = len(msgs[i])
    sb[i, :len(sigs[i])] = np.frombuffer(sigs[i], dtype=np.uint8); sl[i] = len(sigs[i])
sl[B - 1] = sst + 5                                                                            # a stated length beyond the row: invalid before anything is read
verdicts = np.full(B, 7, dtype=np.uint32)
p = lambda a: a.ctypes.data_as(vp)
t0 = time.time()
rc = E.emu_dil_verify(p(key.par), p(tab), p(modk), p(cinv), r2, r2p, sg, p(key.t1), p(key.rho), p(key.Ahat), B, p(mb), stride, p(ml), 0, p(sb), sst, p(sl), flags, p(verdicts))
assert rc == 0, rc
want = [0 if O.dilithium_verify(key, msgs[i], sigs[i]) == 0 else 1 for i in range(B)]
want[B - 1] = 1
assert verdicts.tolist() == want, (verdicts.tolist(), want)
assert want.count(0) >= B // 6 and want.count(1) >= B // 2
print('EMU_OK', tag, B, 'accepted', want.count(0), f'{time.time() - t0:.1f}s')
'''


@pytest.mark.parametrize('tag,B,flags', VERIFY_CASES)
def test_verification_call_equals_the_oracle_asan_ubsan(runs, tag, B, flags):
    """sc_hip_dilithium_set_pubkey + sc_hip_dilithium_verify's default path (the three readers of the inputs as one launch, NTT(c), the fused matrix product with
    c t1 2^d and UseHint, the challenge hash): the oracle's verdict on every signature"""
    _ok(runs[('verify', 'asan', tag, B, flags)])


def test_verification_call_is_race_free_tsan(runs):
    _ok(runs[('verify', 'tsan')])
