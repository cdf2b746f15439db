"""CPU: a ring-TESLA signing call of the engine run lane by lane - every kernel of a planned round as ts_sign_core queues them (csrc/tesla.hip): the round
state on the device, the masks from the tapes, both products a1 y, a2 y from ONE forward transform (the NTT kernel's OP_MULNTT2 form), poly_rounding, the
oracle resuming from the all-zero prefix state, f_function, ts_decide_kernel (sparse products through LDS shift tables, test_w / test_rejection, stopping at
the first failing test), the selection, the packing and the closing wipe.

The kernels' own text (csrc/tesla_sign_dev.h, keccak_dev.h, ntt_kernel.h) is compiled as host code over tests/emu/wave_emu.h (tests/emu_harness.py) with
AddressSanitizer + UBSan and with ThreadSanitizer; signatures, attempt counts and statuses must be the oracle's - whose signatures are pinned against the
reference's (tests/test_oracle_tesla.py) - over several rounds with a shrinking number of messages and a growing number of attempts per message, a tape that
ends inside the rejection loop, and the per-call secrets zero behind the call."""
import pytest

from tests.emu_harness import build, drive_all

DRIVER = r'''
import ctypes as C, sys, time, numpy as np
sys.path.insert(0, sys.argv[1])
from tests.golden_io import load_golden
from tests import oracle_lib as O
from tests.test_emu_ntt import PlanConsts, OP_MULNTT, SMAD, S32, U32
so, tag, B, cap, seed = sys.argv[2], sys.argv[3], int(sys.argv[4]), int(sys.argv[5]), int(sys.argv[6])
E = C.CDLL(so)
vp = C.c_void_p
E.emu_tesla_sign.argtypes = [vp, C.c_int, C.c_int, vp, vp, vp, vp, vp, vp, vp, vp, C.c_uint32, vp, C.c_size_t, vp, vp, C.c_size_t, vp, C.c_uint32, vp, C.c_size_t, vp, vp, vp, vp, vp]
G = load_golden('tesla_ref.txt')
key = O.TeslaKey.from_golden(G, tag)
n, q = key.n, key.q
P = PlanConsts(q, n)
mode = SMAD if P.smad_ok else S32 if P.s32_ok else U32          # sc_hip_plan_create's choice (no multiply-add form at n = 512 for these moduli)
assert not P.mad_ok
tab, modk, c, lazy = P.args(mode, OP_MULNTT)
rng = np.random.default_rng(seed)
W = 12 * n
msgs = [bytes(rng.integers(0, 256, size=int(rng.integers(0, 120)), dtype=np.uint8)) for _ in range(B)]
tapes = [rng.integers(0, 2**32, size=W, dtype=np.uint64).astype(np.uint32) for _ in range(B)]
if B > 2:
    tapes[1] = tapes[1][:n - 3]            # not even one attempt
    tapes[2] = tapes[2][:2 * n + 5]        # two attempts, then dry
stride = 128
mb = np.zeros((B, stride), dtype=np.uint8); ml = np.zeros(B, dtype=np.uint32)
tb = np.zeros((B, W), dtype=np.uint32); tl = np.zeros(B, dtype=np.uint32)
for i in range(B):
    mb[i, :len(msgs[i])] = np.frombuffer(msgs[i], dtype=np.uint8); ml[i] = len(msgs[i])
    tb[i, :len(tapes[i])] = tapes[i]; tl[i] = len(tapes[i])
sst = key.sig_bytes + 16
sigs = np.zeros((B, sst), dtype=np.uint8); siglens = np.zeros(B, dtype=np.uint32); trials = np.zeros(B, dtype=np.uint32); status = np.zeros(B, dtype=np.uint32)
rounds = C.c_uint32(0); residue = C.c_uint32(1)
p = lambda a: a.ctypes.data_as(vp) if a is not None else None
t0 = time.time()
rc = E.emu_tesla_sign(p(key.par), mode, int(lazy), p(tab), p(modk), p(c), p(key.a1), p(key.a2), p(key.s), p(key.e1), p(key.e2), B, p(mb), stride, p(ml), p(tb), W, p(tl), cap,
                      p(sigs), sst, p(siglens), p(trials), p(status), C.byref(rounds), C.byref(residue))
assert rc == 0, rc
multi = 0
for i in range(B):
    orc, es, et, _ = O.tesla_sign(key, msgs[i], tapes[i])
    assert (orc == 0) == (int(status[i]) == 1) and (orc == 2) == (int(status[i]) == 2), (i, orc, int(status[i]))
    if orc == 0:
        assert int(trials[i]) == et and int(siglens[i]) == len(es) and bytes(sigs[i, :len(es)]) == es, (i, int(trials[i]), et)
        multi += et >= 3
    else:
        assert int(siglens[i]) == 0
assert residue.value == 0
print('EMU_OK', tag, B, 'rounds', rounds.value, 'signed', int((status == 1).sum()), 'with 3+ attempts', multi, f'{time.time() - t0:.1f}s')
'''


VERIFY_DRIVER = r'''
import ctypes as C, sys, time, numpy as np
sys.path.insert(0, sys.argv[1])
from tests.golden_io import load_golden
from tests import oracle_lib as O
from tests.test_emu_ntt import PlanConsts, OP_MULNTT, SMAD, S32, U32
so, tag, B, seed = sys.argv[2], sys.argv[3], int(sys.argv[4]), int(sys.argv[5])
E = C.CDLL(so)
vp = C.c_void_p
E.emu_tesla_verify.argtypes = [vp, C.c_int, C.c_int, vp, vp, vp, vp, vp, vp, vp, C.c_uint32, vp, C.c_size_t, vp, vp, C.c_size_t, vp, vp]
G = load_golden('tesla_ref.txt')
key = O.TeslaKey.from_golden(G, tag)
n, q = key.n, key.q
P = PlanConsts(q, n)
mode = SMAD if P.smad_ok else S32 if P.s32_ok else U32
tab, modk, c, lazy = P.args(mode, OP_MULNTT)
rng = np.random.default_rng(seed)
msgs, sigs = [], []
while len(msgs) < B:
    m = bytes(rng.integers(0, 256, size=int(rng.integers(0, 120)), dtype=np.uint8))
    rc, sg, _, _ = O.tesla_sign(key, m, rng.integers(0, 2**32, size=12 * n, dtype=np.uint64).astype(np.uint32))
    if rc != 0:
        continue
    i = len(msgs); kind = i % 5
    s = bytearray(sg)
    if kind == 1: s[int(rng.integers(0, len(s)))] ^= 1 << int(rng.integers(0, 8))       # a flipped bit anywhere
    elif kind == 2: s[len(s) - 1 - int(rng.integers(0, 64))] ^= 0x20                       # in the digest
    elif kind == 3: m = m + b'x'                                                          # another message
    elif kind == 4: s = s[:len(s) - 3]                                                    # truncated
    msgs.append(m); sigs.append(bytes(s))
stride, sst = 128, key.sig_bytes + 16
mb = np.zeros((B, stride), dtype=np.uint8); ml = np.zeros(B, dtype=np.uint32)
sb = np.zeros((B, sst), dtype=np.uint8); sl = np.zeros(B, dtype=np.uint32)
for i in range(B):
    mb[i, :len(msgs[i])] = np.frombuffer(msgs[i], dtype=np.uint8); ml[i] = len(msgs[i])
    sb[i, :len(sigs[i])] = np.frombuffer(sigs[i], dtype=np.uint8); sl[i] = len(sigs[i])
verdicts = np.full(B, 7, dtype=np.uint32)
p = lambda a: a.ctypes.data_as(vp)
t0 = time.time()
rc = E.emu_tesla_verify(p(key.par), mode, int(lazy), p(tab), p(modk), p(c), p(key.a1), p(key.a2), p(key.t1), p(key.t2), B, p(mb), stride, p(ml), p(sb), sst, p(sl), p(verdicts))
assert rc == 0, rc
want = [0 if O.tesla_verify(key, msgs[i], sigs[i]) == 0 else 1 for i in range(B)]
assert verdicts.tolist() == want, (verdicts.tolist(), want)
assert want.count(0) >= B // 5 and want.count(1) >= B // 2
print('EMU_OK', tag, B, 'accepted', want.count(0), f'{time.time() - t0:.1f}s')
'''

VERIFY_CASES = [('tesla0', 11), ('tesla1', 6)]          # (parameter set, signatures - a fifth genuine, the rest tampered four ways)

CASES = [('tesla0', 10, 12), ('tesla0', 5, 20), ('tesla1', 6, 8)]          # (parameter set, messages, attempt slots per round)


@pytest.fixture(scope='module')
def runs():
    asan, tsan = build('emu_tesla_sign', 'address,undefined'), build('emu_tesla_sign', 'thread')
    cases = {c: (DRIVER, [asan, c[0], c[1], c[2], 5 * c[1] + c[2]], 'address,undefined') for c in CASES}
    cases['tsan'] = (DRIVER, [tsan, 'tesla0', 6, 8, 4], 'thread')
    cases.update({('verify',) + c: (VERIFY_DRIVER, [asan, c[0], c[1], 3 * c[1]], 'address,undefined') for c in VERIFY_CASES})
    cases['verify tsan'] = (VERIFY_DRIVER, [tsan, 'tesla0', 6, 2], 'thread')
    return drive_all(cases)


def _ok(r):
    assert r.returncode == 0 and 'EMU_OK' in r.stdout, (r.stdout[-2000:], r.stderr[-4000:])


@pytest.mark.parametrize('tag,B,cap', CASES)
def test_signing_call_equals_the_oracle_asan_ubsan(runs, tag, B, cap):
    _ok(runs[(tag, B, cap)])


def test_signing_call_is_race_free_tsan(runs):
    _ok(runs['tsan'])


@pytest.mark.parametrize('tag,B', VERIFY_CASES)
def test_verification_call_equals_the_oracle_asan_ubsan(runs, tag, B):
    """sc_hip_tesla_verify's path (z's unpacking and F as one launch, a1 z and a2 z from one forward transform, w = a z - t c through the key's shift tables,
    rounding, the oracle compared with the signature's digest): the oracle's verdict on genuine and tampered signatures"""
    _ok(runs[('verify', tag, B)])


def test_verification_call_is_race_free_tsan(runs):
    _ok(runs['verify tsan'])
