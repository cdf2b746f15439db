"""CPU: a BLISS-B signing call of the engine, one key for the batch, run lane by lane - every kernel of a signing round as bl_round_launch queues them
(csrc/bliss.hip): begin, the CDT sampler with its guide and probes, stage, the NTT kernel's fused product with the broadcast public key, round_and_drop, the
oracle (SHA3-512 on the reference's permutation, one lane or two per message), and bl_decide_kernel - greedy_sc from the key's autocorrelation, the exp / cosh
rejection with its reservoir draws, signature_gen, check_norms, the packing - around the reference's open-ended loop.

The kernels' own text (csrc/bliss_sign_dev.h, keccak_dev.h, cdt_guide.h, ntt_kernel.h) is compiled as host code over tests/emu/wave_emu.h
(tests/emu_harness.py) with AddressSanitizer + UBSan and with ThreadSanitizer; signatures, attempt counts and statuses must be the oracle's - whose
signatures are pinned against the reference's (tests/test_oracle_bliss.py) - including a tape that runs dry and an empty message.  One substitution is
made in the copy that is compiled, and only there: the two packing loops that ONE lane runs read the challenge positions out of the other lanes' registers
(`v_readlane`, which the device answers from idle lanes; a thread per lane cannot) - they read the same values from the array the registers were loaded from; and two
pair barriers stand where the two lanes of a hash rely on running in lockstep (REWRITES below)."""
import pytest

from tests.emu_harness import build, drive_all

REWRITES = (('bliss_sign_dev.h', 'bw.put((uint32_t)__builtin_amdgcn_readlane(pos_v, k), P.n_bits);', 'bw.put((uint32_t)cidx[(size_t)aidx * P.kappa + k], P.n_bits);', 2),
            # the two lanes of a hash run in lockstep on the device: a lane reads the half of the sponge state its partner wrote back, and both lanes test the
            # bitmap word before either sets a bit in it - a pair barrier where a thread per lane needs one
            ('keccak_dev.h', 'for (int w = 0; w < 25; w++) st[w * NT + h] = x[w];', 'for (int w = 0; w < 25; w++) st[w * NT + h] = x[w];\n        emu_pair_sync();', 2),
            ('bliss_sign_dev.h', 'wv = myfl[(idx >> 5) * (HT / 2)];', 'wv = myfl[(idx >> 5) * (HT / 2)]; emu_pair_sync();', 1))

DRIVER = r'''
import ctypes as C, sys, time, numpy as np
sys.path.insert(0, sys.argv[1])
from tests.golden_io import load_golden
from tests import oracle_lib as O
from tests.test_emu_ntt import PlanConsts, OP_MULNTT, SMAD, S32, MAD, U32
from libsafecrypto_amd import engine
so, tag, B, two_lane, seed = sys.argv[2], sys.argv[3], int(sys.argv[4]), int(sys.argv[5]), int(sys.argv[6])
E = C.CDLL(so)
vp = C.c_void_p
E.emu_bliss_sign.argtypes = [vp, C.c_int, vp, vp, vp, vp, C.c_int, vp, vp, vp, vp, vp, C.c_uint32, vp, C.c_size_t, vp, vp, C.c_size_t, vp, C.c_int, vp, C.c_size_t, vp, vp, vp, vp]
G = load_golden('bliss_ref.txt')
key = O.BlissKey.from_golden(G, tag)
n, q = key.n, key.q
P = PlanConsts(q, n)
mode = SMAD if P.smad_ok else S32 if P.s32_ok else MAD if P.mad_ok else U32          # sc_hip_plan_create's choice
tab, modk, c, _ = P.args(mode, OP_MULNTT)
# the sampler's table (the oracle's = the reference's) through the product's own guide builder
par = key.par
sigma, tail = (np.array([int(par[12]), int(par[14])], dtype=np.int32).view(np.float32))
cdf = O.cdt_table(float(tail), float(sigma))
L = C.CDLL(engine.LIB_PATH)
L.sc_hip_cdt_guide_host.argtypes = [vp, C.c_int, vp, vp, C.c_size_t, vp, C.c_size_t, vp]
gp = np.zeros(10, dtype=np.uint32); bins = np.zeros(1 << 15, dtype=np.uint16); probe = np.zeros(cdf.size + 8, dtype=np.uint64); blen = C.c_size_t(0)
p = lambda a: a.ctypes.data_as(vp) if a is not None else None
assert L.sc_hip_cdt_guide_host(p(cdf), cdf.size, p(gp), p(bins), bins.size, p(probe), probe.size, C.byref(blen)) == 0
assert 1 <= int(gp[6]) <= 7, gp                                                    # the guide is in use (S probes), as on the device
rng = np.random.default_rng(seed)
W = 3 * 4 * n + 64
msgs = [bytes(rng.integers(0, 256, size=int(rng.integers(1, 90)), dtype=np.uint8)) for _ in range(B)]
tapes = [rng.integers(0, 2**32, size=W, dtype=np.uint64).astype(np.uint32) for _ in range(B)]
if B > 3:
    tapes[1] = tapes[1][:4 * n - 7]        # cannot even sample once
    tapes[2] = tapes[2][:4 * n + 1]        # runs dry inside the decision's draws
    msgs[3] = b''                          # refused by the oracle function
stride = 96
mb = np.zeros((B, stride), dtype=np.uint8); ml = np.zeros(B, dtype=np.uint32)
tb = np.zeros((B, W), dtype=np.uint32); tl = np.zeros(B, dtype=np.uint32)
for i in range(B):
    mb[i, :len(msgs[i])] = np.frombuffer(msgs[i], dtype=np.uint8); ml[i] = len(msgs[i])
    tb[i, :len(tapes[i])] = tapes[i]; tl[i] = len(tapes[i])
sst = 1536
sigs = np.zeros((B, sst), dtype=np.uint8); siglens = np.zeros(B, dtype=np.uint32); trials = np.zeros(B, dtype=np.uint32); status = np.zeros(B, dtype=np.uint32)
rounds = C.c_uint32(0)
t0 = time.time()
rc = E.emu_bliss_sign(p(par), mode, p(tab), p(modk), p(c), p(probe), cdf.size, p(bins), p(gp), p(key.f), p(key.g), p(key.a), B, p(mb), stride, p(ml), p(tb), W, p(tl), two_lane,
                      p(sigs), sst, p(siglens), p(trials), p(status), C.byref(rounds))
assert rc == 0, rc
multi = 0
for i in range(B):
    if B > 3 and i == 3:
        assert int(status[i]) == 3 and int(siglens[i]) == 0
        continue
    orc, es, et, _ = O.bliss_sign(key, msgs[i], tapes[i])
    assert (orc == 0) == (int(status[i]) == 1) and (orc == 2) == (int(status[i]) == 2), (i, orc, int(status[i]))
    if orc == 0:
        assert int(trials[i]) == et and int(siglens[i]) == len(es) and bytes(sigs[i, :len(es)]) == es, i
        multi += et >= 2
    else:
        assert int(siglens[i]) == 0
print('EMU_OK', tag, B, 'rounds', rounds.value, 'signed', int((status == 1).sum()), 'with 2+ attempts', multi, f'{time.time() - t0:.1f}s')
'''


VERIFY_DRIVER = r'''
import ctypes as C, sys, time, numpy as np
sys.path.insert(0, sys.argv[1])
from tests.golden_io import load_golden
from tests import oracle_lib as O
from tests.test_emu_ntt import PlanConsts, OP_MULNTT, SMAD, S32, MAD, U32
so, tag, B, two_lane, seed = sys.argv[2], sys.argv[3], int(sys.argv[4]), int(sys.argv[5]), int(sys.argv[6])
E = C.CDLL(so)
vp = C.c_void_p
E.emu_bliss_verify.argtypes = [vp, C.c_int, vp, vp, vp, vp, C.c_uint32, vp, C.c_size_t, vp, vp, C.c_size_t, vp, C.c_int, vp]
G = load_golden('bliss_ref.txt')
key = O.BlissKey.from_golden(G, tag)
n, q = key.n, key.q
P = PlanConsts(q, n)
mode = SMAD if P.smad_ok else S32 if P.s32_ok else MAD if P.mad_ok else U32
tab, modk, c, _ = P.args(mode, OP_MULNTT)
rng = np.random.default_rng(seed)
msgs, sigs = [], []
while len(msgs) < B:
    m = bytes(rng.integers(0, 256, size=int(rng.integers(1, 90)), dtype=np.uint8))
    tape = rng.integers(0, 2**32, size=6 * 4 * n + 64, dtype=np.uint64).astype(np.uint32)
    rc, sg, _, _ = O.bliss_sign(key, m, tape)
    if rc != 0:
        continue
    i = len(msgs); kind = i % 5
    s = bytearray(sg)
    if kind == 1: s[int(rng.integers(0, len(s)))] ^= 1 << int(rng.integers(0, 8))       # a flipped bit anywhere
    elif kind == 2: s[len(s) - 1 - int(rng.integers(0, 20))] ^= 0x08                       # in the challenge positions
    elif kind == 3: m = m + b'x'                                                          # another message
    elif kind == 4: s[0] ^= 0x80                                                          # the top bit of the first coefficient of t: a norm violation
    msgs.append(m); sigs.append(bytes(s))
stride, sst = 96, 1536
mb = np.zeros((B, stride), dtype=np.uint8); ml = np.zeros(B, dtype=np.uint32)
sb = np.zeros((B, sst), dtype=np.uint8); sl = np.zeros(B, dtype=np.uint32)
for i in range(B):
    mb[i, :len(msgs[i])] = np.frombuffer(msgs[i], dtype=np.uint8); ml[i] = len(msgs[i])
    sb[i, :len(sigs[i])] = np.frombuffer(sigs[i], dtype=np.uint8); sl[i] = len(sigs[i])
verdicts = np.full(B, 7, dtype=np.uint32)
p = lambda a: a.ctypes.data_as(vp)
t0 = time.time()
rc = E.emu_bliss_verify(p(key.par), mode, p(tab), p(modk), p(c), p(key.a), B, p(mb), stride, p(ml), p(sb), sst, p(sl), two_lane, p(verdicts))
assert rc == 0, rc
want = [0 if O.bliss_verify(key, msgs[i], sigs[i]) == 0 else 1 for i in range(B)]
assert verdicts.tolist() == want, (verdicts.tolist(), want)
assert want.count(0) >= B // 5 and want.count(1) >= B // 2
print('EMU_OK', tag, B, 'accepted', want.count(0), f'{time.time() - t0:.1f}s')
'''

VERIFY_CASES = [('bliss1', 11, 1), ('bliss4', 6, 0), ('bliss3', 6, 1)]          # (parameter set, signatures - a fifth genuine, the rest tampered four ways -, two lanes per oracle hash)

CASES = [('bliss1', 12, 1), ('bliss1', 6, 0), ('bliss4', 8, 1), ('bliss3', 6, 1)]          # (parameter set, messages, two lanes per oracle hash)


@pytest.fixture(scope='module')
def runs():
    asan, tsan = build('emu_bliss_sign', 'address,undefined', rewrites=REWRITES), build('emu_bliss_sign', 'thread', rewrites=REWRITES)
    cases = {c: (DRIVER, [asan, c[0], c[1], c[2], 7 * c[1] + c[2]], 'address,undefined') for c in CASES}
    cases['tsan'] = (DRIVER, [tsan, 'bliss1', 6, 0, 3], 'thread')
    cases.update({('verify',) + c: (VERIFY_DRIVER, [asan, c[0], c[1], c[2], 9 * c[1] + c[2]], 'address,undefined') for c in VERIFY_CASES})
    cases['verify tsan'] = (VERIFY_DRIVER, [tsan, 'bliss1', 6, 0, 4], 'thread')
    return drive_all(cases)


def _ok(r):
    assert r.returncode == 0 and 'EMU_OK' in r.stdout, (r.stdout[-2000:], r.stderr[-4000:])


@pytest.mark.parametrize('tag,B,two_lane', CASES)
def test_signing_call_equals_the_oracle_asan_ubsan(runs, tag, B, two_lane):
    _ok(runs[(tag, B, two_lane)])


def test_signing_call_is_race_free_tsan(runs):
    """with one lane per oracle hash: the two lanes of bl_oracle2_kernel keep ONE bitmap of drawn indices and both store the same value to the same word
    - by design (one store instruction on the device), a write-write report for two threads"""
    _ok(runs['tsan'])


@pytest.mark.parametrize('tag,B,two_lane', VERIFY_CASES)
def test_verification_call_equals_the_oracle_asan_ubsan(runs, tag, B, two_lane):
    """sc_hip_bliss_verify's path for raw signatures (unpacking, norm tests, a t on the NTT kernel, z from it and the challenge positions, the oracle compared):
    the oracle's verdict on genuine and tampered signatures"""
    _ok(runs[('verify', tag, B, two_lane)])


def test_verification_call_is_race_free_tsan(runs):
    _ok(runs['verify tsan'])
