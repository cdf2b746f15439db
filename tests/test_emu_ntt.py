"""CPU: the batched NTT kernel - the headline kernel of bench.py and the transform inside every scheme - run lane by lane.

libsafecrypto_amd/csrc/ntt_kernel.h (over ntt_core.h and modarith.h) is compiled as HOST code over tests/emu/wave_emu.h (a host thread per lane, 64 lanes a
wavefront, every wavefront fence a barrier over the wavefront's threads) with AddressSanitizer + UBSan, and once more with ThreadSanitizer - under which a
missing `wave_lds_fence` between the lanes of an LDS exchange is a reported data race.  The GPU pool offers neither sanitizer.  Every instantiation the host
side can select (csrc/sc_hip.hip launch_ntt_l: five ops x five arithmetic modes x lazy / not x the streaming I/O form, n = 256 / 512 / 1024) runs over a
ragged batch with a second trip of the persistent loop, extreme and seeded inputs, and must give the oracle's words - which are pinned against the
reference's (tests/test_oracle.py, tests/golden/ntt_ref.txt).  The plan constants are computed here from their definitions (csrc/sc_hip.hip
sc_hip_plan_create) - this checks the kernel's logic, not its code generation and not the host's table builder (tests/test_gpu_ntt.py does both on the GPU)."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

from tests import oracle_lib as O

from tests.emu_harness import ROOT, build, drive, drive_all

OP_FWD, OP_INV, OP_MULNTT, OP_MUL, OP_MULNTT2 = range(5)
U24, U32, MAD, SMAD, S32 = 1, 2, 3, 4, 5


def brv(x, bits):
    return int(f'{x:0{bits}b}'[::-1], 2) if bits else 0


class PlanConsts:
    """what sc_hip_plan_create computes and run_ntt hands the kernel (csrc/sc_hip.hip), from the definitions"""

    def __init__(self, q, n):
        self.q, self.n = q, n
        self.L = L = {256: 8, 512: 9, 1024: 10}[n]
        g = next(m for m in range(2, q - 1) if pow(m, n, q) == q - 1)          # the reference's root (roots_of_unity.c:67-88)
        self.zf = [pow(g, brv(k, L), q) for k in range(n)]
        self.zi = [pow(z, q - 2, q) for z in self.zf]
        self.ninv = pow(n, q - 2, q)
        self.R = (1 << 32) % q
        qinv = 1
        for _ in range(5):
            qinv = (qinv * (2 - q * qinv)) & 0xFFFFFFFF
        self.qinv = qinv
        self.lazy24 = 2 * q * (L + 2) < (1 << 24)
        self.lazy32 = q < (1 << 25)
        self.mad_ok = q < (1 << 14) and 2 * q * (q - 1) * (L + 1) < (1 << 32)
        self.smad_ok = (n == 256 and 36 * q * q < (1 << 31)) or (n > 256 and q < (1 << 14) and q * q * L < (1 << 31))
        self.s32_ok = q < (1 << 25) and (n == 256 or q < (1 << 24))

    def modes(self):
        return ([U24] if self.q < (1 << 23) else []) + [U32] + ([MAD] if self.mad_ok else []) + ([SMAD] if self.smad_ok else []) + ([S32] if self.s32_ok else [])

    def _shoup(self, w, bits):
        return (w << bits) // self.q

    def _centre(self, w):
        return w - self.q if w > self.q // 2 else w

    def _quad(self, s0, s1, bits, signed):
        if signed:
            s0, s1 = self._centre(s0), self._centre(s1)
        return [x & 0xFFFFFFFF for x in (s0, (s0 << bits) // self.q, s1, (s1 << bits) // self.q)]          # (// floors: the signed companions' rounding)

    def args(self, mode, op):
        q, n = self.q, self.n
        signed = mode in (SMAD, S32)
        bits = 24 if mode == U24 else 32
        cen = self._centre if signed else (lambda w: w)
        tab = [cen(z) for z in self.zf] + [(cen(z) << bits) // q for z in self.zf] + [cen(z) for z in self.zi] + [(cen(z) << bits) // q for z in self.zi]
        tab = np.array([t & 0xFFFFFFFF for t in tab], dtype=np.uint32)
        modk = np.array([q, 2 * q, self.qinv, (1 << 32) // q, (1 << bits) // q, (q - 1) >> 1], dtype=np.uint32)
        inv = (self.ninv, self.ninv * self.zi[1] % q)
        m0 = self.ninv * self.R % q
        mont = (m0, m0 * self.zi[1] % q)
        if mode == SMAD:
            c = self._quad(*inv, 32, q >= 8192)
        elif mode == S32:
            c = self._quad(*(inv if op == OP_INV else mont), 32, True)
        elif op == OP_INV or mode == MAD:
            c = self._quad(*inv, bits, False)
        else:
            c = self._quad(*mont, bits, False)
        lazy = self.lazy24 if mode == U24 else self.lazy32
        return tab, modk, np.array(c, dtype=np.uint32), lazy


DRIVER = r'''
import ctypes as C, sys, numpy as np
sys.path.insert(0, sys.argv[1])
from tests import oracle_lib as O
from tests.test_emu_ntt import PlanConsts, OP_FWD, OP_INV, OP_MULNTT, OP_MUL, OP_MULNTT2, SMAD, S32
so, q, n, extreme = sys.argv[2], int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5])
only = [int(x) for x in sys.argv[6].split(',')] if len(sys.argv) > 6 else None
E = C.CDLL(so)
E.emu_ntt.argtypes = [C.c_int] * 5 + [C.c_uint] + [C.c_void_p] * 3 + [C.c_ulonglong, C.c_void_p, C.c_int] + [C.c_void_p] * 5
E.emu_set_wave_serial(1 if 'thread' in so else 0)        # ThreadSanitizer has 256 thread slots: one 64-lane wavefront at a time (tests/emu/wave_emu.h)
P = PlanConsts(q, n)
pw = 1024 // n
w, r = O.tables(q, n)
rng = np.random.default_rng(q * 131 + n)
p = lambda a: a.ctypes.data_as(C.c_void_p) if a is not None else None
canon = lambda a: np.mod(a.astype(np.int64), q).astype(np.int32)

def inputs(batch):
    a = rng.integers(0, q, size=(batch, n)).astype(np.int32)
    b = rng.integers(0, q, size=(batch, n)).astype(np.int32)
    if extreme:
        rows = [np.full(n, q - 1), np.full(n, 2 * q - 1), np.full(n, -(q - 1)), np.tile([q - 1, -(q - 1)], n // 2), np.full(n, 2**31 - 1), np.full(n, -2**31),
                np.tile([2**31 - 1, -2**31], n // 2), np.concatenate([np.full(n // 2, q - 1), np.zeros(n // 2)]), rng.integers(-2**31, 2**31 - 1, size=n)]
        brow = [np.full(n, q - 1), np.full(n, 2**31 - 1), np.full(n, -2**31), np.full(n, -(q - 1))] + [rng.integers(-2**31, 2**31 - 1, size=n) for _ in range(5)]
        for i in range(min(batch, len(rows))):
            a[i] = np.asarray(rows[i]).astype(np.int64).astype(np.int32); b[i] = np.asarray(brow[i]).astype(np.int64).astype(np.int32)
    return a, b

def run(op, mode, stream, blocks, a, b=None, b2=None, bcast=0, batch_dev=None, bound=None):
    tab, modk, c, lazy = P.args(mode, op)
    batch = a.shape[0]
    out = np.full((batch, n), 0x5A5A5A5A, dtype=np.int32); out2 = np.full((batch, n), 0x5A5A5A5A, dtype=np.int32)
    bd = None if batch_dev is None else np.array([batch_dev], dtype=np.uint32)
    rc = E.emu_ntt(P.L, op, mode, int(lazy), stream, blocks, p(tab), p(modk), p(c), batch if bound is None else bound, p(bd), bcast, p(a), p(b), p(b2), p(out), p(out2))
    return rc, out, out2

ran = 0
for mode in P.modes():
    if only and mode not in only:
        continue
    for stream in (0, 1):
        if stream and mode not in (SMAD, S32):
            continue
        # one workgroup (4 wavefronts), two trips of the persistent loop for some wavefronts, a ragged last group in the plain form (the streaming form takes whole groups)
        batch = pw * 6 if stream else pw * 5 + (1 if pw > 1 else 0) + 1
        a, b = inputs(batch)
        f_exp = O.fwd_ntt(a, q, w)
        rc, f, _ = run(OP_FWD, mode, stream, 1, a)
        assert rc == 0 and np.array_equal(f, f_exp), ('fwd', mode, stream)
        rc, back, _ = run(OP_INV, mode, stream, 1, f)
        assert rc == 0 and np.array_equal(back, canon(a)), ('inv', mode, stream)
        if extreme:                                                                # any int32 representative in the NTT domain as well
            rc, i1, _ = run(OP_INV, mode, stream, 1, a)
            assert rc == 0 and np.array_equal(i1, O.inv_ntt(canon(a), q, w, r)), ('inv raw', mode, stream)
        rc, c1, _ = run(OP_MULNTT, mode, stream, 1, a, b)
        assert rc == 0 and np.array_equal(c1, O.polymul_ntt(a, canon(b), q, w, r)), ('mulntt', mode, stream)
        rc, c2, _ = run(OP_MUL, mode, stream, 1, a, b)
        assert rc == 0 and np.array_equal(c2, O.polymul_ntt(a, O.fwd_ntt(b, q, w), q, w, r)), ('mul', mode, stream)
        ran += 5
        if not stream:
            # a broadcast second operand (a public key), the count only the device knows (a bound on the host), and the pair product of the signed modes
            rc, c3, _ = run(OP_MULNTT, mode, 0, 1, a, b[:1].copy(), bcast=1)
            assert rc == 0 and np.array_equal(c3, O.polymul_ntt(a, np.repeat(canon(b[:1]), batch, axis=0), q, w, r)), ('bcast', mode)
            keep = batch - pw - 1
            big = np.concatenate([a, a[:pw]])
            rc, c4, _ = run(OP_FWD, mode, 0, 2, big, batch_dev=keep, bound=big.shape[0])
            assert rc == 0 and np.array_equal(c4[:keep], f_exp[:keep]) and (c4[keep:] == 0x5A5A5A5A).all(), ('batch_dev', mode)
            if mode in (SMAD, S32):
                rc, o1, o2 = run(OP_MULNTT2, mode, 0, 1, a, b[:1].copy(), b[1:2].copy(), bcast=1)
                assert rc == 0 and np.array_equal(o1, O.polymul_ntt(a, np.repeat(canon(b[:1]), batch, axis=0), q, w, r)), ('pair 1', mode)
                assert np.array_equal(o2, O.polymul_ntt(a, np.repeat(canon(b[1:2]), batch, axis=0), q, w, r)), ('pair 2', mode)
                ran += 1
            ran += 2
print('EMU_OK', ran)
'''

CFGS = [(7681, 256), (8380417, 256), (12289, 512), (12289, 1024), (8399873, 512), (15361, 512)]


TSAN_CASES = [(7681, 256, '4'), (8380417, 256, '5'), (12289, 512, '4,3'), (12289, 1024, '5,2')]


@pytest.fixture(scope='module')
def runs():
    asan, tsan = build('emu_ntt', 'address,undefined'), build('emu_ntt', 'thread')
    cases = {('asan', q, n): (DRIVER, [asan, q, n, 1], 'address,undefined') for q, n in CFGS}
    cases.update({('tsan', q, n): (DRIVER, [tsan, q, n, 0, only], 'thread') for q, n, only in TSAN_CASES})
    return drive_all(cases)


def _ok(r):
    assert r.returncode == 0 and 'EMU_OK' in r.stdout, (r.stdout[-2000:], r.stderr[-4000:])


@pytest.mark.parametrize('q,n', CFGS)
def test_every_instantiation_gives_the_oracles_words_asan_ubsan(runs, q, n):
    _ok(runs[('asan', q, n)])


@pytest.mark.parametrize('q,n,only', TSAN_CASES)
def test_lds_exchanges_are_race_free_tsan(runs, q, n, only):
    """the modes the schemes run (signed) plus one unsigned form per size; seeded inputs"""
    _ok(runs[('tsan', q, n)])


@pytest.mark.parametrize('which', [0, 1])
def test_the_harness_reports_a_removed_fence(tmp_path, which):
    """the harness tested: a copy of the kernel headers with ONE of the two fences of ntt_core.h's LDS exchange (its only two `wave_lds_fence();` calls) taken out (the first: lanes read what other lanes
    have not written yet; the second: the next round's writes overtake this round's reads) must come back from ThreadSanitizer as a data race in `exchange`"""
    so = build('emu_ntt', 'thread', mutate=('ntt_core.h', which), out=str(tmp_path))
    r = drive(DRIVER, [so, 8380417, 256, 0, 5], 'thread', timeout=900)
    assert r.returncode != 0 and 'ThreadSanitizer: data race' in r.stderr and 'exchange' in r.stderr, (r.stdout[-500:], r.stderr[-2000:])
