"""CPU: the product library's HOST side under ThreadSanitizer - the in-library multi-device driver (csrc/multi.cpp: one host thread per worker) and handles shared
between host threads (a plan's sc_hip_matvec scratch: mutex + event), on the library built for the CPU (tests/emu_harness.py build_library('thread'),
tests/emu/hip_host_emu.h).  The kernels' own intra-wavefront ordering is what tests/test_emu_*_sign.py / test_emu_ntt.py check under TSan; this file is about
the host code around them, which the device cannot be asked about either (no sanitizer on the GPU pool).

Round 6 found with it: the lazily filled `static int per_cu` of the NTT and RLWE launch helpers, the CTR-DRBG constants' `ready` flag and the context's CU
count were plain variables written by whichever worker thread came first (the same values, so harmless in practice on x86 - and undefined behaviour as C++);
they are atomics / a thread-safe static now (sc_hip.hip launch_ntt_t, rlwe.hip rl_launch, drbg.hip drbg_const)."""
import os
import re

import pytest

from tests.emu_harness import build_library, drive

DRIVER = r'''
import sys, os, threading
root, lib, what = sys.argv[1], sys.argv[2], sys.argv[3]
sys.path.insert(0, root)
os.environ['SC_EMU_LIB'] = lib
import numpy as np
import tests.conftest                                   # emulation mode: the binding loads the CPU build, torch's "cuda" is the CPU
import torch
from libsafecrypto_amd import Engine, Multi
from tests import oracle_lib as O
from tests.golden_io import load_golden

q, n = 7681, 256
rng = np.random.default_rng(1)
w, r = O.tables(q, n)
if what == 'ntt':
    # the multi-device driver: two worker threads, each with its own context and plan on "device 0", slices of 3 + 2 polynomials
    a = rng.integers(0, q, size=(5, n)).astype(np.int32); bhat = rng.integers(0, q, size=(5, n)).astype(np.int32)
    m = Multi([0, 0])
    assert np.array_equal(m.polymul_ntt_host(q, a, bhat), O.polymul_ntt(a, bhat, q, w, r))
    f = m.ntt_host(q, a)
    assert np.array_equal(f, O.fwd_ntt(a, q, w)) and np.array_equal(m.ntt_host(q, f, inverse=True), a)
    m.close()
    # one context and one plan shared by two host threads: sc_hip_matvec keeps its scratch in the plan
    eng = Engine(0)
    p = eng.plan(q, n)
    A = torch.from_numpy(rng.integers(0, q, size=(2, 2, n)).astype(np.int32)).cuda()
    ys = [torch.from_numpy(rng.integers(0, q, size=(b, 2, n)).astype(np.int32)).cuda() for b in (3, 5)]
    want = [p.matvec(A, y, 2, 2).cpu().numpy() for y in ys]
    out = [None, None]
    def work(i):
        for _ in range(2):
            out[i] = p.matvec(A, ys[i], 2, 2).cpu().numpy()
            p.fwd(ys[i].reshape(-1, n))
            eng.drbg_words(torch.zeros((2, 32), dtype=torch.uint8).cuda(), 16)
    ts = [threading.Thread(target=work, args=(i,)) for i in range(2)]
    [t.start() for t in ts]; [t.join() for t in ts]
    assert all(np.array_equal(o, w_) for o, w_ in zip(out, want))
    p.close(); eng.close()
elif what == 'vtable':
    # the reference-shaped function-pointer tables from two host threads at once (a libsafecrypto context per thread, as the reference asks): the slots of one device are
    # serialised by the binding's own lock; the error counter and the per-device state are shared
    import tests.test_gpu_vtable as V
    errs = []
    def work(args):
        try:
            for _ in range(2):
                V.test_vtable_calls(*args)
        except BaseException as e:
            errs.append(repr(e))
    ts = [threading.Thread(target=work, args=(a,)) for a in ((12289, 512, True), (8380417, 256, False))]
    [t.start() for t in ts]; [t.join() for t in ts]
    assert not errs, errs
else:
    # the multi-device Dilithium driver (set 0): 4 messages over 2 workers, signatures = the oracle's, then verified by the same driver
    G = load_golden('dilithium_ref.txt')
    k = {x: G[f'dil0_{x}'] for x in ('params', 's1', 's2', 't', 'K', 'tr', 'Ahat', 't1', 'rho')}
    msgs = rng.integers(0, 256, size=(4, 64), dtype=np.uint8)
    m = Multi([0, 0])
    m.dilithium(k['params'], k['s1'], k['s2'], k['t'], k['K'], k['tr'], k['Ahat'], k['t1'], k['rho'])
    sigs, sl, tr = m.dilithium_sign_host(msgs, 64)
    key = O.DilKey.from_golden(G, 'dil0')
    for i in range(4):
        assert bytes(sigs[i, :sl[i]]) == O.dilithium_sign(key, bytes(msgs[i]))[0], i
    assert (m.dilithium_verify_host(msgs, 64, sigs, sl) == 0).all()
    m.close()
print('HOST SIDE OK')
'''


def _run(what, timeout):
    so = build_library('thread')
    r = drive(DRIVER.replace('sys.argv[1], sys.argv[2], sys.argv[3]', 'sys.argv[1], sys.argv[2], sys.argv[3]'), [so, what], 'thread', timeout=timeout)
    assert r.returncode == 0 and 'HOST SIDE OK' in r.stdout and not re.search('ThreadSanitizer', r.stderr), (r.stdout[-1500:], r.stderr[-4000:])


def test_multi_driver_and_shared_plan_are_race_free():
    _run('ntt', 900)


def test_vtable_slots_from_two_threads_are_race_free():
    _run('vtable', 900)


@pytest.mark.skipif(os.environ.get('SC_TEST_SLOW_EMU') != '1', reason='about ten minutes of emulation under TSan: SC_TEST_SLOW_EMU=1 (run in round 6: clean)')
def test_multi_dilithium_driver_is_race_free():
    _run('dil', 3000)
