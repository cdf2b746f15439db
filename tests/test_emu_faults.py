"""CPU: every device allocation, copy, and stream / event creation the product makes - FAILED on purpose, one at a time - on the library built for the CPU
(tests/emu_harness.py build_library, ASan + UBSan; fault hooks in tests/emu/hip_host_emu.h: sc_emu_fault / sc_emu_calls / sc_emu_live_allocs).

A scenario is a complete use of one handle type through the product's own binding: create, load keys, one small operation checked against the fixture, destroy.
It runs once to count the calls of a class it makes (N), then N more times with call 0, 1, ... N-1 of that class failing.  Every faulted run must
  * come back as an error of the C ABI (EngineError with a message) or - where the library tolerates the failure, e.g. an explicit pin that falls back to the
    staged copy - as the correct result, never as a wrong result, a crash or a sanitizer report;
  * leave NOTHING allocated once the scenario's handles are destroyed (the shim counts live device / host allocations, streams and events), and no secret table
    half-initialised behind a non-null pointer (a later call then runs on garbage) - which shows as a wrong result in the clean run that follows;
  * leave the context usable: the same scenario, run again without a fault, gives the fixture's bytes.
This is the "a failing call leaves the handle usable" rule of tests/test_gpu_errors.py taken to the failures a test on the device cannot provoke."""
import json
import os
import re

import pytest

from tests.emu_harness import build_library, drive

DRIVER = r'''
import sys, os, json, ctypes as C
root, lib, which, budget, classes = sys.argv[1], sys.argv[2], sys.argv[3].split(','), float(sys.argv[4]), [int(c) for c in sys.argv[5].split(',')]
import time
sys.path.insert(0, root)
os.environ['SC_EMU_LIB'] = lib
import numpy as np
import tests.conftest                                   # emulation mode
import torch
from libsafecrypto_amd import Engine, EngineError, Multi
from libsafecrypto_amd import engine as E
from tests import oracle_lib as O
from tests.golden_io import load_golden

L = E.load_library()
L.sc_emu_calls.restype = C.c_long; L.sc_emu_live_allocs.restype = C.c_long
L.sc_emu_fault.argtypes = [C.c_int, C.c_long]

def dev(a, dt=np.int32):
    return torch.from_numpy(np.ascontiguousarray(a, dtype=dt)).cuda()

def u32(a):
    return (np.asarray(a).astype(np.int64) & 0xFFFFFFFF).astype(np.uint32)

def rows(items):
    stride = max(8, (max(len(m) for m in items) + 7) & ~7)
    b = np.zeros((len(items), stride), dtype=np.uint8)
    for i, m in enumerate(items):
        b[i, :len(m)] = np.frombuffer(bytes(m), dtype=np.uint8)
    return dev(b, np.uint8), dev(np.array([len(m) for m in items], dtype=np.int32))

# ---- scenarios: each creates its own context (so that "nothing stays allocated" includes the context) and closes what it opened on every path ----
def sc_ntt():
    q, n = 7681, 256
    rng = np.random.default_rng(3)
    a = rng.integers(0, q, size=(3, n)).astype(np.int32); b = rng.integers(0, q, size=(3, n)).astype(np.int32)
    w, r = O.tables(q, n)
    eng = Engine(0)
    try:
        p = eng.plan(q, n)
        try:
            assert np.array_equal(p.polymul_ntt(dev(a), dev(b)).cpu().numpy(), O.polymul_ntt(a, b, q, w, r))
            A = rng.integers(0, q, size=(2, 2, n)).astype(np.int32); y = rng.integers(0, q, size=(2, 2, n)).astype(np.int32)
            t = p.matvec(dev(A), dev(y), 2, 2).cpu().numpy()
            yh = O.fwd_ntt(y.reshape(-1, n), q, w).reshape(2, 2, n).astype(np.int64)
            for i in range(2):
                acc = sum(yh[:, j] * A[i, j] for j in range(2)) % q
                assert np.array_equal(t[:, i], O.inv_ntt(acc.astype(np.int32), q, w, r))
            host = p.fwd_host(a) if hasattr(p, 'fwd_host') else None
            if host is not None:
                assert np.array_equal(host, O.fwd_ntt(a, q, w))
        finally:
            p.close()
        s = eng.sampler(13.42, 215.0)
        try:
            u = np.random.default_rng(1).integers(0, 2**63, size=64, dtype=np.uint64)
            assert np.array_equal(s.sample(dev(u.view(np.int64), np.int64)).cpu().numpy(), O.cdt_sample(u, O.cdt_table(13.42, 215.0)))
        finally:
            s.close()
        seeds = np.arange(64, dtype=np.uint8).reshape(2, 32)
        got = eng.expand_matrix_xof(dev(seeds, np.uint8), 1, q, 13).cpu().numpy()
        assert np.array_equal(got[1].reshape(-1), O.ring_q_xof(128, seeds[1], 0, 1, q, 13)[0])
        got = eng.expand_matrix(dev(seeds, np.uint8), 1, q, 13).cpu().numpy()
        assert np.array_equal(got[0].reshape(-1), O.expand_matrix(seeds[0], 1, q, 13))
    finally:
        eng.close()

def sc_dilithium(tuning=0):
    G = load_golden('dilithium_ref.txt')
    key = O.DilKey.from_golden(G, 'dil0')
    msg = bytes(G['dil0_msg0'].astype(np.uint8)); want = bytes(G['dil0_sig0'].astype(np.uint8))
    eng = Engine(0)
    try:
        d = eng.dilithium(key.par)
        try:
            if tuning:
                d.set_tuning(tuning)
            d.set_privkey(key.s1, key.s2, key.t, key.K, key.tr, key.Ahat)
            d.set_pubkey(key.t1, key.rho, key.Ahat)
            sigs, trials = d.sign_host([msg])
            assert sigs[0] == want, 'WRONG SIGNATURE'
            assert d.verify_host([msg], sigs) == [0]
        finally:
            d.close()
    finally:
        eng.close()

def sc_dilithium_tables():
    sc_dilithium(1 << 11)

def sc_dilithium_load(sign=False):
    """creation and key loading only - both key halves, the opt-in finishing tables included (tuning bit 11: built and uploaded inside set_privkey)"""
    G = load_golden('dilithium_ref.txt')
    key = O.DilKey.from_golden(G, 'dil0')
    eng = Engine(0)
    try:
        d = eng.dilithium(key.par)
        try:
            d.set_tuning(1 << 11)
            d.set_privkey(key.s1, key.s2, key.t, key.K, key.tr, key.Ahat)
            d.set_pubkey(key.t1, key.rho, key.Ahat)
            d.set_privkey(key.s1, key.s2, key.t, key.K, key.tr, key.Ahat)          # a second load over the first: the old tables go, new ones come
        finally:
            d.close()
    finally:
        eng.close()

def sc_bliss_tables():
    """the sampler-table setters of a BLISS-B handle: the reference's 64-bit table, then a 128-bit one and a 192-bit one, then back (a setter that fails keeps the table it had)"""
    G = load_golden('bliss_hp_ref.txt'); C64 = load_golden('cdt_ref.txt')
    key = O.BlissKey.from_golden(G, 'blisshp128_1')
    t128 = np.asarray(G['blisshp128_1_table'], dtype=np.uint64); t192 = np.asarray(G['blisshp192_1_table'], dtype=np.uint64)
    t64 = np.asarray(C64['bliss1_s215_t13.42_cdf'], dtype=np.uint64)
    eng = Engine(0)
    try:
        b = eng.bliss(key.par)
        try:
            b.set_key(key.f, key.g, key.a)
            b.set_cdt_table(t64)
            b.set_cdt_table_hp(128, t128)
            b.set_cdt_table_hp(192, t192)
            b.set_cdt_table(t64)
        finally:
            b.close()
        s = eng.sampler_hp(128, t128.reshape(-1, 2))
        try:
            x = np.asarray(load_golden('cdt_hp_ref.txt')['hp128_bliss1_s215_t13.42_words'], dtype=np.uint64)[:64].reshape(32, 2)
            assert np.array_equal(s.sample(dev(x.view(np.int64), np.int64)).cpu().numpy(), O.cdt_hp_sample(x, t128.reshape(-1, 2), 2)), 'WRONG SAMPLES'
        finally:
            s.close()
    finally:
        eng.close()

def sc_tesla_load():
    G = load_golden('tesla_ref.txt')
    key = O.TeslaKey.from_golden(G, 'tesla0')
    eng = Engine(0)
    try:
        t = eng.tesla(key.par)
        try:
            t.set_key(key.a1, key.a2, key.s, key.e1, key.e2, key.t1, key.t2)
            t.set_key(key.a1, key.a2, key.s, key.e1, key.e2, key.t1, key.t2)
        finally:
            t.close()
    finally:
        eng.close()

def sc_kyber():
    G = load_golden('kyber_ref.txt'); tag = 'kyber0'
    eng = Engine(0)
    try:
        ky = eng.kyber(G[f'{tag}_params'])
        try:
            ky.set_pubkey(G[f'{tag}_t_ntt'], G[f'{tag}_Agen'])
            ky.set_privkey(G[f'{tag}_s_ntt'])
            m = G[f'{tag}_m0'].astype(np.uint8)[None]; noise = G[f'{tag}_noise0'].astype(np.uint8)[None]
            u, v = ky.enc(dev(m, np.uint8), dev(noise, np.uint8))
            assert np.array_equal(u[0].cpu().numpy().reshape(-1), G[f'{tag}_u0']) and np.array_equal(v[0].cpu().numpy(), G[f'{tag}_v0']), 'WRONG CIPHERTEXT'
            assert np.array_equal(ky.dec(u, v).cpu().numpy(), m)
        finally:
            ky.close()
    finally:
        eng.close()

def sc_bliss():
    G = load_golden('bliss_ref.txt'); tag = 'bliss1'
    key = O.BlissKey.from_golden(G, tag)
    msg = bytes(G[f'{tag}_msg0'].astype(np.uint8)); tape = u32(G[f'{tag}_tape0']); want = bytes(G[f'{tag}_sig0'].astype(np.uint8))
    eng = Engine(0)
    try:
        b = eng.bliss(key.par)
        try:
            b.set_key(key.f, key.g, key.a)
            mb, ml = rows([msg])
            tb = np.zeros((1, 8192), dtype=np.uint32); tb[0, :min(len(tape), 8192)] = tape[:8192]
            out = b.sign(mb, dev(tb.view(np.int32)), mlens=ml, tape_lens=dev(np.array([min(len(tape), 8192)], dtype=np.int32)))
            sigs, siglens = out[0], out[1]
            assert bytes(sigs[0, :int(siglens[0])].cpu().numpy()) == want, 'WRONG SIGNATURE'
            assert b.verify(mb, sigs, siglens, mlens=ml).cpu().tolist() == [0]
        finally:
            b.close()
    finally:
        eng.close()

def sc_tesla():
    G = load_golden('tesla_ref.txt'); tag = 'tesla0'
    key = O.TeslaKey.from_golden(G, tag)
    msg = bytes(np.asarray(G[f'{tag}_msg0'], dtype=np.uint8)); tape = u32(G[f'{tag}_tape0']); want = bytes(np.asarray(G[f'{tag}_sig0'], dtype=np.uint8))
    eng = Engine(0)
    try:
        t = eng.tesla(key.par)
        try:
            t.set_key(key.a1, key.a2, key.s, key.e1, key.e2, key.t1, key.t2)
            mb, ml = rows([msg])
            tb = np.zeros((1, 8192), dtype=np.uint32); k = min(len(tape), 8192); tb[0, :k] = tape[:k]
            sigs, siglens, trials, status = t.sign(mb, dev(tb.view(np.int32)), mlens=ml, tape_lens=dev(np.array([k], dtype=np.int32)))
            assert bytes(sigs[0, :int(siglens[0])].cpu().numpy()) == want, 'WRONG SIGNATURE'
            assert t.verify(mb, sigs, siglens, mlens=ml).cpu().tolist() == [0]
        finally:
            t.close()
    finally:
        eng.close()

def sc_falcon():
    G = load_golden('falcon_ref.txt'); tag = 'falcon0'
    key = O.FalconKey.from_golden(G, tag)
    msg = bytes(np.asarray(G[f'{tag}_msg0'], dtype=np.uint8)); want = bytes(np.asarray(G[f'{tag}_sig0'], dtype=np.uint8))
    eng = Engine(0)
    try:
        f = eng.falcon(key.par)
        try:
            f.set_privkey(key.sk)
            f.set_pubkey(key.h)
            mb, ml = rows([msg])
            tp = u32(G[f'{tag}_tape0'])[None]
            sigs, siglens, status = f.sign(mb, dev(tp.view(np.int32)), mlens=ml)
            assert bytes(sigs[0, :int(siglens[0])].cpu().numpy()) == want, 'WRONG SIGNATURE'
            assert f.verify(mb, sigs, siglens, mlens=ml).cpu().tolist() == [0]
        finally:
            f.close()
    finally:
        eng.close()

def sc_multi():
    q, n = 7681, 256
    rng = np.random.default_rng(4)
    a = rng.integers(0, q, size=(5, n)).astype(np.int32); b = rng.integers(0, q, size=(5, n)).astype(np.int32)
    w, r = O.tables(q, n)
    m = Multi([0, 0])
    try:
        assert np.array_equal(m.polymul_ntt_host(q, a, b), O.polymul_ntt(a, b, q, w, r)), 'WRONG PRODUCT'
    finally:
        m.close()

SCEN = {'ntt': sc_ntt, 'bliss_tables': sc_bliss_tables, 'dilithium_load': sc_dilithium_load, 'tesla_load': sc_tesla_load, 'dilithium': sc_dilithium, 'dilithium_tables': sc_dilithium_tables, 'kyber': sc_kyber, 'bliss': sc_bliss, 'tesla': sc_tesla, 'falcon': sc_falcon, 'multi': sc_multi}
CLASSES = {0: 'alloc', 1: 'copy', 2: 'create', 3: 'pin'}
report = {}
for name in which:
    fn = SCEN[name]
    base = L.sc_emu_live_allocs()
    before = [L.sc_emu_calls(c) for c in range(4)]
    fn()                                                       # clean run: counts the calls, checks the fixture
    counts = [L.sc_emu_calls(c) - before[c] for c in range(4)]
    assert L.sc_emu_live_allocs() == base, (name, 'clean run leaks', L.sc_emu_live_allocs() - base)
    rep = {'calls': dict(zip(CLASSES.values(), counts)), 'errors': 0, 'tolerated': 0, 'bad': [], 'points': 0, 'classes': [CLASSES[c] for c in classes]}
    for cls, n in enumerate(counts):
        if cls not in classes:
            continue
        # every call of the class in order while the class's time budget lasts (a faulted run costs what the scenario does up to the fault: the early calls - creation,
        # key loading, workspace sizing - are the cheap ones and the many); past the budget, four more points spread over the rest, the last call always among them
        order, t0 = list(range(n)), time.time()
        k = 0
        while k < len(order):
            i = order[k]; k += 1
            if budget > 0 and time.time() - t0 > budget and k < len(order) - 4:
                rest = order[k:]
                order = order[:k] + [rest[(len(rest) - 1) * j // 3] for j in range(4)]
                budget_hit = True
            rep['points'] += 1
            L.sc_emu_fault(cls, i)
            try:
                fn()
                rep['tolerated'] += 1                          # the library went on without that call and the result was still the fixture's
            except EngineError as e:
                rep['errors'] += 1
                if not str(e).strip():
                    rep['bad'].append([CLASSES[cls], i, 'error without a message'])
            except AssertionError as e:
                rep['bad'].append([CLASSES[cls], i, 'WRONG RESULT after a failed call: ' + str(e)[:200]])
            except Exception as e:
                rep['bad'].append([CLASSES[cls], i, type(e).__name__ + ': ' + str(e)[:200]])
            finally:
                for c in range(4):
                    L.sc_emu_fault(c, -1)
            if L.sc_emu_live_allocs() != base:
                rep['bad'].append([CLASSES[cls], i, 'left %d allocation(s) behind' % (L.sc_emu_live_allocs() - base)])
                base = L.sc_emu_live_allocs()
    fn()                                                       # and the library still works
    report[name] = rep
    print('FAULTS', name, json.dumps(rep), flush=True)
print('FAULT SWEEP DONE')
'''

# the default suite: every call of every class for the scenarios whose faulted runs cost seconds; the signing calls of Dilithium and ring-TESLA cost 10 s and more
# per run in the emulation - creation and key loading (where the tables of secrets are built) are swept here, the whole calls with SC_TEST_SLOW_EMU=1 (run in round 6
# on the plain build: dilithium 66 + dilithium_tables 68 + tesla 46 faulted runs, every one a clean error, nothing left allocated)
GROUPS = [['ntt', 'multi', 'dilithium_load', 'bliss_tables'], ['bliss', 'falcon', 'tesla_load', 'kyber']]
SLOW_GROUPS = [['dilithium'], ['dilithium_tables'], ['tesla']]
SLOW = os.environ.get('SC_TEST_SLOW_EMU') == '1'


def sweep(groups, sanitize='address,undefined', budget=0.0, timeout=3000, classes='0,1,2,3'):
    from concurrent.futures import ThreadPoolExecutor
    so = build_library(sanitize)
    with ThreadPoolExecutor(max_workers=2) as ex:              # (children are hundreds of threads each: more of them side by side spend their time in the kernel's futex queues)
        runs = list(ex.map(lambda g: drive(DRIVER, [so, ','.join(g), budget, classes], sanitize, timeout=timeout), groups))
    reps = {}
    for g, r in zip(groups, runs):
        assert r.returncode == 0 and 'FAULT SWEEP DONE' in r.stdout, (g, r.stdout[-2000:], r.stderr[-4000:])
        reps.update({m.group(1): json.loads(m.group(2)) for m in re.finditer(r'^FAULTS (\w+) (\{.*\})$', r.stdout, re.M)})
    return reps


def check(reps, groups):
    assert set(reps) == {n for g in groups for n in g}
    for name, rep in reps.items():
        assert rep['bad'] == [], (name, rep)
        assert rep['calls']['alloc'] > 0 and rep['errors'] > 0 and rep['points'] == sum(n for c, n in rep['calls'].items() if c in rep['classes']), (name, rep)


def test_every_failed_allocation_and_creation_is_a_clean_error_asan_ubsan():
    """allocations, stream / event creation, pinning: where the frees of an error path are - under AddressSanitizer, so that a use after free or a double free is a report"""
    check(sweep(GROUPS, timeout=900, classes='0,2,3'), GROUPS)


def test_every_failed_copy_is_a_clean_error():
    """copies and memsets (the many): on the plain build - a failed copy frees nothing by itself; what it must not leave behind is a half-filled table that a later call
    trusts, which shows as a wrong result in the clean run that closes every scenario"""
    check(sweep(GROUPS, sanitize=None, timeout=900, classes='1'), GROUPS)


@pytest.mark.skipif(not SLOW, reason='whole signing calls of Dilithium and ring-TESLA with every call failed in turn: about an hour of emulation (SC_TEST_SLOW_EMU=1)')
def test_the_same_through_whole_signing_calls():
    check(sweep(SLOW_GROUPS, timeout=5400), SLOW_GROUPS)
    check(sweep(GROUPS, timeout=2400), GROUPS)                  # and the default scenarios' copies under ASan as well
