"""CPU: the library's fork / join logic against an adversarial stream order (tests/emu/hip_host_emu.h, EMU_SIDE_STREAMS=late).

The calls with a key per item, the seeded Falcon signer and every *_host entry run part of their work on streams the library creates for itself, forked from and
joined to the caller's stream with events.  On the device a missing join usually goes unnoticed - the side work happens to be done in time.  Here the CPU build of the
library runs with every such stream held back as long as the HIP model allows: work issued to a stream of the library's own runs only when an event wait on the caller's
stream, a stream / event / device synchronisation or hipFree (a device-wide synchronisation) forces it, while the caller's stream runs at once - so a consumer that was
not made to wait for its producer reads what was there before, and the comparison with the oracle fails.  Every scenario also asserts that work WAS held back
(sc_emu_deferred): a run in which nothing was deferred has shown nothing.  Under ASan + UBSan, so that held-back work touching a buffer freed in the meantime is a report.
EMU_SIDE_STREAMS=null-late is the adversary for the other slip - the caller's stream held back, the library's own at once: side work not made to wait for what the
caller queued before (a missing fork) reads its inputs before they are written.  Both adversaries are shown to work on copies of the library with one wait removed.  DESIGN.md 2."""
import os
import re

import pytest

from tests.emu_harness import build_library, drive, sanitizer_env  # noqa: F401

DRIVER = r'''
import sys, os, ctypes as C
root, lib, which = sys.argv[1], sys.argv[2], sys.argv[3].split(',')
sys.path.insert(0, root)
os.environ['SC_EMU_LIB'] = lib
import numpy as np
import tests.conftest
import torch
from libsafecrypto_amd import Engine
from libsafecrypto_amd import engine as E
from tests import oracle_lib as O
from tests.golden_io import load_golden

L = E.load_library()
L.sc_emu_deferred.restype = C.c_long
sync = L.sc_emu_sync                                             # what a caller's hipDeviceSynchronize does: before every read of a result (the null-late adversary holds the caller's stream back)
dev = lambda a, dt=np.int32: torch.from_numpy(np.ascontiguousarray(a, dtype=dt)).cuda()


def sc_host_entries(eng):
    """every *_host entry moves its chunks through a ring of streams of its own"""
    q, n = 12289, 512
    rng = np.random.default_rng(1)
    a = rng.integers(0, q, size=(5, n)).astype(np.int32); b = rng.integers(0, q, size=(5, n)).astype(np.int32)
    w, r = O.tables(q, n)
    p = eng.plan(q, n)
    f = p.fwd_host(a)
    assert np.array_equal(f, O.fwd_ntt(a, q, w)) and np.array_equal(p.inv_host(f), a)
    assert np.array_equal(p.polymul_ntt_host(a, b), O.polymul_ntt(a, b, q, w, r))
    p.close()


def sc_multi(eng):
    """the multi-device driver: two worker threads, each moving its slice through the *_host ring of streams of its own context"""
    from libsafecrypto_amd import Multi
    q, n = 7681, 256
    rng = np.random.default_rng(4)
    a = rng.integers(0, q, size=(7, n)).astype(np.int32); b = rng.integers(0, q, size=(7, n)).astype(np.int32)
    w, r = O.tables(q, n)
    m = Multi([0, 0])
    assert np.array_equal(m.polymul_ntt_host(q, a, b), O.polymul_ntt(a, b, q, w, r))
    f = m.ntt_host(q, a)
    assert np.array_equal(f, O.fwd_ntt(a, q, w)) and np.array_equal(m.ntt_host(q, f, inverse=True), a)
    m.close()


def sc_kyber_keys(eng):
    """sc_hip_kyber_kem_encaps_keys / _decaps_keys: every key's matrix expanded on a stream of the handle's own beside the hashing"""
    G = load_golden('kyber_ref.txt'); tag = 'kyber0'
    par = G[f'{tag}_params']
    k, q, qb = int(par[5]), int(par[1]), int(par[2])
    ky = eng.kyber(par)
    rng = np.random.default_rng(41)
    B = 3
    seeds = dev(rng.integers(0, 256, size=(B, 32), dtype=np.uint8), np.uint8)
    m = rng.integers(0, 256, size=(B, 32), dtype=np.uint8)
    md = dev(m, np.uint8)                                        # (inputs stay referenced until the synchronisation: torch's allocator does that for a caller on the device)
    for _ in range(2):                                           # twice: the second call finds the handle's stream and events already there
        s_ntt, t, t_ntt, rho0, z = ky.keygen(seeds)
        sync()
        # rho reaches the calls through one more kernel on the caller's stream, with no synchronisation behind it: it is still being written when the matrices' stream
        # is forked (the key generation itself ends in a hipFree, which synchronises the device)
        rho_in, zero = rho0.view(torch.int32), torch.zeros(B * 8, dtype=torch.int32, device=rho0.device).view(B, 8)     # (referenced until the synchronisation below)
        rho = eng.poly_add(rho_in, zero).view(torch.uint8)
        ct, keys = ky.encaps_keys(t, t_ntt, rho, md)
        dk = ky.decaps_keys(s_ntt, t, t_ntt, rho, ct)
        sync()
        assert torch.equal(dk, keys)
        sh, th, tnh, rh, cth, kh = (x.cpu().numpy() for x in (s_ntt, t, t_ntt, rho, ct, keys))
        for i in range(B):
            A = O.expand_matrix(rh[i], k * k, q, qb)
            oc, ok = O.kyber_kem_enc(par, th[i].reshape(-1), tnh[i].reshape(-1), rh[i], A, m[i])
            assert cth[i].tobytes() == oc and kh[i].tobytes() == ok, i
    ky.close()


def sc_dilithium_keys(eng):
    """sc_hip_dilithium_verify_keys (three chains on three streams) and _sign_keys (the matrices on a side stream): a key pair per message"""
    G = load_golden('dilithium_ref.txt'); tag = 'dil0'
    par, eb = G[f'{tag}_params'], int(G[f'{tag}_eta_bits'][0])
    d = eng.dilithium(par)
    rng = np.random.default_rng(7)
    B = 2
    seeds = dev(rng.integers(0, 256, size=(B, 32), dtype=np.uint8), np.uint8)
    msgs = dev(rng.integers(0, 256, size=(B, 64), dtype=np.uint8), np.uint8)
    s1, s2, t, rho, K, tr, t1 = d.keygen(seeds, eb)
    sigs, siglens = d.sign_keys(s1, s2, t, rho, K, tr, msgs, fixed_len=64)[:2]
    sync()
    h = lambda x: x.cpu().numpy()
    for i in range(B):
        Ahat = O.expand_matrix(h(rho)[i], int(par[3]) * int(par[4]), int(par[1]), int(par[2]))
        key = O.DilKey(par, h(s1)[i].reshape(-1), h(s2)[i].reshape(-1), h(t)[i].reshape(-1), h(rho)[i], h(K)[i], h(tr)[i], h(t1)[i].reshape(-1), Ahat)
        want, _ = O.dilithium_sign(key, h(msgs)[i].tobytes())
        assert bytes(h(sigs)[i, :int(siglens[i])]) == want, i
    t1f = t1.reshape(B, -1).contiguous()
    for _ in range(2):
        v = d.verify_keys(t1f, rho, msgs, sigs, siglens, fixed_len=64); sync()
        assert v.cpu().tolist() == [0] * B
    bad = sigs.clone(); bad[0, 40] ^= 1
    v = d.verify_keys(t1f, rho, msgs, bad, siglens, fixed_len=64); sync()
    assert v.cpu().tolist() == [1, 0]
    d.close()


def sc_dilithium_verify_keys(eng):
    """sc_hip_dilithium_verify_keys alone (three chains on three streams), on the fixture's own signatures: the key of the fixture once per message"""
    G = load_golden('dilithium_ref.txt'); tag = 'dil0'
    par = G[f'{tag}_params']
    d = eng.dilithium(par)
    B = 3
    msgs = [bytes(G[f'{tag}_msg{m}'].astype(np.uint8)) for m in range(B)]
    sg = [G[f'{tag}_sig{m}'].astype(np.uint8) for m in range(B)]
    stride = max(8, (max(len(m) for m in msgs) + 7) & ~7)
    mb = np.zeros((B, stride), dtype=np.uint8); sb = np.zeros((B, d.sig_stride), dtype=np.uint8)
    for i in range(B):
        mb[i, :len(msgs[i])] = np.frombuffer(msgs[i], dtype=np.uint8); sb[i, :sg[i].size] = sg[i]
    ml = dev(np.array([len(m) for m in msgs], dtype=np.int32)); sl = dev(np.array([x.size for x in sg], dtype=np.int32))
    t1 = dev(np.tile(G[f'{tag}_t1'].astype(np.int32), (B, 1))); rho = dev(np.tile(G[f'{tag}_rho'].astype(np.uint8), (B, 1)), np.uint8)
    mbd, sbd = dev(mb, np.uint8), dev(sb, np.uint8)
    for _ in range(2):
        v = d.verify_keys(t1, rho, mbd, sbd, sl, mlens=ml); sync()
        assert v.cpu().tolist() == [0] * B
    bad = sb.copy(); bad[1, 40] ^= 1
    badd = dev(bad, np.uint8)
    v = d.verify_keys(t1, rho, mbd, badd, sl, mlens=ml); sync()
    assert v.cpu().tolist() == [0, 1, 0]
    d.close()


def sc_falcon_seeded(eng):
    """sc_hip_falcon_sign_seeded: the tape expansion on a stream of the handle's own beside the hash to the ring"""
    G = load_golden('falcon_ref.txt'); tag = 'falcon0'
    key = O.FalconKey.from_golden(G, tag)
    f = eng.falcon(key.par)
    f.set_privkey(key.sk); f.set_pubkey(key.h)
    rng = np.random.default_rng(9)
    B = 2
    seed0 = dev(rng.integers(0, 256, size=(B, 32), dtype=np.uint8), np.uint8)
    msgs = dev(rng.integers(0, 256, size=(B, 64), dtype=np.uint8), np.uint8)
    seeds = eng.drbg_mem(seed0, 48); sync()
    tapes = eng.drbg_words(seeds, 4 * f.n)
    s0, l0, st0 = f.sign(msgs, tapes, fixed_len=64); sync()
    for _ in range(2):
        seeds2 = eng.drbg_mem(seed0, 48)                         # (the seeds are still being written on the caller's stream when the tape expansion's stream is forked)
        s1, l1, st1 = f.sign_seeded(msgs, seeds2, fixed_len=64); sync()
        assert torch.equal(st0, st1) and torch.equal(l0, l1) and torch.equal(s0, s1)
    v = f.verify(msgs, s1, l1, fixed_len=64); sync()
    assert v[st1 == 1].sum() == 0
    f.close()


SCEN = {'host_entries': sc_host_entries, 'multi': sc_multi, 'dilithium_verify_keys': sc_dilithium_verify_keys, 'kyber_keys': sc_kyber_keys, 'dilithium_keys': sc_dilithium_keys, 'falcon_seeded': sc_falcon_seeded}
for name in which:
    eng = Engine(0)
    before = L.sc_emu_deferred()
    SCEN[name](eng)
    eng.close()
    held = L.sc_emu_deferred() - before
    print('STREAM ORDER', name, held, flush=True)
    assert held > 0 or os.environ.get('EMU_SIDE_STREAMS') not in ('late', 'null-late'), (name, 'nothing was held back: the scenario does not reach a stream of the library')
print('STREAM ORDER DONE')
'''

GROUPS = [['host_entries', 'multi', 'kyber_keys', 'falcon_seeded'], ['dilithium_verify_keys']]
SLOW_GROUPS = [['dilithium_keys']]                        # key generation, signing and verification with a key per message: ten minutes under ASan (SC_TEST_SLOW_EMU=1; run in round 6: passes)


def run(groups, late, sanitize='address,undefined', timeout=2400):
    """late: True / 'late' - the library's own streams held back; 'null-late' - the caller's stream held back; False - streams in issue order"""
    from concurrent.futures import ThreadPoolExecutor
    import subprocess
    so = build_library(sanitize)
    env = sanitizer_env(sanitize)
    if late:
        env['EMU_SIDE_STREAMS'] = late if isinstance(late, str) else 'late'
    from tests.emu_harness import ROOT

    def one(g):
        return subprocess.run(['python3', '-c', DRIVER, ROOT, so, ','.join(g)], env=env, capture_output=True, text=True, timeout=timeout)
    with ThreadPoolExecutor(max_workers=2) as ex:
        runs = list(ex.map(one, groups))
    held = {}
    for g, r in zip(groups, runs):
        assert r.returncode == 0 and 'STREAM ORDER DONE' in r.stdout, (g, r.stdout[-2000:], r.stderr[-4000:])
        held.update({m.group(1): int(m.group(2)) for m in re.finditer(r'^STREAM ORDER (\w+) (\d+)$', r.stdout, re.M)})
    return held


def test_results_do_not_depend_on_how_late_the_library_s_own_streams_run():
    held = run(GROUPS, late=True)
    assert set(held) == {n for g in GROUPS for n in g} and all(v > 0 for v in held.values()), held


def test_results_do_not_depend_on_how_late_the_caller_s_stream_runs():
    """the other adversary: the caller's stream held back, the library's own streams at once - side work that is not made to wait for the caller's earlier work (a missing
    fork) reads its inputs before they are written"""
    groups = [[n for n in g if n not in ('host_entries', 'multi')] for g in GROUPS]          # (the *_host entries issue nothing to the caller's stream: there is nothing of theirs to hold back)
    held = run(groups, late='null-late')
    assert set(held) == {n for g in groups for n in g} and all(v > 0 for v in held.values()), held


@pytest.mark.skipif(os.environ.get('SC_TEST_SLOW_EMU') != '1', reason='Dilithium key generation + signing + verification with a key per message under the adversary: ten minutes (SC_TEST_SLOW_EMU=1)')
@pytest.mark.parametrize('mode', ['late', 'null-late'])
def test_the_same_through_dilithium_signing_with_a_key_per_message(mode):
    held = run(SLOW_GROUPS, late=mode, timeout=5400)
    assert all(v > 0 for v in held.values()), held


MUTANTS = {'kyber': (('kyber.hip', 'if (K.ready) HIPCHK(hipStreamWaitEvent(s, K.ready, 0));', '', 1), 'kyber_keys', 'late'),
           'falcon': (('falcon.hip', 'if (tapes_ready) HIPCHK(hipStreamWaitEvent(s, tapes_ready, 0));', '', 1), 'falcon_seeded', 'late'),
           'kyber_fork': (('kyber.hip', 'HIPCHK(hipStreamWaitEvent(h->side, h->ev_fork, 0));', '', 1), 'kyber_keys', 'null-late'),
           'falcon_fork': (('falcon.hip', 'HIPCHK(hipStreamWaitEvent(f->side, f->ev_fork, 0));', '', 1), 'falcon_seeded', 'null-late')}


_slow = pytest.mark.skipif(os.environ.get('SC_TEST_SLOW_EMU') != '1', reason='further mutants: SC_TEST_SLOW_EMU=1 (run in round 6: each noticed)')


@pytest.mark.parametrize('what', [pytest.param('kyber'), pytest.param('kyber_fork'), pytest.param('falcon', marks=_slow), pytest.param('falcon_fork', marks=_slow)])
def test_the_adversary_notices_a_removed_join(what):
    """the harness itself: a copy of the library with ONE wait taken out - a JOIN (the caller's stream no longer waits for the matrices of Kyber's keys / the tapes of the
    seeded Falcon signer being written on the handle's own stream: noticed when that stream runs late) or a FORK (the handle's stream no longer waits for what the caller
    queued before - the keys' rho, the seeds - : noticed when the caller's stream runs late) - gives wrong results under the matching adversary, and passes when streams
    run in issue order (as it would, most of the time, on the device)"""
    import subprocess
    from tests.emu_harness import LIB_REWRITES, ROOT
    rw, scen, mode = MUTANTS[what]
    so = build_library(None, rewrites=LIB_REWRITES + (rw,))
    res = {}
    for late in (mode, None):
        env = sanitizer_env(None)
        if late:
            env['EMU_SIDE_STREAMS'] = late
        res[late] = subprocess.run(['python3', '-c', DRIVER, ROOT, so, scen], env=env, capture_output=True, text=True, timeout=1200)
    assert res[None].returncode == 0 and 'STREAM ORDER DONE' in res[None].stdout, res[None].stderr[-2000:]
    assert res[mode].returncode != 0 and 'AssertionError' in res[mode].stderr, (res[mode].stdout[-500:], res[mode].stderr[-1500:])
