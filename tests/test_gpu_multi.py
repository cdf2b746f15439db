"""GPU: the N > 1 path of the REAL engine on a one-GPU box.
(1) the in-library multi-device driver (sc_hip_multi_*: one host thread per worker, contiguous slices, results written at the slice's offset)
    with two and three workers on device 0 against one-device results;
(2) two torch.distributed ranks of the real engine: tests/test_gpu_00_two_ranks.py (it has to run before this process opens the GPU)."""
import numpy as np
import pytest

from tests.golden_io import load_golden
from tests import oracle_lib as O
from libsafecrypto_amd.shard import shard_range

pytestmark = pytest.mark.gpu


def _dil_key(G, tag='dil3'):
    return {x: G[f'{tag}_{x}'] for x in ('params', 's1', 's2', 't', 'K', 'tr', 'Ahat', 't1', 'rho')}


@pytest.mark.parametrize('workers', [2, 3])
def test_multi_driver_matches_one_device(workers):
    from libsafecrypto_amd import Engine, Multi
    G = load_golden('dilithium_ref.txt')
    k = _dil_key(G)
    q, n, batch = 7681, 256, 1001                                   # not a multiple of the worker count
    rng = np.random.default_rng(workers)
    a = rng.integers(0, q, size=(batch, n)).astype(np.int32)
    bhat = rng.integers(0, q, size=(batch, n)).astype(np.int32)
    w, r = O.tables(q, n)
    m = Multi([0] * workers)
    assert m.workers == workers
    got = m.polymul_ntt_host(q, a, bhat)
    assert np.array_equal(got, O.polymul_ntt(a, bhat, q, w, r))
    f = m.ntt_host(q, a)
    assert np.array_equal(f, O.fwd_ntt(a, q, w)) and np.array_equal(m.ntt_host(q, f, inverse=True), a)
    # Dilithium-III: 301 messages over the workers == one handle on one device
    msgs = rng.integers(0, 256, size=(301, 64), dtype=np.uint8)
    m.dilithium(k['params'], k['s1'], k['s2'], k['t'], k['K'], k['tr'], k['Ahat'], k['t1'], k['rho'])
    sigs, sl, tr = m.dilithium_sign_host(msgs, 64)
    eng = Engine(0)
    d = eng.dilithium(k['params'])
    d.set_privkey(k['s1'], k['s2'], k['t'], k['K'], k['tr'], k['Ahat']); d.set_pubkey(k['t1'], k['rho'], k['Ahat'])
    one, one_tr = d.sign_host([bytes(x) for x in msgs])
    for i in range(301):
        assert bytes(sigs[i, :sl[i]]) == one[i] and int(tr[i]) == one_tr[i], i
    assert (m.dilithium_verify_host(msgs, 64, sigs, sl) == 0).all()
    bad = sigs.copy(); bad[::3, 100] ^= 1
    v = m.dilithium_verify_host(msgs, 64, bad, sl)
    assert (v[::3] == 1).all() and (np.delete(v, np.s_[::3]) == 0).all()
    d.close(); eng.close(); m.close()


@pytest.mark.parametrize('workers', [1, 3])
def test_multi_bliss_falcon_tesla_kyber_match_one_device(workers):
    """BASELINE configs[3] / [4] in one process (sc_hip_multi_bliss_* / _falcon_* / _tesla_* / _kyber_*): a host batch cut over the workers gives the
    bytes the one-device engine gives for the whole batch - which the other GPU tests pin to the reference.  One worker = the one-device *_host entries."""
    import torch
    from libsafecrypto_amd import Engine, Multi
    rng = np.random.default_rng(40 + workers)
    eng = Engine(0)
    m = Multi([0] * workers)
    u32 = lambda a: (np.asarray(a).astype(np.int64) & 0xFFFFFFFF).astype(np.uint32)
    dev = lambda a: torch.from_numpy(a).cuda()

    # ---- BLISS-B-I: 205 messages, tapes of three fresh attempts
    G = load_golden('bliss_ref.txt')
    B, W = 205, 3 * 2048 + 64
    msgs = rng.integers(0, 256, size=(B, 64), dtype=np.uint8)
    tapes = rng.integers(0, 2**32, size=(B, W), dtype=np.uint64).astype(np.uint32)
    b1 = eng.bliss(G['bliss1_params']); b1.set_key(G['bliss1_f'], G['bliss1_g'], G['bliss1_a'])
    s1, l1, t1, st1 = (x.cpu().numpy() for x in b1.sign(dev(msgs), dev(tapes.view(np.int32)), fixed_len=64))
    mb = m.scheme('bliss', G['bliss1_params']); mb.set_key(G['bliss1_f'], G['bliss1_g'], G['bliss1_a'])
    s, l, t, st = mb.sign_host(msgs, 64, tapes)
    assert np.array_equal(st, st1.astype(np.uint32)) and np.array_equal(l, l1.astype(np.uint32)) and np.array_equal(t, t1.astype(np.uint32))
    assert np.array_equal(s[:, :mb.sig_bytes], s1[:, :mb.sig_bytes]) and int((st == 1).sum()) > B - 3
    v = mb.verify_host(msgs, 64, s, l)
    assert (v[st == 1] == 0).all()
    bad = s.copy(); bad[::4, 300] ^= 2
    assert (mb.verify_host(msgs, 64, bad, l)[::4] == 1).all()
    b1.close()

    # ---- Falcon-512: 130 messages
    G = load_golden('falcon_ref.txt')
    par = G['falcon0_params']; n = int(par[0])
    sk = np.asarray(G['falcon0_sk'], dtype=np.int64).view(np.float64)
    B = 130
    msgs = rng.integers(0, 256, size=(B, 64), dtype=np.uint8)
    tapes = rng.integers(0, 2**32, size=(B, 4 * n), dtype=np.uint64).astype(np.uint32)
    f1 = eng.falcon(par); f1.set_privkey(sk); f1.set_pubkey(G['falcon0_h'])
    s1, l1, st1 = (x.cpu().numpy() for x in f1.sign(dev(msgs), dev(tapes.view(np.int32)), fixed_len=64))
    mf = m.scheme('falcon', par); mf.set_key(sk=sk, pk=G['falcon0_h'])
    s, l, _, st = mf.sign_host(msgs, 64, tapes)
    assert np.array_equal(st, st1.astype(np.uint32)) and np.array_equal(l, l1.astype(np.uint32)) and np.array_equal(s[:, :mf.sig_bytes], s1[:, :mf.sig_bytes])
    v = mf.verify_host(msgs, 64, s, l)
    assert (v[st == 1] == 0).all() and int((st == 1).sum()) > B - 3
    m2 = msgs.copy(); m2[::5, 7] ^= 1
    assert (mf.verify_host(m2, 64, s, l)[::5] == 1).all()
    f1.close()

    # ---- ring-TESLA set 0: 77 messages, tapes of 24 attempts
    G = load_golden('tesla_ref.txt')
    keys = [G[f'tesla0_{x}'] for x in ('a1', 'a2', 's', 'e1', 'e2', 't1', 't2')]
    B = 77
    msgs = rng.integers(0, 256, size=(B, 64), dtype=np.uint8)
    tapes = rng.integers(0, 2**32, size=(B, 24 * 512), dtype=np.uint64).astype(np.uint32)
    t1_ = eng.tesla(G['tesla0_params']); t1_.set_key(*keys)
    s1, l1, tr1, st1 = (x.cpu().numpy() for x in t1_.sign(dev(msgs), dev(tapes.view(np.int32)), fixed_len=64))
    mt = m.scheme('tesla', G['tesla0_params']); mt.set_key(*keys)
    s, l, tr, st = mt.sign_host(msgs, 64, tapes)
    assert np.array_equal(st, st1.astype(np.uint32)) and np.array_equal(tr, tr1.astype(np.uint32)) and np.array_equal(s[:, :mt.sig_bytes], s1[:, :mt.sig_bytes])
    assert (mt.verify_host(msgs, 64, s, l)[st == 1] == 0).all() and int((st == 1).sum()) > B - 3
    t1_.close()

    # ---- Kyber set 1 KEM: 333 values
    G = load_golden('kyber_ref.txt')
    B = 333
    mm = rng.integers(0, 256, size=(B, 32), dtype=np.uint8)
    k1 = eng.kyber(G['kyber1_params']); k1.set_pubkey(G['kyber1_t_ntt'], G['kyber1_Agen']); k1.set_privkey(G['kyber1_s_ntt']); k1.set_kem_pubkey(G['kyber1_t'], G['kyber1_rho'])
    ct1, key1 = (x.cpu().numpy() for x in k1.encaps(dev(mm)))
    mk = m.scheme('kyber', G['kyber1_params']); mk.set_key(t_ntt=G['kyber1_t_ntt'], Agen=G['kyber1_Agen'], t=G['kyber1_t'], rho=G['kyber1_rho'], s_ntt=G['kyber1_s_ntt'])
    ct, key = mk.encaps_host(mm)
    assert np.array_equal(ct[:, :mk.ct_bytes], ct1[:, :mk.ct_bytes]) and np.array_equal(key, key1)
    assert np.array_equal(mk.decaps_host(ct), key)
    k1.close()
    m.close(); eng.close()


@pytest.mark.parametrize('shard', [False, True])
def test_bench_two_ranks_contract_line(tmp_path, shard):
    """bench.py as the driver launches it for N > 1 (torch.distributed.run, one process per rank), here with both ranks on device 0 over gloo
    (--test-one-device): the N > 1 code path - barrier, MAX over ranks, the gathered scheme legs, the weak and the --shard batch arithmetic - must
    produce a contract line the driver can parse"""
    import json, os, subprocess, sys, socket
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    s = socket.socket(); s.bind(('127.0.0.1', 0)); port = s.getsockname()[1]; s.close()
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr', '127.0.0.1', '--master-port', str(port),
           os.path.join(root, 'bench.py'), '--gpus', '2', '--steps', '5', '--warmup', '2', '--regions', '3', '--test-one-device', '--no-cpu-baseline', '--no-live-pmc',
           '--extra-file', str(tmp_path / 'extra.json')] + (['--shard'] if shard else [])
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=root)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    lines = [l for l in r.stdout.strip().splitlines() if l.startswith('{')]
    line = lines[-1]
    assert len(line) < 8000
    d = json.loads(line)
    assert d['n_gpus'] == 2 and d['steps'] == 5 and d['scaling'] == ('strong' if shard else 'weak')
    assert d['config']['batch_per_gpu'] == (32768 if shard else 65536)
    assert d['value'] > 0 and 0 < d['roofline']['frac'] <= 1.0
    agg = d['multi_gpu']['aggregate']
    assert all(v > 0 for v in agg.values()), agg
    assert d['multi_gpu']['batch_per_gpu'] == ({'dil_batch': 8192, 'bliss_batch': 8192, 'falcon_batch': 4096} if shard else {'dil_batch': 16384, 'bliss_batch': 16384, 'falcon_batch': 8192})
