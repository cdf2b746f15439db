"""GPU parity for batched Dilithium sign / verify (C ABI: sc_hip_dilithium_*), against signatures the
reference produced (tests/golden/dilithium_ref.txt) and against the pinned oracle on seeded messages.
Byte-exact: deterministic signing makes a signature a pure function of (key, message)."""
import hashlib

import numpy as np
import pytest

from tests.golden_io import load_golden
from tests import oracle_lib as O

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def eng():
    from libsafecrypto_amd import Engine
    e = Engine(0)
    yield e
    e.close()


@pytest.fixture(scope='module')
def G():
    return load_golden('dilithium_ref.txt')


def golden_msgs(G, tag):
    out, i = [], 0
    while f'{tag}_msg{i}' in G:
        out.append((bytes(G[f'{tag}_msg{i}'].astype(np.uint8)), bytes(G[f'{tag}_sig{i}'].astype(np.uint8)), int(G[f'{tag}_trials'][i])))
        i += 1
    return out


def make(eng, key):
    d = eng.dilithium(key.par)
    d.set_privkey(key.s1, key.s2, key.t, key.K, key.tr, key.Ahat)
    d.set_pubkey(key.t1, key.rho, key.Ahat)
    return d


def test_device_xof_is_the_references_permutation(eng):
    for n in (0, 1, 47, 48, 135, 136, 137, 168, 169, 1000, 1584):
        data = bytes((i * 13 + 5) & 0xFF for i in range(n))
        assert eng.shake(128, data, 200, standard=True) == hashlib.shake_128(data).digest(200)
        assert eng.shake(256, data, 200, standard=True) == hashlib.shake_256(data).digest(200)
        assert eng.shake(128, data, 200) == O.shake(128, data, 200)
        assert eng.shake(256, data, 200) == O.shake(256, data, 200)
    msg = bytes.fromhex('01080f161d242b323940474e555c636a71787f868d949ba2a9b0b7bec5ccd3da02890866240a817ede7ec72775abce636fd6')
    assert eng.shake(256, msg, 48).hex() == ('ba1789c93f9d0640b0f5e41a6f59c1310f643847fc10afeb65ee638e67610bf9'
                                             'dba6b1551cd5eac169615555dbf22a67')     # captured from the reference build


@pytest.mark.parametrize('tag', ['dil3', 'dil1', 'dil0', 'dil2'])
def test_sign_reproduces_reference_signatures(eng, G, tag):
    key = O.DilKey.from_golden(G, tag)
    d = make(eng, key)
    gm = golden_msgs(G, tag)
    sigs, trials = d.sign_host([m for m, _, _ in gm])
    for i, (m, sig, tr) in enumerate(gm):
        assert sigs[i] == sig, (tag, i)
        assert trials[i] == tr
    d.close()


@pytest.mark.parametrize('tag', ['dil3', 'dil1', 'dil0', 'dil2'])
def test_verify_matches_reference_verdicts(eng, G, tag):
    key = O.DilKey.from_golden(G, tag)
    d = make(eng, key)
    gm = golden_msgs(G, tag)
    msgs, sigs, exp = [], [], []
    for i, (m, sig, _) in enumerate(gm):
        bad = bytearray(sig); bad[len(bad) // 2] ^= 0x10
        msgs += [m, m, m + b'x', m]
        sigs += [sig, bytes(bad), sig, sig[:len(sig) - 40]]
        exp += [int(G[f'{tag}_verify'][2 * i]), int(G[f'{tag}_verify'][2 * i + 1]), 1, 1]
    assert exp[0] == 0 and exp[1] == 1
    assert d.verify_host(msgs, sigs) == exp
    d.close()


def test_batch_vs_oracle_seeded(eng, G):
    """257 seeded messages of ragged lengths (0..200 bytes) under the Dilithium-III fixture key: every
    signature equals the oracle's, attempts included; then everything verifies on the device and the oracle"""
    key = O.DilKey.from_golden(G, 'dil3')
    d = make(eng, key)
    rng = np.random.default_rng(2024)
    msgs = [bytes(rng.integers(0, 256, size=int(rng.integers(0, 201)), dtype=np.uint8)) for _ in range(257)]
    sigs, trials = d.sign_host(msgs)
    for i in range(0, 257, 8):                       # the oracle is slow: check a strided sample exactly
        es, et = O.dilithium_sign(key, msgs[i])
        assert sigs[i] == es and trials[i] == et, i
    assert max(trials) > 3                            # the rejection loop really ran several rounds
    assert d.verify_host(msgs, sigs) == [0] * 257
    for i in range(0, 257, 32):
        assert O.dilithium_verify(key, msgs[i], sigs[i]) == 0
    # the forms a batch of this size does not pick by itself give the same bytes: ExpandMask and the challenge hash with one lane per state (what a
    # batch beyond 64 K attempt slots runs; sc_hip_dilithium_set_tuning bit 9), the host looking at the count after every round (bit 8), w = A y
    # between the round's kernels in 24 bits per coefficient (bit 10)
    for tuning in (1 << 9, 1 << 8, (1 << 9) | (1 << 8), 1 << 10, (1 << 10) | (1 << 9)):
        d.set_tuning(tuning)
        s2, t2 = d.sign_host(msgs)
        assert s2 == sigs and t2 == trials, tuning
    d.set_tuning(0)
    d.close()


def test_signature_rows_off_word_boundaries(eng, G):
    """a signature buffer whose row pitch is odd: the verifier's z unpacker reads its groups byte by byte instead of word by word - same verdicts"""
    import torch
    key = O.DilKey.from_golden(G, 'dil3')
    d = make(eng, key)
    g = torch.Generator(device='cuda'); g.manual_seed(77)
    B = 300
    msgs = torch.randint(0, 256, (B, 48), dtype=torch.uint8, device='cuda', generator=g)
    sigs, siglens, _ = d.sign(msgs, fixed_len=48)
    assert d.verify(msgs, sigs, siglens, fixed_len=48).cpu().tolist() == [0] * B
    odd = torch.zeros((B, d.sig_stride + 1), dtype=torch.uint8, device='cuda')
    odd[:, :d.sig_stride] = sigs
    assert d.verify(msgs, odd, siglens, fixed_len=48).cpu().tolist() == [0] * B
    odd[7, 100] ^= 1; odd[200, 3000] ^= 0x40
    v = d.verify(msgs, odd, siglens, fixed_len=48).cpu().tolist()
    assert v[7] == 1 and v[200] == 1 and sum(v) == 2
    d.close()


def test_device_buffers_full_config3_properties(eng, G):
    """BASELINE config 3 shape: 16384 messages of 64 bytes, one key, device-resident buffers.
    Size-independent properties: sign -> verify accepts all; flipping one bit of a message or a
    signature rejects exactly those; a sample equals the oracle byte for byte."""
    import torch
    key = O.DilKey.from_golden(G, 'dil3')
    d = make(eng, key)
    B = 16384
    g = torch.Generator(device='cuda'); g.manual_seed(3)
    msgs = torch.randint(0, 256, (B, 64), dtype=torch.uint8, device='cuda', generator=g)
    sigs, siglens, trials = d.sign(msgs, fixed_len=64)
    torch.cuda.synchronize()
    assert int(siglens.min()) > 3200 and int(siglens.max()) <= d.sig_max
    assert 2.5 < float(trials.float().mean()) < 4.5          # ~3.4 trials per signature
    assert d.workspace_residue() == 0                         # the masks y and A y of every attempt are zeroed behind the call (dilithium.c:1964-1975)
    v = d.verify(msgs, sigs, siglens, fixed_len=64)
    assert int(v.sum()) == 0
    bad_m = msgs.clone(); bad_m[::7, 5] ^= 1
    assert torch.equal(d.verify(bad_m, sigs, siglens, fixed_len=64).bool().cpu(), (torch.arange(B) % 7 == 0))
    bad_s = sigs.clone(); bad_s[::5, 1000] ^= 0x40
    assert torch.equal(d.verify(msgs, bad_s, siglens, fixed_len=64).bool().cpu(), (torch.arange(B) % 5 == 0))
    mh, sh, lh, th = msgs.cpu().numpy(), sigs.cpu().numpy(), siglens.cpu().numpy(), trials.cpu().numpy()
    sample = list(range(0, B, 31)) + [B - 1]                 # 530 of the 16384 signatures, byte for byte against the oracle (attempt counts included)
    for i in sample:
        es, et = O.dilithium_sign(key, bytes(mh[i]))
        assert bytes(sh[i, :lh[i]]) == es and int(th[i]) == et, i
    d.close()


def _ref_driver():
    import os
    drv = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'oracle', '_ref', 'ref_driver')
    if not os.path.exists(drv):
        pytest.skip('oracle/_ref/ref_driver not built')
    return drv


def test_fresh_keys_and_signatures_from_the_reference_run_here(eng, tmp_path):
    """the reference itself, run on the box the test runs on (oracle/_ref/ref_driver travels as a binary): fresh OS-seeded key pairs for all four
    parameter sets, its signatures and verdicts - a fixture nobody has seen before.  The engine reproduces every signature byte for byte."""
    import subprocess
    subprocess.check_call([_ref_driver(), 'dilithium', str(tmp_path)], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    L = load_golden(str(tmp_path / 'dilithium_ref.txt'))
    total = 0
    for tag in ('dil3', 'dil1', 'dil0', 'dil2'):
        key = O.DilKey.from_golden(L, tag)
        d = make(eng, key)
        gm = golden_msgs(L, tag)
        sigs, trials = d.sign_host([m for m, _, _ in gm])
        for i, (m, sig, tr) in enumerate(gm):
            assert sigs[i] == sig and trials[i] == tr, (tag, i)
        assert d.verify_host([m for m, _, _ in gm], [s_ for _, s_, _ in gm]) == [0] * len(gm)
        total += len(gm)
        d.close()
    assert total >= 20


@pytest.mark.parametrize('tag', ['dil3', 'dil1', 'dil0', 'dil2'])
def test_keygen_vs_reference(eng, tag):
    """batched key generation from per-key seeds: the reference's key pair for the fixture's seeds, the oracle's on a seeded batch,
    and a generated key signs and verifies (matrix expanded from its rho on the device)"""
    import torch
    G = load_golden('dilithium_ref.txt')
    par, eb = G[f'{tag}_params'], int(G[f'{tag}_eta_bits'][0])
    d = eng.dilithium(par)
    rng = np.random.default_rng(4)
    B = 67
    seeds = rng.integers(0, 256, size=(B, 32), dtype=np.uint8)
    seeds[0] = np.asarray(G[f'{tag}_kg_seed0'], dtype=np.uint8); seeds[B - 1] = np.asarray(G[f'{tag}_kg_seed1'], dtype=np.uint8)
    out = [x.cpu().numpy() for x in d.keygen(torch.from_numpy(seeds).cuda(), eb)]
    names = ('s1_', 's2_', 't', 'rho', 'K', 'tr', 't1_')
    for row, i in ((0, 0), (B - 1, 1)):
        for got, name in zip(out, names):
            assert np.array_equal(got[row].reshape(-1), G[f'{tag}_kg_{name}{i}']), (tag, i, name)
    for row in (1, 40, 64):
        o = O.dilithium_keygen(par, eb, seeds[row])
        for got, want, name in zip(out, o, names):
            assert np.array_equal(got[row].reshape(-1), want), (row, name)
    s1, s2, t, rho, K, tr, t1 = (x[40] for x in out)
    k, l = int(par[3]), int(par[4])
    A = eng.expand_matrix(torch.from_numpy(rho[None, :].copy()).cuda(), k * l, int(par[1]), int(par[2])).cpu().numpy().reshape(-1)
    d.set_privkey(s1.reshape(-1), s2.reshape(-1), t.reshape(-1), K, tr, A)
    d.set_pubkey(t1.reshape(-1), rho, A)
    msgs = torch.from_numpy(rng.integers(0, 256, size=(9, 64), dtype=np.uint8)).cuda()
    sigs, siglens, trials = d.sign(msgs, fixed_len=64)
    assert d.verify(msgs, sigs, siglens, fixed_len=64).cpu().tolist() == [0] * 9
    d.close()


def _sign_under_generated_keys(eng, par, eb, nkeys, per_key, seed):
    """nkeys key pairs from the batched key generator; per_key 64-byte messages signed under each (one key load per key)"""
    import torch
    from types import SimpleNamespace
    d = eng.dilithium(par)
    rng = np.random.default_rng(seed)
    seeds = torch.from_numpy(rng.integers(0, 256, size=(nkeys, 32), dtype=np.uint8)).cuda()
    s1, s2, t, rho, K, tr, t1 = d.keygen(seeds, eb)
    k, l = int(par[3]), int(par[4])
    A = eng.expand_matrix(rho, k * l, int(par[1]), int(par[2]))
    g = torch.Generator(device='cuda'); g.manual_seed(seed)
    msgs = torch.randint(0, 256, (nkeys * per_key, 64), dtype=torch.uint8, device='cuda', generator=g)
    sigs = torch.zeros((nkeys * per_key, d.sig_stride), dtype=torch.uint8, device='cuda')
    siglens = torch.zeros(nkeys * per_key, dtype=torch.int32, device='cuda')
    keys = []
    for i in range(nkeys):
        h = [x[i].cpu().numpy().reshape(-1) for x in (s1, s2, t, K, tr, t1, rho, A)]
        d.set_privkey(h[0], h[1], h[2], h[3], h[4], h[7])
        sl = slice(i * per_key, (i + 1) * per_key)
        sg, ln, _ = d.sign(msgs[sl], fixed_len=64)
        sigs[sl, :sg.shape[1]] = sg; siglens[sl] = ln
        keys.append(SimpleNamespace(par=np.ascontiguousarray(np.asarray(par), dtype=np.int32), t1=np.ascontiguousarray(h[5], dtype=np.int32),
                                    rho=np.ascontiguousarray(h[6], dtype=np.uint8), Ahat=np.ascontiguousarray(h[7], dtype=np.int32)))
    t1_items = t1.reshape(nkeys, -1).repeat_interleave(per_key, dim=0).contiguous()
    rho_items = rho.repeat_interleave(per_key, dim=0).contiguous()
    return d, keys, msgs, sigs, siglens, t1_items, rho_items


@pytest.mark.parametrize('tag', ['dil3', 'dil1', 'dil0', 'dil2'])
def test_verify_under_distinct_keys(eng, G, tag):
    """sc_hip_dilithium_verify_keys: every item carries its own public key (t1, rho); the matrix, tr and NTT(t1 2^d) are derived per item on the
    device.  Signatures made under 11 generated keys verify; tampered signatures / messages and a signature presented with ANOTHER item's key get
    the oracle's verdict."""
    import torch
    par, eb = G[f'{tag}_params'], int(G[f'{tag}_eta_bits'][0])
    nk, per = 11, 6
    d, keys, msgs, sigs, siglens, t1i, rhoi = _sign_under_generated_keys(eng, par, eb, nk, per, 77)
    B = nk * per
    assert d.verify_keys(t1i, rhoi, msgs, sigs, siglens, fixed_len=64).cpu().tolist() == [0] * B
    bad_s = sigs.clone(); bad_s[1::4, 700] ^= 0x10
    bad_m = msgs.clone(); bad_m[2::4, 9] ^= 0x01
    t1x, rhox = t1i.clone(), rhoi.clone()
    t1x[3::4] = t1i.roll(per, 0)[3::4]; rhox[3::4] = rhoi.roll(per, 0)[3::4]          # the neighbouring key
    got = d.verify_keys(t1x, rhox, bad_m, bad_s, siglens, fixed_len=64).cpu().tolist()
    mh, sh, lh = bad_m.cpu().numpy(), bad_s.cpu().numpy(), siglens.cpu().numpy()
    for i in range(B):
        ki = i // per if i % 4 != 3 else ((i - per) % B) // per
        want = O.dilithium_verify(keys[ki], bytes(mh[i]), bytes(sh[i, :lh[i]]))
        assert got[i] == want, i
        assert want == (0 if i % 4 == 0 else 1)
    d.close()


def test_verify_distinct_keys_full_config3(eng, G):
    """BASELINE config 3, the distinct-key leg: 16384 items of 64 bytes, each with its own (t1, rho) buffer slot (64 generated keys x 256 messages).
    Everything verifies; flipping a signature bit rejects exactly those items; agrees with the one-key path on the same data."""
    import torch
    par, eb = G['dil3_params'], int(G['dil3_eta_bits'][0])
    d, keys, msgs, sigs, siglens, t1i, rhoi = _sign_under_generated_keys(eng, par, eb, 64, 256, 5)
    B = 64 * 256
    assert int(d.verify_keys(t1i, rhoi, msgs, sigs, siglens, fixed_len=64).sum()) == 0
    bad = sigs.clone(); bad[::9, 1234] ^= 0x04
    assert torch.equal(d.verify_keys(t1i, rhoi, msgs, bad, siglens, fixed_len=64).bool().cpu(), (torch.arange(B) % 9 == 0))
    d.set_pubkey(keys[63].t1, keys[63].rho, keys[63].Ahat)
    one = d.verify(msgs[-256:], bad[-256:], siglens[-256:], fixed_len=64)
    many = d.verify_keys(t1i[-256:], rhoi[-256:], msgs[-256:], bad[-256:], siglens[-256:], fixed_len=64)
    assert torch.equal(one, many)
    d.close()


@pytest.mark.parametrize('tag', ['dil3', 'dil1', 'dil0', 'dil2'])
def test_sign_under_distinct_keys(eng, G, tag):
    """sc_hip_dilithium_sign_keys: every message under its own private key (the key generator's output buffers go straight in).  Byte for byte the
    oracle's signature under that key, attempts included; ragged message lengths; then the distinct-key verifier accepts all of them."""
    import torch
    par, eb = G[f'{tag}_params'], int(G[f'{tag}_eta_bits'][0])
    d = eng.dilithium(par)
    rng = np.random.default_rng(99)
    B = 70
    seeds = torch.from_numpy(rng.integers(0, 256, size=(B, 32), dtype=np.uint8)).cuda()
    s1, s2, t, rho, K, tr, t1 = d.keygen(seeds, eb)
    k, l = int(par[3]), int(par[4])
    lens = rng.integers(0, 150, size=B).astype(np.int32)
    mb = rng.integers(0, 256, size=(B, 152), dtype=np.uint8)
    msgs, ml = torch.from_numpy(mb).cuda(), torch.from_numpy(lens).cuda()
    sigs, siglens, trials = d.sign_keys(s1, s2, t, rho, K, tr, msgs, mlens=ml)
    torch.cuda.synchronize()
    assert d.workspace_residue() == 0                         # y, A y and NTT(y) (kept in HBM on this path) are zeroed behind the call
    A = eng.expand_matrix(rho, k * l, int(par[1]), int(par[2])).cpu().numpy()
    h = [x.cpu().numpy() for x in (s1, s2, t, rho, K, tr, t1)]
    sh, lh, th = sigs.cpu().numpy(), siglens.cpu().numpy(), trials.cpu().numpy()
    for i in range(0, B, 3):
        key = O.DilKey(par, h[0][i].reshape(-1), h[1][i].reshape(-1), h[2][i].reshape(-1), h[3][i], h[4][i], h[5][i], h[6][i].reshape(-1), A[i].reshape(-1))
        es, et = O.dilithium_sign(key, bytes(mb[i, :lens[i]]))
        assert bytes(sh[i, :lh[i]]) == es and int(th[i]) == et, i
    assert int(th.max()) > 3
    v = d.verify_keys(t1.reshape(B, -1).contiguous(), rho, msgs, sigs, siglens, mlens=ml)
    assert v.cpu().tolist() == [0] * B
    d.close()


def test_sign_distinct_keys_full_config3(eng, G):
    """BASELINE config 3, the distinct-key leg at full size: 16384 generated key pairs, one 64-byte message each.  Everything verifies under its own
    key and nothing under its neighbour's; the one-key signer reproduces a sample under the same keys."""
    import torch
    par, eb = G['dil3_params'], int(G['dil3_eta_bits'][0])
    d = eng.dilithium(par)
    B = 16384
    g = torch.Generator(device='cuda'); g.manual_seed(8)
    seeds = torch.randint(0, 256, (B, 32), dtype=torch.uint8, device='cuda', generator=g)
    msgs = torch.randint(0, 256, (B, 64), dtype=torch.uint8, device='cuda', generator=g)
    s1, s2, t, rho, K, tr, t1 = d.keygen(seeds, eb)
    sigs, siglens, trials = d.sign_keys(s1, s2, t, rho, K, tr, msgs, fixed_len=64)
    assert 2.5 < float(trials.float().mean()) < 4.5
    t1f = t1.reshape(B, -1).contiguous()
    assert int(d.verify_keys(t1f, rho, msgs, sigs, siglens, fixed_len=64).sum()) == 0
    assert int(d.verify_keys(t1f.roll(1, 0).contiguous(), rho.roll(1, 0).contiguous(), msgs, sigs, siglens, fixed_len=64).sum()) == B
    k, l = int(par[3]), int(par[4])
    for i in (0, 7777, B - 1):
        A = eng.expand_matrix(rho[i:i + 1], k * l, int(par[1]), int(par[2])).cpu().numpy().reshape(-1)
        h = [x[i].cpu().numpy().reshape(-1) for x in (s1, s2, t, K, tr)]
        d.set_privkey(h[0], h[1], h[2], h[3], h[4], A)
        sg, ln, _ = d.sign(msgs[i:i + 1], fixed_len=64)
        assert torch.equal(sg[0, :int(ln[0])], sigs[i, :int(siglens[i])])
    d.close()


def test_untrusted_signature_lengths_and_hint_counts(eng, G):
    """a verifier must not read past a signature's row whatever siglens says, and must reject hint counts above omega before using them
    (dilithium.c:2085-2123): lengths beyond the row pitch, a hint-count field of all ones, and a row pitch that only just holds z"""
    import torch
    tag = 'dil3'
    d = eng.dilithium(G[f'{tag}_params'])
    d.set_privkey(*(G[f'{tag}_{x}'] for x in ('s1', 's2', 't', 'K', 'tr', 'Ahat')))
    d.set_pubkey(G[f'{tag}_t1'], G[f'{tag}_rho'], G[f'{tag}_Ahat'])
    B = 64
    g = torch.Generator(device='cuda'); g.manual_seed(9)
    msgs = torch.randint(0, 256, (B, 64), dtype=torch.uint8, device='cuda', generator=g)
    sigs, siglens, _ = d.sign(msgs, fixed_len=64)
    assert int(d.verify(msgs, sigs, siglens, fixed_len=64).sum()) == 0
    # lengths that claim more bytes than a row holds: invalid (and the neighbour's bytes are never consulted)
    long_lens = siglens.clone(); long_lens[::2] = sigs.shape[1] + 200
    v = d.verify(msgs, sigs, long_lens, fixed_len=64).cpu().numpy()
    assert (v[::2] == 1).all() and (v[1::2] == 0).all()
    # a hint count of 2^omega_bits - 1 (> omega): invalid
    par = np.asarray(G[f'{tag}_params'])
    n, l, z_bits, omega_bits = int(par[0]), int(par[4]), int(par[11]), int(par[14])
    zbytes = l * n * z_bits // 8
    bad = sigs.clone()
    bad[:, zbytes] |= ((1 << omega_bits) - 1) << (8 - omega_bits)
    assert int(d.verify(msgs, bad, siglens, fixed_len=64).min()) == 1
    # rows cut right behind z: every signature is too short for its hints, none may be read past
    cut = sigs[:, :zbytes].contiguous()
    assert int(d.verify(msgs, cut, siglens, fixed_len=64).min()) == 1
    d.close()
