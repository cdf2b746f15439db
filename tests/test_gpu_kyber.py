"""GPU parity for batched Kyber CPA encryption / decryption (C ABI: sc_hip_kyber_*) against the reference's own
ciphertexts (tests/golden/kyber_ref.txt) and the pinned oracle on seeded inputs.  Coefficient-exact."""
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


def make(eng, G, tag):
    ky = eng.kyber(G[f'{tag}_params'])
    ky.set_pubkey(G[f'{tag}_t_ntt'], G[f'{tag}_Agen'])
    ky.set_privkey(G[f'{tag}_s_ntt'])
    return ky


@pytest.mark.parametrize('tag,nmsg', [('kyber1', 6), ('kyber2', 2), ('kyber0', 2)])
def test_enc_dec_vs_reference(eng, tag, nmsg):
    import torch
    G = load_golden('kyber_ref.txt')
    ky = make(eng, G, tag)
    m = np.stack([G[f'{tag}_m{i}'].astype(np.uint8) for i in range(nmsg)])
    noise = np.stack([G[f'{tag}_noise{i}'].astype(np.uint8) for i in range(nmsg)])
    assert noise.shape[1] == ky.noise_bytes
    u, v = ky.enc(torch.from_numpy(m).cuda(), torch.from_numpy(noise).cuda())
    for i in range(nmsg):
        assert np.array_equal(u[i].cpu().numpy().reshape(-1), G[f'{tag}_u{i}']), (tag, i)
        assert np.array_equal(v[i].cpu().numpy(), G[f'{tag}_v{i}']), (tag, i)
    back = ky.dec(u, v).cpu().numpy()
    assert np.array_equal(back, m)
    ky.close()


def test_batch_vs_oracle_and_roundtrip(eng):
    """BASELINE config 2 shape at the scheme level: 65536 encryptions under one Kyber set-1 key; every decryption
    returns the message; a strided sample equals the oracle coefficient for coefficient"""
    import torch
    G = load_golden('kyber_ref.txt')
    ky = make(eng, G, 'kyber1')
    B = 65536
    g = torch.Generator(device='cuda'); g.manual_seed(5)
    m = torch.randint(0, 256, (B, 32), dtype=torch.uint8, device='cuda', generator=g)
    noise = torch.randint(0, 256, (B, ky.noise_bytes), dtype=torch.uint8, device='cuda', generator=g)
    u, v = ky.enc(m, noise)
    assert int(u.max()) < 2048 and int(u.min()) >= 0 and int(v.max()) < 8 and int(v.min()) >= 0
    assert torch.equal(ky.dec(u, v), m)
    mh, nh = m.cpu().numpy(), noise.cpu().numpy()
    for i in range(0, B, 4099):
        eu, ev = O.kyber_enc(G['kyber1_params'], G['kyber1_t_ntt'], G['kyber1_Agen'], mh[i], nh[i])
        assert np.array_equal(u[i].cpu().numpy().reshape(-1), eu) and np.array_equal(v[i].cpu().numpy(), ev)
        assert np.array_equal(O.kyber_dec(G['kyber1_params'], G['kyber1_s_ntt'], eu, ev), mh[i])
    ky.close()


@pytest.mark.parametrize('tag,nmsg', [('kyber1', 6), ('kyber2', 2), ('kyber0', 2)])
def test_kem_vs_reference(eng, tag, nmsg):
    """safecrypto_encapsulation / safecrypto_decapsulation on the device: G hash, DRBG oracle bytes and noise, CPA encryption,
    H hash, packing - the reference's ciphertext bytes and shared keys for the same key pair and m"""
    import torch
    G = load_golden('kyber_ref.txt')
    ky = make(eng, G, tag)
    ky.set_kem_pubkey(G[f'{tag}_t'], G[f'{tag}_rho'])
    u8 = lambda a: np.asarray(a, dtype=np.uint8)
    m = np.stack([u8(G[f'{tag}_kem_m{i}']) for i in range(nmsg)])
    ct, keys = ky.encaps(torch.from_numpy(m).cuda())
    cth, kh = ct.cpu().numpy(), keys.cpu().numpy()
    for i in range(nmsg):
        assert cth[i].tobytes() == u8(G[f'{tag}_kem_ct{i}']).tobytes(), (tag, i)
        assert kh[i].tobytes() == u8(G[f'{tag}_kem_key{i}']).tobytes(), (tag, i)
    dk = ky.decaps(ct).cpu().numpy()
    assert np.array_equal(dk, kh)
    bad = ct.clone(); bad[:, ky.ct_bytes // 2] ^= 0x20
    dkb = ky.decaps(bad).cpu().numpy()
    for i in range(nmsg):
        assert dkb[i].tobytes() == u8(G[f'{tag}_kem_decbad{i}']).tobytes(), (tag, i)
    ky.close()


def test_kem_batch_vs_oracle(eng):
    import torch
    G = load_golden('kyber_ref.txt')
    tag = 'kyber1'
    ky = make(eng, G, tag)
    ky.set_kem_pubkey(G[f'{tag}_t'], G[f'{tag}_rho'])
    rng = np.random.default_rng(8)
    B = 333
    m = rng.integers(0, 256, size=(B, 32), dtype=np.uint8)
    ct, keys = ky.encaps(torch.from_numpy(m).cuda())
    dk = ky.decaps(ct)
    assert torch.equal(dk, keys)
    cth, kh = ct.cpu().numpy(), keys.cpu().numpy()
    for i in (0, 63, 64, 200, 332):
        oc, ok = O.kyber_kem_enc(G[f'{tag}_params'], G[f'{tag}_t'], G[f'{tag}_t_ntt'], G[f'{tag}_rho'], G[f'{tag}_Agen'], m[i])
        assert cth[i].tobytes() == oc and kh[i].tobytes() == ok, i
    ky.close()


@pytest.mark.parametrize('tag', ['kyber1', 'kyber2', 'kyber0'])
def test_keygen_vs_reference(eng, tag):
    """batched key generation from per-key seeds: every component of the reference's key pair for the fixture's seeds, the oracle's
    on a seeded batch, and a generated key pair encapsulates / decapsulates"""
    import torch
    G = load_golden('kyber_ref.txt')
    par = G[f'{tag}_params']
    ky = eng.kyber(par)
    rng = np.random.default_rng(3)
    B = 70
    seeds = rng.integers(0, 256, size=(B, 32), dtype=np.uint8)
    seeds[0] = np.asarray(G[f'{tag}_kg_seed0'], dtype=np.uint8); seeds[B - 1] = np.asarray(G[f'{tag}_kg_seed1'], dtype=np.uint8)
    s_ntt, t, t_ntt, rho, z = (x.cpu().numpy() for x in ky.keygen(torch.from_numpy(seeds).cuda()))
    for row, i in ((0, 0), (B - 1, 1)):
        assert np.array_equal(rho[row], G[f'{tag}_kg_rho{i}']) and np.array_equal(z[row], G[f'{tag}_kg_z{i}'])
        assert np.array_equal(s_ntt[row].reshape(-1), G[f'{tag}_kg_s_ntt{i}'])
        assert np.array_equal(t[row].reshape(-1), G[f'{tag}_kg_t{i}'])
        assert np.array_equal(t_ntt[row].reshape(-1), G[f'{tag}_kg_t_ntt{i}'])
    for row in (1, 33, 64):
        o = O.kyber_keygen(par, seeds[row])
        for got, want in zip((s_ntt[row], t[row], t_ntt[row], rho[row], z[row]), o):
            assert np.array_equal(got.reshape(-1), want)
    # use key 33: expand its matrix, load it, run the KEM round trip
    k = int(par[5])
    A = eng.expand_matrix(torch.from_numpy(rho[33:34].copy()).cuda(), k * k, int(par[1]), int(par[2])).cpu().numpy().reshape(-1)
    ky.set_pubkey(t_ntt[33].reshape(-1), A); ky.set_privkey(s_ntt[33].reshape(-1)); ky.set_kem_pubkey(t[33].reshape(-1), rho[33])
    m = torch.from_numpy(rng.integers(0, 256, size=(5, 32), dtype=np.uint8)).cuda()
    ct, keys = ky.encaps(m)
    assert torch.equal(ky.decaps(ct), keys)
    ky.close()


@pytest.mark.parametrize('tag', ['kyber1', 'kyber2'])
def test_kem_distinct_keys_per_value(eng, tag):
    """SURVEY 8(d) distinct keys: 200 key pairs from the batched keygen, one encapsulation under each.  Ciphertext and shared key
    equal the oracle's under THAT key (its matrix expanded from its own rho); decapsulation under the matching private keys returns
    the keys, under shifted private keys it does not; the one-key entry points agree on a row"""
    import torch
    G = load_golden('kyber_ref.txt')
    par = G[f'{tag}_params']
    k, q, qb = int(par[5]), int(par[1]), int(par[2])
    ky = eng.kyber(par)
    rng = np.random.default_rng(41)
    B = 200
    seeds = torch.from_numpy(rng.integers(0, 256, size=(B, 32), dtype=np.uint8)).cuda()
    s_ntt, t, t_ntt, rho, z = ky.keygen(seeds)
    m = rng.integers(0, 256, size=(B, 32), dtype=np.uint8)
    ct, keys = ky.encaps_keys(t, t_ntt, rho, torch.from_numpy(m).cuda())
    dk = ky.decaps_keys(s_ntt, t, t_ntt, rho, ct)
    assert torch.equal(dk, keys)
    sh, th, tnh, rh, cth, kh = (x.cpu().numpy() for x in (s_ntt, t, t_ntt, rho, ct, keys))
    for i in (0, 1, 63, 64, 127, 199):
        A = O.expand_matrix(rh[i], k * k, q, qb)
        oc, ok = O.kyber_kem_enc(par, th[i].reshape(-1), tnh[i].reshape(-1), rh[i], A, m[i])
        assert cth[i].tobytes() == oc and kh[i].tobytes() == ok, i
        assert O.kyber_kem_dec(par, sh[i].reshape(-1), th[i].reshape(-1), tnh[i].reshape(-1), rh[i], A, oc) == ok
    wrong = ky.decaps_keys(torch.roll(s_ntt, 1, 0).contiguous(), t, t_ntt, rho, ct)
    assert int((wrong == keys).all(dim=1).sum()) == 0
    # the one-key path under key 5 gives row 5
    A5 = O.expand_matrix(rh[5], k * k, q, qb)
    ky.set_pubkey(tnh[5].reshape(-1), A5); ky.set_privkey(sh[5].reshape(-1)); ky.set_kem_pubkey(th[5].reshape(-1), rh[5])
    c1, k1 = ky.encaps(torch.from_numpy(m[5:6]).cuda())
    assert torch.equal(c1[0], ct[5]) and torch.equal(k1[0], keys[5])
    assert torch.equal(ky.decaps(ct[5:6].contiguous())[0], keys[5])
    ky.close()
