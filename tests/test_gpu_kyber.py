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


@pytest.mark.parametrize('tag,nmsg', [('kyber1', 6), ('kyber2', 2)])
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
