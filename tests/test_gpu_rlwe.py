"""GPU parity for batched Ring-LWE encryption (C ABI: sc_hip_rlwe_*): key pairs, ciphertexts and decryptions byte-identical to
what the reference produced from the same CSPRNG word tapes (tests/golden/rlwe_ref.txt), and to the pinned oracle on seeded
tapes; encrypt -> decrypt round trip at a large batch."""
import numpy as np
import pytest

from tests.golden_io import load_golden
from tests import oracle_lib as O

pytestmark = pytest.mark.gpu

CASES = [('rlwe0', 3, 6), ('rlwe1', 2, 4)]


@pytest.fixture(scope='module')
def eng():
    from libsafecrypto_amd import Engine
    e = Engine(0)
    yield e
    e.close()


@pytest.fixture(scope='module')
def G():
    return load_golden('rlwe_ref.txt')


def u32(a):
    return (np.asarray(a).astype(np.int64) & 0xFFFFFFFF).astype(np.uint32)


def dev(a):
    import torch
    a = np.ascontiguousarray(a)
    return torch.from_numpy(a.view(np.int32) if a.dtype == np.uint32 else a).cuda()


@pytest.mark.parametrize('tag,nkeys,nmsg', CASES)
def test_keygen_reproduces_reference(eng, G, tag, nkeys, nmsg):
    import torch
    r = eng.rlwe(G[f'{tag}_params'])
    tapes = np.stack([u32(G[f'{tag}_kg_tape{k}']) for k in range(nkeys)])
    assert tapes.shape[1] == r.keygen_words
    a, p, r2 = r.keygen(dev(tapes))
    torch.cuda.synchronize()
    for k in range(nkeys):
        assert np.array_equal(a[k].cpu().numpy(), G[f'{tag}_kg_a{k}'])
        assert np.array_equal(p[k].cpu().numpy(), G[f'{tag}_kg_p{k}'])
        assert np.array_equal(r2[k].cpu().numpy(), G[f'{tag}_kg_r2_{k}'])
    r.close()


@pytest.mark.parametrize('arith', [0, 3])          # automatic (signed small-modulus butterflies) / the unsigned multiply-add form
@pytest.mark.parametrize('tag,nkeys,nmsg', CASES)
def test_encrypt_decrypt_reproduce_reference(eng, G, tag, nkeys, nmsg, arith):
    import torch
    r = eng.rlwe(G[f'{tag}_params'])
    r.set_arith(arith)
    k = nkeys - 1
    r.set_pubkey(G[f'{tag}_kg_a{k}'], G[f'{tag}_kg_p{k}'])
    r.set_privkey(G[f'{tag}_kg_r2_{k}'])
    msgs = np.stack([np.asarray(G[f'{tag}_msg{m}'], dtype=np.uint8) for m in range(nmsg)])
    tapes = np.stack([u32(G[f'{tag}_tape{m}']) for m in range(nmsg)])
    assert tapes.shape[1] == r.enc_words and msgs.shape[1] == r.msg_bytes
    ct = r.encrypt(dev(msgs), dev(tapes))
    torch.cuda.synchronize()
    for m in range(nmsg):
        assert bytes(ct[m].cpu().numpy()) == bytes(np.asarray(G[f'{tag}_ct{m}'], dtype=np.uint8)), m
    dec = r.decrypt(ct).cpu().numpy()
    assert np.array_equal(dec, msgs)
    rnd = np.stack([np.asarray(G[f'{tag}_randct{m}'], dtype=np.uint8) for m in range(nmsg)])      # coefficients >= q included
    dec = r.decrypt(dev(rnd)).cpu().numpy()
    for m in range(nmsg):
        assert bytes(dec[m]) == bytes(np.asarray(G[f'{tag}_randdec{m}'], dtype=np.uint8)), m
    r.close()


@pytest.mark.parametrize('arith', [0, 3])
@pytest.mark.parametrize('tag,batch', [('rlwe0', 37), ('rlwe1', 21)])
def test_seeded_batch_matches_oracle(eng, G, tag, batch, arith):
    """ragged batch (not a multiple of the messages per wavefront), padded strides, keys made on the device"""
    import torch
    R = O.Rlwe(G[f'{tag}_params'])
    r = eng.rlwe(G[f'{tag}_params'])
    r.set_arith(arith)
    rng = np.random.default_rng(0x51e + batch)
    ktape = rng.integers(0, 1 << 32, size=(3, r.keygen_words + 6), dtype=np.uint64).astype(np.uint32)
    a, p, r2 = (x.cpu().numpy() for x in r.keygen(dev(ktape)))
    for i in range(3):
        rc, oa, op, or2 = R.keygen(ktape[i])
        assert rc == 0 and np.array_equal(a[i], oa) and np.array_equal(p[i], op) and np.array_equal(r2[i], or2)
    r.set_pubkey(a[1], p[1]); r.set_privkey(r2[1])
    msgs = rng.integers(0, 256, size=(batch, r.msg_bytes + 8), dtype=np.uint8)
    tapes = rng.integers(0, 1 << 32, size=(batch, r.enc_words + 10), dtype=np.uint64).astype(np.uint32)
    tapes[0, :8] = 0                                               # draws of 0: the search's lower edge
    tapes[1, :8] = 0xFFFFFFFF                                      # and its upper edge
    dm = dev(msgs)[:, :r.msg_bytes]
    ct = r.encrypt(dm, dev(tapes))
    torch.cuda.synchronize()
    ctn = ct.cpu().numpy()
    for i in range(batch):
        rc, oc = R.encrypt(a[1], p[1], bytes(msgs[i, :r.msg_bytes]), tapes[i])
        assert rc == 0 and bytes(ctn[i]) == oc, i
    dec = r.decrypt(ct).cpu().numpy()
    assert np.array_equal(dec, msgs[:, :r.msg_bytes])
    junk = rng.integers(0, 256, size=(batch, r.ct_bytes), dtype=np.uint8)
    dj = r.decrypt(dev(junk)).cpu().numpy()
    for i in range(batch):
        assert bytes(dj[i]) == R.decrypt(r2[1], bytes(junk[i]))[1], i
    r.close()


def test_round_trip_large_batch(eng, G):
    """encrypt -> decrypt over 65536 messages recovers every bit; linearity in the message: ct(m) - ct(0) depends on m only"""
    import torch
    r = eng.rlwe(G['rlwe0_params'])
    r.set_pubkey(G['rlwe0_kg_a2'], G['rlwe0_kg_p2']); r.set_privkey(G['rlwe0_kg_r2_2'])
    batch = 65536
    g = torch.Generator(device='cuda'); g.manual_seed(7)
    msgs = torch.randint(0, 256, (batch, r.msg_bytes), dtype=torch.uint8, device='cuda', generator=g)
    tapes = torch.randint(-2**31, 2**31 - 1, (batch, r.enc_words), dtype=torch.int32, device='cuda', generator=g)
    ct = r.encrypt(msgs, tapes)
    dec = r.decrypt(ct)
    torch.cuda.synchronize()
    assert torch.equal(dec, msgs)
    ct0 = r.encrypt(torch.zeros_like(msgs), tapes)
    assert torch.equal(ct0[:, :r.ct_bytes // 2], ct[:, :r.ct_bytes // 2])          # c1 does not depend on the message
    assert torch.equal(r.decrypt(ct0), torch.zeros_like(msgs))
    r.close()


def test_errors_are_reported(eng, G):
    import torch
    from libsafecrypto_amd import EngineError
    par = np.asarray(G['rlwe0_params']).copy()
    bad = par.copy(); bad[2] = 12
    with pytest.raises(EngineError):
        eng.rlwe(bad)
    r = eng.rlwe(par)
    msgs = torch.zeros((4, r.msg_bytes), dtype=torch.uint8, device='cuda')
    tapes = torch.zeros((4, r.enc_words), dtype=torch.int32, device='cuda')
    with pytest.raises(EngineError):
        r.encrypt(msgs, tapes)                                    # no public key
    with pytest.raises(EngineError):
        r.decrypt(torch.zeros((4, r.ct_bytes), dtype=torch.uint8, device='cuda'))      # no private key
    r.set_pubkey(G['rlwe0_kg_a0'], G['rlwe0_kg_p0'])
    with pytest.raises(EngineError):
        r.encrypt(msgs, tapes[:, :-2].contiguous())               # tape shorter than one encryption consumes
    r.close()
