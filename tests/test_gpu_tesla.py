"""GPU parity for batched ring-TESLA sign / verify (C ABI: sc_hip_tesla_*): byte-identical to signatures the reference produced
from the same (key, message, CSPRNG word tape), and to the pinned oracle on seeded tapes."""
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
    return load_golden('tesla_ref.txt')


def u32(a):
    return (np.asarray(a).astype(np.int64) & 0xFFFFFFFF).astype(np.uint32)


def pack(msgs, tapes, tape_words):
    import torch
    n = len(msgs)
    stride = max(8, (max(len(m) for m in msgs) + 7) & ~7)
    mb = np.zeros((n, stride), dtype=np.uint8)
    for i, m in enumerate(msgs):
        mb[i, :len(m)] = np.frombuffer(m, dtype=np.uint8)
    tb = np.zeros((n, tape_words), dtype=np.uint32)
    tl = np.zeros(n, dtype=np.int32)
    for i, t in enumerate(tapes):
        k = min(len(t), tape_words)
        tb[i, :k] = t[:k]; tl[i] = k
    return (torch.from_numpy(mb).cuda(), torch.from_numpy(np.array([len(m) for m in msgs], dtype=np.int32)).cuda(),
            torch.from_numpy(tb.view(np.int32)).cuda(), torch.from_numpy(tl).cuda())


def open_key(eng, G, tag='tesla0'):
    key = O.TeslaKey.from_golden(G, tag)
    t = eng.tesla(key.par)
    t.set_key(key.a1, key.a2, key.s, key.e1, key.e2, key.t1, key.t2)
    return key, t


@pytest.mark.parametrize('tag,nmsg', [('tesla0', 8), ('tesla1', 5)])
def test_sign_verify_reproduce_reference(eng, G, tag, nmsg):
    import torch
    key, t = open_key(eng, G, tag)
    msgs = [bytes(np.asarray(G[f'{tag}_msg{m}'], dtype=np.uint8)) for m in range(nmsg)]
    tapes = [u32(G[f'{tag}_tape{m}']) for m in range(nmsg)]
    mb, ml, tb, tl = pack(msgs, tapes, 8192)
    sigs, siglens, trials, status = t.sign(mb, tb, mlens=ml, tape_lens=tl)
    torch.cuda.synchronize()
    assert status.cpu().tolist() == [1] * nmsg
    assert trials.cpu().tolist() == [int(x) for x in G[f'{tag}_trials']]
    sh, lh = sigs.cpu().numpy(), siglens.cpu().numpy()
    for m in range(nmsg):
        assert bytes(sh[m, :lh[m]]) == bytes(np.asarray(G[f'{tag}_sig{m}'], dtype=np.uint8)), m
    assert t.verify(mb, sigs, siglens, mlens=ml).cpu().tolist() == [0] * nmsg
    bad = sigs.clone(); bad[:, key.sig_bytes // 3] ^= 0x08                       # the fixture's tampering: the reference's verdicts, whatever they are
    assert t.verify(mb, bad, siglens, mlens=ml).cpu().tolist() == [int(G[f'{tag}_verify'][2 * m + 1]) for m in range(nmsg)]
    bad = sigs.clone(); bad[:, key.sig_bytes - 1] ^= 1                           # the digest
    assert t.verify(mb, bad, siglens, mlens=ml).cpu().tolist() == [1] * nmsg
    bad = sigs.clone(); bad[:, 0] ^= 0x40                                        # z[0] outside the rejection bound
    assert t.verify(mb, bad, siglens, mlens=ml).cpu().tolist() == [1] * nmsg
    mb2 = mb.clone(); mb2[:, 0] ^= 1
    assert t.verify(mb2, sigs, siglens, mlens=ml).cpu().tolist() == [1] * nmsg
    t.close()


@pytest.mark.parametrize('tag', ['tesla0', 'tesla1'])
def test_batch_vs_oracle_seeded_tapes(eng, G, tag):
    import torch
    key, t = open_key(eng, G, tag)
    rng = np.random.default_rng(5)
    B, W = 300, 12 * 512
    msgs = [bytes(rng.integers(0, 256, size=int(rng.integers(1, 90)), dtype=np.uint8)) for _ in range(B)]
    tapes = [rng.integers(0, 2**32, size=W, dtype=np.uint64).astype(np.uint32) for _ in range(B)]
    tapes[3] = tapes[3][:300]                                                   # not even one attempt
    mb, ml, tb, tl = pack(msgs, tapes, W)
    sigs, siglens, trials, status = t.sign(mb, tb, mlens=ml, tape_lens=tl)
    torch.cuda.synchronize()
    sh, lh, tr, st = sigs.cpu().numpy(), siglens.cpu().numpy(), trials.cpu().numpy(), status.cpu().numpy()
    assert st[3] == 2 and lh[3] == 0
    for i in list(range(0, 30)) + [B - 1]:
        rc, sig, otr, used = O.tesla_sign(key, msgs[i], tapes[i])
        assert (rc, int(st[i])) in ((0, 1), (2, 2)), i
        assert tr[i] == otr, i
        if rc == 0:
            assert bytes(sh[i, :lh[i]]) == sig, i
    ok = st == 1
    v = t.verify(mb, sigs, siglens, mlens=ml).cpu().numpy()
    assert np.array_equal(v[ok], np.zeros(ok.sum(), dtype=v.dtype)) and v[3] == 1
    t.close()


def test_beyond_16k_slots_one_attempt_per_message_rounds(eng, G):
    """a batch above 16 K messages starts with ONE attempt per message per round (the slot rule of sc_hip_tesla_sign) and deepens as the batch thins out:
    a strided sample against the oracle - attempts included, and the messages whose tape ends first - and every signature verifies"""
    import torch
    key, t = open_key(eng, G, 'tesla0')
    B, W = 16500, 10 * 512
    g = torch.Generator(device='cuda'); g.manual_seed(77)
    mb = torch.randint(0, 256, (B, 48), dtype=torch.uint8, device='cuda', generator=g)
    tb = torch.randint(-2**31, 2**31 - 1, (B, W), dtype=torch.int32, device='cuda', generator=g)
    sigs, siglens, trials, status = t.sign(mb, tb, fixed_len=48)
    torch.cuda.synchronize()
    st = status.cpu().numpy()
    assert (st == 1).sum() > 0.95 * B and set(np.unique(st)) <= {1, 2}
    mh, th = mb.cpu().numpy(), tb.cpu().numpy().view(np.uint32)
    sh, lh, tr = sigs.cpu().numpy(), siglens.cpu().numpy(), trials.cpu().numpy()
    picks = list(range(0, B, 1500)) + [int(i) for i in np.nonzero(st == 2)[0][:3]]
    for i in picks:
        rc, sig, otr, used = O.tesla_sign(key, bytes(mh[i]), th[i])
        assert (rc, int(st[i])) in ((0, 1), (2, 2)), i
        assert tr[i] == otr, i
        if rc == 0:
            assert bytes(sh[i, :lh[i]]) == sig, i
    v = t.verify(mb, sigs, siglens, fixed_len=48).cpu().numpy()
    assert int(v[st == 1].sum()) == 0 and int((v[st == 2] == 1).all())
    t.close()
