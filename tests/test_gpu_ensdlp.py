"""GPU parity for batched ENS / DLP signatures.  Verification (sc_hip_ens_dlp_sig_verify on a Falcon-shaped handle): the reference's verdicts on
its own signatures and on tampered copies (tests/golden/ensdlp_ref.txt), the oracle's on further tampering.  Signing (sc_hip_ens_dlp_sig_sign):
the reference's signatures byte for byte given its PRNG word tape, the oracle's on seeded inputs."""
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
    return load_golden('ensdlp_ref.txt')


@pytest.mark.parametrize('tag,nmsg', [('enssig0', 5), ('dlpsig1', 4), ('dlpsig0', 4), ('enssig1', 3)])
def test_verdicts_match_reference_and_oracle(eng, G, tag, nmsg):
    import torch
    par, h = np.asarray(G[f'{tag}_params']), G[f'{tag}_h']
    hp = par.copy(); hp[4] = 0                                   # the handle's `set` field only selects the ring: 0 = x^n + 1
    f = eng.falcon(hp)
    f.set_pubkey(h)
    rng = np.random.default_rng(11)
    msgs, sigs, want = [], [], []
    for m in range(nmsg):
        msg = bytes(np.asarray(G[f'{tag}_msg{m}'], dtype=np.uint8))
        sig = bytearray(np.asarray(G[f'{tag}_sig{m}'], dtype=np.uint8))
        v = [int(x) for x in G[f'{tag}_verify'][3 * m:3 * m + 3]]
        msgs.append(msg); sigs.append(bytes(sig)); want.append(v[0])
        s1 = bytearray(sig); s1[len(sig) // 2] ^= 0x01
        msgs.append(msg); sigs.append(bytes(s1)); want.append(v[1])
        s2 = bytearray(sig); s2[3] ^= 0x80
        msgs.append(msg); sigs.append(bytes(s2)); want.append(v[2])
        for _ in range(6):                                       # random single-bit flips and a changed message: the oracle decides
            s3 = bytearray(sig); s3[int(rng.integers(0, len(sig)))] ^= 1 << int(rng.integers(0, 8))
            msgs.append(msg); sigs.append(bytes(s3)); want.append(O.ensdlp_verify(par, h, msg, bytes(s3)))
        msgs.append(msg + b'!'); sigs.append(bytes(sig)); want.append(O.ensdlp_verify(par, h, msg + b'!', bytes(sig)))
    B = len(msgs)
    stride = 160
    mb = np.zeros((B, stride), dtype=np.uint8); ml = np.zeros(B, dtype=np.int32)
    for i, x in enumerate(msgs):
        mb[i, :len(x)] = np.frombuffer(x, dtype=np.uint8); ml[i] = len(x)
    sb = np.stack([np.frombuffer(x, dtype=np.uint8) for x in sigs])
    sl = np.full(B, sb.shape[1], dtype=np.int32)
    sl[-1] -= 1; want[-1] = 1                                    # a truncated signature
    v = f.ens_dlp_verify(torch.from_numpy(mb).cuda(), torch.from_numpy(sb).cuda(), torch.from_numpy(sl).cuda(), mlens=torch.from_numpy(ml).cuda())
    torch.cuda.synchronize()
    assert v.cpu().tolist() == want
    assert want.count(0) >= nmsg
    f.close()


def _rows(items, stride):
    buf = np.zeros((len(items), stride), dtype=np.uint8); lens = np.zeros(len(items), dtype=np.int32)
    for i, x in enumerate(items):
        buf[i, :len(x)] = np.frombuffer(x, dtype=np.uint8); lens[i] = len(x)
    return buf, lens


@pytest.mark.parametrize('tag,nmsg', [('enssig0', 5), ('dlpsig1', 4), ('dlpsig0', 4), ('enssig1', 3)])
@pytest.mark.parametrize("lanes", [1, 4, 8, 16, 0, 8 + 0x200, 16 + 0x200])      # + 0x200: the register form with the table fetched in its own step (the large-batch form)
def test_signatures_match_reference_and_oracle(eng, G, tag, nmsg, lanes):
    """sc_hip_ens_dlp_sig_sign (Klein's sampler) byte for byte against the reference's signatures given its PRNG word
    tape, and against the oracle on further seeded messages / tapes; batch 77 crosses a wavefront boundary with a ragged tail.  Both forms of the
    kernel (a lane per signature with the state in memory; 4 / 8 / 16 lanes per signature with the state in registers) are forced in turn, then the default choice."""
    import torch
    par = np.asarray(G[f'{tag}_params'])
    n, q = int(par[0]), int(par[1])
    b, bgs, inv = O.ensdlp_basis(n, q, G[f'{tag}_fgFG'])
    s = eng.ens_dlp_signer(par)
    s.set_key(b, bgs, inv)
    if lanes and lanes != 1 and (lanes & 0xff) < n // 64:
        pytest.skip('fewer lanes than n / 64 state registers allow')
    s.set_mapping(lanes)                                          # 1: a lane per signature, state in memory; 4 / 8 / 16 lanes: state in registers; 0: default
    assert s.tape_words == 4 * n and s.sig_bytes == n * int(par[2]) // 8
    rng = np.random.default_rng(23)
    msgs, tapes, want = [], [], []
    for m in range(nmsg):
        msgs.append(bytes(np.asarray(G[f'{tag}_msg{m}'], dtype=np.uint8)))
        tapes.append(np.asarray(G[f'{tag}_tape{m}'], dtype=np.int64).astype(np.uint32))
        want.append(bytes(np.asarray(G[f'{tag}_sig{m}'], dtype=np.uint8)))
    while len(msgs) < 77:
        msg = bytes(rng.integers(0, 256, size=int(rng.integers(0, 150)), dtype=np.uint8))
        tp = rng.integers(0, 1 << 32, size=4 * n, dtype=np.uint64).astype(np.uint32)
        rc, sig = O.ensdlp_sign(par, b, bgs, inv, msg, tp)
        assert rc == 0
        msgs.append(msg); tapes.append(tp); want.append(sig)
    mb, ml = _rows(msgs, 160)
    tb = np.stack(tapes).view(np.int32)
    tl = np.full(len(msgs), 4 * n, dtype=np.int32)
    tl[-1] = 4 * n - 1                                           # one word short: status 2, no signature
    sigs, status = s.sign(torch.from_numpy(mb).cuda(), torch.from_numpy(tb).cuda(), mlens=torch.from_numpy(ml).cuda(), tape_lens=torch.from_numpy(tl).cuda())
    torch.cuda.synchronize()
    st = status.cpu().tolist()
    assert st[:-1] == [1] * (len(msgs) - 1) and st[-1] == 2
    got = sigs.cpu().numpy()
    for i in range(len(msgs) - 1):
        assert bytes(got[i]) == want[i], i
    assert not got[-1].any()
    # what was signed verifies under the public key (the Falcon-shaped verifier handle)
    hp = par.copy(); hp[4] = 0
    f = eng.falcon(hp)
    f.set_pubkey(G[f'{tag}_h'])
    sl = torch.full((len(msgs),), s.sig_bytes, dtype=torch.int32, device='cuda')
    v = f.ens_dlp_verify(torch.from_numpy(mb).cuda(), sigs, sl, mlens=torch.from_numpy(ml).cuda())
    assert v.cpu().tolist()[:-1] == [0] * (len(msgs) - 1)
    f.close()
    s.close()


@pytest.mark.parametrize('tag,nmsg', [('enssig0', 5), ('dlpsig1', 4), ('dlpsig0', 4), ('enssig1', 3)])
def test_signer_keyed_from_private_key_alone(eng, G, tag, nmsg):
    """sc_hip_ens_dlp_signer_set_privkey derives the basis, its Gram-Schmidt matrix and the row norms from f | g | F | G itself (the product's own
    host code, not the oracle's): the reference's signatures come out byte for byte"""
    import torch
    par = np.asarray(G[f'{tag}_params'])
    n = int(par[0])
    s = eng.ens_dlp_signer(par)
    s.set_privkey(G[f'{tag}_fgFG'])
    msgs = [bytes(np.asarray(G[f'{tag}_msg{m}'], dtype=np.uint8)) for m in range(nmsg)]
    mb, ml = _rows(msgs, 160)
    tb = np.stack([np.asarray(G[f'{tag}_tape{m}'], dtype=np.int64).astype(np.uint32) for m in range(nmsg)]).view(np.int32)
    sigs, status = s.sign(torch.from_numpy(mb).cuda(), torch.from_numpy(tb).cuda(), mlens=torch.from_numpy(ml).cuda())
    torch.cuda.synchronize()
    assert status.cpu().tolist() == [1] * nmsg
    for m in range(nmsg):
        assert bytes(sigs[m].cpu().numpy()) == bytes(np.asarray(G[f'{tag}_sig{m}'], dtype=np.uint8)), m
    s.close()


def test_signer_rejects_bad_arguments(eng, G):
    import torch
    from libsafecrypto_amd import EngineError
    par = np.asarray(G['dlpsig0_params'])
    n, q = int(par[0]), int(par[1])
    bad = par.copy(); bad[0] = 128
    with pytest.raises(EngineError):
        eng.ens_dlp_signer(bad)
    s = eng.ens_dlp_signer(par)
    msgs = torch.zeros((2, 16), dtype=torch.uint8, device='cuda')
    tapes = torch.zeros((2, 4 * n), dtype=torch.int32, device='cuda')
    with pytest.raises(EngineError):
        s.sign(msgs, tapes, fixed_len=4)                         # no key loaded
    b, bgs, inv = O.ensdlp_basis(n, q, G['dlpsig0_fgFG'])
    s.set_key(b, bgs, inv)
    with pytest.raises(EngineError):
        s.sign(msgs, tapes[:, :4 * n - 2].contiguous(), fixed_len=4)      # rows shorter than 4n words
    s.close()
