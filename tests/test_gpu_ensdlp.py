"""GPU parity for batched ENS / DLP signature verification (sc_hip_ens_dlp_sig_verify on a Falcon-shaped handle): the reference's verdicts on
its own signatures and on tampered copies (tests/golden/ensdlp_ref.txt), the oracle's on further tampering."""
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


@pytest.mark.parametrize('tag,nmsg', [('enssig0', 5), ('dlpsig1', 4)])
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
