"""GPU parity for the static-Huffman coding of one-polynomial signatures (C ABI: sc_hip_huffman_*): the reference's own raw / coded signature pairs
(tests/golden/huff_rows_ref.txt: Falcon-512 / -1024, ring-TESLA), the oracle's recoders on engine-made signatures at batch scale, sign -> encode ->
decode -> verify round trips, and streams that end early."""
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
def H():
    return load_golden('huff_rows_ref.txt')


def rows_of(blobs, stride):
    import torch
    a = np.zeros((len(blobs), stride), dtype=np.uint8)
    for i, b in enumerate(blobs):
        a[i, :len(b)] = np.frombuffer(b, dtype=np.uint8)
    return torch.from_numpy(a).cuda(), torch.from_numpy(np.array([len(b) for b in blobs], dtype=np.int32)).cuda()


@pytest.mark.parametrize('tag,count', [('falcon0', 4), ('falcon2', 2), ('tesla0', 4)])
def test_reference_pairs_both_ways(eng, H, tag, count):
    n, bits, tail, _ = (int(x) for x in H[f'{tag}_layout'])
    code, ln = O.huffman_table_for(H, bits)
    hr = eng.huffman_rows(code, ln, bits)
    raw = [bytes(H[f'{tag}_raw{i}'].astype(np.uint8)) for i in range(count)]
    coded = [bytes(H[f'{tag}_coded{i}'].astype(np.uint8)) for i in range(count)]
    stride = (len(raw[0]) + 15) & ~15
    rows, lens = rows_of(raw, stride)
    hr.encode(rows, lens, n, tail)
    rh, lh = rows.cpu().numpy(), lens.cpu().numpy()
    for i in range(count):
        assert bytes(rh[i, :lh[i]]) == coded[i], (tag, i)
        assert not rh[i, lh[i]:len(raw[0])].any()                       # nothing of the raw row is left behind the coded one
    crow, clen = rows_of(coded, stride)
    out, ol, bad = hr.decode(crow, clen, n, tail)
    oh = out.cpu().numpy()
    assert bad.cpu().tolist() == [0] * count and ol.cpu().tolist() == [len(raw[0])] * count
    for i in range(count):
        assert bytes(oh[i, :len(raw[0])]) == raw[i], (tag, i)
    clen2 = clen.clone(); clen2[0] = clen2[0] // 2                       # a stream that ends early is reported, nothing is read past it
    assert hr.decode(crow, clen2, n, tail)[2].cpu().tolist() == [1] + [0] * (count - 1)
    hr.close()


def test_falcon_sign_encode_decode_verify_at_batch_scale(eng, H):
    """2048 Falcon-512 signatures made by the engine: coded rows equal the oracle's recoding of the raw ones (a strided sample byte for byte, every
    length), decode gives the raw rows back, and the verifier accepts them; the coded signatures are ~16 % shorter"""
    import torch
    G = load_golden('falcon_ref.txt')
    key = O.FalconKey.from_golden(G, 'falcon0')
    f = eng.falcon(key.par); f.set_privkey(np.asarray(G['falcon0_sk'], dtype=np.int64).view(np.float64)); f.set_pubkey(G['falcon0_h'])
    B, n, bits = 2048, key.n, 12
    g = torch.Generator(device='cuda'); g.manual_seed(21)
    msgs = torch.randint(0, 256, (B, 64), dtype=torch.uint8, device='cuda', generator=g)
    tapes = torch.randint(-2**31, 2**31 - 1, (B, 4 * n), dtype=torch.int32, device='cuda', generator=g)
    sigs, siglens, status = f.sign(msgs, tapes, fixed_len=64)
    raw = sigs.clone()
    code, ln = O.huffman_table_for(H, bits)
    hr = eng.huffman_rows(code, ln, bits)
    st = status.clone()
    hr.encode(sigs, siglens, n, 0, status=st)
    assert torch.equal(st, status)                                        # no coefficient outside the alphabet
    rawh, ch, lh, sth = raw.cpu().numpy(), sigs.cpu().numpy(), siglens.cpu().numpy(), status.cpu().numpy()
    ok = sth == 1
    assert ok.sum() > B - 20 and 0.80 < lh[ok].mean() / f.sig_bytes < 0.90
    for i in list(range(0, B, 37)):
        if not ok[i]:
            continue
        rc, enc = O.huff_row_encode(bytes(rawh[i, :f.sig_bytes]), n, bits, 0, (code, ln))
        assert rc == 0 and bytes(ch[i, :lh[i]]) == enc, i
    out, ol, bad = hr.decode(sigs, siglens, n, 0)
    assert int(bad[status == 1].sum()) == 0
    assert torch.equal(out[status == 1][:, :f.sig_bytes], raw[status == 1][:, :f.sig_bytes])
    v = f.verify(msgs, out, ol, fixed_len=64)
    assert int(v[status == 1].sum()) == 0
    hr.close(); f.close()
