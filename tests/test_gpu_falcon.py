"""GPU parity for the batched Falcon tree sampler / sign / verify (C ABI: sc_hip_falcon_*): s1, s2 and signature bytes
identical to what the reference produced from the same (key, target or message, CSPRNG word tape) - tests/golden/falcon_ref.txt -
and to the pinned oracle on seeded batches."""
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
    return load_golden('falcon_ref.txt')


def u32(a):
    return (np.asarray(a).astype(np.int64) & 0xFFFFFFFF).astype(np.uint32)


def open_key(eng, G, tag):
    key = O.FalconKey.from_golden(G, tag)
    f = eng.falcon(key.par)
    f.set_privkey(key.sk)
    f.set_pubkey(key.h)
    return key, f


def pack_msgs(msgs):
    import torch
    stride = max(8, (max(len(m) for m in msgs) + 7) & ~7)
    mb = np.zeros((len(msgs), stride), dtype=np.uint8)
    for i, m in enumerate(msgs):
        mb[i, :len(m)] = np.frombuffer(m, dtype=np.uint8)
    return torch.from_numpy(mb).cuda(), torch.from_numpy(np.array([len(m) for m in msgs], dtype=np.int32)).cuda()


@pytest.mark.parametrize('set_,nsamp', [(0, 2), (2, 1)])
def test_sampler_reproduces_reference(eng, G, set_, nsamp):
    import torch
    tag = f'falcon{set_}'
    key, f = open_key(eng, G, tag)
    c = np.stack([np.asarray(G[f'{tag}_c{m}'], dtype=np.int32) for m in range(nsamp)])
    tp = np.stack([u32(G[f'{tag}_ctape{m}']) for m in range(nsamp)])
    s1, s2, status = f.sample(torch.from_numpy(c).cuda(), torch.from_numpy(tp.view(np.int32)).cuda())
    torch.cuda.synchronize()
    assert status.cpu().tolist() == [0] * nsamp
    for m in range(nsamp):
        assert np.array_equal(s1[m].cpu().numpy(), np.asarray(G[f'{tag}_s1_{m}'], dtype=np.int32)), m
        assert np.array_equal(s2[m].cpu().numpy(), np.asarray(G[f'{tag}_s2_{m}'], dtype=np.int32)), m
    f.close()


@pytest.mark.parametrize('set_,nmsg', [(0, 4), (2, 2)])
def test_sign_verify_reproduce_reference(eng, G, set_, nmsg):
    import torch
    tag = f'falcon{set_}'
    key, f = open_key(eng, G, tag)
    msgs = [bytes(np.asarray(G[f'{tag}_msg{m}'], dtype=np.uint8)) for m in range(nmsg)]
    mb, ml = pack_msgs(msgs)
    tp = np.stack([u32(G[f'{tag}_tape{m}']) for m in range(nmsg)])
    sigs, siglens, status = f.sign(mb, torch.from_numpy(tp.view(np.int32)).cuda(), mlens=ml)
    torch.cuda.synchronize()
    assert status.cpu().tolist() == [1] * nmsg
    sh, lh = sigs.cpu().numpy(), siglens.cpu().numpy()
    for m in range(nmsg):
        assert bytes(sh[m, :lh[m]]) == bytes(np.asarray(G[f'{tag}_sig{m}'], dtype=np.uint8)), m
    assert f.workspace_residue() == 0        # the sampler's working polynomials and s1 | s2 before packing are zeroed inside the call
    f.sig_stride = key.sig_bytes + 1         # rows that do not start on word boundaries take the byte-wise packer: same bytes
    sg2, sl2, _ = f.sign(mb, torch.from_numpy(tp.view(np.int32)).cuda(), mlens=ml)
    assert torch.equal(sl2, siglens) and all(bytes(sg2[m, :lh[m]].cpu().numpy()) == bytes(sh[m, :lh[m]]) for m in range(nmsg))
    f.sig_stride = (key.sig_bytes + 15) & ~15
    assert f.verify(mb, sigs, siglens, mlens=ml).cpu().tolist() == [int(G[f'{tag}_verify'][2 * m]) for m in range(nmsg)] == [0] * nmsg
    bad = sigs.clone(); bad[:, key.sig_bytes // 3] ^= 0x10          # the fixture's tampering
    assert f.verify(mb, bad, siglens, mlens=ml).cpu().tolist() == [int(G[f'{tag}_verify'][2 * m + 1]) for m in range(nmsg)] == [1] * nmsg
    f.close()


def test_fresh_keys_and_signatures_from_the_reference_run_here(eng, tmp_path):
    """the reference run on the spot (oracle/_ref/ref_driver falcon): fresh key pairs for sets 0 and 2 (Falcon-512 / -1024) from its own keygen,
    its sampler outputs and signatures with the CSPRNG words they took - a fixture nobody has seen; both tests above run against it"""
    import os, subprocess
    drv = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'oracle', '_ref', 'ref_driver')
    if not os.path.exists(drv):
        pytest.skip('oracle/_ref/ref_driver not built')
    subprocess.check_call([drv, 'falcon', str(tmp_path)], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    L = load_golden(str(tmp_path / 'falcon_ref.txt'))
    for set_, nsamp, nmsg in ((0, 2, 4), (2, 1, 2)):
        test_sampler_reproduces_reference(eng, L, set_, nsamp)
        test_sign_verify_reproduce_reference(eng, L, set_, nmsg)


def test_batch_vs_oracle_seeded(eng, G):
    """a ragged batch (not a multiple of the workgroup), seeded tapes: every signature equals the oracle's, short tapes are
    reported, signatures verify, a wrong message does not"""
    import torch
    key, f = open_key(eng, G, 'falcon0')
    rng = np.random.default_rng(99)
    B = 333
    msgs = [bytes(rng.integers(0, 256, size=int(rng.integers(1, 120)), dtype=np.uint8)) for _ in range(B)]
    tp = rng.integers(0, 2**32, size=(B, 4 * key.n), dtype=np.uint64).astype(np.uint32)
    tl = np.full(B, 4 * key.n, dtype=np.int32); tl[7] = 4 * key.n - 1
    mb, ml = pack_msgs(msgs)
    sigs, siglens, status = f.sign(mb, torch.from_numpy(tp.view(np.int32)).cuda(), mlens=ml, tape_lens=torch.from_numpy(tl).cuda())
    torch.cuda.synchronize()
    sh, lh, st = sigs.cpu().numpy(), siglens.cpu().numpy(), status.cpu().numpy()
    assert st[7] == 2 and lh[7] == 0
    for i in list(range(0, 40)) + [B - 1]:
        if i == 7:
            continue
        rc, sig = O.falcon_sign(key, msgs[i], tp[i])
        assert (rc, st[i]) in ((0, 1), (1, 4)), i
        if rc == 0:
            assert bytes(sh[i, :lh[i]]) == sig, i
    ok = st == 1
    assert ok.sum() >= B - 5
    v = f.verify(mb, sigs, siglens, mlens=ml).cpu().numpy()
    assert np.array_equal(v[ok], np.zeros(ok.sum(), dtype=v.dtype)) and v[7] == 1
    mb2 = mb.clone(); mb2[:, 0] ^= 1
    assert f.verify(mb2, sigs, siglens, mlens=ml).cpu().numpy().min() == 1
    f.close()


@pytest.mark.parametrize('lanes,variant', [(1, None), (8, None), (64, (8 | 1 << 8, 7)), (64, (8 | 2 << 8, 7)), (64, (8, 6)), (64, (4, 6)), (64, (0, 0))])
def test_every_mapping_reproduces_reference(eng, G, lanes, variant):
    """the ways of laying the sampler over the machine (sc_hip_falcon_set_mapping: a lane per signature, 8 lanes per signature, a wavefront
    per 8 or 4 signatures with the tree below 128 or 64 coefficients in LDS, one or two wavefronts per workgroup (spw | nw << 8) -
    sc_hip_falcon_set_variant; (0, 0) = what the launcher picks);
    each must give the reference's s1, s2 and, on a seeded ragged batch, the oracle's, for n = 1024 and n = 512"""
    import torch
    for tag in ('falcon2', 'falcon0'):
        key, f = open_key(eng, G, tag)
        f.set_mapping(lanes)
        if variant:
            f.set_variant(*variant)
        c = np.asarray(G[f'{tag}_c0'], dtype=np.int32)[None, :]
        tp = u32(G[f'{tag}_ctape0'])[None, :]
        s1, s2, status = f.sample(torch.from_numpy(c).cuda(), torch.from_numpy(tp.view(np.int32)).cuda())
        assert np.array_equal(s1[0].cpu().numpy(), np.asarray(G[f'{tag}_s1_0'], dtype=np.int32))
        assert np.array_equal(s2[0].cpu().numpy(), np.asarray(G[f'{tag}_s2_0'], dtype=np.int32))
        rng = np.random.default_rng(lanes + (10 * variant[0] + variant[1] if variant else 0))
        B = 77                                                   # ragged: the last wavefront is partly empty in every mapping
        cs = rng.integers(0, key.q, size=(B, key.n), dtype=np.int32)
        tps = rng.integers(0, 2**32, size=(B, 4 * key.n), dtype=np.uint64).astype(np.uint32)
        tl = np.full(B, 4 * key.n, dtype=np.int32); tl[5] = 17; tl[76] = 4 * key.n - 1
        s1, s2, status = f.sample(torch.from_numpy(cs).cuda(), torch.from_numpy(tps.view(np.int32)).cuda(), tape_lens=torch.from_numpy(tl).cuda())
        s1, s2, st = s1.cpu().numpy(), s2.cpu().numpy(), status.cpu().numpy()
        assert st[5] == 2 and st[76] == 2 and int((st == 0).sum()) == B - 2
        for i in (0, 1, 4, 6, 38, 72, 75):
            rc, o1, o2 = O.falcon_sample(key, cs[i], tps[i])
            assert rc == 0 and np.array_equal(s1[i], o1) and np.array_equal(s2[i], o2), (tag, i)
        f.close()


def test_tapes_narrower_than_a_signature_needs(eng, G):
    """a tape row pitch below 4n words: every signature reports status 2, nothing is read past a row"""
    import torch
    key, f = open_key(eng, G, 'falcon0')
    B = 9
    cs = np.zeros((B, key.n), dtype=np.int32)
    tps = np.zeros((B, 4 * key.n - 4), dtype=np.int32)
    s1, s2, status = f.sample(torch.from_numpy(cs).cuda(), torch.from_numpy(tps).cuda())
    assert status.cpu().tolist() == [2] * B
    f.close()


def test_full_size_config5(eng, G):
    """BASELINE configs[4] as stated: Falcon-1024 (reference set 2), batch 8192, the default mapping (no override).  A strided sample of
    the signatures equals the oracle's byte for byte, every signature verifies, every flipped message is rejected, and the two first-
    generation mappings agree with the default one on the whole batch."""
    import torch
    key, f = open_key(eng, G, 'falcon2')
    B = 8192
    g = torch.Generator(device='cuda'); g.manual_seed(55)
    msgs = torch.randint(0, 256, (B, 64), dtype=torch.uint8, device='cuda', generator=g)
    tapes = torch.randint(-2**31, 2**31 - 1, (B, 4 * key.n), dtype=torch.int32, device='cuda', generator=g)
    sigs, siglens, status = f.sign(msgs, tapes, fixed_len=64)
    torch.cuda.synchronize()
    st, lh = status.cpu().numpy(), siglens.cpu().numpy()
    ok = st == 1
    assert set(np.unique(st)) <= {1, 4} and ok.sum() >= B - B // 200        # 4 = the reference's norm-bound failure (it does not retry)
    mh, th, sh = msgs.cpu().numpy(), tapes.cpu().numpy().view(np.uint32), sigs.cpu().numpy()
    for i in list(range(0, B, 31)) + [B - 1]:                 # 266 of the 8192 signatures, byte for byte against the oracle
        rc, sig = O.falcon_sign(key, bytes(mh[i]), th[i])
        assert (rc, st[i]) in ((0, 1), (1, 4)), i
        if rc == 0:
            assert bytes(sh[i, :lh[i]]) == sig, i
    assert f.workspace_residue() == 0                         # (the sampler and the finishing kernel zero what they are the last readers of)
    v = f.verify(msgs, sigs, siglens, fixed_len=64).cpu().numpy()
    assert np.array_equal(v[ok], np.zeros(int(ok.sum()), dtype=v.dtype)) and (v[~ok] == 1).all()
    m2 = msgs.clone(); m2[:, 63] ^= 0x80
    assert int(f.verify(m2, sigs, siglens, fixed_len=64).min()) == 1
    for lanes in (1, 8):
        f.set_mapping(lanes)
        s2_, l2_, st2_ = f.sign(msgs, tapes, fixed_len=64)
        assert torch.equal(s2_, sigs) and torch.equal(l2_, siglens) and torch.equal(st2_, status), lanes
        assert f.workspace_residue() == 0, lanes              # (these mappings leave the workspace to the finishing kernel)
    f.close()


def test_distinct_keys_from_the_reference_keygen(eng, tmp_path):
    """SURVEY 8(d) distinct keys: 64 Falcon-512 key pairs generated HERE by the reference's own keygen (oracle/_ref/ref_driver falcon_keys; the
    binary travels to the GPU box), each with a signature the reference produced under it and the CSPRNG words that took.  The key set is loaded
    once; a batch naming a key per message reproduces every reference signature, the oracle's on seeded tapes, verifies under the right keys only,
    and reports an index outside the set"""
    import os, subprocess, torch
    drv = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'oracle', '_ref', 'ref_driver')
    if not os.path.exists(drv):
        pytest.skip('oracle/_ref/ref_driver not built')
    NK = 64
    out = os.path.join(str(tmp_path), 'fk.txt')
    subprocess.check_call([drv, 'falcon_keys', '0', str(NK), out], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    K = load_golden(out)
    keys = [O.FalconKey(K['params'], K[f'k{i}_sk'], K[f'k{i}_h']) for i in range(NK)]
    n = keys[0].n
    f = eng.falcon(keys[0].par)
    f.set_privkeys(np.stack([k.sk for k in keys]))
    f.set_pubkeys(np.stack([k.h for k in keys]))
    # (a) the reference's own signature under every key, then (b) three more messages per key on seeded tapes, keys interleaved
    rng = np.random.default_rng(5)
    msgs = [bytes(np.asarray(K[f'k{i}_msg'], dtype=np.uint8)) for i in range(NK)]
    tapes = [u32(K[f'k{i}_tape'])[:4 * n] for i in range(NK)]
    key_of = list(range(NK))
    for r in range(3):
        for i in range(NK):
            msgs.append(bytes(rng.integers(0, 256, size=int(rng.integers(1, 100)), dtype=np.uint8)))
            tapes.append(rng.integers(0, 2**32, size=4 * n, dtype=np.uint64).astype(np.uint32))
            key_of.append((i * 7 + r) % NK)
    msgs.append(b'x'); tapes.append(tapes[0]); key_of.append(NK)              # no such key
    B = len(msgs)
    mb, ml = pack_msgs(msgs)
    ko = torch.tensor(key_of, dtype=torch.int32, device='cuda')
    tp = torch.from_numpy(np.stack(tapes).view(np.int32)).cuda()
    sigs, siglens, status = f.sign_keys(ko, mb, tp, mlens=ml)
    torch.cuda.synchronize()
    sh, lh, st = sigs.cpu().numpy(), siglens.cpu().numpy(), status.cpu().numpy()
    assert st[B - 1] == 6 and lh[B - 1] == 0
    for i in range(NK):
        want = bytes(np.asarray(K[f'k{i}_sig'], dtype=np.uint8))
        assert (int(K[f'k{i}_status'][0]) == 0) == (st[i] == 1), i
        if st[i] == 1:
            assert bytes(sh[i, :lh[i]]) == want, i
    for i in range(NK, B - 1):
        rc, sig = O.falcon_sign(keys[key_of[i]], msgs[i], tapes[i])
        assert (rc, st[i]) in ((0, 1), (1, 4)), i
        if rc == 0:
            assert bytes(sh[i, :lh[i]]) == sig, i
    ok = torch.from_numpy(st == 1).cuda()
    assert int(ok.sum()) >= B - 6
    v = f.verify_keys(ko, mb, sigs, siglens, mlens=ml)
    assert int(v[ok].sum()) == 0 and int(v[B - 1]) == 1
    wrong = ((ko + 1) % NK).to(torch.int32).contiguous()
    assert int((f.verify_keys(wrong, mb, sigs, siglens, mlens=ml) == 0).sum()) == 0
    # messages grouped by key (runs of 8 and more: the wavefront-per-8 form serves a workgroup whose signatures share a key) with a ragged run in the
    # middle (those workgroups fall back to the 8-lanes-per-signature form): same bytes as signing each message alone under its key
    runs = [(k, 16) for k in range(0, NK, 3)] + [(5, 3), (9, 13)] + [(k, 8) for k in range(1, NK, 7)]
    gk = [k for k, cnt in runs for _ in range(cnt)]
    GB = len(gk)
    gm = [bytes(rng.integers(0, 256, size=40, dtype=np.uint8)) for _ in range(GB)]
    gt = rng.integers(0, 2**32, size=(GB, 4 * n), dtype=np.uint64).astype(np.uint32)
    gmb, gml = pack_msgs(gm)
    gko = torch.tensor(gk, dtype=torch.int32, device='cuda')
    gs, gl, gst = f.sign_keys(gko, gmb, torch.from_numpy(gt.view(np.int32)).cuda(), mlens=gml)
    gsh, glh, gsth = gs.cpu().numpy(), gl.cpu().numpy(), gst.cpu().numpy()
    for i in list(range(0, GB, 5)) + list(range(GB - 20, GB)):
        rc, sig = O.falcon_sign(keys[gk[i]], gm[i], gt[i])
        assert (rc, gsth[i]) in ((0, 1), (1, 4)), i
        if rc == 0:
            assert bytes(gsh[i, :glh[i]]) == sig, i
    assert int(f.verify_keys(gko, gmb, gs, gl, mlens=gml)[torch.from_numpy(gsth == 1).cuda()].sum()) == 0
    # fixed-length messages and full tapes (no length arrays), keys interleaved so that the call regroups the batch by key: the same signatures as the
    # same messages presented grouped by key (which are signed where they lie)
    FB = 8 * NK + 5
    fm = torch.from_numpy(rng.integers(0, 256, size=(FB, 48), dtype=np.uint8)).cuda()
    ft = torch.from_numpy(rng.integers(0, 2**32, size=(FB, 4 * n), dtype=np.uint64).astype(np.uint32).view(np.int32)).cuda()
    fk = (torch.arange(FB, device='cuda') * 5 % NK).to(torch.int32)
    s_il, l_il, st_il = f.sign_keys(fk, fm, ft, fixed_len=48)
    order = torch.argsort(fk.to(torch.int64), stable=True)
    s_gr, l_gr, st_gr = f.sign_keys(fk[order].contiguous(), fm[order].contiguous(), ft[order].contiguous(), fixed_len=48)
    assert torch.equal(st_il[order], st_gr) and torch.equal(l_il[order], l_gr) and int((st_gr == 1).sum()) >= FB - 8
    assert torch.equal(s_il[order], s_gr)
    assert int(f.verify_keys(fk, fm, s_il, l_il, fixed_len=48)[st_il == 1].sum()) == 0
    # the one-key calls use key 0 of the set
    s0, l0, t0 = f.sign(mb[:1], tp[:1], mlens=ml[:1])
    assert int(t0[0]) == st[0] and bytes(s0[0, :int(l0[0])].cpu().numpy()) == bytes(sh[0, :lh[0]])
    f.close()
