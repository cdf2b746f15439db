"""GPU parity for batched DLP-IBE extract (C ABI: sc_hip_dlp_ibe_extract on a Falcon-shaped handle): user secret keys byte-identical to
the ones the reference extracted under its own master key from the same CSPRNG words (tests/golden/dlp_ref.txt), and to the pinned
oracle on seeded tapes."""
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
    return load_golden('dlp_ref.txt')


def u32(a):
    return (np.asarray(a).astype(np.int64) & 0xFFFFFFFF).astype(np.uint32)


def open_master(eng, G, T):
    par = np.asarray(G[T + '_params'])
    n, q, q_bits, n_bits, set_, ubits = (int(x) for x in par[:6])
    fp = np.array([n, q, q_bits, n_bits, 0, 0, 0, 0], dtype=np.int32)       # Falcon-shaped parameter record; bd unused by extract
    f = eng.falcon(fp)
    tree = np.asarray(G[T + '_tree'], dtype=np.int64).view(np.float64)
    f.set_privkey(tree)
    return f, par, tree, ubits


@pytest.mark.parametrize('T', ['dlp0', 'dlp1'])
def test_extract_reproduces_reference_and_oracle(eng, G, T):
    import torch
    f, par, tree, ubits = open_master(eng, G, T)
    n = f.n
    rng = np.random.default_rng(0xd1b)
    B = 90
    ids = [bytes(np.asarray(G[f'{T}_id{m}'], dtype=np.uint8)) for m in range(4)] + [bytes(rng.integers(0, 256, size=int(rng.integers(1, 64)), dtype=np.uint8)) for _ in range(B - 4)]
    tapes = rng.integers(0, 2**32, size=(B, 4 * n), dtype=np.uint64).astype(np.uint32)
    lens = np.full(B, 4 * n, dtype=np.int32)
    for m in range(4):
        tapes[m] = u32(G[f'{T}_tape{m}'])
    lens[7] = 4 * n - 4                                            # one leaf short
    stride = 80
    ib = np.zeros((B, stride), dtype=np.uint8); il = np.zeros(B, dtype=np.int32)
    for i, x in enumerate(ids):
        ib[i, :len(x)] = np.frombuffer(x, dtype=np.uint8); il[i] = len(x)
    sk, st = f.dlp_extract(ubits, torch.from_numpy(ib).cuda(), torch.from_numpy(tapes.view(np.int32)).cuda(), idlens=torch.from_numpy(il).cuda(),
                           tape_lens=torch.from_numpy(lens).cuda())
    torch.cuda.synchronize()
    skn, stn = sk.cpu().numpy(), st.cpu().numpy()
    for m in range(4):
        assert stn[m] == 1 and bytes(skn[m]) == bytes(np.asarray(G[f'{T}_sk{m}'], dtype=np.uint8)), m
    assert stn[7] == 2
    for i in range(4, B):
        if i == 7:
            continue
        rc, osk = O.dlp_extract(par, tree, ids[i], tapes[i])
        assert rc == 0 and stn[i] == 1 and bytes(skn[i]) == osk, i
    f.close()


@pytest.mark.parametrize('T', ['dlp0', 'dlp1'])
def test_extract_both_lane_mappings_agree(eng, G, T):
    """a batch beyond the 8-lanes-per-identity range of the sampler takes the one-lane kernels: same keys for the same inputs"""
    import torch
    f, par, tree, ubits = open_master(eng, G, T)
    n = f.n
    g = torch.Generator(device='cuda'); g.manual_seed(9)
    B = 20000
    ids = torch.randint(0, 256, (B, 16), dtype=torch.uint8, device='cuda', generator=g)
    tapes = torch.randint(-2**31, 2**31 - 1, (B, 4 * n), dtype=torch.int32, device='cuda', generator=g)
    sk, st = f.dlp_extract(ubits, ids, tapes, fixed_len=16)
    sk2, st2 = f.dlp_extract(ubits, ids[:64].contiguous(), tapes[:64].contiguous(), fixed_len=16)
    torch.cuda.synchronize()
    assert int((st == 1).sum()) == B and torch.equal(sk[:64], sk2)
    for i in (0, 63):
        rc, osk = O.dlp_extract(par, tree, bytes(ids[i].cpu().numpy()), (tapes[i].cpu().numpy().astype(np.int64) & 0xFFFFFFFF).astype(np.uint32))
        assert rc == 0 and bytes(sk[i].cpu().numpy()) == osk
    f.close()


@pytest.mark.parametrize('T', ['dlp0', 'dlp1'])
def test_encrypt_decrypt_reproduce_reference_and_oracle(eng, G, T):
    """dlp_ibe_encrypt / _decrypt: the reference's ciphertexts from its own CSPRNG words, seeded batches against the oracle (one identity
    per message), decryption under the matching user key, a short tape reported"""
    import torch
    par = np.asarray(G[T + '_params'])
    d = eng.dlp(par)
    d.set_pubkey(G[T + '_h_ntt'])
    n = d.n
    rng = np.random.default_rng(0xd1e)
    B = 75
    W = d.enc_words + 3
    ids = [bytes(np.asarray(G[f'{T}_id{m}'], dtype=np.uint8)) for m in range(4)]
    ids += [ids[2]] * (B - 4)                                    # the rest go to identity 2, whose user key is loaded below
    msgs = rng.integers(0, 256, size=(B, n // 8), dtype=np.uint8)
    tapes = rng.integers(0, 2**32, size=(B, W), dtype=np.uint64).astype(np.uint32)
    lens = np.full(B, W, dtype=np.int32)
    for m in range(4):
        t = u32(G[f'{T}_etape{m}']); tapes[m, :len(t)] = t; lens[m] = len(t)
        msgs[m] = np.asarray(G[f'{T}_msg{m}'], dtype=np.uint8)
    lens[9] = 60                                                  # runs out inside the reservoir draws
    ib = np.zeros((B, 80), dtype=np.uint8); il = np.zeros(B, dtype=np.int32)
    for i, x in enumerate(ids):
        ib[i, :len(x)] = np.frombuffer(x, dtype=np.uint8); il[i] = len(x)
    ct, st = d.encrypt(torch.from_numpy(ib).cuda(), torch.from_numpy(msgs).cuda(), torch.from_numpy(tapes.view(np.int32)).cuda(),
                       idlens=torch.from_numpy(il).cuda(), tape_lens=torch.from_numpy(lens).cuda())
    torch.cuda.synchronize()
    ctn, stn = ct.cpu().numpy(), st.cpu().numpy()
    for m in range(4):
        assert stn[m] == 1 and bytes(ctn[m]) == bytes(np.asarray(G[f'{T}_ct{m}'], dtype=np.uint8)), m
    assert stn[9] == 2
    for i in range(4, B):
        if i == 9:
            continue
        rc, oc = O.dlp_encrypt(par, G[T + '_h_ntt'], ids[i], bytes(msgs[i]), tapes[i])
        assert rc == 0 and stn[i] == 1 and bytes(ctn[i]) == oc, i
    d.set_user_key(G[T + '_uk_ntt2'])
    dec = d.decrypt(ct).cpu().numpy()
    for i in [2] + list(range(4, B)):
        if i == 9:
            continue
        assert bytes(dec[i]) == bytes(msgs[i]) == O.dlp_decrypt(par, G[T + '_uk_ntt2'], bytes(ctn[i]))[1], i
    junk = rng.integers(0, 256, size=(16, d.ct_bytes), dtype=np.uint8)
    dj = d.decrypt(torch.from_numpy(junk).cuda()).cpu().numpy()
    for i in range(16):
        assert bytes(dj[i]) == O.dlp_decrypt(par, G[T + '_uk_ntt2'], bytes(junk[i]))[1], i
    d.close()
