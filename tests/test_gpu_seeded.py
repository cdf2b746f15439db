"""GPU parity for the SEEDED signing entries (C ABI: sc_hip_{bliss,falcon,tesla}_sign_seeded): message i draws from the reference's AES-256
CTR-DRBG seeded with row i inside the call (prng.c:963-1107, ctr_drbg.c:37-197).  The device generator itself is pinned against the reference's
streams in test_gpu_drbg.py; here every seeded call must give, byte for byte, what the tape-fed call gives over the expanded streams of the same
seeds - whose signatures are pinned against the reference's in test_gpu_{bliss,falcon,tesla}.py - and the oracle's generator must agree on the tapes."""
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


def _batch(B, seed, seed_len=32):
    import torch
    g = torch.Generator(device='cuda'); g.manual_seed(seed)
    msgs = torch.randint(0, 256, (B, 64), dtype=torch.uint8, device='cuda', generator=g)
    seeds = torch.randint(0, 256, (B, seed_len), dtype=torch.uint8, device='cuda', generator=g)
    return msgs, seeds


def test_tapes_of_the_seeds_match_the_oracle(eng):
    """the word tapes the seeded calls stand for: device expansion (the lanes-per-generator kernel, every width) against the oracle's CTR-DRBG"""
    import torch
    rng = np.random.default_rng(77)
    for B, nwords in ((3, 4), (5, 260), (64, 2052), (700, 4096), (9, 16384)):
        seeds = rng.integers(0, 256, size=(B, 32), dtype=np.uint8)
        got = eng.drbg_words(torch.from_numpy(seeds).cuda(), nwords).cpu().numpy().view(np.uint32)
        for i in (0, B // 2, B - 1):
            want = O.drbg_words(seeds[i].tobytes(), nwords)
            assert np.array_equal(got[i], want), (B, nwords, i)


@pytest.mark.parametrize('B', [1, 37, 3000])
@pytest.mark.parametrize('spec', [1, 0, 2, 3])
def test_bliss_seeded_equals_tape_fed(eng, B, spec):
    import torch
    G = load_golden('bliss_ref.txt')
    b = eng.bliss(G['bliss1_params']); b.set_key(G['bliss1_f'], G['bliss1_g'], G['bliss1_a'])
    b.set_speculation(spec)
    msgs, seeds = _batch(B, 100 + B)
    W = (int(b.L.sc_hip_bliss_tape_words(512, 26)) + 3) & ~3
    tapes = eng.drbg_words(seeds, W)
    s0, l0, t0, st0 = b.sign(msgs, tapes, fixed_len=64)
    s1, l1, t1, st1 = b.sign_seeded(msgs, seeds, fixed_len=64)
    assert torch.equal(st0, st1) and torch.equal(l0, l1) and torch.equal(t0, t1) and torch.equal(s0, s1)
    assert int((st1 == 1).sum()) >= B - 1
    if B >= 3000:
        assert int(t1.max()) > 2                      # some message went on to its later pairs: the list-driven fills ran
    v = b.verify(msgs, s1, l1, fixed_len=64)
    assert int(v[st1 == 1].sum()) == 0
    # a second call on the same handle (rows zeroed behind the first) gives the same bytes
    s2, l2, t2, st2 = b.sign_seeded(msgs, seeds, fixed_len=64)
    assert torch.equal(s1, s2) and torch.equal(t1, t2)
    # the handle's tape rows hold nothing after the call
    b.close()


def test_bliss_seeded_signature_is_the_oracles(eng):
    """one step further than tape equality: the oracle signs from the oracle's own expansion of the seed"""
    import torch
    G = load_golden('bliss_ref.txt')
    key = O.BlissKey.from_golden(G, 'bliss1')
    b = eng.bliss(key.par); b.set_key(key.f, key.g, key.a)
    msgs, seeds = _batch(6, 5)
    s1, l1, t1, st1 = b.sign_seeded(msgs, seeds, fixed_len=64)
    for i in range(6):
        tape = O.drbg_words(seeds[i].cpu().numpy().tobytes(), 4 * 2100)
        rc, want, trials, used = O.bliss_sign(key, msgs[i].cpu().numpy().tobytes(), tape)
        assert rc == 0 and int(st1[i]) == 1 and int(t1[i]) == trials
        assert bytes(s1[i, :int(l1[i])].cpu().numpy()) == want
    b.close()


@pytest.mark.parametrize('tag,B', [('falcon0', 50), ('falcon2', 9), ('falcon2', 1100)])
def test_falcon_seeded_equals_tape_fed(eng, tag, B):
    import torch
    G = load_golden('falcon_ref.txt')
    f = eng.falcon(G[f'{tag}_params'])
    f.set_privkey(np.asarray(G[f'{tag}_sk'], dtype=np.int64).view(np.float64)); f.set_pubkey(G[f'{tag}_h'])
    msgs, seeds = _batch(B, 200 + B, seed_len=48)
    tapes = eng.drbg_words(seeds, 4 * f.n)
    s0, l0, st0 = f.sign(msgs, tapes, fixed_len=64)
    s1, l1, st1 = f.sign_seeded(msgs, seeds, fixed_len=64)
    assert torch.equal(st0, st1) and torch.equal(l0, l1) and torch.equal(s0, s1)
    ok = st1 == 1
    assert int(ok.sum()) >= B - max(1, B // 50)
    v = f.verify(msgs, s1, l1, fixed_len=64)
    assert int(v[ok].sum()) == 0
    f.close()


@pytest.mark.parametrize('B', [1, 33, 2500])
def test_tesla_seeded_equals_tape_fed(eng, B):
    import torch
    G = load_golden('tesla_ref.txt')
    ts = eng.tesla(G['tesla0_params'])
    ts.set_key(*(G[f'tesla0_{x}'] for x in ('a1', 'a2', 's', 'e1', 'e2', 't1', 't2')))
    msgs, seeds = _batch(B, 300 + B, seed_len=16)
    tapes = eng.drbg_words(seeds, 64 * 512)
    s0, l0, t0, st0 = ts.sign(msgs, tapes, fixed_len=64)
    s1, l1, t1, st1 = ts.sign_seeded(msgs, seeds, fixed_len=64)
    done = st0 == 1                                         # (a message past 64 attempts on the tape - p ~ 2e-11 - would differ by construction)
    assert int(done.sum()) == B
    assert torch.equal(st0, st1) and torch.equal(l0, l1) and torch.equal(t0, t1) and torch.equal(s0, s1)
    v = ts.verify(msgs, s1, l1, fixed_len=64)
    assert int(v.sum()) == 0
    ts.close()
