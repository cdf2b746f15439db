"""GPU parity for the remaining sc_ntt / sc_poly / sc_vec entries (csrc/ops.hip): against the reference's own outputs
(tests/golden/ops_ref.txt) and against the oracle on larger seeded batches."""
import numpy as np
import pytest

from tests.golden_io import load_golden
from tests import oracle_lib as O

pytestmark = pytest.mark.gpu
CASES = [(12289, 512), (8380417, 256)]


@pytest.fixture(scope='module')
def eng():
    from libsafecrypto_amd import Engine
    e = Engine(0)
    yield e
    e.close()


@pytest.fixture(scope='module')
def G():
    return load_golden('ops_ref.txt')


def arr(G, q, n, name):
    return np.ascontiguousarray(np.asarray(G[f'q{q}_n{n}_{name}'], dtype=np.int32))


def dev(a):
    import torch
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


@pytest.mark.parametrize('q,n', CASES)
def test_invert_div_pwr_match_reference(eng, G, q, n):
    plan = eng.plan(q, n)
    a, b = arr(G, q, n, 'a').reshape(4, n), arr(G, q, n, 'b').reshape(4, n)
    v = dev(a)
    rc = plan.invert(v)
    assert rc.cpu().tolist() == [int(x) for x in G[f'q{q}_n{n}_invert_rc']]
    got = v.cpu().numpy()
    ref = arr(G, q, n, 'invert').reshape(4, n)
    assert np.array_equal(got[:3], ref[:3])
    z = n // 3                                  # failing polynomial: done before the first zero, untouched (raw input) after it
    assert np.array_equal(got[3][:z], ref[3][:z]) and np.array_equal(got[3][z:], a[3][z:])
    num = dev(b)
    rc = plan.div(num, dev(a))
    assert rc.cpu().tolist() == [int(x) for x in G[f'q{q}_n{n}_div_rc']]
    got, ref = num.cpu().numpy(), arr(G, q, n, 'div').reshape(4, n)
    assert np.array_equal(got[:3], ref[:3]) and np.array_equal(got[3][:z], ref[3][:z]) and np.array_equal(got[3][z:], b[3][z:])
    for e, ex in enumerate(G[f'q{q}_pwr_exponents']):
        assert np.array_equal(plan.pwr(dev(a[0]), int(ex)).cpu().numpy(), arr(G, q, n, f'pwr{e}')), ex
    # signed, non-canonical inputs: poly 2 of a
    assert np.array_equal(plan.pwr(dev(a[2]), 3).cpu().numpy(), np.array([pow(int(x) % q, 3, q) for x in a[2]], dtype=np.int32))
    plan.close()


@pytest.mark.parametrize('q,n', CASES)
def test_sparse_poly_vec_match_reference(eng, G, q, n):
    import torch
    a, b = arr(G, q, n, 'a').reshape(4, n), arr(G, q, n, 'b').reshape(4, n)
    pos = arr(G, q, n, 'sparse_pos').reshape(4, 11)
    assert np.array_equal(eng.mult_sparse(dev(b), dev(pos)).cpu().numpy(), arr(G, q, n, 'sparse').reshape(4, n))
    assert np.array_equal(eng.poly_mul_scalar(dev(a), -3).cpu().numpy().reshape(-1), arr(G, q, n, 'mul_scalar_m3'))
    assert np.array_equal(eng.poly_mod_negate(dev(b), q).cpu().numpy().reshape(-1), arr(G, q, n, 'mod_negate'))
    p = dev(a)
    eng.poly_add_scalar(p, dev(np.arange(1000, 1004, dtype=np.int32)))
    assert np.array_equal(p.cpu().numpy().reshape(-1), arr(G, q, n, 'add_scalar'))
    p = dev(a); eng.poly_add(p, dev(b), out=p)                                  # add_single_32: out aliases the first input
    assert np.array_equal(p.cpu().numpy().reshape(-1), arr(G, q, n, 'add_single'))
    p = dev(a); eng.poly_sub(p, dev(b), out=p)
    assert np.array_equal(p.cpu().numpy().reshape(-1), arr(G, q, n, 'sub_single'))
    chg = a.copy(); chg[1, 17] ^= 0x40
    x = np.stack([a[0], a[1], a[2], a[3]]); y = np.stack([a[0], chg[1], b[2], chg[3]])
    assert eng.poly_cmp_not_equal(dev(x), dev(y)).cpu().tolist() == [int(v) for v in G[f'q{q}_n{n}_cmp']]
    ms = eng.vec_absmax(dev(b)).cpu().tolist() + eng.vec_scalar(dev(a), dev(b)).cpu().tolist()
    assert ms == [int(v) for v in G[f'q{q}_n{n}_absmax_scalar']]
    torch.cuda.synchronize()


def test_batches_vs_oracle(eng):
    """ragged sizes, int16 sparse inputs, zeros scattered through an invert batch"""
    L = O.lib()
    rng = np.random.default_rng(2024)
    q, n, B = 12289, 512, 301
    plan = eng.plan(q, n)
    a = rng.integers(-3 * q, 3 * q, size=(B, n), dtype=np.int32)
    zeros_at = {5: 100, 17: 0, 300: 511}
    for k in range(B):
        a[k][a[k] % q == 0] += 1
    for k, z in zeros_at.items():
        a[k, z] = -q if k == 5 else 0
    v = dev(a); rc = plan.invert(v).cpu().numpy(); got = v.cpu().numpy()
    for k in list(range(0, 24)) + [300]:
        w = a[k].copy()
        assert L.orc_invert(w, n, q) == rc[k] == (1 if k in zeros_at else 0), k
        z = zeros_at.get(k, n)
        L.orc_normalize(w[:z].copy(), z, q)
        ww = w[:z].copy(); L.orc_normalize(ww, z, q)
        assert np.array_equal(got[k][:z], ww) and np.array_equal(got[k][z:], a[k][z:]), k
    t16 = rng.integers(-300, 300, size=(B, n), dtype=np.int16)
    pos = np.stack([rng.permutation(n)[:19] for _ in range(B)]).astype(np.int32)
    sp = eng.mult_sparse(dev(t16), dev(pos)).cpu().numpy()
    for k in (0, 1, 150, 300):
        w = np.zeros(n, dtype=np.int32)
        L.orc_mult_sparse(w, n, 19, t16[k].astype(np.int32), pos[k].copy())
        assert np.array_equal(sp[k], w), k
    big = rng.integers(-2**31, 2**31 - 1, size=(B, 1000), dtype=np.int64).astype(np.int32)     # n not a multiple of 64, wrapping sums
    big2 = rng.integers(-2**31, 2**31 - 1, size=(B, 1000), dtype=np.int64).astype(np.int32)
    sc = eng.vec_scalar(dev(big), dev(big2)).cpu().numpy(); am = eng.vec_absmax(dev(a)).cpu().numpy()
    for k in (0, 7, 300):
        assert sc[k] == L.orc_vec_scalar(big[k].copy(), big2[k].copy(), 1000) and am[k] == L.orc_vec_absmax(a[k].copy(), n)
    flat = rng.integers(-2**31, 2**31 - 1, size=1003, dtype=np.int64).astype(np.int32)                # length not a multiple of 4
    o = flat.copy(); L.orc_poly_mul_scalar(o, o.size, 77777)
    import torch
    buf = torch.zeros(1004, dtype=torch.int32, device='cuda'); buf[:1003] = dev(flat)
    assert np.array_equal(eng.poly_mul_scalar(buf[:1003], 77777).cpu().numpy(), o)
    plan.close()


@pytest.mark.parametrize('n,c', [(512, [0, 154]), (256, [38, 140]), (512, [1, 1, 3, 5, 8, 12, 17, 24, 31, 38, 44, 47]), (1024, [0, 0, 0, 0, 1, 3, 9, 22, 46, 80, 118, 150])])
def test_poly_uniform_matches_oracle(eng, n, c):
    """utils_arith_poly_t uniform_32 (poly_32_uniform_rand) with the coefficient lists of BLISS-B and ENS-KEM: polynomial, words consumed, underrun"""
    import torch
    rng = np.random.default_rng(n + len(c))
    B, W = 130, 2 * sum(c) + 40
    tapes = rng.integers(0, 2**32, size=(B, W), dtype=np.uint64).astype(np.uint32)
    lens = np.full(B, W, dtype=np.int32)
    lens[5] = sum(c) // 2                                        # cannot finish
    out, used, st = eng.poly_uniform(n, c, torch.from_numpy(tapes.view(np.int32)).cuda(), torch.from_numpy(lens).cuda())
    torch.cuda.synchronize()
    on, un, sn = out.cpu().numpy(), used.cpu().numpy(), st.cpu().numpy()
    for i in range(B):
        rc, v, u = O.poly_uniform(n, c, tapes[i, :lens[i]])
        assert (rc, int(sn[i])) in ((0, 1), (2, 2)) and un[i] == u, i
        assert np.array_equal(on[i], v if rc == 0 else np.zeros(n, dtype=np.int32)), i
    assert sn[5] == 2
    k = len(c)
    for j, cnt in enumerate(c):                                   # the value histogram is the list itself
        assert int((np.abs(on[0]) == k - j).sum()) == cnt
