"""CPU: the oracle's pwr/invert/div, sparse product and utils_arith_poly / utils_arith_vectors restatements against
what the reference's own function-pointer tables returned on the same inputs (tests/golden/ops_ref.txt, oracle/ref_driver.c)."""
import numpy as np
import pytest

from tests.golden_io import load_golden
from tests import oracle_lib as O

CASES = [(12289, 512), (8380417, 256)]


@pytest.fixture(scope='module')
def G():
    return load_golden('ops_ref.txt')


def arr(G, q, n, name):
    return np.ascontiguousarray(np.asarray(G[f'q{q}_n{n}_{name}'], dtype=np.int32))


@pytest.mark.parametrize('q,n', CASES)
def test_invert_div_pwr(G, q, n):
    L = O.lib()
    a, b = arr(G, q, n, 'a').reshape(4, n), arr(G, q, n, 'b').reshape(4, n)
    inv, dv = arr(G, q, n, 'invert').reshape(4, n), arr(G, q, n, 'div').reshape(4, n)
    for k in range(4):
        v = a[k].copy()
        assert L.orc_invert(v, n, q) == int(G[f'q{q}_n{n}_invert_rc'][k])
        L.orc_normalize(v, n, q)
        assert np.array_equal(v, inv[k]), k
        num = b[k].copy()
        assert L.orc_div(num, a[k].copy(), n, q) == int(G[f'q{q}_n{n}_div_rc'][k])
        L.orc_normalize(num, n, q)
        assert np.array_equal(num, dv[k]), k
    assert list(G[f'q{q}_n{n}_invert_rc']) == [0, 0, 0, 1]
    # the failing polynomial: inverted up to the first zero, untouched (normalised) after it
    z = n // 3
    assert np.array_equal(inv[3][z:], np.mod(a[3][z:], q)) and (inv[3][:z].astype(np.int64) * a[3][:z] % q == 1).all()
    for e, ex in enumerate(G[f'q{q}_pwr_exponents']):
        out = np.array([L.orc_pwr(int(x), int(ex), q) for x in a[0]], dtype=np.int32)
        L.orc_normalize(out, n, q)
        assert np.array_equal(out, arr(G, q, n, f'pwr{e}')), ex


@pytest.mark.parametrize('q,n', CASES)
def test_sparse_poly_vec(G, q, n):
    L = O.lib()
    a, b = arr(G, q, n, 'a').reshape(4, n), arr(G, q, n, 'b').reshape(4, n)
    pos = arr(G, q, n, 'sparse_pos').reshape(4, 11)
    sp = arr(G, q, n, 'sparse').reshape(4, n)
    for k in range(4):
        v = np.zeros(n, dtype=np.int32)
        L.orc_mult_sparse(v, n, 11, b[k].copy(), pos[k].copy())
        assert np.array_equal(v, sp[k]), k
    flat_a, flat_b = a.reshape(-1).copy(), b.reshape(-1).copy()
    o = flat_a.copy(); L.orc_poly_mul_scalar(o, o.size, -3)
    assert np.array_equal(o, arr(G, q, n, 'mul_scalar_m3'))
    o = a.copy()
    for k in range(4):
        L.orc_poly_add_scalar(o[k], n, 1000 + k)
    assert np.array_equal(o.reshape(-1), arr(G, q, n, 'add_scalar'))
    o = np.zeros_like(flat_a); L.orc_poly_mod_negate(o, o.size, q, flat_b)
    assert np.array_equal(o, arr(G, q, n, 'mod_negate'))
    o = flat_a.copy(); L.orc_poly_add_single(o, o.size, flat_b)
    assert np.array_equal(o, arr(G, q, n, 'add_single'))
    o = flat_a.copy(); L.orc_poly_sub_single(o, o.size, flat_b)
    assert np.array_equal(o, arr(G, q, n, 'sub_single'))
    chg = a.copy(); chg[1, 17] ^= 0x40
    cmp_ = [L.orc_poly_cmp_not_equal(a[0].copy(), a[0].copy(), n), L.orc_poly_cmp_not_equal(a[1].copy(), chg[1].copy(), n),
            L.orc_poly_cmp_not_equal(a[2].copy(), b[2].copy(), n), L.orc_poly_cmp_not_equal(a[3].copy(), chg[3].copy(), n)]
    assert cmp_ == [int(x) for x in G[f'q{q}_n{n}_cmp']]
    ms = [L.orc_vec_absmax(b[k].copy(), n) for k in range(4)] + [L.orc_vec_scalar(a[k].copy(), b[k].copy(), n) for k in range(4)]
    assert ms == [int(x) for x in G[f'q{q}_n{n}_absmax_scalar']]
