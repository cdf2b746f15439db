"""GPU: the reference-shaped single-polynomial vtable (include/sc_hip_vtable.h) called exactly as a
scheme calls sc->sc_ntt->fwd_ntt_32_16(v, &ntt, t, w) etc. (e.g. bliss_b.c:1378-1384)."""
import ctypes as C

import numpy as np
import pytest

from tests import oracle_lib as O

pytestmark = pytest.mark.gpu


class U32(C.Structure):
    _pack_ = 1
    _fields_ = [('q', C.c_int32), ('q_inv', C.c_uint32), ('m', C.c_int32), ('k', C.c_int32)]


class U64(C.Structure):
    _pack_ = 1
    _fields_ = [('q', C.c_int64), ('q_inv', C.c_uint64), ('m', C.c_int64), ('k', C.c_int64)]


class U(C.Union):
    _pack_ = 1
    _fields_ = [('ntt64', U64), ('ntt32', U32)]


class Params(C.Structure):           # ntt_params_t, src/utils/arith/ntt.h:91-103 (packed)
    _pack_ = 1
    _fields_ = [('q_dbl', C.c_double), ('inv_q_dbl', C.c_double), ('inv_q_flt', C.c_float), ('n', C.c_size_t), ('u', U)]


i32p, i16p, pp = C.POINTER(C.c_int32), C.POINTER(C.c_int16), C.POINTER(Params)
F = C.CFUNCTYPE


class VT(C.Structure):
    _fields_ = [
        ('mul_32_pointwise', F(None, i32p, pp, i32p, i32p)), ('mul_32_pointwise_16', F(None, i32p, pp, i32p, i16p)),
        ('center_32', F(None, i32p, C.c_size_t, pp)), ('normalize_32', F(None, i32p, C.c_size_t, pp)),
        ('fwd_ntt_32_32', F(None, i32p, pp, i32p, i32p)), ('inv_ntt_32_32', F(None, i32p, pp, i32p, i32p, i32p)),
        ('fwd_ntt_32_32_large', F(None, i32p, pp, i32p, i32p)), ('inv_ntt_32_32_large', F(None, i32p, pp, i32p, i32p, i32p)),
        ('fwd_ntt_32_16', F(None, i32p, pp, i32p, i16p)), ('inv_ntt_32_16', F(None, i32p, pp, i32p, i16p, i16p)),
        ('fwd_ntt_32_16_large', F(None, i32p, pp, i32p, i16p)), ('inv_ntt_32_16_large', F(None, i32p, pp, i32p, i16p, i16p)),
        ('mul_32_sparse', F(None, i32p, C.c_size_t, C.c_uint16, i32p, i32p)), ('mul_32_sparse_16', F(None, i32p, C.c_size_t, C.c_uint16, i16p, i32p)),
        ('invert_32', F(C.c_int32, i32p, pp, C.c_size_t)), ('div_32', F(C.c_int32, i32p, i32p, pp, C.c_size_t)),
    ]


def P(a):
    return a.ctypes.data_as(i32p)


def test_params_layout_matches_reference():
    assert C.sizeof(Params) == 60 and Params.n.offset == 20 and Params.u.offset == 28   # packed ntt_params_t


@pytest.mark.parametrize('q,n,is16', [(12289, 512, True), (8380417, 256, False)])
def test_vtable_calls(q, n, is16):
    from libsafecrypto_amd import load_library
    L = load_library()
    L.sc_hip_utils_arith_ntt32.restype = C.POINTER(VT)
    vt = L.sc_hip_utils_arith_ntt32()
    assert vt, L.sc_hip_last_error()
    vt = vt.contents
    L.sc_hip_vtable_errors.restype = C.c_ulong
    e0 = L.sc_hip_vtable_errors()
    p = Params()
    p.n = n
    p.u.ntt32.q = q
    p.q_dbl = q
    p.inv_q_dbl = 1.0 / q
    w, r = O.tables(q, n)
    w16, r16 = w.astype(np.int16), r.astype(np.int16)
    rng = np.random.default_rng(q)
    t = rng.integers(-12, 13, size=n).astype(np.int32)
    a = rng.integers(0, q, size=n).astype(np.int32)
    v = np.zeros(n, dtype=np.int32)
    if is16:
        vt.fwd_ntt_32_16(P(v), C.byref(p), P(t), w16.ctypes.data_as(i16p))
    else:
        vt.fwd_ntt_32_32(P(v), C.byref(p), P(t), P(w))
    exp_f = O.fwd_ntt(t[None, :], q, w)[0]
    assert np.array_equal(v, exp_f)
    # v = v o a, in place (aliasing v == t is allowed by the reference)
    if is16 and q < 32768:
        vt.mul_32_pointwise_16(P(v), C.byref(p), P(v), a.astype(np.int16).ctypes.data_as(i16p))
    else:
        vt.mul_32_pointwise(P(v), C.byref(p), P(v), P(a))
    exp_pw = ((exp_f.astype(np.int64) * a) % q).astype(np.int32)
    assert np.array_equal(v, exp_pw)
    if is16:
        vt.inv_ntt_32_16(P(v), C.byref(p), P(v), w16.ctypes.data_as(i16p), r16.ctypes.data_as(i16p))
    else:
        vt.inv_ntt_32_32(P(v), C.byref(p), P(v), P(w), P(r))
    assert np.array_equal(v, O.inv_ntt(exp_pw[None, :], q, w, r)[0])
    big = rng.integers(-2**31, 2**31 - 1, size=3 * n).astype(np.int32)   # normalize_32(t, l*n) style call
    nrm, cen = big.copy(), big.copy()
    vt.normalize_32(P(nrm), nrm.size, C.byref(p))
    vt.center_32(P(cen), cen.size, C.byref(p))
    e1, e2 = big.copy(), big.copy()
    O.lib().orc_normalize(e1, e1.size, q)
    O.lib().orc_center(e2, e2.size, q)
    assert np.array_equal(nrm, e1) and np.array_equal(cen, e2)
    # ring-TESLA style sparse products (ring_tesla.c:856-871,1015) and BLISS-keygen style inversion (bliss_b.c:1193)
    pos = rng.permutation(n)[:11].astype(np.int32)
    sp, exp_sp = np.zeros(n, dtype=np.int32), np.zeros(n, dtype=np.int32)
    vt.mul_32_sparse(P(sp), n, 11, P(t), P(pos))
    O.lib().orc_mult_sparse(exp_sp, n, 11, t.copy(), pos.copy())
    assert np.array_equal(sp, exp_sp)
    t16 = t.astype(np.int16)
    vt.mul_32_sparse_16(P(sp), n, 11, t16.ctypes.data_as(i16p), P(pos))
    assert np.array_equal(sp, exp_sp)
    nz = rng.integers(1, q, size=n).astype(np.int32)
    iv = nz.copy()
    assert vt.invert_32(P(iv), C.byref(p), n) == 0
    assert ((iv.astype(np.int64) * nz) % q == 1).all()
    num = a.copy()
    assert vt.div_32(P(num), P(nz), C.byref(p), n) == 0
    assert np.array_equal(num, (a.astype(np.int64) * iv % q).astype(np.int32))
    nz[n // 2] = 0
    half = nz.copy()
    assert vt.invert_32(P(half), C.byref(p), n) == 1                     # SC_FUNC_FAILURE, stopped at the zero
    assert np.array_equal(half[n // 2:], nz[n // 2:]) and np.array_equal(half[:n // 2], iv[:n // 2])
    assert L.sc_hip_vtable_errors() == e0
