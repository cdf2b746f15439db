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


# ---- the layout-identical tables: utils_arith_ntt_t (77 slots), utils_arith_poly_t (35), utils_arith_vec_t (4), utils_sampling_t ----

VOIDP = C.c_void_p


class NTT77(C.Structure):            # utils_arith_ntt_t, src/utils/arith/ntt.h:217-298 (packed; every slot a pointer)
    _pack_ = 1
    _fields_ = [('slots_16', VOIDP * 18),
                ('modn_32', VOIDP), ('muln_32', VOIDP), ('sqrn_32', VOIDP),
                ('mul_32_sparse', F(None, i32p, C.c_size_t, C.c_uint16, i32p, i32p)), ('mul_32_sparse_16', F(None, i32p, C.c_size_t, C.c_uint16, i16p, i32p)),
                ('mul_32_pointwise', F(None, i32p, pp, i32p, i32p)), ('mul_32_pointwise_16', F(None, i32p, pp, i32p, i16p)),
                ('mul_32_scalar', F(None, i32p, pp, i32p, C.c_int32)),
                ('fft_32_32', F(None, i32p, pp, i32p)), ('fft_32_32_large', F(None, i32p, pp, i32p)),
                ('fft_32_16', F(None, i32p, pp, i16p)), ('fft_32_16_large', F(None, i32p, pp, i16p)),
                ('pwr_32', VOIDP),
                ('invert_32', F(C.c_int32, i32p, pp, C.c_size_t)), ('div_32', F(C.c_int32, i32p, i32p, pp, C.c_size_t)),
                ('flip_32', F(None, i32p, pp)), ('center_32', F(None, i32p, C.c_size_t, pp)), ('normalize_32', F(None, i32p, C.c_size_t, pp)),
                ('fwd_ntt_32_32', F(None, i32p, pp, i32p, i32p)), ('inv_ntt_32_32', F(None, i32p, pp, i32p, i32p, i32p)),
                ('fwd_ntt_32_32_large', F(None, i32p, pp, i32p, i32p)), ('inv_ntt_32_32_large', F(None, i32p, pp, i32p, i32p, i32p)),
                ('fwd_ntt_32_16', F(None, i32p, pp, i32p, i16p)), ('inv_ntt_32_16', F(None, i32p, pp, i32p, i16p, i16p)),
                ('fwd_ntt_32_16_large', F(None, i32p, pp, i32p, i16p)), ('inv_ntt_32_16_large', F(None, i32p, pp, i32p, i16p, i16p)),
                ('slots_limb', VOIDP * 33)]


class POLY35(C.Structure):           # utils_arith_poly_t, src/utils/arith/arith.h:74-111
    _pack_ = 1
    _fields_ = [('slots_8_16', VOIDP * 15), ('copy_32', VOIDP), ('reset_32', VOIDP), ('add_32_scalar', VOIDP), ('sub_32_scalar', VOIDP),
                ('mul_32_scalar', F(None, i32p, C.c_size_t, C.c_int32)),
                ('add_32', F(None, i32p, C.c_size_t, i32p, i32p)), ('sub_32', F(None, i32p, C.c_size_t, i32p, i32p)),
                ('add_single_32', F(None, i32p, C.c_size_t, i32p)), ('sub_single_32', F(None, i32p, C.c_size_t, i32p)),
                ('mul_32', VOIDP), ('mod_negate_32', F(None, i32p, C.c_size_t, C.c_int32, i32p)),
                ('cmp_not_equal_32', F(C.c_int32, i32p, i32p, C.c_size_t)), ('uniform_32', VOIDP), ('degree_32', VOIDP), ('slots_z2', VOIDP * 6)]


class VEC4(C.Structure):             # utils_arith_vec_t, arith.h:123-128
    _pack_ = 1
    _fields_ = [('absmax_32', F(C.c_int32, i32p, C.c_size_t)), ('scalar_32', F(C.c_int32, i32p, i32p, C.c_size_t)),
                ('absmax_16', F(C.c_int32, i16p, C.c_size_t)), ('scalar_16', F(C.c_int32, i16p, i16p, C.c_size_t))]


def _bitrev(n):
    bits = n.bit_length() - 1
    return np.array([int(format(i, f'0{bits}b')[::-1], 2) for i in range(n)])


@pytest.mark.parametrize('q,n', [(12289, 512), (8380417, 256), (7681, 256), (12289, 1024)])
def test_full_ntt_table_new_slots(q, n):
    """utils_arith_ntt_t-shaped table: 77 slots; the slots round 1 did not serve (flip_32, fft_32_*, mul_32_scalar) against their definitions
    (ntt.c:584-604, ntt_template.c.in:1144-1244 as a transform of the bit-reversed input, ntt.c:425-468)"""
    from libsafecrypto_amd import load_library
    L = load_library()
    assert C.sizeof(NTT77) == 77 * 8
    tab = NTT77()
    marker = (VOIDP * 77)(*[VOIDP(0x1000 + i) for i in range(77)])          # a stand-in base table: unserved slots must come through untouched
    L.sc_hip_utils_arith_ntt_fill.argtypes = [C.c_void_p, C.c_void_p]
    assert L.sc_hip_utils_arith_ntt_fill(C.byref(tab), marker) == 0
    raw = (VOIDP * 77).from_buffer(tab)
    served = set(range(18 + 3, 18 + 12)) | set(range(18 + 13, 18 + 26))
    for i in range(77):
        if i not in served:
            assert raw[i] == 0x1000 + i, i
        else:
            assert raw[i] not in (None, 0x1000 + i), i
    L.sc_hip_vtable_errors.restype = C.c_ulong
    e0 = L.sc_hip_vtable_errors()
    p = Params(); p.n = n; p.u.ntt32.q = q; p.q_dbl = q; p.inv_q_dbl = 1.0 / q
    w, r = O.tables(q, n)
    rng = np.random.default_rng(q + n)
    # flip_32: v[0] = -v[0], v[i] <-> v[n-i], one +q / -q
    x = rng.integers(-q + 1, 2 * q, size=n).astype(np.int32)
    v = x.copy()
    tab.flip_32(P(v), C.byref(p))
    e = np.empty(n, dtype=np.int64); e[0] = -int(x[0]); e[1:] = x[:0:-1]
    e = np.where(e < 0, e + q, e); e = np.where(e >= q, e - q, e)
    assert np.array_equal(v, e.astype(np.int32))
    # fft_32_32 / fft_32_16: X[k] = sum_m x[m] g^(2mk) of the un-bit-reversed input; with the twist it is the forward transform
    xn = rng.integers(0, q, size=n).astype(np.int32)
    br = _bitrev(n)
    a = xn[br].copy()                                                        # bit-reversed order in, natural order out
    tw = ((xn.astype(np.int64) * w) % q).astype(np.int32)                    # what fwd_ntt feeds the fft: t o w, shuffled
    want = O.fwd_ntt(xn[None, :], q, w)[0]
    b = tw[br].copy()
    tab.fft_32_32(P(b), C.byref(p), P(w))
    assert np.array_equal(b, want)
    b = tw[br].copy()
    tab.fft_32_16_large(P(b), C.byref(p), w.astype(np.int16).ctypes.data_as(i16p))
    assert np.array_equal(b, want)
    # fft is linear and fixes the delta: fft(bitrev(e0)) = all ones
    d = np.zeros(n, dtype=np.int32); d[0] = 1
    tab.fft_32_32_large(P(d), C.byref(p), P(w))
    assert (d == 1).all()
    del a
    # mul_32_scalar
    t = rng.integers(-q, q, size=n).astype(np.int32)
    out = np.zeros(n, dtype=np.int32)
    tab.mul_32_scalar(P(out), C.byref(p), P(t), 12345 % q)
    assert np.array_equal(out, ((t.astype(np.int64) * (12345 % q)) % q).astype(np.int32))
    assert L.sc_hip_vtable_errors() == e0


def test_poly_and_vec_tables():
    from libsafecrypto_amd import load_library
    L = load_library()
    assert C.sizeof(POLY35) == 35 * 8 and C.sizeof(VEC4) == 32
    pt, vt = POLY35(), VEC4()
    L.sc_hip_utils_arith_poly_fill.argtypes = [C.c_void_p, C.c_void_p]
    L.sc_hip_utils_arith_vec_fill.argtypes = [C.c_void_p]
    assert L.sc_hip_utils_arith_poly_fill(C.byref(pt), None) == 0 and L.sc_hip_utils_arith_vec_fill(C.byref(vt)) == 0
    L.sc_hip_vtable_errors.restype = C.c_ulong
    e0 = L.sc_hip_vtable_errors()
    rng = np.random.default_rng(5)
    for n in (1, 7, 512, 1536):
        a = rng.integers(-2**31, 2**31 - 1, size=n).astype(np.int32); b = rng.integers(-2**31, 2**31 - 1, size=n).astype(np.int32)
        o = np.zeros(n, dtype=np.int32)
        with np.errstate(over='ignore'):
            pt.add_32(P(o), n, P(a), P(b)); assert np.array_equal(o, a + b)
            pt.sub_32(P(o), n, P(a), P(b)); assert np.array_equal(o, a - b)
            c = a.copy(); pt.add_single_32(P(c), n, P(b)); assert np.array_equal(c, a + b)
            c = a.copy(); pt.sub_single_32(P(c), n, P(b)); assert np.array_equal(c, a - b)
            c = a.copy(); pt.mul_32_scalar(P(c), n, -77); assert np.array_equal(c, a * np.int32(-77))
            pt.mod_negate_32(P(o), n, 12289, P(a)); assert np.array_equal(o, np.int32(12289) - a)
            assert pt.cmp_not_equal_32(P(a), P(a.copy()), n) == 0
            c = a.copy(); c[n // 2] ^= 4
            assert pt.cmp_not_equal_32(P(a), P(c), n) != 0
            s = rng.integers(-3000, 3000, size=n).astype(np.int32); t = rng.integers(-3000, 3000, size=n).astype(np.int32)
            assert vt.absmax_32(P(s), n) == int(np.abs(s).max())
            assert vt.scalar_32(P(s), P(t), n) == int(np.sum(s.astype(np.int64) * t).astype(np.int64).astype(np.int32))
            s16, t16 = s.astype(np.int16), t.astype(np.int16)
            assert vt.absmax_16(s16.ctypes.data_as(i16p), n) == int(np.abs(s).max())
            assert vt.scalar_16(s16.ctypes.data_as(i16p), t16.ctypes.data_as(i16p), n) == int(np.sum(s.astype(np.int64) * t).astype(np.int64).astype(np.int32))
    assert L.sc_hip_vtable_errors() == e0


def test_sampler_table_pulls_words_in_order():
    """utils_sampling_t-shaped table: vector_32 draws n words from the callback, in order, and maps them through the CDT"""
    from libsafecrypto_amd import load_library
    L = load_library()

    class SAMP(C.Structure):         # utils_sampling_t, src/utils/sampling/sampling.h:68-86 (packed)
        _pack_ = 1
    SAMP._fields_ = [('create', VOIDP), ('destroy', F(C.c_int32, C.POINTER(VOIDP))), ('get_prng', VOIDP), ('sample', F(C.c_int32, VOIDP)),
                     ('vector_16', F(C.c_int32, C.POINTER(SAMP), i16p, C.c_size_t, C.c_int32)), ('vector_32', F(C.c_int32, C.POINTER(SAMP), i32p, C.c_size_t, C.c_int32)),
                     ('precision', C.c_int32), ('dimension', C.c_int32), ('bootstrapped', C.c_int32), ('tail', C.c_float), ('sigma', C.c_float), ('sigma2', C.c_float),
                     ('gauss', VOIDP), ('prng_ctx', VOIDP), ('discard', C.c_uint32), ('bootstrap', VOIDP)]
    assert C.sizeof(SAMP) == 100
    words = np.random.default_rng(77).integers(0, 2**63, size=4096, dtype=np.uint64)
    state = {'i': 0}
    RNG = F(C.c_uint64, VOIDP)

    def draw(_ctx):
        v = int(words[state['i']]); state['i'] += 1
        return v
    cb = RNG(draw)
    h = VOIDP()
    L.sc_hip_sampler_create.argtypes = [C.c_float, C.c_float, C.c_void_p, C.c_int, RNG, C.c_void_p, C.POINTER(VOIDP)]
    assert L.sc_hip_sampler_create(13.42, 215.0, None, 0, cb, None, C.byref(h)) == 0, L.sc_hip_last_error()
    tab = SAMP()
    L.sc_hip_utils_sampling_fill.argtypes = [C.c_void_p, C.c_void_p]
    assert L.sc_hip_utils_sampling_fill(C.byref(tab), h) == 0
    table = O.cdt_table(13.42, 215.0)
    v = np.zeros(512, dtype=np.int32)
    assert tab.vector_32(C.byref(tab), P(v), 512, 0) == 0 and state['i'] == 512
    assert np.array_equal(v, O.cdt_sample(words[:512], table))
    v16 = np.zeros(100, dtype=np.int16)
    assert tab.vector_16(C.byref(tab), v16.ctypes.data_as(i16p), 100, 3) == 0 and state['i'] == 612
    assert np.array_equal(v16.astype(np.int32), O.cdt_sample(words[512:612], table) + 3)
    one = tab.sample(tab.gauss)
    assert one == int(O.cdt_sample(words[612:613], table)[0]) and state['i'] == 613
    tab.discard = 2                                                      # sample discarding draws prng_32 words: refused (and counted) until the sampler has the generator's prng_32
    L.sc_hip_vtable_errors.restype = C.c_ulong
    e0 = L.sc_hip_vtable_errors()
    assert tab.vector_32(C.byref(tab), P(v), 4, 0) == 1 and L.sc_hip_vtable_errors() == e0 + 1
    g = VOIDP(tab.gauss)
    assert tab.destroy(C.byref(g)) == 0 and not g.value


def test_sampler_side_channel_variants_match_the_reference():
    """shuffle / blinding fillers and sample discarding (sampling.c:107-228) behind the utils_sampling_t mirror: the vectors the reference itself filled from a
    seeded CSPRNG (tests/golden/sampling_ref.txt), reproduced from the same pool words delivered through the prng_32 / prng_64 callbacks; and the oracle"""
    from libsafecrypto_amd import load_library
    from tests.golden_io import load_golden
    L = load_library()
    G = load_golden('sampling_ref.txt')

    class SAMP(C.Structure):
        _pack_ = 1
    SAMP._fields_ = [('create', VOIDP), ('destroy', F(C.c_int32, C.POINTER(VOIDP))), ('get_prng', VOIDP), ('sample', F(C.c_int32, VOIDP)),
                     ('vector_16', F(C.c_int32, C.POINTER(SAMP), i16p, C.c_size_t, C.c_int32)), ('vector_32', F(C.c_int32, C.POINTER(SAMP), i32p, C.c_size_t, C.c_int32)),
                     ('precision', C.c_int32), ('dimension', C.c_int32), ('bootstrapped', C.c_int32), ('tail', C.c_float), ('sigma', C.c_float), ('sigma2', C.c_float),
                     ('gauss', VOIDP), ('prng_ctx', VOIDP), ('discard', C.c_uint32), ('bootstrap', VOIDP)]
    RNG64, RNG32 = F(C.c_uint64, VOIDP), F(C.c_uint32, VOIDP)
    L.sc_hip_sampler_create.argtypes = [C.c_float, C.c_float, C.c_void_p, C.c_int, RNG64, C.c_void_p, C.POINTER(VOIDP)]
    L.sc_hip_sampler_set_rng32.argtypes = [C.c_void_p, RNG32]
    L.sc_hip_sampler_set_rng32.restype = None
    L.sc_hip_utils_sampling_fill_sca.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
    plain = O.cdt_table(13.42, 215.0)
    blind = np.zeros(plain.size, dtype=np.uint64)
    assert L.sc_hip_cdt_build_table_blinding(13.42, 215.0, blind.ctypes.data, blind.size) == 0
    seen = set()
    for k in range(int(G['sca_cases'][0])):
        blinding, discard, n, centre = (int(x) for x in G[f'sca{k}_par'])
        tape = (np.asarray(G[f'sca{k}_tape']).astype(np.int64) & 0xFFFFFFFF).astype(np.uint32)
        st = {'i': 0}

        def d32(_c):
            w = int(tape[st['i']]); st['i'] += 1
            return w

        def d64(_c):                                              # prng_64: two pool words, high word first (prng.c:963-979)
            hi = int(tape[st['i']]); lo = int(tape[st['i'] + 1]); st['i'] += 2
            return (hi << 32) | lo
        c64, c32 = RNG64(d64), RNG32(d32)
        table = blind if blinding == 1 else plain
        h = VOIDP()
        assert L.sc_hip_sampler_create(13.42, 215.0, table.ctypes.data, table.size, c64, None, C.byref(h)) == 0, L.sc_hip_last_error()
        L.sc_hip_sampler_set_rng32(h, c32)
        tab = SAMP()
        assert L.sc_hip_utils_sampling_fill_sca(C.byref(tab), h, blinding) == 0
        tab.discard = discard
        v = np.full(n, 12345, dtype=np.int32)
        assert tab.vector_32(C.byref(tab), P(v), n, centre) == 0, L.sc_hip_last_error()
        assert np.array_equal(v, G[f'sca{k}_v']), (k, blinding, discard)
        want, used = O.sample_vector_sca(n, centre, blinding, discard, tape, table)
        assert np.array_equal(v, want) and used == st['i']        # the same words consumed, in the same order
        if k in (5, 10):                                          # the 16-bit fillers, on a fresh stretch of the same tape
            st['i'] = 0
            v16 = np.zeros(n, dtype=np.int16)
            assert tab.vector_16(C.byref(tab), v16.ctypes.data_as(i16p), n, centre) == 0
            assert np.array_equal(v16.astype(np.int32), G[f'sca{k}_v'])
        seen.add((blinding, discard))
        g = VOIDP(tab.gauss)
        assert tab.destroy(C.byref(g)) == 0
    assert len(seen) == 12
