"""ctypes binding of oracle/libsc_oracle.so - the CPU checker (tests / smoke / cpu_baseline only)."""
import ctypes as C
import os
import subprocess
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_LIB = None

i32p = np.ctypeslib.ndpointer(dtype=np.int32, flags='C_CONTIGUOUS')
u64p = np.ctypeslib.ndpointer(dtype=np.uint64, flags='C_CONTIGUOUS')


def lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(ROOT, 'oracle', 'libsc_oracle.so')
        if not os.path.exists(path):
            subprocess.check_call(['make', '-C', os.path.join(ROOT, 'oracle'), 'libsc_oracle.so'])
        L = C.CDLL(path)
        L.orc_find_root.restype = C.c_int32
        L.orc_find_root.argtypes = [C.c_int32, C.c_int]
        L.orc_make_tables.argtypes = [C.c_int32, C.c_int, i32p, i32p]
        L.orc_modn.restype = C.c_int32
        L.orc_modn.argtypes = [C.c_int32, C.c_int32]
        L.orc_muln.restype = C.c_int32
        L.orc_muln.argtypes = [C.c_int32, C.c_int32, C.c_int32]
        L.orc_fwd_ntt.argtypes = [i32p, C.c_int, C.c_int32, i32p, i32p]
        L.orc_inv_ntt.argtypes = [i32p, C.c_int, C.c_int32, i32p, i32p, i32p]
        L.orc_normalize.argtypes = [i32p, C.c_size_t, C.c_int32]
        L.orc_center.argtypes = [i32p, C.c_size_t, C.c_int32]
        L.orc_mul_pointwise.argtypes = [i32p, C.c_int, C.c_int32, i32p, i32p]
        L.orc_polymul.argtypes = [i32p, C.c_int, C.c_int32, i32p, i32p, i32p, i32p]
        L.orc_polymul_ntt.argtypes = [i32p, C.c_int, C.c_int32, i32p, i32p, i32p, i32p]
        L.orc_mult_sparse.argtypes = [i32p, C.c_int, C.c_int, i32p, i32p]
        L.orc_cdt_bits.restype = C.c_int
        L.orc_cdt_bits.argtypes = [C.c_float, C.c_float]
        L.orc_cdt_create.argtypes = [C.c_float, C.c_float, u64p, C.c_int]
        L.orc_cdt_sample.restype = C.c_int32
        L.orc_cdt_sample.argtypes = [C.c_uint64, u64p, C.c_int]
        L.orc_cdt_sample_vector.argtypes = [i32p, C.c_size_t, u64p, u64p, C.c_int, C.c_int32]
        L.orc_batch_fwd_ntt.argtypes = [i32p, i32p, C.c_size_t, C.c_int, C.c_int32, i32p]
        L.orc_batch_inv_ntt.argtypes = [i32p, i32p, C.c_size_t, C.c_int, C.c_int32, i32p, i32p]
        L.orc_batch_polymul_ntt.argtypes = [i32p, i32p, i32p, C.c_size_t, C.c_int, C.c_int32, i32p, i32p]
        _LIB = L
    return _LIB


def tables(q, n):
    w = np.zeros(n, dtype=np.int32)
    r = np.zeros(n, dtype=np.int32)
    assert lib().orc_make_tables(q, n, w, r) == 0
    return w, r


def fwd_ntt(a, q, w):
    """batched canonical forward NTT ([batch][n] int32)"""
    a = np.ascontiguousarray(a, dtype=np.int32)
    out = np.empty_like(a)
    n = a.shape[-1]
    lib().orc_batch_fwd_ntt(out.reshape(-1), a.reshape(-1), a.size // n, n, q, w)
    return out


def inv_ntt(a, q, w, r):
    a = np.ascontiguousarray(a, dtype=np.int32)
    out = np.empty_like(a)
    n = a.shape[-1]
    lib().orc_batch_inv_ntt(out.reshape(-1), a.reshape(-1), a.size // n, n, q, w, r)
    return out


def polymul_ntt(a, bhat, q, w, r):
    a = np.ascontiguousarray(a, dtype=np.int32)
    bhat = np.ascontiguousarray(bhat, dtype=np.int32)
    out = np.empty_like(a)
    n = a.shape[-1]
    lib().orc_batch_polymul_ntt(out.reshape(-1), a.reshape(-1), bhat.reshape(-1), a.size // n, n, q, w, r)
    return out


def cdt_table(tail, sigma):
    bits = lib().orc_cdt_bits(tail, sigma)
    cdf = np.zeros(1 << bits, dtype=np.uint64)
    lib().orc_cdt_create(tail, sigma, cdf, 1 << bits)
    return cdf


def cdt_sample(x, cdf, centre=0):
    x = np.ascontiguousarray(x, dtype=np.uint64)
    v = np.empty(x.size, dtype=np.int32)
    lib().orc_cdt_sample_vector(v, x.size, x, cdf, cdf.size, centre)
    return v
