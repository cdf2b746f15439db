"""ctypes binding of oracle/libsc_oracle.so - the CPU checker (tests / smoke / cpu_baseline only)."""
import ctypes as C
import os
import subprocess
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_LIB = None

i32p = np.ctypeslib.ndpointer(dtype=np.int32, flags='C_CONTIGUOUS')
u64p = np.ctypeslib.ndpointer(dtype=np.uint64, flags='C_CONTIGUOUS')
u32p_g = np.ctypeslib.ndpointer(dtype=np.uint32, flags='C_CONTIGUOUS')


def lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(ROOT, 'oracle', 'libsc_oracle.so')
        srcs = [os.path.join(ROOT, 'oracle', f) for f in ('sc_oracle.c', 'sc_oracle_keccak.c', 'sc_oracle_dilithium.c', 'sc_oracle_bliss.c', 'sc_oracle_kyber.c', 'sc_oracle_falcon.c', 'sc_oracle_drbg.c', 'sc_oracle_tesla.c', 'sc_oracle_rlwe.c', 'sc_oracle_ens.c', 'sc_oracle_mlwe_xof.c', 'sc_oracle.h')]
        if not os.path.exists(path) or any(os.path.getmtime(f) > os.path.getmtime(path) for f in srcs):
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
        L.orc_pwr.restype = C.c_int32
        L.orc_pwr.argtypes = [C.c_int32, C.c_int32, C.c_int32]
        L.orc_invert.argtypes = [i32p, C.c_size_t, C.c_int32]
        L.orc_div.argtypes = [i32p, i32p, C.c_size_t, C.c_int32]
        L.orc_poly_mul_scalar.argtypes = [i32p, C.c_size_t, C.c_int32]
        L.orc_poly_add_scalar.argtypes = [i32p, C.c_size_t, C.c_int32]
        L.orc_poly_mod_negate.argtypes = [i32p, C.c_size_t, C.c_int32, i32p]
        L.orc_poly_add_single.argtypes = [i32p, C.c_size_t, i32p]
        L.orc_poly_sub_single.argtypes = [i32p, C.c_size_t, i32p]
        L.orc_poly_cmp_not_equal.restype = C.c_int32
        L.orc_poly_cmp_not_equal.argtypes = [i32p, i32p, C.c_size_t]
        L.orc_vec_absmax.restype = C.c_int32
        L.orc_vec_absmax.argtypes = [i32p, C.c_size_t]
        L.orc_vec_scalar.restype = C.c_int32
        L.orc_vec_scalar.argtypes = [i32p, i32p, C.c_size_t]
        L.orc_cdt_bits.restype = C.c_int
        L.orc_cdt_bits.argtypes = [C.c_float, C.c_float]
        L.orc_cdt_create.argtypes = [C.c_float, C.c_float, u64p, C.c_int]
        L.orc_cdt_create_blinding.argtypes = [C.c_float, C.c_float, u64p, C.c_int]
        L.orc_cdt32_create.argtypes = [C.c_float, C.c_float, u32p_g, C.c_int]
        L.orc_cdt32_sample_vector.argtypes = [i32p, C.c_size_t, u32p_g, u32p_g, C.c_int, C.c_int32]
        L.orc_cdt_sample.restype = C.c_int32
        L.orc_cdt_sample.argtypes = [C.c_uint64, u64p, C.c_int]
        L.orc_cdt_sample_vector.argtypes = [i32p, C.c_size_t, u64p, u64p, C.c_int, C.c_int32]
        L.orc_sample_vector_sca.restype = C.c_long
        L.orc_sample_vector_sca.argtypes = [i32p, C.c_size_t, C.c_int32, C.c_int, C.c_uint32, np.ctypeslib.ndpointer(dtype=np.uint32, flags='C_CONTIGUOUS'), C.c_size_t, u64p, C.c_int]
        L.orc_batch_fwd_ntt.argtypes = [i32p, i32p, C.c_size_t, C.c_int, C.c_int32, i32p]
        L.orc_batch_inv_ntt.argtypes = [i32p, i32p, C.c_size_t, C.c_int, C.c_int32, i32p, i32p]
        L.orc_batch_polymul_ntt.argtypes = [i32p, i32p, i32p, C.c_size_t, C.c_int, C.c_int32, i32p, i32p]
        u8p = np.ctypeslib.ndpointer(dtype=np.uint8, flags='C_CONTIGUOUS')
        L.orc_shake.argtypes = [C.c_int, u8p, C.c_size_t, u8p, C.c_size_t]
        L.orc_dil_decompose.argtypes = [i32p, i32p, i32p, C.c_size_t, C.c_int, C.c_int32]
        L.orc_dilithium_crh_t1.argtypes = [u8p, i32p, C.c_size_t, C.c_int, u8p]
        L.orc_dilithium_sign.argtypes = [i32p, i32p, i32p, i32p, u8p, u8p, i32p, u8p, C.c_size_t, u8p,
                                         C.POINTER(C.c_size_t), C.POINTER(C.c_int32)]
        L.orc_dilithium_keygen.argtypes = [i32p, C.c_int, u8p, C.c_size_t, i32p, i32p, i32p, u8p, u8p, u8p, i32p]
        L.orc_dilithium_verify.argtypes = [i32p, i32p, u8p, i32p, u8p, C.c_size_t, u8p, C.c_size_t]
        i16p = np.ctypeslib.ndpointer(dtype=np.int16, flags='C_CONTIGUOUS')
        u32p = np.ctypeslib.ndpointer(dtype=np.uint32, flags='C_CONTIGUOUS')
        L.orc_sha3.argtypes = [C.c_int, u8p, C.c_size_t, u8p]
        L.orc_bliss_sign.argtypes = [i32p, i16p, i16p, i16p, u8p, C.c_size_t, u32p, C.c_size_t, u8p, C.POINTER(C.c_size_t),
                                     C.POINTER(C.c_int32), C.POINTER(C.c_size_t)]
        L.orc_bliss_verify.argtypes = [i32p, i16p, u8p, C.c_size_t, u8p, C.c_size_t]
        u16p = np.ctypeslib.ndpointer(dtype=np.uint16, flags='C_CONTIGUOUS')
        L.orc_bliss_sig_to_huffman.argtypes = [i32p, u8p, C.c_size_t, u32p, u16p, C.c_uint32, u32p, u16p, C.c_uint32, u8p, C.POINTER(C.c_size_t)]
        L.orc_bliss_sig_from_huffman.argtypes = [i32p, u8p, C.c_size_t, u32p, u16p, C.c_uint32, u32p, u16p, C.c_uint32, u8p, C.POINTER(C.c_size_t)]
        L.orc_bac_dist.argtypes = [u64p, C.c_float, C.c_size_t]; L.orc_bac_dist.restype = None
        L.orc_bliss_sig_to_bac.argtypes = [i32p, u8p, C.c_size_t, u64p, u64p, u8p, C.POINTER(C.c_size_t)]
        L.orc_bliss_sig_from_bac.argtypes = [i32p, u8p, C.c_size_t, u64p, u64p, u8p, C.POINTER(C.c_size_t)]
        L.orc_huff_row_encode.argtypes = [u8p, C.c_size_t, C.c_int, C.c_int, C.c_int, u32p, u16p, C.c_uint32, u8p, C.POINTER(C.c_size_t)]
        L.orc_huff_row_decode.argtypes = [u8p, C.c_size_t, C.c_int, C.c_int, C.c_int, u32p, u16p, C.c_uint32, u8p, C.POINTER(C.c_size_t)]
        L.orc_bliss_keygen.argtypes = [i32p, C.c_int, C.c_int, u32p, C.c_size_t, i16p, i16p, i16p, C.POINTER(C.c_size_t), C.POINTER(C.c_int32)]
        L.orc_kyber_cpa_enc.argtypes = [i32p, i32p, i32p, u8p, u8p, i32p, i32p]
        L.orc_kyber_cpa_dec.argtypes = [i32p, i32p, i32p, i32p, u8p]
        L.orc_kyber_keygen.argtypes = [i32p, u8p, C.c_size_t, i32p, i32p, i32p, u8p, u8p]
        L.orc_expand_matrix.argtypes = [u8p, C.c_size_t, C.c_int, C.c_int32, C.c_int, i32p]
        L.orc_kyber_ct_bytes.restype = C.c_size_t
        L.orc_kyber_ct_bytes.argtypes = [i32p]
        L.orc_kyber_kem_enc.argtypes = [i32p, i32p, i32p, u8p, i32p, u8p, u8p, u8p]
        L.orc_kyber_kem_dec.argtypes = [i32p, i32p, i32p, i32p, u8p, i32p, u8p, u8p]
        L.orc_drbg_create.restype = C.c_void_p
        L.orc_drbg_create.argtypes = [u8p, C.c_size_t]
        L.orc_drbg_destroy.argtypes = [C.c_void_p]
        L.orc_drbg_mem.argtypes = [C.c_void_p, u8p, C.c_int32]
        L.orc_drbg_32.restype = C.c_uint32
        L.orc_drbg_32.argtypes = [C.c_void_p]
        L.orc_drbg_64.restype = C.c_uint64
        L.orc_drbg_64.argtypes = [C.c_void_p]
        L.orc_aes256_ecb.argtypes = [u8p, u8p, u8p]
        L.orc_tesla_sig_bytes.restype = C.c_size_t
        L.orc_tesla_sig_bytes.argtypes = [i32p]
        L.orc_tesla_sign.argtypes = [i32p, i32p, i32p, i16p, i16p, i16p, u8p, C.c_size_t, u32p, C.c_size_t, u8p, C.POINTER(C.c_size_t),
                                     C.POINTER(C.c_int32), C.POINTER(C.c_size_t)]
        L.orc_tesla_verify.argtypes = [i32p, i32p, i32p, i32p, i32p, u8p, C.c_size_t, u8p, C.c_size_t]
        for fn in ('orc_rlwe_kg_words', 'orc_rlwe_enc_words', 'orc_rlwe_ct_bytes'):
            getattr(L, fn).restype = C.c_size_t
            getattr(L, fn).argtypes = [i32p]
        L.orc_rlwe_keygen.argtypes = [i32p, u64p, C.c_int, u32p, C.c_size_t, i16p, i16p, i16p]
        L.orc_rlwe_encrypt.argtypes = [i32p, u64p, C.c_int, i16p, i16p, u8p, u32p, C.c_size_t, u8p]
        L.orc_rlwe_decrypt.argtypes = [i32p, i16p, u8p, C.c_size_t, u8p]
        L.orc_poly_uniform.argtypes = [i32p, C.c_int, np.ctypeslib.ndpointer(dtype=np.uint16, flags='C_CONTIGUOUS'), C.c_int, u32p, C.c_size_t, C.POINTER(C.c_size_t)]
        f32p = np.ctypeslib.ndpointer(dtype=np.float32, flags='C_CONTIGUOUS')
        L.orc_ensdlp_expand_basis.argtypes = [C.c_int, i32p, i32p]
        L.orc_ensdlp_gso.argtypes = [C.c_int, C.c_int, i32p, f32p, f32p]
        L.orc_ensdlp_gso.restype = None
        L.orc_ensdlp_sign.argtypes = [i32p, i32p, f32p, f32p, u8p, C.c_size_t, u32p, C.c_size_t, u8p]
        L.orc_ensdlp_verify.argtypes = [i32p, np.ctypeslib.ndpointer(dtype=np.int16, flags='C_CONTIGUOUS'), u8p, C.c_size_t, u8p, C.c_size_t]
        L.orc_ensdlp_rec_m1_bits.restype = C.c_size_t
        L.orc_ensdlp_rec_m1_bits.argtypes = [i32p]
        L.orc_ensdlp_sign_recovery.argtypes = [i32p, i32p, f32p, f32p, u8p, C.c_size_t, u32p, C.c_size_t, u8p, u8p, C.POINTER(C.c_size_t)]
        L.orc_ensdlp_verify_recovery.argtypes = [i32p, np.ctypeslib.ndpointer(dtype=np.int16, flags='C_CONTIGUOUS'), u8p, C.POINTER(C.c_size_t), C.c_size_t, u8p, C.c_size_t]
        L.orc_dlp_ct_bytes.restype = C.c_size_t
        L.orc_dlp_ct_bytes.argtypes = [i32p]
        L.orc_dlp_encrypt.argtypes = [i32p, i32p, u8p, C.c_size_t, u8p, u32p, C.c_size_t, u8p]
        L.orc_dlp_decrypt.argtypes = [i32p, i32p, u8p, C.c_size_t, u8p]
        L.orc_dlp_extract.argtypes = [i32p, np.ctypeslib.ndpointer(dtype=np.float64, flags='C_CONTIGUOUS'), u8p, C.c_size_t, u32p, C.c_size_t, u8p]
        for fn in ('orc_ens_ct_bytes', 'orc_ens_key_bytes'):
            getattr(L, fn).restype = C.c_size_t
            getattr(L, fn).argtypes = [i32p]
        L.orc_ens_encaps.argtypes = [i32p, i16p, u32p, C.c_size_t, u8p, u8p, C.POINTER(C.c_size_t)]
        L.orc_ens_decaps.argtypes = [i32p, i16p, u32p, u8p, C.c_size_t, u8p]
        f64p = np.ctypeslib.ndpointer(dtype=np.float64, flags='C_CONTIGUOUS')
        f32p = np.ctypeslib.ndpointer(dtype=np.float32, flags='C_CONTIGUOUS')
        L.orc_falcon_gm.restype = C.POINTER(C.c_double)
        L.orc_falcon_fft.argtypes = [f64p, C.c_uint]
        L.orc_falcon_ifft.argtypes = [f64p, C.c_uint]
        L.orc_falcon_sample.argtypes = [f64p, C.c_int, C.c_int, C.c_int, i32p, u32p, C.c_size_t, i32p, i32p]
        L.orc_falcon_hash_to_ring.argtypes = [i32p, C.c_int, C.c_int, C.c_int, u8p, C.c_size_t]
        L.orc_falcon_sign.argtypes = [i32p, f64p, u8p, C.c_size_t, u32p, C.c_size_t, u8p, C.POINTER(C.c_size_t)]
        L.orc_falcon_verify.argtypes = [i32p, i16p, u8p, C.c_size_t, u8p, C.c_size_t]
        L.orc_falcon_leaf_sigmas.argtypes = [f64p, C.c_int, C.c_int, f32p]
        _LIB = L
    return _LIB


class FalconKey:
    """one Falcon key as the fixture provides it: params, expanded private key (doubles), public key h"""

    def __init__(self, par, sk_bits, h):
        self.par = np.ascontiguousarray(np.asarray(par), dtype=np.int32)
        self.sk = np.ascontiguousarray(np.asarray(sk_bits, dtype=np.int64)).view(np.float64)
        self.h = np.ascontiguousarray(np.asarray(h), dtype=np.int16)
        self.n, self.q, self.q_bits, self.logn = (int(x) for x in self.par[:4])
        self.bd = float(self.par[5:6].view(np.float32)[0])
        self.sig_bytes = (self.n * (self.q_bits - 2) + 7) // 8

    @classmethod
    def from_golden(cls, G, tag):
        return cls(G[f'{tag}_params'], G[f'{tag}_sk'], G[f'{tag}_h'])


def _tape(tape):
    return np.ascontiguousarray(np.asarray(tape).astype(np.int64) & 0xFFFFFFFF, dtype=np.uint32)


def falcon_sample(key, c, tape):
    s1, s2 = np.zeros(key.n, dtype=np.int32), np.zeros(key.n, dtype=np.int32)
    tp = _tape(tape)
    rc = lib().orc_falcon_sample(key.sk, key.n, key.q, key.logn, np.ascontiguousarray(np.asarray(c), dtype=np.int32), tp, tp.size, s1, s2)
    return rc, s1, s2


def falcon_hash_to_ring(key, msg):
    c = np.zeros(key.n, dtype=np.int32)
    m, ml = _msgbuf(msg)
    lib().orc_falcon_hash_to_ring(c, key.n, key.q, key.q_bits, m, ml)
    return c


def falcon_sign(key, msg, tape):
    sig = np.zeros(key.sig_bytes + 8, dtype=np.uint8)
    sl = C.c_size_t(0)
    m, ml = _msgbuf(msg)
    tp = _tape(tape)
    rc = lib().orc_falcon_sign(key.par, key.sk, m, ml, tp, tp.size, sig, C.byref(sl))
    return rc, bytes(sig[:sl.value])


def falcon_verify(key, msg, sig):
    m, ml = _msgbuf(msg)
    s = np.ascontiguousarray(np.frombuffer(bytes(sig), dtype=np.uint8))
    return lib().orc_falcon_verify(key.par, key.h, m, ml, s, s.size)


def falcon_leaf_sigmas(key):
    out = np.zeros(key.n, dtype=np.float32)
    lib().orc_falcon_leaf_sigmas(key.sk, key.n, key.logn, out)
    return out


def kyber_enc(par, t_ntt, Agen, m, noise):
    f = lambda a, dt: np.ascontiguousarray(np.asarray(a), dtype=dt)
    par = f(par, np.int32)
    n, k = int(par[0]), int(par[5])
    u, v = np.zeros(k * n, dtype=np.int32), np.zeros(n, dtype=np.int32)
    assert lib().orc_kyber_cpa_enc(par, f(t_ntt, np.int32), f(Agen, np.int32), f(m, np.uint8), f(noise, np.uint8), u, v) == 0
    return u, v


def kyber_dec(par, s_ntt, u, v):
    f = lambda a, dt: np.ascontiguousarray(np.asarray(a), dtype=dt)
    m = np.zeros(32, dtype=np.uint8)
    assert lib().orc_kyber_cpa_dec(f(par, np.int32), f(s_ntt, np.int32), f(u, np.int32), f(v, np.int32), m) == 0
    return m


class BlissKey:
    def __init__(self, par, f, g, a):
        self.par = np.ascontiguousarray(np.asarray(par), dtype=np.int32)
        self.f, self.g, self.a = (np.ascontiguousarray(np.asarray(x), dtype=np.int16) for x in (f, g, a))
        self.n, self.q = int(par[0]), int(par[1])

    @classmethod
    def from_golden(cls, G, tag):
        return cls(*(G[f'{tag}_{x}'] for x in ('params', 'f', 'g', 'a')))


def bliss_sign(key, msg, tape):
    sig = np.zeros(4096, dtype=np.uint8)
    sl, tr, used = C.c_size_t(0), C.c_int32(0), C.c_size_t(0)
    m, ml = _msgbuf(msg)
    tp = np.ascontiguousarray(np.asarray(tape).astype(np.int64) & 0xFFFFFFFF, dtype=np.uint32)
    rc = lib().orc_bliss_sign(key.par, key.f, key.g, key.a, m, ml, tp, tp.size, sig, C.byref(sl), C.byref(tr), C.byref(used))
    return rc, bytes(sig[:sl.value]), tr.value, used.value


def bliss_sign_hp(key, msg, tape, table, words):
    """bliss_b_sign with the sampler at 128- / 192-bit precision: table uint64 [size][words], words = 2 / 3 prng_64 draws a sample"""
    sig = np.zeros(4096, dtype=np.uint8)
    sl, tr, used = C.c_size_t(0), C.c_int32(0), C.c_size_t(0)
    m, ml = _msgbuf(msg)
    tp = np.ascontiguousarray(np.asarray(tape).astype(np.int64) & 0xFFFFFFFF, dtype=np.uint32)
    tb = np.ascontiguousarray(np.asarray(table, dtype=np.uint64).reshape(-1, words))
    L = lib()
    rc = L.orc_bliss_sign_hp(key.par.ctypes.data_as(C.c_void_p), key.f.ctypes.data_as(C.c_void_p), key.g.ctypes.data_as(C.c_void_p), key.a.ctypes.data_as(C.c_void_p),
                             m.ctypes.data_as(C.c_void_p) if hasattr(m, 'ctypes') else m, C.c_size_t(int(ml)), tp.ctypes.data_as(C.c_void_p), C.c_size_t(tp.size), int(words),
                             tb.ctypes.data_as(C.c_void_p), int(tb.shape[0]), sig.ctypes.data_as(C.c_void_p), C.byref(sl), C.byref(tr), C.byref(used))
    return rc, bytes(sig[:sl.value]), tr.value, used.value


def bliss_verify(key, msg, sig):
    m, ml = _msgbuf(msg)
    s = np.ascontiguousarray(np.frombuffer(bytes(sig), dtype=np.uint8))
    return lib().orc_bliss_verify(key.par, key.a, m, ml, s, s.size)


def huffman_tables(G, z1_bits, z2_bits):
    """the two static code tables a BLISS signature is coded with, picked as entropy_huffman.c:22-34 picks them (signed: the table of
    min(bits, 7) - 1 bits; 2 bits for anything below 4), as ((code, bits), (code, bits)) from the fixture's huffN_* records"""
    def pick(bits):
        hb = min(bits, 7)
        t = {7: 6, 6: 5, 5: 4, 4: 3}.get(hb, 2)
        return (np.ascontiguousarray(G[f'huff{t}_code'].astype(np.int64) & 0xFFFFFFFF, dtype=np.uint32), np.ascontiguousarray(G[f'huff{t}_bits'], dtype=np.uint16))
    return pick(z1_bits), pick(z2_bits)


def bliss_sig_to_huffman(key, raw, tabs):
    (c1, l1), (c2, l2) = tabs
    r = np.ascontiguousarray(np.frombuffer(bytes(raw), dtype=np.uint8)); out = np.zeros(8192, dtype=np.uint8); n = C.c_size_t(0)
    rc = lib().orc_bliss_sig_to_huffman(key.par, r, r.size, c1, l1, c1.size, c2, l2, c2.size, out, C.byref(n))
    return rc, bytes(out[:n.value])


def bliss_sig_from_huffman(key, coded, tabs):
    (c1, l1), (c2, l2) = tabs
    r = np.ascontiguousarray(np.frombuffer(bytes(coded), dtype=np.uint8)); out = np.zeros(8192, dtype=np.uint8); n = C.c_size_t(0)
    rc = lib().orc_bliss_sig_from_huffman(key.par, r if r.size else np.zeros(1, dtype=np.uint8), r.size, c1, l1, c1.size, c2, l2, c2.size, out, C.byref(n))
    return rc, bytes(out[:n.value])


def bac_dists(G, tag):
    """the arithmetic coder's two distributions (bliss_b.c:397-398) from the fixture's *_dist_t / *_dist_z records (64-bit entries as pairs of words)"""
    f = lambda k: np.ascontiguousarray(np.ascontiguousarray(G[k], dtype=np.int32).view(np.uint64))
    return f(f'{tag}_dist_t'), f(f'{tag}_dist_z')


def bac_dist(sigma, bits):
    """gauss_freq_bac_64 (bac.c:83-122) computed here, on this host"""
    d = np.zeros(1 << bits, dtype=np.uint64)
    lib().orc_bac_dist(d, float(sigma), d.size)
    return d


def bliss_sig_to_bac(key, raw, dists):
    r = np.ascontiguousarray(np.frombuffer(bytes(raw), dtype=np.uint8)); out = np.zeros(8192, dtype=np.uint8); n = C.c_size_t(0)
    rc = lib().orc_bliss_sig_to_bac(key.par, r, r.size, dists[0], dists[1], out, C.byref(n))
    return rc, bytes(out[:n.value])


def bliss_sig_from_bac(key, coded, dists):
    r = np.ascontiguousarray(np.frombuffer(bytes(coded), dtype=np.uint8)); out = np.zeros(8192, dtype=np.uint8); n = C.c_size_t(0)
    rc = lib().orc_bliss_sig_from_bac(key.par, r if r.size else np.zeros(1, dtype=np.uint8), r.size, dists[0], dists[1], out, C.byref(n))
    return rc, bytes(out[:n.value])


def huffman_table_for(H, bits):
    """the static table entropy_huffman.c:27-31 walks for fields of `bits` bits: huff_table_gaussian_<min(bits, 7) - 1>, _2 below 4 bits"""
    k = min(bits, 7) - 1
    if k < 3:
        k = 2
    return ((np.asarray(H[f'huff{k}_code']).astype(np.int64) & 0xFFFFFFFF).astype(np.uint32), np.asarray(H[f'huff{k}_bits']).astype(np.uint16))


def huff_row_encode(raw, n, bits, tail, tab):
    code, ln = tab
    r = np.ascontiguousarray(np.frombuffer(bytes(raw), dtype=np.uint8)); out = np.zeros(2 * r.size + 64, dtype=np.uint8); m = C.c_size_t(0)
    rc = lib().orc_huff_row_encode(r, r.size, n, bits, tail, code, ln, code.size, out, C.byref(m))
    return rc, bytes(out[:m.value])


def huff_row_decode(coded, n, bits, tail, tab):
    code, ln = tab
    r = np.ascontiguousarray(np.frombuffer(bytes(coded), dtype=np.uint8)); out = np.zeros((n * bits + 7) // 8 + tail + 64, dtype=np.uint8); m = C.c_size_t(0)
    rc = lib().orc_huff_row_decode(r if r.size else np.zeros(1, dtype=np.uint8), r.size, n, bits, tail, code, ln, code.size, out, C.byref(m))
    return rc, bytes(out[:m.value])


class DilKey:
    """one Dilithium key pair + expanded matrix, as the fixture (or a test) provides it"""

    def __init__(self, par, s1, s2, t, rho, K, tr, t1, Ahat):
        f = lambda a, dt: np.ascontiguousarray(np.asarray(a), dtype=dt)
        self.par = f(par, np.int32)
        self.s1, self.s2, self.t, self.t1, self.Ahat = (f(x, np.int32) for x in (s1, s2, t, t1, Ahat))
        self.rho, self.K, self.tr = (f(x, np.uint8) for x in (rho, K, tr))
        self.n, self.q, self.k, self.l = int(par[0]), int(par[1]), int(par[3]), int(par[4])

    @classmethod
    def from_golden(cls, G, tag):
        return cls(*(G[f'{tag}_{x}'] for x in ('params', 's1', 's2', 't', 'rho', 'K', 'tr', 't1', 'Ahat')))


def _msgbuf(msg):
    m = np.frombuffer(bytes(msg), dtype=np.uint8) if len(msg) else np.zeros(0, dtype=np.uint8)
    return np.ascontiguousarray(np.concatenate([m, np.zeros(1, dtype=np.uint8)])), len(msg)


def dilithium_sign(key, msg):
    sig = np.zeros(8192, dtype=np.uint8)
    sl, tr = C.c_size_t(0), C.c_int32(0)
    m, ml = _msgbuf(msg)
    rc = lib().orc_dilithium_sign(key.par, key.s1, key.s2, key.t, key.K, key.tr, key.Ahat, m, ml, sig, C.byref(sl), C.byref(tr))
    assert rc == 0
    return bytes(sig[:sl.value]), tr.value


def dilithium_attempts(key, msg, kappa0, count):
    """(y [count][l][n] as ExpandMask writes it, w = A y [count][k][n] canonical, c [count][n] int8) of attempts kappa0 .. kappa0 + count - 1"""
    y = np.zeros((count, key.l, key.n), dtype=np.int32)
    w = np.zeros((count, key.k, key.n), dtype=np.int32)
    c = np.zeros((count, key.n), dtype=np.int8)
    m, ml = _msgbuf(msg)
    p = lambda a: a.ctypes.data_as(C.c_void_p)
    f = lib().orc_dilithium_attempts
    f.argtypes = [C.c_void_p] * 5 + [C.c_size_t, C.c_uint32, C.c_uint32] + [C.c_void_p] * 3
    rc = f(p(key.par), p(key.K), p(key.tr), p(key.Ahat), p(m), ml, kappa0, count, p(y), p(w), p(c))
    assert rc == 0
    return y, w, c


def dilithium_verify(key, msg, sig):
    m, ml = _msgbuf(msg)
    s = np.ascontiguousarray(np.frombuffer(bytes(sig), dtype=np.uint8))
    return lib().orc_dilithium_verify(key.par, key.t1, key.rho, key.Ahat, m, ml, s, s.size)


def shake(bits, data, outlen, standard=False):
    """standard=True: FIPS 202; False: the reference's permutation (see sc_oracle_dilithium.c)"""
    C.c_int.in_dll(lib(), 'orc_keccak_standard').value = 1 if standard else 0
    try:
        return _shake(bits, data, outlen)
    finally:
        C.c_int.in_dll(lib(), 'orc_keccak_standard').value = 0


def _shake(bits, data, outlen):
    m, ml = _msgbuf(data)
    out = np.zeros(outlen, dtype=np.uint8)
    lib().orc_shake(bits, m, ml, out, outlen)
    return bytes(out)


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


def cdt_table_blinding(tail, sigma):
    """the table of a sampler created with BLINDING_SAMPLES: the plain one's size, sigma / sqrt(2)"""
    bits = lib().orc_cdt_bits(tail, sigma)
    cdf = np.zeros(1 << bits, dtype=np.uint64)
    lib().orc_cdt_create_blinding(tail, sigma, cdf, 1 << bits)
    return cdf


def cdt32_table(tail, sigma):
    """gaussian_cdf_create_32 restated (float accumulators)"""
    bits = lib().orc_cdt_bits(tail, sigma)
    cdf = np.zeros(1 << bits, dtype=np.uint32)
    lib().orc_cdt32_create(tail, sigma, cdf, 1 << bits)
    return cdf


def cdt32_sample(x, cdf, centre=0):
    x = np.ascontiguousarray(x, dtype=np.uint32); cdf = np.ascontiguousarray(cdf, dtype=np.uint32)
    v = np.empty(x.size, dtype=np.int32)
    lib().orc_cdt32_sample_vector(v, x.size, x, cdf, cdf.size, centre)
    return v


def sample_vector_sca(n, centre, blinding, discard, tape, cdf):
    """sampling.c:107-228: (vector, words consumed); blinding 0 NORMAL / 1 BLINDING / 2 SHUFFLE, discard 0 / 2 / 4 / 6"""
    tp = _tape(tape)
    cdf = np.ascontiguousarray(cdf, dtype=np.uint64)
    v = np.zeros(n, dtype=np.int32)
    used = lib().orc_sample_vector_sca(v, n, centre, blinding, discard, tp, tp.size, cdf, cdf.size)
    return v, used


def cdt_sample(x, cdf, centre=0):
    x = np.ascontiguousarray(x, dtype=np.uint64)
    v = np.empty(x.size, dtype=np.int32)
    lib().orc_cdt_sample_vector(v, x.size, x, cdf, cdf.size, centre)
    return v


class Drbg:
    """the reference's AES-256 CTR-DRBG seeded as create_csprng does"""

    def __init__(self, seed):
        s = np.ascontiguousarray(np.frombuffer(bytes(seed), dtype=np.uint8))
        self.h = lib().orc_drbg_create(s, s.size)

    def mem(self, n):
        out = np.zeros(max(n, 1), dtype=np.uint8)
        lib().orc_drbg_mem(self.h, out, n)
        return bytes(out[:n])

    def u32(self):
        return lib().orc_drbg_32(self.h)

    def u64(self):
        return lib().orc_drbg_64(self.h)

    def close(self):
        if self.h:
            lib().orc_drbg_destroy(self.h)
            self.h = None


def drbg_words(seed, nwords):
    """the first nwords prng_32 draws of the generator create_csprng makes of `seed`: the tape a seeded signing call stands for"""
    d = Drbg(seed)
    out = np.array([d.u32() for _ in range(nwords)], dtype=np.int64)
    d.close()
    return (out & 0xFFFFFFFF).astype(np.uint32)


def kyber_kem_enc(par, t, t_ntt, rho, Agen, m):
    f = lambda a, dt: np.ascontiguousarray(np.asarray(a), dtype=dt)
    par = f(par, np.int32)
    ct, key = np.zeros(lib().orc_kyber_ct_bytes(par), dtype=np.uint8), np.zeros(32, dtype=np.uint8)
    assert lib().orc_kyber_kem_enc(par, f(t, np.int32), f(t_ntt, np.int32), f(rho, np.uint8), f(Agen, np.int32), f(m, np.uint8), ct, key) == 0
    return bytes(ct), bytes(key)


def kyber_kem_dec(par, s_ntt, t, t_ntt, rho, Agen, ct):
    f = lambda a, dt: np.ascontiguousarray(np.asarray(a), dtype=dt)
    key = np.zeros(32, dtype=np.uint8)
    c = np.ascontiguousarray(np.frombuffer(bytes(ct), dtype=np.uint8))
    assert lib().orc_kyber_kem_dec(f(par, np.int32), f(s_ntt, np.int32), f(t, np.int32), f(t_ntt, np.int32), f(rho, np.uint8), f(Agen, np.int32), c, key) == 0
    return bytes(key)


class TeslaKey:
    """ring-TESLA key pair + the reference's public constants, as the fixture provides them"""

    def __init__(self, par, a1, a2, s, e1, e2, t1, t2):
        f = lambda a, dt: np.ascontiguousarray(np.asarray(a), dtype=dt)
        self.par = f(par, np.int32)
        self.a1, self.a2, self.t1, self.t2 = (f(x, np.int32) for x in (a1, a2, t1, t2))
        self.s, self.e1, self.e2 = (f(x, np.int16) for x in (s, e1, e2))
        self.n, self.q = int(par[0]), int(par[2])
        self.sig_bytes = lib().orc_tesla_sig_bytes(self.par)

    @classmethod
    def from_golden(cls, G, tag):
        return cls(*(G[f'{tag}_{x}'] for x in ('params', 'a1', 'a2', 's', 'e1', 'e2', 't1', 't2')))


def tesla_sign(key, msg, tape):
    sig = np.zeros(key.sig_bytes + 8, dtype=np.uint8)
    sl, tr, used = C.c_size_t(0), C.c_int32(0), C.c_size_t(0)
    m, ml = _msgbuf(msg)
    tp = _tape(tape)
    rc = lib().orc_tesla_sign(key.par, key.a1, key.a2, key.s, key.e1, key.e2, m, ml, tp, tp.size, sig, C.byref(sl), C.byref(tr), C.byref(used))
    return rc, bytes(sig[:sl.value]), tr.value, used.value


def tesla_verify(key, msg, sig):
    m, ml = _msgbuf(msg)
    s = np.ascontiguousarray(np.frombuffer(bytes(sig), dtype=np.uint8))
    return lib().orc_tesla_verify(key.par, key.a1, key.a2, key.t1, key.t2, m, ml, s, s.size)


class Ens:
    """ENS-KEM parameter set (oracle/sc_oracle_ens.c)"""

    def __init__(self, par):
        self.par = np.ascontiguousarray(np.asarray(par), dtype=np.int32)
        self.q, self.q_bits, self.n = int(par[1]), int(par[2]), int(par[3])
        self.ct_bytes = lib().orc_ens_ct_bytes(self.par)
        self.key_bytes = lib().orc_ens_key_bytes(self.par)

    def encaps(self, h, tape):
        ct = np.zeros(self.ct_bytes, dtype=np.uint8); key = np.zeros(self.key_bytes, dtype=np.uint8)
        used = C.c_size_t(0)
        tp = _tape(tape)
        rc = lib().orc_ens_encaps(self.par, np.ascontiguousarray(np.asarray(h), dtype=np.int16), tp, tp.size, ct, key, C.byref(used))
        return rc, bytes(ct), bytes(key), used.value

    def decaps(self, g, inv_g, ct):
        c = np.ascontiguousarray(np.frombuffer(bytes(ct), dtype=np.uint8))
        key = np.zeros(self.key_bytes, dtype=np.uint8)
        ig = np.ascontiguousarray((np.asarray(inv_g).astype(np.int64) & 0xFFFFFFFF).astype(np.uint32))
        rc = lib().orc_ens_decaps(self.par, np.ascontiguousarray(np.asarray(g), dtype=np.int16), ig, c, c.size, key)
        return rc, bytes(key)


def dlp_extract(par, tree, ident, tape):
    par = np.ascontiguousarray(np.asarray(par), dtype=np.int32)
    sk = np.zeros(int(par[0]) * int(par[5]) // 8, dtype=np.uint8)
    i, il = _msgbuf(ident)
    tp = _tape(tape)
    rc = lib().orc_dlp_extract(par, np.ascontiguousarray(tree, dtype=np.float64), i, il, tp, tp.size, sk)
    return rc, bytes(sk)


def poly_uniform(n, c, tape):
    v = np.zeros(n, dtype=np.int32)
    cc = np.ascontiguousarray(np.asarray(c), dtype=np.uint16)
    used = C.c_size_t(0)
    tp = _tape(tape)
    rc = lib().orc_poly_uniform(v, n, cc, cc.size, tp, tp.size, C.byref(used))
    return rc, v, used.value


def ensdlp_verify(par, h, msg, sig):
    par = np.ascontiguousarray(np.asarray(par), dtype=np.int32)
    m, ml = _msgbuf(msg)
    s = np.ascontiguousarray(np.frombuffer(bytes(sig), dtype=np.uint8))
    return lib().orc_ensdlp_verify(par, np.ascontiguousarray(np.asarray(h), dtype=np.int16), m, ml, s, s.size)


def ensdlp_basis(n, q, fgFG):
    """(b, b_gs, inv_norm) as create_gpv_matrices leaves them in the reference context: int32 [2n, 2n], float32 [2n, 2n], float32 [2n]"""
    k = np.ascontiguousarray(np.asarray(fgFG), dtype=np.int32)
    b = np.zeros((2 * n, 2 * n), dtype=np.int32)
    bgs = np.zeros((2 * n, 2 * n), dtype=np.float32)
    inv = np.zeros(2 * n, dtype=np.float32)
    lib().orc_ensdlp_expand_basis(n, k, b.reshape(-1))
    lib().orc_ensdlp_gso(n, q, k, bgs.reshape(-1), inv)
    return b, bgs, inv


def ensdlp_sign(par, b, bgs, inv, msg, tape):
    par = np.ascontiguousarray(np.asarray(par), dtype=np.int32)
    m, ml = _msgbuf(msg)
    tp = _tape(tape)
    sig = np.zeros((int(par[0]) * int(par[2]) + 7) // 8, dtype=np.uint8)
    rc = lib().orc_ensdlp_sign(par, b.reshape(-1), bgs.reshape(-1), inv, m, ml, tp, tp.size, sig)
    return rc, bytes(sig)


def ensdlp_rec_m1_bytes(par):
    par = np.ascontiguousarray(np.asarray(par), dtype=np.int32)
    return (int(lib().orc_ensdlp_rec_m1_bits(par)) + 7) // 8


def ensdlp_sign_recovery(par, b, bgs, inv, msg, tape):
    """(rc, signature s1 | s2, m2) - ens_dlp_sig_sign_recovery"""
    par = np.ascontiguousarray(np.asarray(par), dtype=np.int32)
    m, ml = _msgbuf(msg)
    tp = _tape(tape)
    sig = np.zeros((2 * int(par[0]) * int(par[2]) + 7) // 8, dtype=np.uint8)
    m2 = np.zeros(ml + 2, dtype=np.uint8)
    m2len = C.c_size_t(0)
    rc = lib().orc_ensdlp_sign_recovery(par, b.reshape(-1), bgs.reshape(-1), inv, m, ml, tp, tp.size, sig, m2, C.byref(m2len))
    return rc, bytes(sig), bytes(m2[:m2len.value])


def ensdlp_verify_recovery(par, h, m2, sig, fill=0):
    """(verdict, rebuilt message) - ens_dlp_sig_verify_recovery over a buffer that holds m2 and `fill` bytes behind it"""
    par = np.ascontiguousarray(np.asarray(par), dtype=np.int32)
    cap = ensdlp_rec_m1_bytes(par) + len(m2) + 8
    buf = np.full(cap, fill, dtype=np.uint8)
    buf[:len(m2)] = np.frombuffer(bytes(m2), dtype=np.uint8)
    ln = C.c_size_t(len(m2))
    s = np.ascontiguousarray(np.frombuffer(bytes(sig), dtype=np.uint8))
    rc = lib().orc_ensdlp_verify_recovery(par, np.ascontiguousarray(np.asarray(h), dtype=np.int16), buf, C.byref(ln), cap, s, s.size)
    return rc, bytes(buf[:ln.value])


def dlp_encrypt(par, h_ntt, ident, msg, tape):
    par = np.ascontiguousarray(np.asarray(par), dtype=np.int32)
    ct = np.zeros(lib().orc_dlp_ct_bytes(par), dtype=np.uint8)
    i, il = _msgbuf(ident)
    m = np.ascontiguousarray(np.frombuffer(bytes(msg), dtype=np.uint8))
    tp = _tape(tape)
    rc = lib().orc_dlp_encrypt(par, np.ascontiguousarray(np.asarray(h_ntt), dtype=np.int32), i, il, m, tp, tp.size, ct)
    return rc, bytes(ct)


def dlp_decrypt(par, uk_ntt, ct):
    par = np.ascontiguousarray(np.asarray(par), dtype=np.int32)
    c = np.ascontiguousarray(np.frombuffer(bytes(ct), dtype=np.uint8))
    msg = np.zeros(int(par[0]) // 8, dtype=np.uint8)
    rc = lib().orc_dlp_decrypt(par, np.ascontiguousarray(np.asarray(uk_ntt), dtype=np.int32), c, c.size, msg)
    return rc, bytes(msg)


def bliss_keygen(par, nz, tape):
    par = np.ascontiguousarray(np.asarray(par), dtype=np.int32)
    n = int(par[0])
    f, g, a = (np.zeros(n, dtype=np.int16) for _ in range(3))
    used, att = C.c_size_t(0), C.c_int32(0)
    tp = _tape(tape)
    rc = lib().orc_bliss_keygen(par, int(nz[0]), int(nz[1]), tp, tp.size, f, g, a, C.byref(used), C.byref(att))
    return rc, f, g, a, used.value, att.value


class Rlwe:
    """Ring-LWE encryption parameter set + sampler table (oracle/sc_oracle_rlwe.c)"""

    def __init__(self, par):
        self.par = np.ascontiguousarray(np.asarray(par), dtype=np.int32)
        self.q, self.q_bits, self.n = int(par[1]), int(par[2]), int(par[3])
        sig, tail = (float(x) for x in self.par[8:10].view(np.float32))
        self.sigma, self.tail = sig, tail
        self.cdf = cdt_table(tail, sig)
        self.kg_words = lib().orc_rlwe_kg_words(self.par)
        self.enc_words = lib().orc_rlwe_enc_words(self.par)
        self.ct_bytes = lib().orc_rlwe_ct_bytes(self.par)

    def keygen(self, tape):
        a, p, r2 = (np.zeros(self.n, dtype=np.int16) for _ in range(3))
        tp = _tape(tape)
        rc = lib().orc_rlwe_keygen(self.par, self.cdf, self.cdf.size, tp, tp.size, a, p, r2)
        return rc, a, p, r2

    def encrypt(self, a, p, msg, tape):
        f = lambda x: np.ascontiguousarray(np.asarray(x), dtype=np.int16)
        ct = np.zeros(self.ct_bytes, dtype=np.uint8)
        m = np.ascontiguousarray(np.frombuffer(bytes(msg), dtype=np.uint8))
        assert m.size == self.n // 8
        tp = _tape(tape)
        rc = lib().orc_rlwe_encrypt(self.par, self.cdf, self.cdf.size, f(a), f(p), m, tp, tp.size, ct)
        return rc, bytes(ct)

    def decrypt(self, r2, ct):
        c = np.ascontiguousarray(np.frombuffer(bytes(ct), dtype=np.uint8))
        msg = np.zeros(self.n // 8, dtype=np.uint8)
        rc = lib().orc_rlwe_decrypt(self.par, np.ascontiguousarray(np.asarray(r2), dtype=np.int16), c, c.size, msg)
        return rc, bytes(msg)


def kyber_keygen(par, seed):
    f = lambda a, dt: np.ascontiguousarray(np.asarray(a), dtype=dt)
    par = f(par, np.int32)
    kn = int(par[5]) * int(par[0])
    s_ntt, t, t_ntt = (np.zeros(kn, dtype=np.int32) for _ in range(3))
    rho, z = np.zeros(32, dtype=np.uint8), np.zeros(32, dtype=np.uint8)
    sd = f(seed, np.uint8)
    assert lib().orc_kyber_keygen(par, sd, sd.size, s_ntt, t, t_ntt, rho, z) == 0
    return s_ntt, t, t_ntt, rho, z


def expand_matrix(rho, npoly, q, q_bits):
    r = np.ascontiguousarray(np.asarray(rho), dtype=np.uint8)
    out = np.zeros(npoly * 256, dtype=np.int32)
    lib().orc_expand_matrix(r, r.size, npoly, q, q_bits, out)
    return out


def dilithium_keygen(par, eta_bits, seed):
    f = lambda a, dt: np.ascontiguousarray(np.asarray(a), dtype=dt)
    par = f(par, np.int32)
    n, k, l = int(par[0]), int(par[3]), int(par[4])
    s1, s2, t, t1 = np.zeros(l * n, np.int32), np.zeros(k * n, np.int32), np.zeros(k * n, np.int32), np.zeros(k * n, np.int32)
    rho, K, tr = np.zeros(32, np.uint8), np.zeros(32, np.uint8), np.zeros(48, np.uint8)
    sd = f(seed, np.uint8)
    assert lib().orc_dilithium_keygen(par, int(eta_bits), sd, sd.size, s1, s2, t, rho, K, tr, t1) == 0
    return s1, s2, t, rho, K, tr, t1


# ---- module-LWE, XOF-driven variants (oracle/sc_oracle_mlwe_xof.c); every call takes the stream position to start from and returns (values, position after) ----
def _xof_args(shake_bits, seed, skip):
    sd = np.ascontiguousarray(np.asarray(seed), dtype=np.uint8)
    return int(shake_bits), sd.ctypes.data_as(C.c_void_p), C.c_size_t(sd.size), C.c_size_t(int(skip)), sd


def ring_q_xof(shake_bits, seed, skip, npoly, q, q_bits, n=256):
    L = lib(); L.orc_ring_q_xof.restype = C.c_size_t
    out = np.zeros(npoly * n, dtype=np.int32)
    a = _xof_args(shake_bits, seed, skip)
    pos = L.orc_ring_q_xof(a[0], a[1], a[2], a[3], int(npoly), int(n), C.c_int32(int(q)), int(q_bits), out.ctypes.data_as(C.c_void_p))
    return out, int(pos)


def sample_xof(kind, shake_bits, seed, skip, eta, bits, count, fill=99):
    """kind: 'uniform' | 'small' | 'binomial' (uniform_rand_sample_xof / uniform_rand_sample_small_xof / binomial_rand_sample_xof); the output starts
    as `fill` everywhere so that what the reference's function does not write shows"""
    L = lib()
    out = np.full(count, fill, dtype=np.int32)
    a = _xof_args(shake_bits, seed, skip)
    op = out.ctypes.data_as(C.c_void_p)
    if kind == 'uniform':
        L.orc_uniform_sample_xof.restype = C.c_size_t
        pos = L.orc_uniform_sample_xof(a[0], a[1], a[2], a[3], C.c_int32(int(eta)), int(bits), op, C.c_size_t(count))
    elif kind == 'small':
        L.orc_uniform_sample_small_xof.restype = C.c_size_t
        pos = L.orc_uniform_sample_small_xof(a[0], a[1], a[2], a[3], C.c_int32(int(eta)), int(bits), op, C.c_size_t(count))
    else:
        L.orc_binomial_sample_xof.restype = C.c_size_t
        pos = L.orc_binomial_sample_xof(a[0], a[1], a[2], a[3], C.c_int32(int(eta)), op, C.c_size_t(count))
    return out, int(pos)


def rand_product_xof(shake_bits, seed, skip, q, q_bits, k, l, transposed, y, n=256, width=16):
    L = lib(); L.orc_rand_product_xof.restype = C.c_size_t
    w, r = tables(q, n)
    yy = np.ascontiguousarray(np.asarray(y), dtype=np.int32)
    t = np.zeros(k * n, dtype=np.int32)
    a = _xof_args(shake_bits, seed, skip)
    pos = L.orc_rand_product_xof(a[0], a[1], a[2], a[3], C.c_int32(int(q)), int(q_bits), int(k), int(l), 1 if transposed else 0,
                                 yy.ctypes.data_as(C.c_void_p), t.ctypes.data_as(C.c_void_p), int(n), w.ctypes.data_as(C.c_void_p), r.ctypes.data_as(C.c_void_p), int(width))
    return t, int(pos)


def cdt_hp_sample(x, tab, words, centre=0):
    """x uint64 [n][words] (limb 0 first), tab uint64 [size][words] -> int32 [n]: gaussian_cdf_sample_128 / _192 as the reference computes them"""
    xx = np.ascontiguousarray(np.asarray(x, dtype=np.uint64).reshape(-1, words)); tt = np.ascontiguousarray(np.asarray(tab, dtype=np.uint64).reshape(-1, words))
    v = np.zeros(xx.shape[0], dtype=np.int32)
    lib().orc_cdt_hp_sample_vector(v.ctypes.data_as(C.c_void_p), C.c_size_t(xx.shape[0]), xx.ctypes.data_as(C.c_void_p), tt.ctypes.data_as(C.c_void_p), int(tt.shape[0]), int(words), C.c_int32(int(centre)))
    return v
