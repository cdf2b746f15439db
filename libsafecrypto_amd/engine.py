"""ctypes binding of include/sc_hip.h.  No CPU fallback: if the library or a gfx950 device
is missing every compute call raises EngineError."""
import ctypes as C
import os

ROOT = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(ROOT, 'lib', 'libsc_hip.so')

# every symbol include/sc_hip.h declares (the CPU test-suite checks the library exports them)
SYMBOLS = [
    'sc_hip_ctx_create', 'sc_hip_ctx_destroy', 'sc_hip_last_error', 'sc_hip_device_count', 'sc_hip_synchronize',
    'sc_hip_plan_create', 'sc_hip_plan_destroy', 'sc_hip_plan_q', 'sc_hip_plan_n', 'sc_hip_plan_root',
    'sc_hip_plan_tables', 'sc_hip_plan_set_arith',
    'sc_hip_ntt_fwd', 'sc_hip_ntt_inv', 'sc_hip_pointwise', 'sc_hip_polymul_ntt', 'sc_hip_polymul_ntt_pair', 'sc_hip_polymul',
    'sc_hip_normalize', 'sc_hip_center', 'sc_hip_poly_add', 'sc_hip_poly_sub', 'sc_hip_matvec',
    'sc_hip_cdt_create', 'sc_hip_cdt_table_size', 'sc_hip_cdt_build_table', 'sc_hip_cdt_build_table_blinding', 'sc_hip_cdt_create_from_table',
    'sc_hip_cdt_destroy', 'sc_hip_cdt_size', 'sc_hip_cdt_table', 'sc_hip_cdt_sample',
    'sc_hip_cdt32_build_table', 'sc_hip_cdt32_create', 'sc_hip_cdt32_create_from_table', 'sc_hip_cdt32_table', 'sc_hip_cdt32_sample',
    'sc_hip_ntt_fwd_host', 'sc_hip_ntt_inv_host', 'sc_hip_polymul_ntt_host', 'sc_hip_cdt_sample_host',
    'sc_hip_pointwise_host', 'sc_hip_normalize_host', 'sc_hip_center_host',
    'sc_hip_dilithium_create', 'sc_hip_dilithium_destroy', 'sc_hip_dilithium_set_privkey', 'sc_hip_dilithium_set_pubkey',
    'sc_hip_dilithium_sig_max', 'sc_hip_dilithium_sign', 'sc_hip_dilithium_verify', 'sc_hip_dilithium_verify_keys', 'sc_hip_dilithium_sign_keys', 'sc_hip_dilithium_sign_host',
    'sc_hip_dilithium_verify_host', 'sc_hip_dilithium_keygen',
    'sc_hip_kyber_create', 'sc_hip_kyber_destroy', 'sc_hip_kyber_set_pubkey', 'sc_hip_kyber_set_privkey', 'sc_hip_kyber_noise_bytes',
    'sc_hip_kyber_cpa_enc', 'sc_hip_kyber_cpa_dec', 'sc_hip_kyber_set_kem_pubkey', 'sc_hip_kyber_ct_bytes', 'sc_hip_kyber_kem_encaps', 'sc_hip_kyber_kem_decaps', 'sc_hip_kyber_kem_encaps_keys', 'sc_hip_kyber_kem_decaps_keys', 'sc_hip_kyber_keygen',
    'sc_hip_bliss_create', 'sc_hip_bliss_destroy', 'sc_hip_bliss_set_key', 'sc_hip_bliss_sig_bytes', 'sc_hip_bliss_sign', 'sc_hip_bliss_sign_seeded', 'sc_hip_bliss_verify', 'sc_hip_bliss_sign_keys', 'sc_hip_bliss_verify_keys', 'sc_hip_bliss_keygen', 'sc_hip_bliss_set_tie_margin', 'sc_hip_bliss_set_cdt_table', 'sc_hip_bliss_set_cdt_table_hp', 'sc_hip_bliss_tape_words_hp', 'sc_hip_bliss_set_entropy', 'sc_hip_bliss_set_entropy_bac', 'sc_hip_bliss_tape_words',
    'sc_hip_falcon_create', 'sc_hip_falcon_destroy', 'sc_hip_falcon_set_privkey', 'sc_hip_falcon_set_pubkey', 'sc_hip_falcon_set_privkeys', 'sc_hip_falcon_set_pubkeys', 'sc_hip_falcon_sign_keys', 'sc_hip_falcon_verify_keys', 'sc_hip_falcon_sig_bytes',
    'sc_hip_falcon_sample', 'sc_hip_falcon_sign', 'sc_hip_falcon_sign_seeded', 'sc_hip_falcon_verify', 'sc_hip_falcon_set_mapping',
    'sc_hip_ntt_pwr', 'sc_hip_ntt_flip', 'sc_hip_ntt_invert', 'sc_hip_ntt_div', 'sc_hip_mult_sparse', 'sc_hip_mult_sparse_16',
    'sc_hip_poly_mul_scalar', 'sc_hip_poly_mod_negate', 'sc_hip_poly_add_scalar', 'sc_hip_poly_cmp_not_equal',
    'sc_hip_vec_absmax', 'sc_hip_vec_scalar', 'sc_hip_ntt_invert_host', 'sc_hip_ntt_div_host', 'sc_hip_mult_sparse_host',
    'sc_hip_drbg_mem', 'sc_hip_drbg_words', 'sc_hip_mlwe_expand_matrix', 'sc_hip_mlwe_expand_matrix_xof', 'sc_hip_mlwe_sample_xof', 'sc_hip_cdt_hp_create_from_table', 'sc_hip_cdt_hp_precision', 'sc_hip_cdt_hp_table', 'sc_hip_cdt_hp_sample', 'sc_hip_cdt_hp_sample_host',
    'sc_hip_tesla_create', 'sc_hip_tesla_destroy', 'sc_hip_tesla_set_key', 'sc_hip_tesla_sig_bytes', 'sc_hip_tesla_sign', 'sc_hip_tesla_sign_seeded', 'sc_hip_tesla_verify',
    'sc_hip_rlwe_create', 'sc_hip_rlwe_destroy', 'sc_hip_rlwe_set_pubkey', 'sc_hip_rlwe_set_privkey', 'sc_hip_rlwe_msg_bytes', 'sc_hip_rlwe_ct_bytes',
    'sc_hip_poly_uniform', 'sc_hip_dlp_ibe_extract', 'sc_hip_ens_dlp_sig_verify', 'sc_hip_ens_dlp_signer_create', 'sc_hip_ens_dlp_signer_destroy', 'sc_hip_ens_dlp_signer_set_key', 'sc_hip_ens_dlp_signer_set_privkey', 'sc_hip_ens_dlp_signer_set_mapping',
    'sc_hip_ens_dlp_sig_bytes', 'sc_hip_ens_dlp_sig_tape_words', 'sc_hip_ens_dlp_sig_sign', 'sc_hip_ens_dlp_sig_rec_bytes', 'sc_hip_ens_dlp_sig_rec_m1_bytes', 'sc_hip_ens_dlp_sig_sign_recovery', 'sc_hip_ens_dlp_sig_verify_recovery', 'sc_hip_dlp_create', 'sc_hip_dlp_destroy', 'sc_hip_dlp_set_pubkey', 'sc_hip_dlp_set_user_key', 'sc_hip_dlp_msg_bytes',
    'sc_hip_dlp_ct_bytes', 'sc_hip_dlp_enc_words', 'sc_hip_dlp_encrypt', 'sc_hip_dlp_decrypt',
    'sc_hip_ens_create', 'sc_hip_ens_destroy', 'sc_hip_ens_set_pubkey', 'sc_hip_ens_set_privkey', 'sc_hip_ens_ct_bytes', 'sc_hip_ens_key_bytes',
    'sc_hip_ens_encaps', 'sc_hip_ens_decaps',
    'sc_hip_rlwe_enc_words', 'sc_hip_rlwe_keygen_words', 'sc_hip_rlwe_encrypt', 'sc_hip_rlwe_decrypt', 'sc_hip_rlwe_keygen',
]
SYMBOLS += ['sc_hip_shard_range', 'sc_hip_multi_create', 'sc_hip_multi_destroy', 'sc_hip_multi_workers', 'sc_hip_multi_ctx', 'sc_hip_multi_plan_create', 'sc_hip_multi_plan_destroy',
            'sc_hip_multi_ntt_fwd_host', 'sc_hip_multi_ntt_inv_host', 'sc_hip_multi_polymul_ntt_host', 'sc_hip_multi_dilithium_create', 'sc_hip_multi_dilithium_destroy',
            'sc_hip_multi_dilithium_set_privkey', 'sc_hip_multi_dilithium_set_pubkey', 'sc_hip_multi_dilithium_sig_max', 'sc_hip_multi_dilithium_sign_host',
            'sc_hip_multi_dilithium_verify_host']
SYMBOLS += ['sc_hip_huffman_create', 'sc_hip_huffman_destroy', 'sc_hip_huffman_encode_rows', 'sc_hip_huffman_decode_rows']
SYMBOLS += [f'sc_hip_{k}_{f}_host' for k in ('bliss', 'tesla', 'falcon') for f in ('sign', 'verify')] + ['sc_hip_kyber_kem_encaps_host', 'sc_hip_kyber_kem_decaps_host']
SYMBOLS += [f'sc_hip_multi_{k}_{f}' for k in ('bliss', 'tesla') for f in ('create', 'destroy', 'set_key', 'sig_bytes', 'sign_host', 'verify_host')] + ['sc_hip_multi_bliss_set_cdt_table']
SYMBOLS += [f'sc_hip_multi_falcon_{f}' for f in ('create', 'destroy', 'set_privkey', 'set_pubkey', 'sig_bytes', 'sign_host', 'verify_host')]
SYMBOLS += [f'sc_hip_multi_kyber_{f}' for f in ('create', 'destroy', 'set_pubkey', 'set_privkey', 'ct_bytes', 'kem_encaps_host', 'kem_decaps_host')]
# include/sc_hip_bench.h (measurement / self-test hooks, not part of the drop-in ABI)
BENCH_SYMBOLS = ['sc_hip_bench_set_host_pin', 'sc_hip_bench_set_ntt_stream', 'sc_hip_dilithium_workspace_residue', 'sc_hip_falcon_workspace_residue', 'sc_hip_polymul_ntt_steps', 'sc_hip_time_op', 'sc_hip_shake_host', 'sc_hip_falcon_sample_profile', 'sc_hip_falcon_set_variant', 'sc_hip_dilithium_set_tuning', 'sc_hip_dilithium_fin_tables_host', 'sc_hip_cdt_guide_host', 'sc_hip_bliss_set_speculation', 'sc_hip_rlwe_set_arith']
# include/sc_hip_vtable.h
VTABLE_SYMBOLS = ['sc_hip_sampler_create_hp', 'sc_hip_sampler_set_rng32', 'sc_hip_utils_sampling_fill_sca', 'sc_hip_utils_arith_ntt32', 'sc_hip_vtable_errors', 'sc_hip_utils_arith_ntt_fill', 'sc_hip_utils_arith_poly_fill', 'sc_hip_utils_arith_vec_fill',
                  'sc_hip_sampler_create', 'sc_hip_sampler_set_rng', 'sc_hip_sampler_destroy', 'sc_hip_utils_sampling_fill', 'sc_hip_vtable_bind_device']


class EngineError(RuntimeError):
    pass


def build_cdt_table(tail, sigma):
    """host-only: the table gaussian_cdf_create_64 would build on THIS host"""
    import numpy as np
    L = load_library()
    size = L.sc_hip_cdt_table_size(tail, sigma)
    t = np.empty(size, dtype=np.uint64)
    _chk(L, L.sc_hip_cdt_build_table(tail, sigma, C.c_void_p(t.ctypes.data), size), 'cdt_build_table')
    return t


_LIB = None


def load_library(path=LIB_PATH):
    global _LIB
    if _LIB is not None:
        return _LIB
    if not os.path.exists(path):
        raise EngineError(f'{path} is missing: run __graft_entry__.build() (make -C libsafecrypto_amd/csrc); '
                          'there is no CPU fallback')
    # torch's wheel bundles its own libamdhip64.so.7: import it first so that this process holds
    # ONE HIP runtime (ours resolves the same SONAME to the already-loaded copy); two runtimes in
    # one process leave the second without a visible GPU.  A pure C host links /opt/rocm's.
    import torch  # noqa: F401
    L = C.CDLL(path)
    vp, i32, sz = C.c_void_p, C.c_int32, C.c_size_t
    L.sc_hip_bench_set_host_pin.argtypes = [C.c_int]
    L.sc_hip_bench_set_host_pin.restype = None
    L.sc_hip_bench_set_ntt_stream.argtypes = [C.c_int]
    L.sc_hip_bench_set_ntt_stream.restype = None
    if os.environ.get('SC_TEST_HOST_PIN') in ('0', '2'):     # test harness only (tools/dbg/bisect_pin.sh): the library itself reads no environment
        L.sc_hip_bench_set_host_pin(int(os.environ['SC_TEST_HOST_PIN']))
    L.sc_hip_last_error.restype = C.c_char_p
    L.sc_hip_ctx_create.argtypes = [C.c_int, C.POINTER(vp)]
    L.sc_hip_ctx_destroy.argtypes = [vp]
    L.sc_hip_ctx_destroy.restype = None
    L.sc_hip_synchronize.argtypes = [vp, vp]
    L.sc_hip_plan_create.argtypes = [vp, i32, C.c_int, C.POINTER(vp)]
    L.sc_hip_plan_destroy.argtypes = [vp]
    L.sc_hip_plan_destroy.restype = None
    L.sc_hip_plan_q.argtypes = [vp]
    L.sc_hip_plan_q.restype = i32
    L.sc_hip_plan_n.argtypes = [vp]
    L.sc_hip_plan_root.argtypes = [vp]
    L.sc_hip_plan_root.restype = i32
    L.sc_hip_plan_tables.argtypes = [vp, vp, vp]
    L.sc_hip_plan_set_arith.argtypes = [vp, C.c_int]
    for name in ('sc_hip_ntt_fwd', 'sc_hip_ntt_inv'):
        getattr(L, name).argtypes = [vp, sz, vp, vp, vp]
    for name in ('sc_hip_polymul_ntt', 'sc_hip_polymul', 'sc_hip_pointwise'):
        getattr(L, name).argtypes = [vp, sz, vp, vp, vp, vp]
    L.sc_hip_polymul_ntt_pair.argtypes = [vp, sz, vp, vp, vp, vp, vp, vp]
    for name in ('sc_hip_normalize', 'sc_hip_center'):
        getattr(L, name).argtypes = [vp, sz, vp, vp, vp]
    for name in ('sc_hip_poly_add', 'sc_hip_poly_sub'):
        getattr(L, name).argtypes = [vp, sz, vp, vp, vp, vp]
    L.sc_hip_matvec.argtypes = [vp, sz, C.c_int, C.c_int, C.c_int, vp, vp, vp, vp]
    L.sc_hip_cdt_create.argtypes = [vp, C.c_float, C.c_float, C.POINTER(vp)]
    L.sc_hip_cdt_table_size.argtypes = [C.c_float, C.c_float]
    L.sc_hip_cdt_build_table.argtypes = [C.c_float, C.c_float, vp, C.c_int]
    L.sc_hip_cdt_build_table_blinding.argtypes = [C.c_float, C.c_float, vp, C.c_int]
    L.sc_hip_cdt_create_from_table.argtypes = [vp, vp, C.c_int, C.POINTER(vp)]
    L.sc_hip_cdt_destroy.argtypes = [vp]
    L.sc_hip_cdt_destroy.restype = None
    L.sc_hip_cdt_size.argtypes = [vp]
    L.sc_hip_cdt_table.argtypes = [vp, vp]
    L.sc_hip_cdt_sample.argtypes = [vp, sz, vp, i32, vp, vp]
    L.sc_hip_cdt32_build_table.argtypes = [C.c_float, C.c_float, vp, C.c_int]
    L.sc_hip_cdt32_create.argtypes = [vp, C.c_float, C.c_float, C.POINTER(vp)]
    L.sc_hip_cdt32_create_from_table.argtypes = [vp, vp, C.c_int, C.POINTER(vp)]
    L.sc_hip_cdt32_table.argtypes = [vp, vp]
    L.sc_hip_cdt32_sample.argtypes = [vp, sz, vp, i32, vp, vp]
    L.sc_hip_ntt_fwd_host.argtypes = [vp, sz, vp, vp]
    L.sc_hip_ntt_inv_host.argtypes = [vp, sz, vp, vp]
    L.sc_hip_polymul_ntt_host.argtypes = [vp, sz, vp, vp, vp]
    L.sc_hip_cdt_sample_host.argtypes = [vp, sz, vp, i32, vp]
    L.sc_hip_dilithium_create.argtypes = [vp, vp, C.POINTER(vp)]
    L.sc_hip_dilithium_destroy.argtypes = [vp]
    L.sc_hip_dilithium_destroy.restype = None
    L.sc_hip_dilithium_set_privkey.argtypes = [vp, vp, vp, vp, vp, vp, vp]
    L.sc_hip_dilithium_set_pubkey.argtypes = [vp, vp, vp, vp]
    L.sc_hip_dilithium_sig_max.argtypes = [vp]
    L.sc_hip_dilithium_sig_max.restype = sz
    L.sc_hip_dilithium_sign.argtypes = [vp, sz, vp, sz, vp, C.c_uint32, vp, sz, vp, vp, vp]
    L.sc_hip_dilithium_verify.argtypes = [vp, sz, vp, sz, vp, C.c_uint32, vp, sz, vp, vp, vp]
    L.sc_hip_dilithium_verify_keys.argtypes = [vp, sz, vp, vp, vp, sz, vp, C.c_uint32, vp, sz, vp, vp, vp]
    L.sc_hip_dilithium_sign_keys.argtypes = [vp, sz, vp, vp, vp, vp, vp, vp, vp, sz, vp, C.c_uint32, vp, sz, vp, vp, vp]
    L.sc_hip_dilithium_sign_host.argtypes = [vp, sz, vp, sz, vp, C.c_uint32, vp, sz, vp, vp]
    L.sc_hip_dilithium_verify_host.argtypes = [vp, sz, vp, sz, vp, C.c_uint32, vp, sz, vp, vp]
    L.sc_hip_shake_host.argtypes = [vp, C.c_int, C.c_int, vp, sz, vp, sz]
    L.sc_hip_kyber_create.argtypes = [vp, vp, C.POINTER(vp)]
    L.sc_hip_kyber_destroy.argtypes = [vp]
    L.sc_hip_kyber_destroy.restype = None
    L.sc_hip_kyber_set_pubkey.argtypes = [vp, vp, vp]
    L.sc_hip_kyber_set_privkey.argtypes = [vp, vp]
    L.sc_hip_kyber_noise_bytes.argtypes = [vp]
    L.sc_hip_kyber_noise_bytes.restype = sz
    L.sc_hip_kyber_cpa_enc.argtypes = [vp, sz, vp, vp, sz, vp, vp, vp]
    L.sc_hip_kyber_cpa_dec.argtypes = [vp, sz, vp, vp, vp, vp]
    L.sc_hip_bliss_create.argtypes = [vp, vp, C.POINTER(vp)]
    L.sc_hip_bliss_destroy.argtypes = [vp]
    L.sc_hip_bliss_destroy.restype = None
    L.sc_hip_bliss_set_key.argtypes = [vp, vp, vp, vp]
    L.sc_hip_bliss_sig_bytes.argtypes = [vp]
    L.sc_hip_bliss_sig_bytes.restype = sz
    L.sc_hip_bliss_sign.argtypes = [vp, sz, vp, sz, vp, C.c_uint32, vp, sz, vp, vp, sz, vp, vp, vp, vp]
    L.sc_hip_bliss_sign_seeded.argtypes = [vp, sz, vp, sz, vp, C.c_uint32, vp, sz, sz, vp, sz, vp, vp, vp, vp]
    L.sc_hip_bliss_verify.argtypes = [vp, sz, vp, sz, vp, C.c_uint32, vp, sz, vp, vp, vp]
    L.sc_hip_bliss_set_tie_margin.argtypes = [vp, C.c_double]
    L.sc_hip_bliss_set_speculation.argtypes = [vp, C.c_int]
    L.sc_hip_rlwe_set_arith.argtypes = [vp, C.c_int]
    L.sc_hip_bliss_tape_words.argtypes = [C.c_int, C.c_uint]
    L.sc_hip_bliss_tape_words.restype = sz
    L.sc_hip_bliss_tape_words_hp.argtypes = [C.c_int, C.c_uint, C.c_int]
    L.sc_hip_bliss_tape_words_hp.restype = sz
    L.sc_hip_bliss_set_cdt_table_hp.argtypes = [vp, C.c_int, vp, C.c_int]
    L.sc_hip_bliss_set_entropy.argtypes = [vp, C.c_int, vp, vp, C.c_uint32, vp, vp, C.c_uint32]
    L.sc_hip_bliss_set_entropy_bac.argtypes = [vp, vp, vp]
    L.sc_hip_bliss_sign_keys.argtypes = [vp, sz, vp, vp, vp, vp, sz, vp, C.c_uint32, vp, sz, vp, vp, sz, vp, vp, vp, vp]
    L.sc_hip_bliss_verify_keys.argtypes = [vp, sz, vp, vp, sz, vp, C.c_uint32, vp, sz, vp, vp, vp]
    L.sc_hip_bliss_keygen.argtypes = [vp, sz, C.c_int, C.c_int, vp, sz, vp, vp, vp, vp, vp, vp, vp]
    L.sc_hip_ntt_pwr.argtypes = [vp, sz, vp, i32, vp, vp]
    L.sc_hip_ntt_flip.argtypes = [vp, sz, vp, vp, vp]
    L.sc_hip_ntt_invert.argtypes = [vp, sz, sz, vp, vp, vp]
    L.sc_hip_ntt_div.argtypes = [vp, sz, sz, vp, vp, vp, vp]
    L.sc_hip_mult_sparse.argtypes = [vp, sz, C.c_int, C.c_int, vp, vp, vp, vp]
    L.sc_hip_mult_sparse_16.argtypes = [vp, sz, C.c_int, C.c_int, vp, vp, vp, vp]
    L.sc_hip_poly_mul_scalar.argtypes = [vp, sz, vp, i32, vp, vp]
    L.sc_hip_poly_mod_negate.argtypes = [vp, sz, i32, vp, vp, vp]
    L.sc_hip_poly_add_scalar.argtypes = [vp, sz, sz, vp, vp, vp]
    L.sc_hip_poly_cmp_not_equal.argtypes = [vp, sz, sz, vp, vp, vp, vp]
    L.sc_hip_vec_absmax.argtypes = [vp, sz, sz, vp, vp, vp]
    L.sc_hip_vec_scalar.argtypes = [vp, sz, sz, vp, vp, vp, vp]
    L.sc_hip_kyber_set_kem_pubkey.argtypes = [vp, vp, vp]
    L.sc_hip_kyber_ct_bytes.argtypes = [vp]
    L.sc_hip_kyber_ct_bytes.restype = sz
    L.sc_hip_kyber_kem_encaps.argtypes = [vp, sz, vp, vp, sz, vp, vp]
    L.sc_hip_kyber_kem_decaps.argtypes = [vp, sz, vp, sz, vp, vp]
    L.sc_hip_kyber_kem_encaps_keys.argtypes = [vp, sz, vp, vp, vp, vp, vp, sz, vp, vp]
    L.sc_hip_kyber_kem_decaps_keys.argtypes = [vp, sz, vp, vp, vp, vp, vp, sz, vp, vp]
    L.sc_hip_dilithium_keygen.argtypes = [vp, sz, C.c_int, vp, sz, sz, vp, vp, vp, vp, vp, vp, vp, vp]
    L.sc_hip_kyber_keygen.argtypes = [vp, sz, vp, sz, sz, vp, vp, vp, vp, vp, vp]
    L.sc_hip_tesla_create.argtypes = [vp, vp, C.POINTER(vp)]
    L.sc_hip_tesla_destroy.argtypes = [vp]
    L.sc_hip_tesla_destroy.restype = None
    L.sc_hip_tesla_set_key.argtypes = [vp, vp, vp, vp, vp, vp, vp, vp]
    L.sc_hip_tesla_sig_bytes.argtypes = [vp]
    L.sc_hip_tesla_sig_bytes.restype = sz
    L.sc_hip_tesla_sign.argtypes = [vp, sz, vp, sz, vp, C.c_uint32, vp, sz, vp, vp, sz, vp, vp, vp, vp]
    L.sc_hip_tesla_sign_seeded.argtypes = [vp, sz, vp, sz, vp, C.c_uint32, vp, sz, sz, vp, sz, vp, vp, vp, vp]
    L.sc_hip_tesla_verify.argtypes = [vp, sz, vp, sz, vp, C.c_uint32, vp, sz, vp, vp, vp]
    L.sc_hip_polymul_ntt_steps.argtypes = [vp, sz, vp, vp, vp, C.c_int, vp]
    L.sc_hip_poly_uniform.argtypes = [vp, sz, C.c_int, vp, C.c_int, vp, sz, vp, vp, vp, vp, vp]
    L.sc_hip_ens_dlp_sig_verify.argtypes = [vp, sz, vp, sz, vp, C.c_uint32, vp, sz, vp, vp, vp]
    L.sc_hip_ens_dlp_signer_create.argtypes = [vp, vp, C.POINTER(vp)]
    L.sc_hip_ens_dlp_signer_destroy.argtypes = [vp]
    L.sc_hip_ens_dlp_signer_destroy.restype = None
    L.sc_hip_ens_dlp_signer_set_key.argtypes = [vp, vp, vp, vp]
    L.sc_hip_ens_dlp_signer_set_privkey.argtypes = [vp, vp]
    L.sc_hip_ens_dlp_signer_set_mapping.argtypes = [vp, C.c_int]
    L.sc_hip_dilithium_set_tuning.argtypes = [vp, C.c_int, C.c_int]
    L.sc_hip_ens_dlp_sig_bytes.argtypes = [vp]
    L.sc_hip_ens_dlp_sig_bytes.restype = sz
    L.sc_hip_ens_dlp_sig_tape_words.argtypes = [vp]
    L.sc_hip_ens_dlp_sig_tape_words.restype = sz
    L.sc_hip_ens_dlp_sig_sign.argtypes = [vp, sz, vp, sz, vp, C.c_uint32, vp, sz, vp, vp, sz, vp, vp]
    for fn in ('sc_hip_ens_dlp_sig_rec_bytes', 'sc_hip_ens_dlp_sig_rec_m1_bytes'):
        getattr(L, fn).argtypes = [vp]
        getattr(L, fn).restype = sz
    L.sc_hip_ens_dlp_sig_sign_recovery.argtypes = [vp, sz, vp, sz, vp, C.c_uint32, vp, sz, vp, vp, sz, vp, sz, vp, vp, vp]
    L.sc_hip_ens_dlp_sig_verify_recovery.argtypes = [vp, sz, vp, sz, vp, vp, sz, vp, vp, sz, vp, vp, vp]
    L.sc_hip_dlp_ibe_extract.argtypes = [vp, sz, C.c_int, vp, sz, vp, C.c_uint32, vp, sz, vp, vp, sz, vp, vp]
    L.sc_hip_dlp_create.argtypes = [vp, vp, C.POINTER(vp)]
    L.sc_hip_dlp_destroy.argtypes = [vp]
    L.sc_hip_dlp_destroy.restype = None
    L.sc_hip_dlp_set_pubkey.argtypes = [vp, vp]
    L.sc_hip_dlp_set_user_key.argtypes = [vp, vp]
    for fn in ('sc_hip_dlp_msg_bytes', 'sc_hip_dlp_ct_bytes', 'sc_hip_dlp_enc_words'):
        getattr(L, fn).argtypes = [vp]
        getattr(L, fn).restype = sz
    L.sc_hip_dlp_encrypt.argtypes = [vp, sz, vp, sz, vp, C.c_uint32, vp, sz, vp, sz, vp, vp, sz, vp, vp]
    L.sc_hip_dlp_decrypt.argtypes = [vp, sz, vp, sz, vp, sz, vp]
    L.sc_hip_ens_create.argtypes = [vp, vp, C.POINTER(vp)]
    L.sc_hip_ens_destroy.argtypes = [vp]
    L.sc_hip_ens_destroy.restype = None
    L.sc_hip_ens_set_pubkey.argtypes = [vp, vp]
    L.sc_hip_ens_set_privkey.argtypes = [vp, vp, vp]
    for fn in ('sc_hip_ens_ct_bytes', 'sc_hip_ens_key_bytes'):
        getattr(L, fn).argtypes = [vp]
        getattr(L, fn).restype = sz
    L.sc_hip_ens_encaps.argtypes = [vp, sz, vp, sz, vp, vp, sz, vp, sz, vp, vp, vp]
    L.sc_hip_ens_decaps.argtypes = [vp, sz, vp, sz, vp, sz, vp]
    L.sc_hip_rlwe_create.argtypes = [vp, vp, C.POINTER(vp)]
    L.sc_hip_rlwe_destroy.argtypes = [vp]
    L.sc_hip_rlwe_destroy.restype = None
    L.sc_hip_rlwe_set_pubkey.argtypes = [vp, vp, vp]
    L.sc_hip_rlwe_set_privkey.argtypes = [vp, vp]
    for fn in ('sc_hip_rlwe_msg_bytes', 'sc_hip_rlwe_ct_bytes', 'sc_hip_rlwe_enc_words', 'sc_hip_rlwe_keygen_words'):
        getattr(L, fn).argtypes = [vp]
        getattr(L, fn).restype = sz
    L.sc_hip_rlwe_encrypt.argtypes = [vp, sz, vp, sz, vp, sz, vp, sz, vp]
    L.sc_hip_rlwe_decrypt.argtypes = [vp, sz, vp, sz, vp, sz, vp]
    L.sc_hip_rlwe_keygen.argtypes = [vp, sz, vp, sz, vp, vp, vp, vp]
    L.sc_hip_mlwe_expand_matrix.argtypes = [vp, sz, vp, sz, sz, C.c_int, i32, C.c_int, vp, vp]
    L.sc_hip_cdt_hp_create_from_table.argtypes = [vp, C.c_int, vp, C.c_int, C.POINTER(vp)]
    L.sc_hip_cdt_hp_precision.argtypes = [vp]
    L.sc_hip_cdt_hp_table.argtypes = [vp, vp]
    L.sc_hip_cdt_hp_sample.argtypes = [vp, sz, vp, i32, vp, vp]
    L.sc_hip_mlwe_expand_matrix_xof.argtypes = [vp, sz, vp, sz, sz, C.c_int, sz, C.c_int, C.c_int, i32, C.c_int, vp, vp]
    L.sc_hip_mlwe_sample_xof.argtypes = [vp, C.c_int, sz, vp, sz, sz, C.c_int, sz, i32, C.c_int, sz, vp, sz, C.POINTER(C.c_size_t), vp]
    L.sc_hip_drbg_mem.argtypes = [vp, sz, vp, sz, sz, vp, sz, sz, vp]
    L.sc_hip_drbg_words.argtypes = [vp, sz, vp, sz, sz, vp, sz, sz, vp]
    L.sc_hip_falcon_create.argtypes = [vp, vp, C.POINTER(vp)]
    L.sc_hip_falcon_destroy.argtypes = [vp]
    L.sc_hip_falcon_destroy.restype = None
    L.sc_hip_falcon_set_privkey.argtypes = [vp, vp]
    L.sc_hip_falcon_set_pubkey.argtypes = [vp, vp]
    L.sc_hip_falcon_set_privkeys.argtypes = [vp, sz, vp]
    L.sc_hip_falcon_set_pubkeys.argtypes = [vp, sz, vp]
    L.sc_hip_falcon_sign_keys.argtypes = [vp, sz, vp, vp, sz, vp, C.c_uint32, vp, sz, vp, vp, sz, vp, vp, vp]
    L.sc_hip_falcon_verify_keys.argtypes = [vp, sz, vp, vp, sz, vp, C.c_uint32, vp, sz, vp, vp, vp]
    L.sc_hip_falcon_sig_bytes.argtypes = [vp]
    L.sc_hip_falcon_sig_bytes.restype = sz
    L.sc_hip_falcon_set_mapping.argtypes = [vp, C.c_int]
    L.sc_hip_shard_range.argtypes = [sz, C.c_int, C.c_int, C.POINTER(sz), C.POINTER(sz)]
    L.sc_hip_shard_range.restype = None
    L.sc_hip_multi_create.argtypes = [vp, C.c_int, C.POINTER(vp)]
    L.sc_hip_multi_destroy.argtypes = [vp]
    L.sc_hip_multi_destroy.restype = None
    L.sc_hip_multi_workers.argtypes = [vp]
    L.sc_hip_multi_ctx.argtypes = [vp, C.c_int]
    L.sc_hip_multi_ctx.restype = vp
    L.sc_hip_multi_plan_create.argtypes = [vp, i32, C.c_int, C.POINTER(vp)]
    L.sc_hip_multi_plan_destroy.argtypes = [vp]
    L.sc_hip_multi_plan_destroy.restype = None
    for name in ('sc_hip_multi_ntt_fwd_host', 'sc_hip_multi_ntt_inv_host'):
        getattr(L, name).argtypes = [vp, sz, vp, vp]
    L.sc_hip_multi_polymul_ntt_host.argtypes = [vp, sz, vp, vp, vp]
    L.sc_hip_multi_dilithium_create.argtypes = [vp, vp, C.POINTER(vp)]
    L.sc_hip_multi_dilithium_destroy.argtypes = [vp]
    L.sc_hip_multi_dilithium_destroy.restype = None
    L.sc_hip_multi_dilithium_set_privkey.argtypes = [vp, vp, vp, vp, vp, vp, vp]
    L.sc_hip_multi_dilithium_set_pubkey.argtypes = [vp, vp, vp, vp]
    L.sc_hip_multi_dilithium_sig_max.argtypes = [vp]
    L.sc_hip_multi_dilithium_sig_max.restype = sz
    L.sc_hip_multi_dilithium_sign_host.argtypes = [vp, sz, vp, sz, vp, C.c_uint32, vp, sz, vp, vp]
    L.sc_hip_multi_dilithium_verify_host.argtypes = [vp, sz, vp, sz, vp, C.c_uint32, vp, sz, vp, vp]
    L.sc_hip_falcon_sample_profile.argtypes = [vp, sz, vp, vp, sz, vp, vp, vp, vp]
    L.sc_hip_falcon_set_variant.argtypes = [vp, C.c_int, C.c_int]
    L.sc_hip_falcon_sample.argtypes = [vp, sz, vp, vp, sz, vp, vp, vp, vp, vp]
    L.sc_hip_falcon_sign.argtypes = [vp, sz, vp, sz, vp, C.c_uint32, vp, sz, vp, vp, sz, vp, vp, vp]
    L.sc_hip_falcon_sign_seeded.argtypes = [vp, sz, vp, sz, vp, C.c_uint32, vp, sz, sz, vp, sz, vp, vp, vp]
    L.sc_hip_falcon_verify.argtypes = [vp, sz, vp, sz, vp, C.c_uint32, vp, sz, vp, vp, vp]
    L.sc_hip_time_op.argtypes = [vp, C.c_int, sz, vp, vp, vp, vp, C.c_int, C.POINTER(C.c_float)]
    _LIB = L
    return L


def _chk(L, rc, what):
    if rc != 0:
        raise EngineError(f'{what}: {L.sc_hip_last_error().decode()}')


def _ptr(t):
    """device (torch tensor) or host (numpy array) pointer"""
    if hasattr(t, 'data_ptr'):
        return C.c_void_p(t.data_ptr())
    return C.c_void_p(t.ctypes.data)


def _stream():
    import torch
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _rows(t):
    """row stride (in elements) of a [batch][row] tensor; a size-1 leading dimension may carry any stride"""
    assert t.dim() == 2 and t.stride(1) == 1
    return t.stride(0) if t.shape[0] > 1 else t.shape[1]


def shard_range(total, parts, index):
    """sc_hip_shard_range: (begin, count) of slice `index` of `parts` (host-only, no device needed)"""
    L = load_library()
    b, c = C.c_size_t(), C.c_size_t()
    L.sc_hip_shard_range(total, parts, index, C.byref(b), C.byref(c))
    return b.value, c.value


class Multi:
    """The in-library multi-device driver (sc_hip_multi_*): one host thread per worker, contiguous slices of a host batch, no collective.
    devices: HIP device ordinals, one worker each (an ordinal may repeat)."""

    def __init__(self, devices):
        import numpy as np
        self.L = load_library()
        d = np.ascontiguousarray(np.asarray(devices), dtype=np.int32)
        h = C.c_void_p()
        _chk(self.L, self.L.sc_hip_multi_create(_ptr(d), d.size, C.byref(h)), 'sc_hip_multi_create')
        self.h = h
        self.workers = self.L.sc_hip_multi_workers(h)
        self._plans, self._dil, self._schemes = {}, None, []

    def scheme(self, kind, params):
        """kind: 'bliss' | 'falcon' | 'tesla' | 'kyber' - the scheme handle replicated on every worker (sc_hip_multi_<kind>_*)"""
        m = MultiScheme(self, kind, params)
        self._schemes.append(m)
        return m

    def close(self):
        for m in self._schemes:
            m.close()
        self._schemes = []
        for p in self._plans.values():
            self.L.sc_hip_multi_plan_destroy(p)
        self._plans = {}
        if self._dil:
            self.L.sc_hip_multi_dilithium_destroy(self._dil); self._dil = None
        if self.h:
            self.L.sc_hip_multi_destroy(self.h); self.h = None

    def _plan(self, q, n):
        if (q, n) not in self._plans:
            p = C.c_void_p()
            _chk(self.L, self.L.sc_hip_multi_plan_create(self.h, q, n, C.byref(p)), 'sc_hip_multi_plan_create')
            self._plans[(q, n)] = p
        return self._plans[(q, n)]

    def polymul_ntt_host(self, q, a, bhat):
        import numpy as np
        a = np.ascontiguousarray(a, dtype=np.int32); bhat = np.ascontiguousarray(bhat, dtype=np.int32)
        out = np.empty_like(a)
        _chk(self.L, self.L.sc_hip_multi_polymul_ntt_host(self._plan(q, a.shape[1]), a.shape[0], _ptr(a), _ptr(bhat), _ptr(out)), 'multi_polymul_ntt_host')
        return out

    def ntt_host(self, q, a, inverse=False):
        import numpy as np
        a = np.ascontiguousarray(a, dtype=np.int32)
        out = np.empty_like(a)
        f = self.L.sc_hip_multi_ntt_inv_host if inverse else self.L.sc_hip_multi_ntt_fwd_host
        _chk(self.L, f(self._plan(q, a.shape[1]), a.shape[0], _ptr(a), _ptr(out)), 'multi_ntt_host')
        return out

    def dilithium(self, params, s1, s2, t, K, tr, Ahat, t1, rho):
        import numpy as np
        par = np.ascontiguousarray(np.asarray(params), dtype=np.int32)
        d = C.c_void_p()
        _chk(self.L, self.L.sc_hip_multi_dilithium_create(self.h, _ptr(par), C.byref(d)), 'sc_hip_multi_dilithium_create')
        self._dil = d
        c32 = lambda x: np.ascontiguousarray(np.asarray(x), dtype=np.int32)
        c8 = lambda x: np.ascontiguousarray(np.asarray(x), dtype=np.uint8)
        A, a1, a2, at, aK, atr, at1, arho = c32(Ahat), c32(s1), c32(s2), c32(t), c8(K), c8(tr), c32(t1), c8(rho)      # converted copies must outlive the calls
        _chk(self.L, self.L.sc_hip_multi_dilithium_set_privkey(d, _ptr(a1), _ptr(a2), _ptr(at), _ptr(aK), _ptr(atr), _ptr(A)), 'multi_dilithium_set_privkey')
        _chk(self.L, self.L.sc_hip_multi_dilithium_set_pubkey(d, _ptr(at1), _ptr(arho), _ptr(A)), 'multi_dilithium_set_pubkey')
        self.sig_max = self.L.sc_hip_multi_dilithium_sig_max(d)

    def dilithium_sign_host(self, msgs, fixed_len):
        """msgs uint8 [batch][stride] (numpy) -> signatures uint8 [batch][sig_max], lengths, trials"""
        import numpy as np
        msgs = np.ascontiguousarray(msgs, dtype=np.uint8)
        n = msgs.shape[0]
        sigs = np.zeros((n, self.sig_max), dtype=np.uint8); sl = np.zeros(n, dtype=np.uint32); tr = np.zeros(n, dtype=np.uint32)
        _chk(self.L, self.L.sc_hip_multi_dilithium_sign_host(self._dil, n, _ptr(msgs), msgs.shape[1], C.c_void_p(), fixed_len, _ptr(sigs), self.sig_max, _ptr(sl), _ptr(tr)), 'multi_dilithium_sign_host')
        return sigs, sl, tr

    def dilithium_verify_host(self, msgs, fixed_len, sigs, siglens):
        import numpy as np
        msgs = np.ascontiguousarray(msgs, dtype=np.uint8); sigs = np.ascontiguousarray(sigs, dtype=np.uint8); sl = np.ascontiguousarray(siglens, dtype=np.uint32)
        v = np.ones(msgs.shape[0], dtype=np.uint32)
        _chk(self.L, self.L.sc_hip_multi_dilithium_verify_host(self._dil, msgs.shape[0], _ptr(msgs), msgs.shape[1], C.c_void_p(), fixed_len, _ptr(sigs), sigs.shape[1], _ptr(sl), _ptr(v)), 'multi_dilithium_verify_host')
        return v


class HuffmanRows:
    """Static-Huffman coding of signature rows that are one polynomial of `field_bits`-bit signed fields + raw tail bytes (sc_hip_huffman_*):
    Falcon, ENS / DLP signatures, ring-TESLA with SC_FLAG_0_ENTROPY_HUFFMAN.  codes / bits: the reference's static table for the field width."""

    def __init__(self, eng, codes, bits, field_bits):
        import numpy as np
        self.eng, self.L, self.field_bits = eng, eng.L, int(field_bits)
        c = np.ascontiguousarray(np.asarray(codes), dtype=np.uint32); b = np.ascontiguousarray(np.asarray(bits), dtype=np.uint16)
        h = C.c_void_p()
        _chk(self.L, self.L.sc_hip_huffman_create(eng.h, _ptr(c), _ptr(b), c.size, self.field_bits, C.byref(h)), 'sc_hip_huffman_create')
        self.h = h
        self.L.sc_hip_huffman_destroy.restype = None
        self.L.sc_hip_huffman_destroy.argtypes = [C.c_void_p]
        sz, vp = C.c_size_t, C.c_void_p
        self.L.sc_hip_huffman_encode_rows.argtypes = [vp, sz, vp, sz, vp, vp, C.c_int, C.c_int, vp]
        self.L.sc_hip_huffman_decode_rows.argtypes = [vp, sz, vp, sz, vp, vp, sz, vp, vp, C.c_int, C.c_int, vp]

    def close(self):
        if self.h:
            self.L.sc_hip_huffman_destroy(self.h)
            self.h = None

    def encode(self, rows, lens, n, tail=0, status=None):
        """rows uint8 [B][stride] (device, raw signatures; coded in place), lens int32 [B] (device, receives the coded lengths)"""
        _chk(self.L, self.L.sc_hip_huffman_encode_rows(self.h, rows.shape[0], _ptr(rows), _rows(rows), _ptr(lens), _ptr(status) if status is not None else C.c_void_p(),
                                                       n, tail, _stream()), 'huffman_encode_rows')
        return rows, lens

    def decode(self, rows, lens, n, tail=0):
        """coded rows (device) -> raw rows [B][raw bytes rounded up to 16], raw lengths, malformed flags"""
        import torch
        B = rows.shape[0]
        rawb = (n * self.field_bits + 7) // 8 + tail
        stride = (rawb + 15) & ~15
        out = torch.zeros((B, stride), dtype=torch.uint8, device=rows.device)
        ol = torch.zeros(B, dtype=torch.int32, device=rows.device); bad = torch.zeros(B, dtype=torch.int32, device=rows.device)
        _chk(self.L, self.L.sc_hip_huffman_decode_rows(self.h, B, _ptr(rows), _rows(rows), _ptr(lens), _ptr(out), stride, _ptr(ol), _ptr(bad), n, tail, _stream()), 'huffman_decode_rows')
        return out, ol, bad


class MultiScheme:
    """BLISS-B / Falcon / ring-TESLA / Kyber under one key on every worker of a Multi; host batches (numpy) are sharded by the library.
    With one worker this is the one-device *_host entry (sc_hip_<kind>_sign_host, ...) behind one fan-out thread."""

    def __init__(self, multi, kind, params):
        import numpy as np
        assert kind in ('bliss', 'falcon', 'tesla', 'kyber')
        self.multi, self.L, self.kind = multi, multi.L, kind
        self.par = np.ascontiguousarray(np.asarray(params), dtype=np.int32)
        self.f = lambda name: getattr(self.L, f'sc_hip_multi_{kind}_{name}')
        h = C.c_void_p()
        _chk(self.L, self.f('create')(multi.h, _ptr(self.par), C.byref(h)), f'sc_hip_multi_{kind}_create')
        self.h = h
        self.f('destroy').restype = None
        self.f('destroy').argtypes = [C.c_void_p]
        sz, vp, u32 = C.c_size_t, C.c_void_p, C.c_uint32
        if kind == 'kyber':
            self.f('ct_bytes').restype = sz; self.f('ct_bytes').argtypes = [vp]
            self.f('kem_encaps_host').argtypes = [vp, sz, vp, vp, sz, vp]
            self.f('kem_decaps_host').argtypes = [vp, sz, vp, sz, vp]
        else:
            self.f('sig_bytes').restype = sz; self.f('sig_bytes').argtypes = [vp]
            self.f('sign_host').argtypes = [vp, sz, vp, sz, vp, u32, vp, sz, vp, vp, sz, vp, vp] + ([vp] if kind != 'falcon' else [])
            self.f('verify_host').argtypes = [vp, sz, vp, sz, vp, u32, vp, sz, vp, vp]

    def close(self):
        if self.h:
            self.f('destroy')(self.h)
            self.h = None

    def set_cdt_table(self, precision_bits, table):
        """BLISS-B: the reference context's own sampler table on every worker (64-bit entries, or [size][2 | 3] limbs of a 128- / 192-bit sampler)"""
        import numpy as np
        assert self.kind == 'bliss'
        t = np.ascontiguousarray(np.asarray(table), dtype=np.uint64)
        self.L.sc_hip_multi_bliss_set_cdt_table.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int]
        _chk(self.L, self.L.sc_hip_multi_bliss_set_cdt_table(self.h, int(precision_bits), _ptr(t), t.size // (int(precision_bits) // 64)), 'multi_bliss_set_cdt_table')

    def set_key(self, *a, **kw):
        import numpy as np
        c = lambda x, dt: np.ascontiguousarray(np.asarray(x), dtype=dt) if x is not None else None
        p = lambda x: _ptr(x) if x is not None else C.c_void_p()
        if self.kind == 'bliss':
            arrs = [c(a[0], np.int16), c(a[1], np.int16), c(a[2], np.int16)]
            _chk(self.L, self.f('set_key')(self.h, *[p(x) for x in arrs]), 'multi_bliss_set_key')
        elif self.kind == 'tesla':
            dts = [np.int32, np.int32, np.int16, np.int16, np.int16, np.int32, np.int32]
            arrs = [c(x, dt) for x, dt in zip(a, dts)]
            _chk(self.L, self.f('set_key')(self.h, *[p(x) for x in arrs]), 'multi_tesla_set_key')
        elif self.kind == 'falcon':
            sk, pk = c(kw.get('sk'), np.float64), c(kw.get('pk'), np.int16)
            if sk is not None:
                _chk(self.L, self.f('set_privkey')(self.h, _ptr(sk)), 'multi_falcon_set_privkey')
            if pk is not None:
                _chk(self.L, self.f('set_pubkey')(self.h, _ptr(pk)), 'multi_falcon_set_pubkey')
        else:
            t_ntt, Agen, t, rho, s_ntt = (kw.get(k) for k in ('t_ntt', 'Agen', 't', 'rho', 's_ntt'))
            arrs = [c(t_ntt, np.int32), c(Agen, np.int32), c(t, np.int32), c(rho, np.uint8)]
            _chk(self.L, self.f('set_pubkey')(self.h, *[p(x) for x in arrs]), 'multi_kyber_set_pubkey')
            if s_ntt is not None:
                sa = c(s_ntt, np.int32)
                _chk(self.L, self.f('set_privkey')(self.h, _ptr(sa)), 'multi_kyber_set_privkey')
        if self.kind == 'kyber':
            self.ct_bytes = self.f('ct_bytes')(self.h)
        else:
            self.sig_bytes = self.f('sig_bytes')(self.h)
            self.sig_stride = (self.sig_bytes + 15) & ~15

    def sign_host(self, msgs, fixed_len, tapes):
        """msgs uint8 [B][stride], tapes uint32 [B][W] (numpy) -> sigs uint8 [B][sig_stride], siglens, trials (None for Falcon), status"""
        import numpy as np
        msgs = np.ascontiguousarray(msgs, dtype=np.uint8); tapes = np.ascontiguousarray(tapes, dtype=np.uint32)
        n = msgs.shape[0]
        sigs = np.zeros((n, self.sig_stride), dtype=np.uint8); sl = np.zeros(n, dtype=np.uint32); tr = np.zeros(n, dtype=np.uint32); st = np.zeros(n, dtype=np.uint32)
        args = [self.h, n, _ptr(msgs), msgs.shape[1], C.c_void_p(), fixed_len, _ptr(tapes), tapes.shape[1], C.c_void_p(), _ptr(sigs), self.sig_stride, _ptr(sl)]
        args += [_ptr(st)] if self.kind == 'falcon' else [_ptr(tr), _ptr(st)]
        _chk(self.L, self.f('sign_host')(*args), f'multi_{self.kind}_sign_host')
        return sigs, sl, (None if self.kind == 'falcon' else tr), st

    def verify_host(self, msgs, fixed_len, sigs, siglens):
        import numpy as np
        msgs = np.ascontiguousarray(msgs, dtype=np.uint8); sigs = np.ascontiguousarray(sigs, dtype=np.uint8); sl = np.ascontiguousarray(siglens, dtype=np.uint32)
        v = np.ones(msgs.shape[0], dtype=np.uint32)
        _chk(self.L, self.f('verify_host')(self.h, msgs.shape[0], _ptr(msgs), msgs.shape[1], C.c_void_p(), fixed_len, _ptr(sigs), sigs.shape[1], _ptr(sl), _ptr(v)), f'multi_{self.kind}_verify_host')
        return v

    def encaps_host(self, m):
        import numpy as np
        m = np.ascontiguousarray(m, dtype=np.uint8)
        n = m.shape[0]
        stride = (self.ct_bytes + 15) & ~15
        ct = np.zeros((n, stride), dtype=np.uint8); keys = np.zeros((n, 32), dtype=np.uint8)
        _chk(self.L, self.f('kem_encaps_host')(self.h, n, _ptr(m), _ptr(ct), stride, _ptr(keys)), 'multi_kyber_kem_encaps_host')
        return ct, keys

    def decaps_host(self, ct):
        import numpy as np
        ct = np.ascontiguousarray(ct, dtype=np.uint8)
        keys = np.zeros((ct.shape[0], 32), dtype=np.uint8)
        _chk(self.L, self.f('kem_decaps_host')(self.h, ct.shape[0], _ptr(ct), ct.shape[1], _ptr(keys)), 'multi_kyber_kem_decaps_host')
        return keys


class Engine:
    """One per (process, device): mirrors the per-context wiring safecrypto_create does
    (sc->sc_ntt / sc->sc_gauss, src/schemes/sig/dilithium/dilithium.c:207-209)."""

    def __init__(self, device=0):
        self.L = load_library()
        h = C.c_void_p()
        _chk(self.L, self.L.sc_hip_ctx_create(device, C.byref(h)), 'sc_hip_ctx_create')
        self.h = h
        self.device = device

    def close(self):
        if self.h:
            self.L.sc_hip_ctx_destroy(self.h)
            self.h = None

    def plan(self, q, n):
        return NttPlan(self, q, n)

    def kyber(self, params):
        return Kyber(self, params)

    def huffman_rows(self, codes, bits, field_bits):
        return HuffmanRows(self, codes, bits, field_bits)

    def bliss(self, params):
        return Bliss(self, params)

    def falcon(self, params):
        return Falcon(self, params)

    def tesla(self, params):
        return Tesla(self, params)

    def rlwe(self, params):
        return Rlwe(self, params)

    def poly_uniform(self, n, c, tapes, tape_lens=None):
        """utils_arith_poly_t uniform_32 for a batch: tapes [batch][words] (device) -> polynomials int32 [batch][n], words used, status"""
        import numpy as np, torch
        cc = np.ascontiguousarray(np.asarray(c), dtype=np.uint16)
        batch, dev = tapes.shape[0], tapes.device
        out = torch.zeros((batch, n), dtype=torch.int32, device=dev)
        used = torch.zeros(batch, dtype=torch.int32, device=dev)
        status = torch.zeros(batch, dtype=torch.int32, device=dev)
        _chk(self.L, self.L.sc_hip_poly_uniform(self.h, batch, n, _ptr(cc), cc.size, _ptr(tapes), _rows(tapes), _ptr(tape_lens) if tape_lens is not None else C.c_void_p(),
                                                _ptr(out), _ptr(used), _ptr(status), _stream()), 'poly_uniform')
        return out, used, status

    def ens(self, params):
        return Ens(self, params)

    def ens_dlp_signer(self, params):
        return EnsDlpSigner(self, params)

    def dlp(self, params):
        return Dlp(self, params)

    def dilithium(self, params):
        return Dilithium(self, params)

    def shake(self, bits, data, outlen, standard=False):
        """the reference's XOF (or FIPS 202 with standard=True) evaluated on the device"""
        import numpy as np
        d = np.frombuffer(bytes(data), dtype=np.uint8) if len(data) else np.zeros(1, dtype=np.uint8)
        out = np.zeros(outlen, dtype=np.uint8)
        _chk(self.L, self.L.sc_hip_shake_host(self.h, bits, 1 if standard else 0, _ptr(d), len(data), _ptr(out), outlen), 'shake_host')
        return bytes(out)

    def sampler(self, tail=None, sigma=None, table=None):
        return CdtSampler(self, tail, sigma, table)

    def sampler32(self, tail=None, sigma=None, table=None):
        return CdtSampler32(self, tail, sigma, table)

    def sampler_hp(self, precision_bits, table):
        return CdtSamplerHp(self, precision_bits, table)

    def synchronize(self):
        _chk(self.L, self.L.sc_hip_synchronize(self.h, _stream()), 'sc_hip_synchronize')

    def poly_add(self, a, b, out=None):
        import torch
        out = torch.empty_like(a) if out is None else out
        _chk(self.L, self.L.sc_hip_poly_add(self.h, a.numel(), _ptr(a), _ptr(b), _ptr(out), _stream()), 'poly_add')
        return out

    def poly_sub(self, a, b, out=None):
        import torch
        out = torch.empty_like(a) if out is None else out
        _chk(self.L, self.L.sc_hip_poly_sub(self.h, a.numel(), _ptr(a), _ptr(b), _ptr(out), _stream()), 'poly_sub')
        return out

    # the reference's AES-256 CTR-DRBG, one generator per seed row (uint8 [batch][seed_len] device tensor)
    def drbg_mem(self, seeds, nbytes):
        import torch
        out = torch.empty((seeds.shape[0], nbytes), dtype=torch.uint8, device=seeds.device)
        _chk(self.L, self.L.sc_hip_drbg_mem(self.h, seeds.shape[0], _ptr(seeds), _rows(seeds), seeds.shape[1], _ptr(out), nbytes, nbytes, _stream()), 'drbg_mem')
        return out

    def expand_matrix(self, rho, npoly, q, q_bits):
        """rho uint8 [nkeys][len] (device) -> int32 [nkeys][npoly][256]: the matrix polynomials in generation order"""
        import torch
        out = torch.empty((rho.shape[0], npoly, 256), dtype=torch.int32, device=rho.device)
        _chk(self.L, self.L.sc_hip_mlwe_expand_matrix(self.h, rho.shape[0], _ptr(rho), _rows(rho), rho.shape[1], npoly, int(q), int(q_bits), _ptr(out), _stream()), 'mlwe_expand_matrix')
        return out

    # the XOF-driven forms (module_lwe.c:330-367,449-512,550-560): SHAKE128 / SHAKE256 of each seed row, read from byte `skip` of the squeezed stream
    def expand_matrix_xof(self, seeds, npoly, q, q_bits, shake_bits=128, skip=0, n=256):
        """seeds uint8 [nkeys][len] (device) -> int32 [nkeys][npoly][n]: uniform_random_ring_q_xof npoly times on xof_absorb(seed)"""
        import torch
        out = torch.empty((seeds.shape[0], npoly, n), dtype=torch.int32, device=seeds.device)
        _chk(self.L, self.L.sc_hip_mlwe_expand_matrix_xof(self.h, seeds.shape[0], _ptr(seeds), _rows(seeds), seeds.shape[1], int(shake_bits), int(skip), int(npoly), int(n), int(q), int(q_bits),
                                                          _ptr(out), _stream()), 'mlwe_expand_matrix_xof')
        return out

    XOF_KINDS = {'uniform': 0, 'small': 1, 'binomial': 2}

    def sample_xof(self, kind, seeds, eta, bits, count, shake_bits=128, skip=0, out=None, fill=0):
        """uniform_rand_sample_xof / uniform_rand_sample_small_xof / binomial_rand_sample_xof for a stream per seed row -> (int32 [nseeds][count], stream position after);
        entries the reference's function does not write keep what `out` held (a fresh `out` is filled with `fill`)"""
        import torch
        if out is None:
            out = torch.full((seeds.shape[0], count), int(fill), dtype=torch.int32, device=seeds.device)
        pos = C.c_size_t(0)
        _chk(self.L, self.L.sc_hip_mlwe_sample_xof(self.h, self.XOF_KINDS[kind], seeds.shape[0], _ptr(seeds), _rows(seeds), seeds.shape[1], int(shake_bits), int(skip), int(eta), int(bits),
                                                   int(count), _ptr(out), out.stride(0), C.byref(pos), _stream()), 'mlwe_sample_xof')
        return out, int(pos.value)

    def drbg_words(self, seeds, nwords):
        import torch
        out = torch.empty((seeds.shape[0], nwords), dtype=torch.int32, device=seeds.device)
        _chk(self.L, self.L.sc_hip_drbg_words(self.h, seeds.shape[0], _ptr(seeds), _rows(seeds), seeds.shape[1], _ptr(out), nwords, nwords, _stream()), 'drbg_words')
        return out

    # utils_arith_poly / utils_arith_vectors entries (device tensors, int32; per-polynomial ops take [batch][n])
    def poly_mul_scalar(self, a, scalar, out=None):
        import torch
        out = torch.empty_like(a) if out is None else out
        _chk(self.L, self.L.sc_hip_poly_mul_scalar(self.h, a.numel(), _ptr(a), int(scalar), _ptr(out), _stream()), 'poly_mul_scalar')
        return out

    def poly_mod_negate(self, a, q, out=None):
        import torch
        out = torch.empty_like(a) if out is None else out
        _chk(self.L, self.L.sc_hip_poly_mod_negate(self.h, a.numel(), int(q), _ptr(a), _ptr(out), _stream()), 'poly_mod_negate')
        return out

    def poly_add_scalar(self, polys, scalars):
        """in place: coefficient 0 of polynomial i += scalars[i]"""
        _chk(self.L, self.L.sc_hip_poly_add_scalar(self.h, polys.shape[0], polys.shape[1], _ptr(polys), _ptr(scalars), _stream()), 'poly_add_scalar')
        return polys

    def _per_poly(self, fn, what, a, b=None):
        import torch
        out = torch.empty(a.shape[0], dtype=torch.int32, device=a.device)
        args = [self.h, a.shape[0], a.shape[1], _ptr(a)] + ([_ptr(b)] if b is not None else []) + [_ptr(out), _stream()]
        _chk(self.L, fn(*args), what)
        return out

    def poly_cmp_not_equal(self, a, b):
        return self._per_poly(self.L.sc_hip_poly_cmp_not_equal, 'poly_cmp_not_equal', a, b)

    def vec_absmax(self, v):
        return self._per_poly(self.L.sc_hip_vec_absmax, 'vec_absmax', v)

    def vec_scalar(self, t, u):
        return self._per_poly(self.L.sc_hip_vec_scalar, 'vec_scalar', t, u)

    def mult_sparse(self, t, pos):
        """t int32 or int16 [batch][n], pos int32 [batch][omega] -> int32 [batch][n] (mul_32_sparse / mul_32_sparse_16)"""
        import torch
        out = torch.empty(t.shape, dtype=torch.int32, device=t.device)
        fn = self.L.sc_hip_mult_sparse_16 if t.dtype == torch.int16 else self.L.sc_hip_mult_sparse
        _chk(self.L, fn(self.h, t.shape[0], t.shape[1], pos.shape[1], _ptr(t), _ptr(pos), _ptr(out), _stream()), 'mult_sparse')
        return out


class NttPlan:
    """(q, n) context: the GPU counterpart of ntt_params_t + w/r tables."""

    def __init__(self, eng, q, n):
        self.eng, self.L = eng, eng.L
        h = C.c_void_p()
        _chk(self.L, self.L.sc_hip_plan_create(eng.h, q, n, C.byref(h)), 'sc_hip_plan_create')
        self.h, self.q, self.n = h, q, n

    def close(self):
        if self.h:
            self.L.sc_hip_plan_destroy(self.h)
            self.h = None

    @property
    def root(self):
        return self.L.sc_hip_plan_root(self.h)

    def tables(self):
        import numpy as np
        w = np.empty(self.n, dtype=np.int32)
        r = np.empty(self.n, dtype=np.int32)
        _chk(self.L, self.L.sc_hip_plan_tables(self.h, _ptr(w), _ptr(r)), 'plan_tables')
        return w, r

    def set_arith(self, mode):
        _chk(self.L, self.L.sc_hip_plan_set_arith(self.h, mode), 'plan_set_arith')

    def _batch(self, t):
        assert t.dtype.is_floating_point is False and t.element_size() == 4 and t.is_contiguous()
        assert t.numel() % self.n == 0
        return t.numel() // self.n

    def _out(self, a, out):
        import torch
        return torch.empty_like(a) if out is None else out

    def fwd(self, a, out=None):
        out = self._out(a, out)
        _chk(self.L, self.L.sc_hip_ntt_fwd(self.h, self._batch(a), _ptr(a), _ptr(out), _stream()), 'ntt_fwd')
        return out

    def inv(self, a, out=None):
        out = self._out(a, out)
        _chk(self.L, self.L.sc_hip_ntt_inv(self.h, self._batch(a), _ptr(a), _ptr(out), _stream()), 'ntt_inv')
        return out

    def polymul_ntt(self, a, bhat, out=None):
        out = self._out(a, out)
        _chk(self.L, self.L.sc_hip_polymul_ntt(self.h, self._batch(a), _ptr(a), _ptr(bhat), _ptr(out), _stream()), 'polymul_ntt')
        return out

    def polymul_ntt_pair(self, a, b1hat, b2hat):
        """(INTT(NTT(a) o b1hat), INTT(NTT(a) o b2hat)) for two stored NTT-domain polynomials (n words each), one forward transform per input"""
        import torch
        o1, o2 = torch.empty_like(a), torch.empty_like(a)
        _chk(self.L, self.L.sc_hip_polymul_ntt_pair(self.h, self._batch(a), _ptr(a), _ptr(b1hat), _ptr(b2hat), _ptr(o1), _ptr(o2), _stream()), 'polymul_ntt_pair')
        return o1, o2

    def polymul_ntt_steps(self, a, bhat, out, steps):
        """`steps` fused products on the same buffers as one HIP graph launch on torch's current (non-default) stream"""
        _chk(self.L, self.L.sc_hip_polymul_ntt_steps(self.h, self._batch(a), _ptr(a), _ptr(bhat), _ptr(out), int(steps), _stream()), 'polymul_ntt_steps')
        return out

    def polymul(self, a, b, out=None):
        out = self._out(a, out)
        _chk(self.L, self.L.sc_hip_polymul(self.h, self._batch(a), _ptr(a), _ptr(b), _ptr(out), _stream()), 'polymul')
        return out

    def matvec(self, Ahat, y, k, l, transposed=False):
        """t[b][i] = INTT(sum_j A[i][j] o NTT(y[b][j])) - create_rand_product_* with the matrix expanded"""
        import torch
        batch = y.numel() // (l * self.n)
        t = torch.empty((batch, k, self.n), dtype=torch.int32, device=y.device)
        _chk(self.L, self.L.sc_hip_matvec(self.h, batch, k, l, 1 if transposed else 0, _ptr(Ahat), _ptr(y), _ptr(t), _stream()), 'matvec')
        return t

    def rand_product_xof(self, seed, y, k, l, q_bits, transposed=False, shake_bits=128, skip=0):
        """create_rand_product_32_xof / _16_xof (module_lwe.c:751-912) under one seed: the matrix drawn from SHAKE(seed) in the loops' order, then the product.
        seed uint8 [len] or [1][len] (device); y int32 [batch][l][n]; -> (t int32 [batch][k][n] canonical, stream position after)"""
        A = self.eng.expand_matrix_xof(seed.reshape(1, -1), k * l, self.q, q_bits, shake_bits, skip, self.n)
        return self.matvec(A, y, k, l, transposed), int(skip) + k * l * self.n * 4

    def pointwise(self, a, b, out=None):
        out = self._out(a, out)
        _chk(self.L, self.L.sc_hip_pointwise(self.h, a.numel(), _ptr(a), _ptr(b), _ptr(out), _stream()), 'pointwise')
        return out

    def normalize(self, a, out=None):
        out = self._out(a, out)
        _chk(self.L, self.L.sc_hip_normalize(self.h, a.numel(), _ptr(a), _ptr(out), _stream()), 'normalize')
        return out

    def center(self, a, out=None):
        out = self._out(a, out)
        _chk(self.L, self.L.sc_hip_center(self.h, a.numel(), _ptr(a), _ptr(out), _stream()), 'center')
        return out

    def pwr(self, x, e, out=None):
        out = self._out(x, out)
        _chk(self.L, self.L.sc_hip_ntt_pwr(self.h, x.numel(), _ptr(x), int(e), _ptr(out), _stream()), 'ntt_pwr')
        return out

    def flip(self, x):
        """flip_32 (ntt32_flip_generic) of x [batch][n]"""
        import torch
        out = torch.empty_like(x)
        _chk(self.L, self.L.sc_hip_ntt_flip(self.h, self._batch(x), _ptr(x), _ptr(out), _stream()), 'ntt_flip')
        return out

    def invert(self, v):
        """in place on v [batch][n]; returns the per-polynomial return codes (0 ok / 1 a coefficient was 0 mod q)"""
        import torch
        rc = torch.empty(v.shape[0], dtype=torch.int32, device=v.device)
        _chk(self.L, self.L.sc_hip_ntt_invert(self.h, v.shape[0], v.shape[1], _ptr(v), _ptr(rc), _stream()), 'ntt_invert')
        return rc

    def div(self, num, den):
        import torch
        rc = torch.empty(num.shape[0], dtype=torch.int32, device=num.device)
        _chk(self.L, self.L.sc_hip_ntt_div(self.h, num.shape[0], num.shape[1], _ptr(num), _ptr(den), _ptr(rc), _stream()), 'ntt_div')
        return rc

    # host-buffer entry points (numpy in / numpy out), the shape the reference's vtable has
    def fwd_host(self, a):
        import numpy as np
        a = np.ascontiguousarray(a, dtype=np.int32)
        out = np.empty_like(a)
        _chk(self.L, self.L.sc_hip_ntt_fwd_host(self.h, a.size // self.n, _ptr(a), _ptr(out)), 'ntt_fwd_host')
        return out

    def inv_host(self, a):
        import numpy as np
        a = np.ascontiguousarray(a, dtype=np.int32)
        out = np.empty_like(a)
        _chk(self.L, self.L.sc_hip_ntt_inv_host(self.h, a.size // self.n, _ptr(a), _ptr(out)), 'ntt_inv_host')
        return out

    def polymul_ntt_host(self, a, bhat):
        import numpy as np
        a = np.ascontiguousarray(a, dtype=np.int32)
        bhat = np.ascontiguousarray(bhat, dtype=np.int32)
        out = np.empty_like(a)
        _chk(self.L, self.L.sc_hip_polymul_ntt_host(self.h, a.size // self.n, _ptr(a), _ptr(bhat), _ptr(out)), 'polymul_ntt_host')
        return out

    def time_op(self, op, a, b, out, iters):
        ms = C.c_float()
        bp = _ptr(b) if b is not None else C.c_void_p()
        _chk(self.L, self.L.sc_hip_time_op(self.h, op, self._batch(a), _ptr(a), bp, _ptr(out), _stream(), iters, C.byref(ms)), 'time_op')
        return ms.value


class CdtSampler:
    """create_sampler(CDF_GAUSSIAN_SAMPLING, SAMPLING_64BIT, ...) counterpart."""

    def __init__(self, eng, tail=None, sigma=None, table=None):
        self.eng, self.L = eng, eng.L
        h = C.c_void_p()
        if table is not None:
            import numpy as np
            t = np.ascontiguousarray(table, dtype=np.uint64)
            _chk(self.L, self.L.sc_hip_cdt_create_from_table(eng.h, _ptr(t), t.size, C.byref(h)), 'sc_hip_cdt_create_from_table')
        else:
            _chk(self.L, self.L.sc_hip_cdt_create(eng.h, tail, sigma, C.byref(h)), 'sc_hip_cdt_create')
        self.h = h
        self.size = self.L.sc_hip_cdt_size(h)

    def close(self):
        if self.h:
            self.L.sc_hip_cdt_destroy(self.h)
            self.h = None

    def table(self):
        import numpy as np
        t = np.empty(self.size, dtype=np.uint64)
        _chk(self.L, self.L.sc_hip_cdt_table(self.h, _ptr(t)), 'cdt_table')
        return t

    def sample(self, uniform, centre=0, out=None):
        """uniform: int64 device tensor holding the u64 words bit-for-bit"""
        import torch
        assert uniform.element_size() == 8 and uniform.is_contiguous()
        out = torch.empty(uniform.numel(), dtype=torch.int32, device=uniform.device) if out is None else out
        _chk(self.L, self.L.sc_hip_cdt_sample(self.h, uniform.numel(), _ptr(uniform), centre, _ptr(out), _stream()), 'cdt_sample')
        return out

    def sample_host(self, uniform, centre=0):
        import numpy as np
        u = np.ascontiguousarray(uniform, dtype=np.uint64)
        out = np.empty(u.size, dtype=np.int32)
        _chk(self.L, self.L.sc_hip_cdt_sample_host(self.h, u.size, _ptr(u), centre, _ptr(out)), 'cdt_sample_host')
        return out


class CdtSampler32:
    """create_sampler(CDF_GAUSSIAN_SAMPLING, SAMPLING_32BIT, ...) counterpart: 32-bit table entries, one 32-bit word per sample."""

    def __init__(self, eng, tail=None, sigma=None, table=None):
        self.eng, self.L = eng, eng.L
        h = C.c_void_p()
        if table is not None:
            import numpy as np
            t = np.ascontiguousarray(table, dtype=np.uint32)
            _chk(self.L, self.L.sc_hip_cdt32_create_from_table(eng.h, _ptr(t), t.size, C.byref(h)), 'sc_hip_cdt32_create_from_table')
        else:
            _chk(self.L, self.L.sc_hip_cdt32_create(eng.h, tail, sigma, C.byref(h)), 'sc_hip_cdt32_create')
        self.h = h
        self.size = self.L.sc_hip_cdt_size(h)

    def close(self):
        if self.h:
            self.L.sc_hip_cdt_destroy(self.h)
            self.h = None

    def table(self):
        import numpy as np
        t = np.empty(self.size, dtype=np.uint32)
        _chk(self.L, self.L.sc_hip_cdt32_table(self.h, _ptr(t)), 'cdt32_table')
        return t

    def sample(self, uniform, centre=0, out=None):
        """uniform: int32 device tensor holding the u32 words bit-for-bit"""
        import torch
        assert uniform.element_size() == 4 and uniform.is_contiguous()
        out = torch.empty(uniform.numel(), dtype=torch.int32, device=uniform.device) if out is None else out
        _chk(self.L, self.L.sc_hip_cdt32_sample(self.h, uniform.numel(), _ptr(uniform), centre, _ptr(out), _stream()), 'cdt32_sample')
        return out


class CdtSamplerHp:
    """create_sampler(CDF_GAUSSIAN_SAMPLING, SAMPLING_128BIT / SAMPLING_192BIT, ...) counterpart over the caller's table: uint64 [size][precision / 64] limbs, least
    significant first; precision / 64 64-bit words per sample (limb 0 first)."""

    def __init__(self, eng, precision_bits, table):
        import numpy as np
        self.eng, self.L, self.words = eng, eng.L, int(precision_bits) // 64
        t = np.ascontiguousarray(table, dtype=np.uint64)
        h = C.c_void_p()
        _chk(self.L, self.L.sc_hip_cdt_hp_create_from_table(eng.h, int(precision_bits), _ptr(t), t.size // max(self.words, 1), C.byref(h)), 'sc_hip_cdt_hp_create_from_table')
        self.h = h
        self.size = self.L.sc_hip_cdt_size(h)
        self.precision = self.L.sc_hip_cdt_hp_precision(h)

    def close(self):
        if self.h:
            self.L.sc_hip_cdt_destroy(self.h)
            self.h = None

    def table(self):
        import numpy as np
        t = np.empty((self.size, self.words), dtype=np.uint64)
        _chk(self.L, self.L.sc_hip_cdt_hp_table(self.h, _ptr(t)), 'cdt_hp_table')
        return t

    def sample(self, uniform, centre=0, out=None):
        """uniform: int64 device tensor [count][words] holding the u64 draws bit-for-bit"""
        import torch
        assert uniform.element_size() == 8 and uniform.is_contiguous() and uniform.numel() % self.words == 0
        n = uniform.numel() // self.words
        out = torch.empty(n, dtype=torch.int32, device=uniform.device) if out is None else out
        _chk(self.L, self.L.sc_hip_cdt_hp_sample(self.h, n, _ptr(uniform), centre, _ptr(out), _stream()), 'cdt_hp_sample')
        return out


class Dilithium:
    """Batched Dilithium sign / verify under one key (dilithium_sign / dilithium_verify counterpart).
    params: the 16 ints {n,q,q_bits,k,l,d,weight_of_c,gamma_1,gamma_1_bits,gamma_2,eta,z_bits,beta,omega,omega_bits,set}."""

    def __init__(self, eng, params):
        import numpy as np
        self.eng, self.L = eng, eng.L
        self.par = np.ascontiguousarray(np.asarray(params), dtype=np.int32)
        assert self.par.size == 16
        h = C.c_void_p()
        _chk(self.L, self.L.sc_hip_dilithium_create(eng.h, _ptr(self.par), C.byref(h)), 'sc_hip_dilithium_create')
        self.h = h
        self.sig_max = self.L.sc_hip_dilithium_sig_max(h)
        self.sig_stride = (self.sig_max + 15) & ~15

    def workspace_residue(self):
        """non-zero words of per-call secrets left in the signer's workspace (test hook)"""
        c = C.c_ulonglong(0)
        _chk(self.L, self.L.sc_hip_dilithium_workspace_residue(self.h, C.byref(c)), 'dilithium_workspace_residue')
        return int(c.value)

    def set_tuning(self, slots_per_message=0, unfused=False):
        """measurement hook (include/sc_hip_bench.h)"""
        _chk(self.L, self.L.sc_hip_dilithium_set_tuning(self.h, int(slots_per_message), 1 if unfused else 0), 'dilithium_set_tuning')

    def close(self):
        if self.h:
            self.L.sc_hip_dilithium_destroy(self.h)
            self.h = None

    @staticmethod
    def _a(x, dt):
        import numpy as np
        return np.ascontiguousarray(np.asarray(x), dtype=dt)

    def set_privkey(self, s1, s2, t, K, tr, Ahat):
        import numpy as np
        a = [self._a(s1, np.int32), self._a(s2, np.int32), self._a(t, np.int32), self._a(K, np.uint8), self._a(tr, np.uint8), self._a(Ahat, np.int32)]
        _chk(self.L, self.L.sc_hip_dilithium_set_privkey(self.h, *[_ptr(x) for x in a]), 'dilithium_set_privkey')

    def set_pubkey(self, t1, rho, Ahat):
        import numpy as np
        a = [self._a(t1, np.int32), self._a(rho, np.uint8), self._a(Ahat, np.int32)]
        _chk(self.L, self.L.sc_hip_dilithium_set_pubkey(self.h, *[_ptr(x) for x in a]), 'dilithium_set_pubkey')

    def keygen(self, seeds, eta_bits):
        """seeds uint8 [batch][len] (device) -> s1 [batch][l][n], s2, t, t1 [batch][k][n] int32; rho, K [batch][32]; tr [batch][48]"""
        import torch
        n, k, l = int(self.par[0]), int(self.par[3]), int(self.par[4])
        batch, dev = seeds.shape[0], seeds.device
        s1 = torch.empty((batch, l, n), dtype=torch.int32, device=dev)
        s2, t, t1 = (torch.empty((batch, k, n), dtype=torch.int32, device=dev) for _ in range(3))
        rho, K = (torch.empty((batch, 32), dtype=torch.uint8, device=dev) for _ in range(2))
        tr = torch.empty((batch, 48), dtype=torch.uint8, device=dev)
        _chk(self.L, self.L.sc_hip_dilithium_keygen(self.h, batch, int(eta_bits), _ptr(seeds), _rows(seeds), seeds.shape[1], _ptr(s1), _ptr(s2), _ptr(t),
                                                    _ptr(rho), _ptr(K), _ptr(tr), _ptr(t1), _stream()), 'dilithium_keygen')
        return s1, s2, t, rho, K, tr, t1

    # device-buffer entry points: torch uint8 [batch][stride] messages / signatures, int32 lengths
    def sign(self, msgs, mlens=None, fixed_len=0):
        import torch
        batch = msgs.shape[0]
        sigs = torch.zeros((batch, self.sig_stride), dtype=torch.uint8, device=msgs.device)
        siglens = torch.zeros(batch, dtype=torch.int32, device=msgs.device)
        trials = torch.zeros(batch, dtype=torch.int32, device=msgs.device)
        _chk(self.L, self.L.sc_hip_dilithium_sign(self.h, batch, _ptr(msgs), _rows(msgs), _ptr(mlens) if mlens is not None else C.c_void_p(),
                                                  fixed_len, _ptr(sigs), self.sig_stride, _ptr(siglens), _ptr(trials), _stream()), 'dilithium_sign')
        return sigs, siglens, trials

    def verify(self, msgs, sigs, siglens, mlens=None, fixed_len=0):
        import torch
        batch = msgs.shape[0]
        verdicts = torch.ones(batch, dtype=torch.int32, device=msgs.device)
        _chk(self.L, self.L.sc_hip_dilithium_verify(self.h, batch, _ptr(msgs), _rows(msgs), _ptr(mlens) if mlens is not None else C.c_void_p(),
                                                    fixed_len, _ptr(sigs), _rows(sigs), _ptr(siglens), _ptr(verdicts), _stream()), 'dilithium_verify')
        return verdicts

    def sign_keys(self, s1, s2, t, rho, K, tr, msgs, mlens=None, fixed_len=0):
        """every message under its own private key; key tensors (device, contiguous) in the layouts keygen() returns"""
        import torch
        batch = msgs.shape[0]
        for x in (s1, s2, t, rho, K, tr):
            assert x.shape[0] == batch and x.is_contiguous()
        sigs = torch.zeros((batch, self.sig_stride), dtype=torch.uint8, device=msgs.device)
        siglens = torch.zeros(batch, dtype=torch.int32, device=msgs.device)
        trials = torch.zeros(batch, dtype=torch.int32, device=msgs.device)
        _chk(self.L, self.L.sc_hip_dilithium_sign_keys(self.h, batch, _ptr(s1), _ptr(s2), _ptr(t), _ptr(rho), _ptr(K), _ptr(tr), _ptr(msgs), _rows(msgs),
                                                       _ptr(mlens) if mlens is not None else C.c_void_p(), fixed_len, _ptr(sigs), self.sig_stride, _ptr(siglens), _ptr(trials),
                                                       _stream()), 'dilithium_sign_keys')
        return sigs, siglens, trials

    def verify_keys(self, t1, rho, msgs, sigs, siglens, mlens=None, fixed_len=0):
        """every signature under its own public key: t1 int32 [batch][k*n], rho uint8 [batch][32] (device)"""
        import torch
        batch = msgs.shape[0]
        assert t1.shape[0] == batch and rho.shape[0] == batch and t1.is_contiguous() and rho.is_contiguous() and rho.shape[1] == 32
        verdicts = torch.ones(batch, dtype=torch.int32, device=msgs.device)
        _chk(self.L, self.L.sc_hip_dilithium_verify_keys(self.h, batch, _ptr(t1), _ptr(rho), _ptr(msgs), _rows(msgs), _ptr(mlens) if mlens is not None else C.c_void_p(),
                                                         fixed_len, _ptr(sigs), _rows(sigs), _ptr(siglens), _ptr(verdicts), _stream()), 'dilithium_verify_keys')
        return verdicts

    # host-buffer entry points: python bytes in / out (the shape safecrypto_sign / safecrypto_verify have)
    def _pack_msgs(self, msgs):
        import numpy as np
        stride = max(8, (max((len(m) for m in msgs), default=0) + 7) & ~7)
        buf = np.zeros((len(msgs), stride), dtype=np.uint8)
        for i, m in enumerate(msgs):
            buf[i, :len(m)] = np.frombuffer(bytes(m), dtype=np.uint8)
        lens = np.array([len(m) for m in msgs], dtype=np.uint32)
        return buf, stride, lens

    def sign_host(self, msgs):
        import numpy as np
        buf, stride, lens = self._pack_msgs(msgs)
        n = len(msgs)
        sigs = np.zeros((n, self.sig_stride), dtype=np.uint8)
        sl = np.zeros(n, dtype=np.uint32)
        tr = np.zeros(n, dtype=np.uint32)
        _chk(self.L, self.L.sc_hip_dilithium_sign_host(self.h, n, _ptr(buf), stride, _ptr(lens), 0, _ptr(sigs), self.sig_stride, _ptr(sl), _ptr(tr)), 'dilithium_sign_host')
        return [bytes(sigs[i, :sl[i]]) for i in range(n)], [int(x) for x in tr]

    def verify_host(self, msgs, sigs):
        import numpy as np
        buf, stride, lens = self._pack_msgs(msgs)
        n = len(msgs)
        sstride = max(self.sig_stride, (max(len(s) for s in sigs) + 15) & ~15)
        sb = np.zeros((n, sstride), dtype=np.uint8)
        for i, s_ in enumerate(sigs):
            sb[i, :len(s_)] = np.frombuffer(bytes(s_), dtype=np.uint8)
        sl = np.array([len(s_) for s_ in sigs], dtype=np.uint32)
        v = np.ones(n, dtype=np.uint32)
        _chk(self.L, self.L.sc_hip_dilithium_verify_host(self.h, n, _ptr(buf), stride, _ptr(lens), 0, _ptr(sb), sstride, _ptr(sl), _ptr(v)), 'dilithium_verify_host')
        return [int(x) for x in v]


class Bliss:
    """Batched BLISS-B sign / verify under one key (bliss_b_sign / bliss_b_verify counterpart).
    params: the 16 words of the fixture record {n,q,q_bits,n_bits,d,p,kappa,b_inf,b_l2,z1_bits,z2_bits,set, f32 sigma, f32 M, f32 tail, 0}."""

    def __init__(self, eng, params):
        import numpy as np
        self.eng, self.L = eng, eng.L
        self.par = np.ascontiguousarray(np.asarray(params), dtype=np.int32)
        assert self.par.size == 16
        h = C.c_void_p()
        _chk(self.L, self.L.sc_hip_bliss_create(eng.h, _ptr(self.par), C.byref(h)), 'sc_hip_bliss_create')
        self.h = h
        self.n = int(self.par[0])
        self.sig_bytes = self.L.sc_hip_bliss_sig_bytes(h)
        self.sig_stride = (self.sig_bytes + 15) & ~15

    def set_cdt_table(self, cdf):
        """the reference's own 64-bit CDF table (uint64) instead of the one tabulated on this host"""
        import numpy as np
        t = np.ascontiguousarray(np.asarray(cdf), dtype=np.uint64)
        _chk(self.L, self.L.sc_hip_bliss_set_cdt_table(self.h, _ptr(t), t.size), 'bliss_set_cdt_table')

    def set_cdt_table_hp(self, precision_bits, table):
        """the table of a context created with SC_FLAG_0_SAMPLE_128BIT / _192BIT (uint64 [size][precision / 64] limbs): a sample then takes precision / 64 prng_64 draws of the tape"""
        import numpy as np
        t = np.ascontiguousarray(np.asarray(table), dtype=np.uint64)
        _chk(self.L, self.L.sc_hip_bliss_set_cdt_table_hp(self.h, int(precision_bits), _ptr(t), t.size // (int(precision_bits) // 64)), 'bliss_set_cdt_table_hp')

    def set_tie_margin(self, relative):
        """relative margin around the rejection threshold inside which a draw is reported (status 5) instead of decided; default 2^-40"""
        _chk(self.L, self.L.sc_hip_bliss_set_tie_margin(self.h, float(relative)), 'bliss_set_tie_margin')

    def set_speculation(self, on):
        """measurement / test hook: 0 / False = the reference's rejection loop round by round; 1 / True (default) = a message's later pairs of attempts side by
        side; 2 = as 1 with every third such message sent through the in-order redo path"""
        _chk(self.L, self.L.sc_hip_bliss_set_speculation(self.h, int(on)), 'bliss_set_speculation')

    def set_entropy(self, tables):
        """the coding of the signatures (sc->coding_signature.type): None = raw fields; ((codes_t, bits_t), (codes_z, bits_z)) = SC_ENTROPY_HUFFMAN_STATIC with
        the two static tables the reference's coder walks for t and for z (src/utils/entropy/huffman.h), given as per-symbol code words and lengths"""
        import numpy as np
        if tables is None:
            _chk(self.L, self.L.sc_hip_bliss_set_entropy(self.h, 0, None, None, 0, None, None, 0), 'bliss_set_entropy')
            return
        (c1, l1), (c2, l2) = tables
        c1, c2 = (np.ascontiguousarray(np.asarray(x), dtype=np.uint32) for x in (c1, c2))
        l1, l2 = (np.ascontiguousarray(np.asarray(x), dtype=np.uint16) for x in (l1, l2))
        assert c1.size == l1.size and c2.size == l2.size
        _chk(self.L, self.L.sc_hip_bliss_set_entropy(self.h, 1, _ptr(c1), _ptr(l1), c1.size, _ptr(c2), _ptr(l2), c2.size), 'bliss_set_entropy')

    def set_entropy_bac(self, dist_t, dist_z):
        """SC_ENTROPY_BAC: the arithmetic coder's two distributions (uint64 [2^z1_bits], [2^z2_bits]) as the reference's context holds them (sc->dist[2], sc->dist[3])"""
        import numpy as np
        dt = np.ascontiguousarray(np.asarray(dist_t), dtype=np.uint64); dz = np.ascontiguousarray(np.asarray(dist_z), dtype=np.uint64)
        assert dt.size == 1 << int(self.par[9]) and dz.size == 1 << int(self.par[10])
        _chk(self.L, self.L.sc_hip_bliss_set_entropy_bac(self.h, _ptr(dt), _ptr(dz)), 'bliss_set_entropy_bac')

    def close(self):
        if self.h:
            self.L.sc_hip_bliss_destroy(self.h)
            self.h = None

    def set_key(self, f, g, a):
        import numpy as np
        fa = np.ascontiguousarray(np.asarray(f), dtype=np.int16) if f is not None else None
        ga = np.ascontiguousarray(np.asarray(g), dtype=np.int16) if g is not None else None
        aa = np.ascontiguousarray(np.asarray(a), dtype=np.int16)
        _chk(self.L, self.L.sc_hip_bliss_set_key(self.h, _ptr(fa) if fa is not None else C.c_void_p(), _ptr(ga) if ga is not None else C.c_void_p(), _ptr(aa)), 'bliss_set_key')

    def sign(self, msgs, tapes, mlens=None, fixed_len=0, tape_lens=None):
        """msgs uint8 [batch][stride], tapes int32/uint32 [batch][tape_stride] (device tensors)"""
        import torch
        batch = msgs.shape[0]
        dev = msgs.device
        sigs = torch.zeros((batch, self.sig_stride), dtype=torch.uint8, device=dev)
        siglens = torch.zeros(batch, dtype=torch.int32, device=dev)
        trials = torch.zeros(batch, dtype=torch.int32, device=dev)
        status = torch.zeros(batch, dtype=torch.int32, device=dev)
        nul = C.c_void_p()
        _chk(self.L, self.L.sc_hip_bliss_sign(self.h, batch, _ptr(msgs), _rows(msgs), _ptr(mlens) if mlens is not None else nul, fixed_len,
                                              _ptr(tapes), _rows(tapes), _ptr(tape_lens) if tape_lens is not None else nul,
                                              _ptr(sigs), self.sig_stride, _ptr(siglens), _ptr(trials), _ptr(status), _stream()), 'bliss_sign')
        return sigs, siglens, trials, status

    def sign_seeded(self, msgs, seeds, mlens=None, fixed_len=0):
        """sign with the reference's CSPRNG inside the call: seeds uint8 [batch][seed_len], message i draws from the AES-256 CTR-DRBG seeded with row i"""
        import torch
        batch = msgs.shape[0]
        dev = msgs.device
        sigs = torch.zeros((batch, self.sig_stride), dtype=torch.uint8, device=dev)
        siglens = torch.zeros(batch, dtype=torch.int32, device=dev)
        trials = torch.zeros(batch, dtype=torch.int32, device=dev)
        status = torch.zeros(batch, dtype=torch.int32, device=dev)
        nul = C.c_void_p()
        _chk(self.L, self.L.sc_hip_bliss_sign_seeded(self.h, batch, _ptr(msgs), _rows(msgs), _ptr(mlens) if mlens is not None else nul, fixed_len,
                                                     _ptr(seeds), _rows(seeds), seeds.shape[1],
                                                     _ptr(sigs), self.sig_stride, _ptr(siglens), _ptr(trials), _ptr(status), _stream()), 'bliss_sign_seeded')
        return sigs, siglens, trials, status

    def verify(self, msgs, sigs, siglens, mlens=None, fixed_len=0):
        import torch
        batch = msgs.shape[0]
        verdicts = torch.ones(batch, dtype=torch.int32, device=msgs.device)
        _chk(self.L, self.L.sc_hip_bliss_verify(self.h, batch, _ptr(msgs), _rows(msgs), _ptr(mlens) if mlens is not None else C.c_void_p(), fixed_len,
                                                _ptr(sigs), _rows(sigs), _ptr(siglens), _ptr(verdicts), _stream()), 'bliss_verify')
        return verdicts

    def sign_keys(self, f, g, a, msgs, tapes, mlens=None, fixed_len=0, tape_lens=None):
        """a key pair per message: f, g, a int16 [batch][n] device tensors (what keygen() returns)"""
        import torch
        batch = msgs.shape[0]
        dev = msgs.device
        for k in (f, g, a):
            assert k.dtype == torch.int16 and k.is_contiguous() and tuple(k.shape) == (batch, int(self.par[0]))
        sigs = torch.zeros((batch, self.sig_stride), dtype=torch.uint8, device=dev)
        siglens = torch.zeros(batch, dtype=torch.int32, device=dev)
        trials = torch.zeros(batch, dtype=torch.int32, device=dev)
        status = torch.zeros(batch, dtype=torch.int32, device=dev)
        nul = C.c_void_p()
        _chk(self.L, self.L.sc_hip_bliss_sign_keys(self.h, batch, _ptr(f), _ptr(g), _ptr(a), _ptr(msgs), _rows(msgs), _ptr(mlens) if mlens is not None else nul, fixed_len,
                                                   _ptr(tapes), _rows(tapes), _ptr(tape_lens) if tape_lens is not None else nul,
                                                   _ptr(sigs), self.sig_stride, _ptr(siglens), _ptr(trials), _ptr(status), _stream()), 'bliss_sign_keys')
        return sigs, siglens, trials, status

    def verify_keys(self, a, msgs, sigs, siglens, mlens=None, fixed_len=0):
        import torch
        batch = msgs.shape[0]
        assert a.dtype == torch.int16 and a.is_contiguous() and tuple(a.shape) == (batch, int(self.par[0]))
        verdicts = torch.ones(batch, dtype=torch.int32, device=msgs.device)
        _chk(self.L, self.L.sc_hip_bliss_verify_keys(self.h, batch, _ptr(a), _ptr(msgs), _rows(msgs), _ptr(mlens) if mlens is not None else C.c_void_p(), fixed_len,
                                                     _ptr(sigs), _rows(sigs), _ptr(siglens), _ptr(verdicts), _stream()), 'bliss_verify_keys')
        return verdicts

    def keygen(self, nz, tapes, tape_lens=None):
        """batched bliss_b_keygen: tapes [batch][words] (device) -> f, g, a int16 [batch][n], attempts, status"""
        import torch
        batch, dev = tapes.shape[0], tapes.device
        n = int(self.par[0])
        f, g, a = (torch.zeros((batch, n), dtype=torch.int16, device=dev) for _ in range(3))
        attempts = torch.zeros(batch, dtype=torch.int32, device=dev)
        status = torch.zeros(batch, dtype=torch.int32, device=dev)
        _chk(self.L, self.L.sc_hip_bliss_keygen(self.h, batch, int(nz[0]), int(nz[1]), _ptr(tapes), _rows(tapes), _ptr(tape_lens) if tape_lens is not None else C.c_void_p(),
                                                _ptr(f), _ptr(g), _ptr(a), _ptr(attempts), _ptr(status), _stream()), 'bliss_keygen')
        return f, g, a, attempts, status


class Tesla:
    """Batched ring-TESLA sign / verify under one key (ring_tesla_sign / ring_tesla_verify counterpart).
    params: the 12 words {n, n_bits, q, q_bits, b, b_bits, omega, d, bound, u, set, 0}."""

    def __init__(self, eng, params):
        import numpy as np
        self.eng, self.L = eng, eng.L
        self.par = np.ascontiguousarray(np.asarray(params), dtype=np.int32)
        assert self.par.size == 12
        h = C.c_void_p()
        _chk(self.L, self.L.sc_hip_tesla_create(eng.h, _ptr(self.par), C.byref(h)), 'sc_hip_tesla_create')
        self.h = h
        self.n = int(self.par[0])
        self.sig_bytes = self.L.sc_hip_tesla_sig_bytes(h)
        self.sig_stride = (self.sig_bytes + 15) & ~15

    def close(self):
        if self.h:
            self.L.sc_hip_tesla_destroy(self.h)
            self.h = None

    def set_key(self, a1, a2, s=None, e1=None, e2=None, t1=None, t2=None):
        import numpy as np
        f = lambda a, dt: np.ascontiguousarray(np.asarray(a), dtype=dt) if a is not None else None
        arrs = [f(a1, np.int32), f(a2, np.int32), f(s, np.int16), f(e1, np.int16), f(e2, np.int16), f(t1, np.int32), f(t2, np.int32)]
        _chk(self.L, self.L.sc_hip_tesla_set_key(self.h, *[_ptr(a) if a is not None else C.c_void_p() for a in arrs]), 'tesla_set_key')

    def sign(self, msgs, tapes, mlens=None, fixed_len=0, tape_lens=None):
        import torch
        batch, dev = msgs.shape[0], msgs.device
        sigs = torch.zeros((batch, self.sig_stride), dtype=torch.uint8, device=dev)
        siglens = torch.zeros(batch, dtype=torch.int32, device=dev)
        trials = torch.zeros(batch, dtype=torch.int32, device=dev)
        status = torch.zeros(batch, dtype=torch.int32, device=dev)
        nul = C.c_void_p()
        _chk(self.L, self.L.sc_hip_tesla_sign(self.h, batch, _ptr(msgs), _rows(msgs), _ptr(mlens) if mlens is not None else nul, fixed_len,
                                              _ptr(tapes), _rows(tapes), _ptr(tape_lens) if tape_lens is not None else nul,
                                              _ptr(sigs), self.sig_stride, _ptr(siglens), _ptr(trials), _ptr(status), _stream()), 'tesla_sign')
        return sigs, siglens, trials, status

    def sign_seeded(self, msgs, seeds, mlens=None, fixed_len=0):
        """sign with the reference's CSPRNG inside the call (seeds uint8 [batch][seed_len]): no tape exists, the masks come straight out of the cipher"""
        import torch
        batch, dev = msgs.shape[0], msgs.device
        sigs = torch.zeros((batch, self.sig_stride), dtype=torch.uint8, device=dev)
        siglens = torch.zeros(batch, dtype=torch.int32, device=dev)
        trials = torch.zeros(batch, dtype=torch.int32, device=dev)
        status = torch.zeros(batch, dtype=torch.int32, device=dev)
        nul = C.c_void_p()
        _chk(self.L, self.L.sc_hip_tesla_sign_seeded(self.h, batch, _ptr(msgs), _rows(msgs), _ptr(mlens) if mlens is not None else nul, fixed_len,
                                                     _ptr(seeds), _rows(seeds), seeds.shape[1],
                                                     _ptr(sigs), self.sig_stride, _ptr(siglens), _ptr(trials), _ptr(status), _stream()), 'tesla_sign_seeded')
        return sigs, siglens, trials, status

    def verify(self, msgs, sigs, siglens, mlens=None, fixed_len=0):
        import torch
        batch = msgs.shape[0]
        verdicts = torch.ones(batch, dtype=torch.int32, device=msgs.device)
        _chk(self.L, self.L.sc_hip_tesla_verify(self.h, batch, _ptr(msgs), _rows(msgs), _ptr(mlens) if mlens is not None else C.c_void_p(), fixed_len,
                                                _ptr(sigs), _rows(sigs), _ptr(siglens), _ptr(verdicts), _stream()), 'tesla_verify')
        return verdicts


class Dlp:
    """Batched DLP-IBE encrypt (identity per message) / decrypt (one user key).  params: {n, q, q_bits, n_bits, set, ubits, m_scale, l}."""

    def __init__(self, eng, params):
        import numpy as np
        self.eng, self.L = eng, eng.L
        self.par = np.ascontiguousarray(np.asarray(params), dtype=np.int32)
        assert self.par.size == 8
        h = C.c_void_p()
        _chk(self.L, self.L.sc_hip_dlp_create(eng.h, _ptr(self.par), C.byref(h)), 'sc_hip_dlp_create')
        self.h = h
        self.n = int(self.par[0])
        self.msg_bytes = self.L.sc_hip_dlp_msg_bytes(h)
        self.ct_bytes = self.L.sc_hip_dlp_ct_bytes(h)
        self.enc_words = self.L.sc_hip_dlp_enc_words(h)

    def close(self):
        if self.h:
            self.L.sc_hip_dlp_destroy(self.h)
            self.h = None

    def set_pubkey(self, h_ntt):
        import numpy as np
        a = np.ascontiguousarray(np.asarray(h_ntt), dtype=np.int32)
        assert a.size == self.n
        _chk(self.L, self.L.sc_hip_dlp_set_pubkey(self.h, _ptr(a)), 'dlp_set_pubkey')

    def set_user_key(self, uk_ntt):
        import numpy as np
        a = np.ascontiguousarray(np.asarray(uk_ntt), dtype=np.int32)
        assert a.size == self.n
        _chk(self.L, self.L.sc_hip_dlp_set_user_key(self.h, _ptr(a)), 'dlp_set_user_key')

    def encrypt(self, ids, msgs, tapes, idlens=None, fixed_len=0, tape_lens=None):
        import torch
        batch, dev = msgs.shape[0], msgs.device
        ct = torch.zeros((batch, self.ct_bytes), dtype=torch.uint8, device=dev)
        status = torch.zeros(batch, dtype=torch.int32, device=dev)
        nul = C.c_void_p()
        _chk(self.L, self.L.sc_hip_dlp_encrypt(self.h, batch, _ptr(ids), _rows(ids), _ptr(idlens) if idlens is not None else nul, fixed_len, _ptr(msgs), _rows(msgs),
                                               _ptr(tapes), _rows(tapes), _ptr(tape_lens) if tape_lens is not None else nul, _ptr(ct), self.ct_bytes, _ptr(status), _stream()),
             'dlp_encrypt')
        return ct, status

    def decrypt(self, ct):
        import torch
        batch = ct.shape[0]
        msgs = torch.zeros((batch, self.msg_bytes), dtype=torch.uint8, device=ct.device)
        _chk(self.L, self.L.sc_hip_dlp_decrypt(self.h, batch, _ptr(ct), _rows(ct), _ptr(msgs), self.msg_bytes, _stream()), 'dlp_decrypt')
        return msgs


class Ens:
    """Batched ENS-KEM encapsulation / decapsulation under one key pair (ens_kem_encapsulation / _decapsulation counterpart).
    params: the 20 words {set, q, q_bits, n, n_bits, 0, 0, coeff_rnd[13]}."""

    def __init__(self, eng, params):
        import numpy as np
        self.eng, self.L = eng, eng.L
        self.par = np.ascontiguousarray(np.asarray(params), dtype=np.int32)
        assert self.par.size == 20
        h = C.c_void_p()
        _chk(self.L, self.L.sc_hip_ens_create(eng.h, _ptr(self.par), C.byref(h)), 'sc_hip_ens_create')
        self.h = h
        self.n = int(self.par[3])
        self.ct_bytes = self.L.sc_hip_ens_ct_bytes(h)
        self.key_bytes = self.L.sc_hip_ens_key_bytes(h)

    def close(self):
        if self.h:
            self.L.sc_hip_ens_destroy(self.h)
            self.h = None

    def set_pubkey(self, h):
        import numpy as np
        a = np.ascontiguousarray(np.asarray(h), dtype=np.int16)
        assert a.size == self.n
        _chk(self.L, self.L.sc_hip_ens_set_pubkey(self.h, _ptr(a)), 'ens_set_pubkey')

    def set_privkey(self, g, inv_g):
        import numpy as np
        a = np.ascontiguousarray(np.asarray(g), dtype=np.int16)
        b = np.ascontiguousarray((np.asarray(inv_g).astype(np.int64) & 0xFFFFFFFF).astype(np.uint32))
        assert a.size == self.n and b.size == self.n // 32
        _chk(self.L, self.L.sc_hip_ens_set_privkey(self.h, _ptr(a), _ptr(b)), 'ens_set_privkey')

    def encaps(self, tapes, tape_lens=None):
        """tapes [batch][words] (device) -> ciphertexts, keys, words used, status"""
        import torch
        batch, dev = tapes.shape[0], tapes.device
        ct = torch.zeros((batch, self.ct_bytes), dtype=torch.uint8, device=dev)
        keys = torch.zeros((batch, self.key_bytes), dtype=torch.uint8, device=dev)
        used = torch.zeros(batch, dtype=torch.int32, device=dev)
        status = torch.zeros(batch, dtype=torch.int32, device=dev)
        _chk(self.L, self.L.sc_hip_ens_encaps(self.h, batch, _ptr(tapes), _rows(tapes), _ptr(tape_lens) if tape_lens is not None else C.c_void_p(),
                                              _ptr(ct), self.ct_bytes, _ptr(keys), self.key_bytes, _ptr(used), _ptr(status), _stream()), 'ens_encaps')
        return ct, keys, used, status

    def decaps(self, ct):
        import torch
        batch = ct.shape[0]
        keys = torch.zeros((batch, self.key_bytes), dtype=torch.uint8, device=ct.device)
        _chk(self.L, self.L.sc_hip_ens_decaps(self.h, batch, _ptr(ct), _rows(ct), _ptr(keys), self.key_bytes, _stream()), 'ens_decaps')
        return keys


class Rlwe:
    """Batched Ring-LWE encryption under one key pair and batched key generation (rlwe_enc_encrypt / _decrypt / _keygen
    counterpart).  params: the 12 words {set, q, q_bits, n, n_bits, m_scale, o_scale_0, o_scale_1, f32 sigma, f32 tail, 0, 0}."""

    def __init__(self, eng, params):
        import numpy as np
        self.eng, self.L = eng, eng.L
        self.par = np.ascontiguousarray(np.asarray(params), dtype=np.int32)
        assert self.par.size == 12
        h = C.c_void_p()
        _chk(self.L, self.L.sc_hip_rlwe_create(eng.h, _ptr(self.par), C.byref(h)), 'sc_hip_rlwe_create')
        self.h = h
        self.n = int(self.par[3])
        self.msg_bytes = self.L.sc_hip_rlwe_msg_bytes(h)
        self.ct_bytes = self.L.sc_hip_rlwe_ct_bytes(h)
        self.enc_words = self.L.sc_hip_rlwe_enc_words(h)
        self.keygen_words = self.L.sc_hip_rlwe_keygen_words(h)

    def close(self):
        if self.h:
            self.L.sc_hip_rlwe_destroy(self.h)
            self.h = None

    def set_pubkey(self, a, p):
        import numpy as np
        f = lambda x: np.ascontiguousarray(np.asarray(x), dtype=np.int16)
        a, p = f(a), f(p)
        assert a.size == self.n and p.size == self.n
        _chk(self.L, self.L.sc_hip_rlwe_set_pubkey(self.h, _ptr(a), _ptr(p)), 'rlwe_set_pubkey')

    def set_privkey(self, r2):
        import numpy as np
        r2 = np.ascontiguousarray(np.asarray(r2), dtype=np.int16)
        assert r2.size == self.n
        _chk(self.L, self.L.sc_hip_rlwe_set_privkey(self.h, _ptr(r2)), 'rlwe_set_privkey')

    def set_arith(self, mode):
        """test hook (sc_hip_bench.h): 0 automatic, 3 the unsigned multiply-add form of the encrypt kernel"""
        _chk(self.L, self.L.sc_hip_rlwe_set_arith(self.h, int(mode)), 'rlwe_set_arith')

    def encrypt(self, msgs, tapes):
        """msgs uint8 [batch][n/8], tapes int32/uint32 [batch][>= 6n] (device) -> ciphertexts uint8 [batch][ct_bytes]"""
        import torch
        batch = msgs.shape[0]
        ct = torch.empty((batch, self.ct_bytes), dtype=torch.uint8, device=msgs.device)
        _chk(self.L, self.L.sc_hip_rlwe_encrypt(self.h, batch, _ptr(msgs), _rows(msgs), _ptr(tapes), _rows(tapes), _ptr(ct), self.ct_bytes, _stream()), 'rlwe_encrypt')
        return ct

    def decrypt(self, ct):
        import torch
        batch = ct.shape[0]
        msgs = torch.empty((batch, self.msg_bytes), dtype=torch.uint8, device=ct.device)
        _chk(self.L, self.L.sc_hip_rlwe_decrypt(self.h, batch, _ptr(ct), _rows(ct), _ptr(msgs), self.msg_bytes, _stream()), 'rlwe_decrypt')
        return msgs

    def keygen(self, tapes):
        """tapes [batch][>= keygen_words] -> a, p, r2 int16 [batch][n] in the reference's storage format"""
        import torch
        batch, dev = tapes.shape[0], tapes.device
        a, p, r2 = (torch.empty((batch, self.n), dtype=torch.int16, device=dev) for _ in range(3))
        _chk(self.L, self.L.sc_hip_rlwe_keygen(self.h, batch, _ptr(tapes), _rows(tapes), _ptr(a), _ptr(p), _ptr(r2), _stream()), 'rlwe_keygen')
        return a, p, r2


class EnsDlpSigner:
    """Batched ENS / DLP signing under one private key (ens_dlp_sig_sign counterpart): Klein's sampler over the expanded basis and its
    single-precision Gram-Schmidt orthogonalisation.  params: the 8 words {n, q, q_bits, n_bits, set, f32 bd, 0, 0}."""

    def __init__(self, eng, params):
        import numpy as np
        self.eng, self.L = eng, eng.L
        self.par = np.ascontiguousarray(np.asarray(params), dtype=np.int32)
        assert self.par.size == 8
        h = C.c_void_p()
        _chk(self.L, self.L.sc_hip_ens_dlp_signer_create(eng.h, _ptr(self.par), C.byref(h)), 'sc_hip_ens_dlp_signer_create')
        self.h = h
        self.n = int(self.par[0])
        self.sig_bytes = self.L.sc_hip_ens_dlp_sig_bytes(h)
        self.tape_words = self.L.sc_hip_ens_dlp_sig_tape_words(h)

    def close(self):
        if self.h:
            self.L.sc_hip_ens_dlp_signer_destroy(self.h)
            self.h = None

    def set_key(self, b, b_gs, inv_norm):
        """host arrays as the reference context holds them: b int32 [2n, 2n], b_gs float32 [2n, 2n], inv_norm float32 [2n]"""
        import numpy as np
        b = np.ascontiguousarray(np.asarray(b), dtype=np.int32)
        g = np.ascontiguousarray(np.asarray(b_gs), dtype=np.float32)
        v = np.ascontiguousarray(np.asarray(inv_norm), dtype=np.float32)
        assert b.size == 4 * self.n * self.n and g.size == b.size and v.size == 2 * self.n
        _chk(self.L, self.L.sc_hip_ens_dlp_signer_set_key(self.h, _ptr(b), _ptr(g), _ptr(v)), 'ens_dlp_signer_set_key')

    def set_mapping(self, lanes):
        """0: chosen from the batch size; 1: a lane per signature (state in HBM); 4 / 8 / 16 lanes (state in registers)"""
        _chk(self.L, self.L.sc_hip_ens_dlp_signer_set_mapping(self.h, int(lanes)), 'ens_dlp_signer_set_mapping')

    def set_privkey(self, fgFG):
        """f | g | F | G (int32, 4n): the matrices are derived at key load as the reference's create_gpv_matrices does"""
        import numpy as np
        k = np.ascontiguousarray(np.asarray(fgFG), dtype=np.int32)
        assert k.size == 4 * self.n
        _chk(self.L, self.L.sc_hip_ens_dlp_signer_set_privkey(self.h, _ptr(k)), 'ens_dlp_signer_set_privkey')

    def sign(self, msgs, tapes, mlens=None, fixed_len=0, tape_lens=None):
        """msgs uint8 [batch][stride], tapes int32/uint32 [batch][>= 4n] -> signatures uint8 [batch][n * q_bits / 8], status"""
        import torch
        batch, dev = msgs.shape[0], msgs.device
        sigs = torch.zeros((batch, self.sig_bytes), dtype=torch.uint8, device=dev)
        status = torch.zeros(batch, dtype=torch.int32, device=dev)
        nul = C.c_void_p()
        _chk(self.L, self.L.sc_hip_ens_dlp_sig_sign(self.h, batch, _ptr(msgs), _rows(msgs), _ptr(mlens) if mlens is not None else nul, fixed_len,
                                                    _ptr(tapes), _rows(tapes), _ptr(tape_lens) if tape_lens is not None else nul,
                                                    _ptr(sigs), self.sig_bytes, _ptr(status), _stream()), 'ens_dlp_sig_sign')
        return sigs, status

    def sign_recovery(self, msgs, tapes, mlens=None, fixed_len=0, tape_lens=None):
        """with message recovery (the handle's params carry k_recovery in word 6): -> signatures uint8 [batch][2 n q_bits / 8] (s1 | s2), the remainders
        m2 uint8 [batch][stride] and their lengths, status"""
        import torch
        batch, dev = msgs.shape[0], msgs.device
        nb = self.L.sc_hip_ens_dlp_sig_rec_bytes(self.h)
        sigs = torch.zeros((batch, nb), dtype=torch.uint8, device=dev)
        m2 = torch.zeros((batch, max(8, msgs.shape[1])), dtype=torch.uint8, device=dev)
        m2lens = torch.zeros(batch, dtype=torch.int32, device=dev)
        status = torch.zeros(batch, dtype=torch.int32, device=dev)
        nul = C.c_void_p()
        _chk(self.L, self.L.sc_hip_ens_dlp_sig_sign_recovery(self.h, batch, _ptr(msgs), _rows(msgs), _ptr(mlens) if mlens is not None else nul, fixed_len,
                                                             _ptr(tapes), _rows(tapes), _ptr(tape_lens) if tape_lens is not None else nul,
                                                             _ptr(sigs), nb, _ptr(m2), _rows(m2), _ptr(m2lens), _ptr(status), _stream()), 'ens_dlp_sig_sign_recovery')
        return sigs, m2, m2lens, status


class Falcon:
    """Batched Falcon (sets 0 and 2) tree sampler / sign / verify under one key (gaussian_sample_with_tree, falcon_sig_sign,
    falcon_sig_verify counterpart).  params: the 8 words {n, q, q_bits, n_bits, set, f32 bd, 0, 0}."""

    def __init__(self, eng, params):
        import numpy as np
        self.eng, self.L = eng, eng.L
        self.par = np.ascontiguousarray(np.asarray(params), dtype=np.int32)
        assert self.par.size == 8
        h = C.c_void_p()
        _chk(self.L, self.L.sc_hip_falcon_create(eng.h, _ptr(self.par), C.byref(h)), 'sc_hip_falcon_create')
        self.h = h
        self.n = int(self.par[0])
        self.sig_bytes = self.L.sc_hip_falcon_sig_bytes(h)
        self.sig_stride = (self.sig_bytes + 15) & ~15

    def close(self):
        if self.h:
            self.L.sc_hip_falcon_destroy(self.h)
            self.h = None

    def workspace_residue(self):
        """non-zero words of per-call secrets left in the signer's workspace (test hook)"""
        c = C.c_ulonglong(0)
        _chk(self.L, self.L.sc_hip_falcon_workspace_residue(self.h, C.byref(c)), 'falcon_workspace_residue')
        return int(c.value)

    def set_privkey(self, sk):
        """sk: float64 [(n_bits+5)*n], the expanded key (b00|b01|b10|b11|tree)"""
        import numpy as np
        a = np.ascontiguousarray(np.asarray(sk), dtype=np.float64)
        assert a.size == (int(self.par[3]) + 5) * self.n
        _chk(self.L, self.L.sc_hip_falcon_set_privkey(self.h, _ptr(a)), 'falcon_set_privkey')

    def set_pubkey(self, h):
        import numpy as np
        a = np.ascontiguousarray(np.asarray(h), dtype=np.int16)
        assert a.size == self.n
        _chk(self.L, self.L.sc_hip_falcon_set_pubkey(self.h, _ptr(a)), 'falcon_set_pubkey')

    def set_privkeys(self, sks):
        """a key set: float64 [nkeys][(n_bits+5)*n]"""
        import numpy as np
        a = np.ascontiguousarray(np.asarray(sks), dtype=np.float64)
        per = (int(self.par[3]) + 5) * self.n
        assert a.ndim == 2 and a.shape[1] == per
        _chk(self.L, self.L.sc_hip_falcon_set_privkeys(self.h, a.shape[0], _ptr(a)), 'falcon_set_privkeys')

    def set_pubkeys(self, hs):
        import numpy as np
        a = np.ascontiguousarray(np.asarray(hs), dtype=np.int16)
        assert a.ndim == 2 and a.shape[1] == self.n
        _chk(self.L, self.L.sc_hip_falcon_set_pubkeys(self.h, a.shape[0], _ptr(a)), 'falcon_set_pubkeys')

    def sign_keys(self, key_of, msgs, tapes, mlens=None, fixed_len=0, tape_lens=None):
        """key_of int32 [batch] (device): the index, in the loaded key set, of the key each message is signed under"""
        import torch
        batch, dev = msgs.shape[0], msgs.device
        assert key_of.dtype == torch.int32 and key_of.numel() == batch and key_of.is_contiguous()
        sigs = torch.zeros((batch, self.sig_stride), dtype=torch.uint8, device=dev)
        siglens = torch.zeros(batch, dtype=torch.int32, device=dev)
        status = torch.zeros(batch, dtype=torch.int32, device=dev)
        nul = C.c_void_p()
        _chk(self.L, self.L.sc_hip_falcon_sign_keys(self.h, batch, _ptr(key_of), _ptr(msgs), _rows(msgs), _ptr(mlens) if mlens is not None else nul, fixed_len,
                                                    _ptr(tapes), _rows(tapes), _ptr(tape_lens) if tape_lens is not None else nul,
                                                    _ptr(sigs), self.sig_stride, _ptr(siglens), _ptr(status), _stream()), 'falcon_sign_keys')
        return sigs, siglens, status

    def verify_keys(self, key_of, msgs, sigs, siglens, mlens=None, fixed_len=0):
        import torch
        batch = msgs.shape[0]
        assert key_of.dtype == torch.int32 and key_of.numel() == batch and key_of.is_contiguous()
        verdicts = torch.ones(batch, dtype=torch.int32, device=msgs.device)
        _chk(self.L, self.L.sc_hip_falcon_verify_keys(self.h, batch, _ptr(key_of), _ptr(msgs), _rows(msgs), _ptr(mlens) if mlens is not None else C.c_void_p(), fixed_len,
                                                      _ptr(sigs), _rows(sigs), _ptr(siglens), _ptr(verdicts), _stream()), 'falcon_verify_keys')
        return verdicts

    def set_variant(self, spw, lm):
        """measurement hook: pin the default mapping's variant (signatures per wavefront 8 / 4, LDS sub-tree level 7 / 6); 0, 0 = per batch"""
        _chk(self.L, self.L.sc_hip_falcon_set_variant(self.h, int(spw), int(lm)), 'falcon_set_variant')

    def sample_profile(self, c, tapes):
        """measurement hook: the default mapping's sampler, returns (s1, s2, cycles[6]) - shader cycles per phase summed over wavefronts"""
        import numpy as np, torch
        batch, dev = c.shape[0], c.device
        s1 = torch.empty((batch, self.n), dtype=torch.int32, device=dev)
        s2 = torch.empty((batch, self.n), dtype=torch.int32, device=dev)
        cyc = np.zeros(6, dtype=np.uint64)
        _chk(self.L, self.L.sc_hip_falcon_sample_profile(self.h, batch, _ptr(c), _ptr(tapes), _rows(tapes), _ptr(s1), _ptr(s2), _ptr(cyc), _stream()), 'falcon_sample_profile')
        return s1, s2, cyc

    def set_mapping(self, lanes):
        """0 / 64: 8 signatures per wavefront, lower tree in LDS (default); 1, 8: lanes per signature, state in HBM"""
        _chk(self.L, self.L.sc_hip_falcon_set_mapping(self.h, int(lanes)), 'falcon_set_mapping')

    def ens_dlp_verify(self, msgs, sigs, siglens, mlens=None, fixed_len=0):
        """ENS / DLP signature verification (the handle holds that scheme's parameters and public key)"""
        import torch
        batch = msgs.shape[0]
        verdicts = torch.ones(batch, dtype=torch.int32, device=msgs.device)
        _chk(self.L, self.L.sc_hip_ens_dlp_sig_verify(self.h, batch, _ptr(msgs), _rows(msgs), _ptr(mlens) if mlens is not None else C.c_void_p(), fixed_len,
                                                      _ptr(sigs), _rows(sigs), _ptr(siglens), _ptr(verdicts), _stream()), 'ens_dlp_sig_verify')
        return verdicts

    def ens_dlp_verify_recovery(self, sigs, siglens, m2=None, m2lens=None, out_bytes=0):
        """ENS / DLP signatures with message recovery (params word 6 = k_recovery, h loaded): -> verdicts (0 valid), the rebuilt messages uint8
        [batch][out_bytes or what the longest can need] and their lengths (0 for a refused signature)"""
        import torch
        batch, dev = sigs.shape[0], sigs.device
        k, qb = int(self.par[6]), int(self.par[2])
        m1 = ((self.n - k) * (qb - 1) + 7) // 8
        stride = int(out_bytes) if out_bytes else m1 + (m2.shape[1] if m2 is not None else 0)
        out = torch.zeros((batch, stride), dtype=torch.uint8, device=dev)
        outlens = torch.zeros(batch, dtype=torch.int32, device=dev)
        verdicts = torch.ones(batch, dtype=torch.int32, device=dev)
        nul = C.c_void_p()
        _chk(self.L, self.L.sc_hip_ens_dlp_sig_verify_recovery(self.h, batch, _ptr(sigs), _rows(sigs), _ptr(siglens),
                                                               _ptr(m2) if m2 is not None else nul, _rows(m2) if m2 is not None else 0, _ptr(m2lens) if m2lens is not None else nul,
                                                               _ptr(out), stride, _ptr(outlens), _ptr(verdicts), _stream()), 'ens_dlp_sig_verify_recovery')
        return verdicts, out, outlens

    def dlp_extract(self, ubits, ids, tapes, idlens=None, fixed_len=0, tape_lens=None):
        """DLP-IBE extract (the handle holds the DLP-IBE parameters and the master tree): ids uint8 [batch][stride], tapes [batch][>= 4n]
        -> user secret keys uint8 [batch][n * ubits / 8], status"""
        import torch
        batch, dev = ids.shape[0], ids.device
        nbytes = self.n * ubits // 8
        sk = torch.zeros((batch, nbytes), dtype=torch.uint8, device=dev)
        status = torch.zeros(batch, dtype=torch.int32, device=dev)
        nul = C.c_void_p()
        _chk(self.L, self.L.sc_hip_dlp_ibe_extract(self.h, batch, int(ubits), _ptr(ids), _rows(ids), _ptr(idlens) if idlens is not None else nul, fixed_len,
                                                   _ptr(tapes), _rows(tapes), _ptr(tape_lens) if tape_lens is not None else nul, _ptr(sk), nbytes, _ptr(status), _stream()),
             'dlp_ibe_extract')
        return sk, status

    def sample(self, c, tapes, tape_lens=None, want_s1=True):
        """c int32 [batch][n], tapes int32/uint32 [batch][>= 4n] (device) -> s1, s2 int32 [batch][n], status"""
        import torch
        batch, dev = c.shape[0], c.device
        s1 = torch.empty((batch, self.n), dtype=torch.int32, device=dev) if want_s1 else None
        s2 = torch.empty((batch, self.n), dtype=torch.int32, device=dev)
        status = torch.zeros(batch, dtype=torch.int32, device=dev)
        nul = C.c_void_p()
        _chk(self.L, self.L.sc_hip_falcon_sample(self.h, batch, _ptr(c), _ptr(tapes), _rows(tapes), _ptr(tape_lens) if tape_lens is not None else nul,
                                                 _ptr(s1) if want_s1 else nul, _ptr(s2), _ptr(status), _stream()), 'falcon_sample')
        return s1, s2, status

    def sign(self, msgs, tapes, mlens=None, fixed_len=0, tape_lens=None):
        import torch
        batch, dev = msgs.shape[0], msgs.device
        sigs = torch.zeros((batch, self.sig_stride), dtype=torch.uint8, device=dev)
        siglens = torch.zeros(batch, dtype=torch.int32, device=dev)
        status = torch.zeros(batch, dtype=torch.int32, device=dev)
        nul = C.c_void_p()
        _chk(self.L, self.L.sc_hip_falcon_sign(self.h, batch, _ptr(msgs), _rows(msgs), _ptr(mlens) if mlens is not None else nul, fixed_len,
                                               _ptr(tapes), _rows(tapes), _ptr(tape_lens) if tape_lens is not None else nul,
                                               _ptr(sigs), self.sig_stride, _ptr(siglens), _ptr(status), _stream()), 'falcon_sign')
        return sigs, siglens, status

    def sign_seeded(self, msgs, seeds, mlens=None, fixed_len=0):
        """sign with the reference's CSPRNG inside the call (seeds uint8 [batch][seed_len]): the 4n words of a signature are expanded on the device"""
        import torch
        batch, dev = msgs.shape[0], msgs.device
        sigs = torch.zeros((batch, self.sig_stride), dtype=torch.uint8, device=dev)
        siglens = torch.zeros(batch, dtype=torch.int32, device=dev)
        status = torch.zeros(batch, dtype=torch.int32, device=dev)
        nul = C.c_void_p()
        _chk(self.L, self.L.sc_hip_falcon_sign_seeded(self.h, batch, _ptr(msgs), _rows(msgs), _ptr(mlens) if mlens is not None else nul, fixed_len,
                                                      _ptr(seeds), _rows(seeds), seeds.shape[1],
                                                      _ptr(sigs), self.sig_stride, _ptr(siglens), _ptr(status), _stream()), 'falcon_sign_seeded')
        return sigs, siglens, status

    def verify(self, msgs, sigs, siglens, mlens=None, fixed_len=0):
        import torch
        batch = msgs.shape[0]
        verdicts = torch.ones(batch, dtype=torch.int32, device=msgs.device)
        _chk(self.L, self.L.sc_hip_falcon_verify(self.h, batch, _ptr(msgs), _rows(msgs), _ptr(mlens) if mlens is not None else C.c_void_p(), fixed_len,
                                                 _ptr(sigs), _rows(sigs), _ptr(siglens), _ptr(verdicts), _stream()), 'falcon_verify')
        return verdicts


class Kyber:
    """Batched Kyber (round 1) CPA encryption / decryption under one key (kyber_cpa_enc / kyber_cpa_dec counterpart).
    params: {n,q,q_bits,q_inv,q_norm,k,eta,d_u,d_v,d_t,set,0,0,0,0,0}."""

    def __init__(self, eng, params):
        import numpy as np
        self.eng, self.L = eng, eng.L
        self.par = np.ascontiguousarray(np.asarray(params), dtype=np.int32)
        assert self.par.size == 16
        h = C.c_void_p()
        _chk(self.L, self.L.sc_hip_kyber_create(eng.h, _ptr(self.par), C.byref(h)), 'sc_hip_kyber_create')
        self.h = h
        self.n, self.k = int(self.par[0]), int(self.par[5])
        self.noise_bytes = self.L.sc_hip_kyber_noise_bytes(h)

    def close(self):
        if self.h:
            self.L.sc_hip_kyber_destroy(self.h)
            self.h = None

    def set_pubkey(self, t_ntt, Agen):
        import numpy as np
        a = [np.ascontiguousarray(np.asarray(x), dtype=np.int32) for x in (t_ntt, Agen)]
        _chk(self.L, self.L.sc_hip_kyber_set_pubkey(self.h, _ptr(a[0]), _ptr(a[1])), 'kyber_set_pubkey')

    def set_privkey(self, s_ntt):
        import numpy as np
        a = np.ascontiguousarray(np.asarray(s_ntt), dtype=np.int32)
        _chk(self.L, self.L.sc_hip_kyber_set_privkey(self.h, _ptr(a)), 'kyber_set_privkey')

    def set_kem_pubkey(self, t, rho):
        import numpy as np
        a = np.ascontiguousarray(np.asarray(t), dtype=np.int32)
        r = np.ascontiguousarray(np.asarray(rho), dtype=np.uint8)
        assert a.size == self.k * self.n and r.size == 32
        _chk(self.L, self.L.sc_hip_kyber_set_kem_pubkey(self.h, _ptr(a), _ptr(r)), 'kyber_set_kem_pubkey')
        self.ct_bytes = self.L.sc_hip_kyber_ct_bytes(self.h)

    def keygen(self, seeds):
        """seeds uint8 [batch][len] (device) -> s_ntt, t, t_ntt int32 [batch][k][n]; rho, z uint8 [batch][32]"""
        import torch
        batch, dev = seeds.shape[0], seeds.device
        s_ntt, t, t_ntt = (torch.empty((batch, self.k, self.n), dtype=torch.int32, device=dev) for _ in range(3))
        rho, z = (torch.empty((batch, 32), dtype=torch.uint8, device=dev) for _ in range(2))
        _chk(self.L, self.L.sc_hip_kyber_keygen(self.h, batch, _ptr(seeds), _rows(seeds), seeds.shape[1], _ptr(s_ntt), _ptr(t), _ptr(t_ntt), _ptr(rho), _ptr(z), _stream()), 'kyber_keygen')
        return s_ntt, t, t_ntt, rho, z

    def encaps(self, m):
        """m uint8 [batch][32] (device) -> ciphertexts uint8 [batch][ct_bytes], keys uint8 [batch][32]"""
        import torch
        batch = m.shape[0]
        ct = torch.empty((batch, self.ct_bytes), dtype=torch.uint8, device=m.device)
        keys = torch.empty((batch, 32), dtype=torch.uint8, device=m.device)
        _chk(self.L, self.L.sc_hip_kyber_kem_encaps(self.h, batch, _ptr(m), _ptr(ct), self.ct_bytes, _ptr(keys), _stream()), 'kyber_kem_encaps')
        return ct, keys

    def decaps(self, ct):
        import torch
        batch = ct.shape[0]
        keys = torch.empty((batch, 32), dtype=torch.uint8, device=ct.device)
        _chk(self.L, self.L.sc_hip_kyber_kem_decaps(self.h, batch, _ptr(ct), _rows(ct), _ptr(keys), _stream()), 'kyber_kem_decaps')
        return keys

    def _key_rows(self, batch, *polys):
        import torch
        for x in polys:
            assert x.dtype == torch.int32 and x.is_contiguous() and tuple(x.shape) == (batch, self.k, self.n)

    def encaps_keys(self, t, t_ntt, rho, m):
        """a public key per value: t, t_ntt int32 [batch][k][n], rho uint8 [batch][32] (what keygen() returns), m uint8 [batch][32]"""
        import torch
        batch = m.shape[0]
        self._key_rows(batch, t, t_ntt)
        assert rho.dtype == torch.uint8 and rho.is_contiguous() and tuple(rho.shape) == (batch, 32)
        ctb = self.L.sc_hip_kyber_ct_bytes(self.h)
        ct = torch.empty((batch, ctb), dtype=torch.uint8, device=m.device)
        keys = torch.empty((batch, 32), dtype=torch.uint8, device=m.device)
        _chk(self.L, self.L.sc_hip_kyber_kem_encaps_keys(self.h, batch, _ptr(t), _ptr(t_ntt), _ptr(rho), _ptr(m), _ptr(ct), ctb, _ptr(keys), _stream()), 'kyber_kem_encaps_keys')
        return ct, keys

    def decaps_keys(self, s_ntt, t, t_ntt, rho, ct):
        import torch
        batch = ct.shape[0]
        self._key_rows(batch, s_ntt, t, t_ntt)
        assert rho.dtype == torch.uint8 and rho.is_contiguous() and tuple(rho.shape) == (batch, 32)
        keys = torch.empty((batch, 32), dtype=torch.uint8, device=ct.device)
        _chk(self.L, self.L.sc_hip_kyber_kem_decaps_keys(self.h, batch, _ptr(s_ntt), _ptr(t), _ptr(t_ntt), _ptr(rho), _ptr(ct), _rows(ct), _ptr(keys), _stream()), 'kyber_kem_decaps_keys')
        return keys

    def enc(self, m, noise):
        """m uint8 [batch][32], noise uint8 [batch][stride >= noise_bytes] (device) -> u int32 [batch][k][n], v int32 [batch][n]"""
        import torch
        batch = m.shape[0]
        u = torch.empty((batch, self.k, self.n), dtype=torch.int32, device=m.device)
        v = torch.empty((batch, self.n), dtype=torch.int32, device=m.device)
        _chk(self.L, self.L.sc_hip_kyber_cpa_enc(self.h, batch, _ptr(m), _ptr(noise), _rows(noise), _ptr(u), _ptr(v), _stream()), 'kyber_cpa_enc')
        return u, v

    def dec(self, u, v):
        import torch
        batch = v.shape[0]
        m = torch.empty((batch, 32), dtype=torch.uint8, device=v.device)
        _chk(self.L, self.L.sc_hip_kyber_cpa_dec(self.h, batch, _ptr(u), _ptr(v), _ptr(m), _stream()), 'kyber_cpa_dec')
        return m
