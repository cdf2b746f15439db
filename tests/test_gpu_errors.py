"""GPU: error behaviour of the C ABI - bad arguments come back as SC_HIP_FAILURE with a message in sc_hip_last_error (never a
crash, never a silent fallback), and a failing call leaves the handle usable."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def eng():
    from libsafecrypto_amd import Engine
    e = Engine(0)
    yield e
    e.close()


def err(L):
    return L.sc_hip_last_error().decode()


def test_plan_rejects_bad_moduli_and_sizes(eng):
    L = eng.L
    h = C.c_void_p()
    for q, n in ((7681, 128), (7681, 257), (7680, 256), (12289, 2048), (2, 256), (7681 * 3, 256), (40961, 1024 * 4)):
        assert L.sc_hip_plan_create(eng.h, q, n, C.byref(h)) != 0, (q, n)
        assert 'plan_create' in err(L)
    assert L.sc_hip_plan_create(eng.h, 7681, 256, None) != 0
    p = eng.plan(7681, 256)                                   # the context still works
    assert L.sc_hip_plan_set_arith(p.h, 7) != 0 and 'mode' in err(L)
    assert L.sc_hip_plan_set_arith(p.h, 3) == 0
    p.close()
    big = eng.plan(8399873, 512)                              # ring-TESLA's modulus, above 2^23
    assert L.sc_hip_plan_set_arith(big.h, 3) != 0 and 'q < 2^14' in err(L)
    assert L.sc_hip_plan_set_arith(big.h, 1) != 0 and 'q < 2^23' in err(L)
    big.close()


def test_null_buffers_and_missing_keys(eng):
    import torch
    from tests.golden_io import load_golden
    L = eng.L
    p = eng.plan(12289, 512)
    a = torch.zeros((4, 512), dtype=torch.int32, device='cuda')
    nul = C.c_void_p()
    assert L.sc_hip_ntt_fwd(p.h, 4, nul, C.c_void_p(a.data_ptr()), nul) != 0 and 'null' in err(L)
    assert L.sc_hip_ntt_fwd(nul, 4, C.c_void_p(a.data_ptr()), C.c_void_p(a.data_ptr()), nul) != 0
    assert L.sc_hip_ntt_fwd(p.h, 0, nul, nul, nul) == 0       # an empty batch is a no-op
    assert L.sc_hip_ntt_invert(p.h, 1, 0, C.c_void_p(a.data_ptr()), C.c_void_p(a.data_ptr()), nul) != 0
    p.close()
    G = load_golden('falcon_ref.txt')
    f = eng.falcon(G['falcon0_params'])
    msgs = torch.zeros((2, 64), dtype=torch.uint8, device='cuda')
    tapes = torch.zeros((2, 2048), dtype=torch.int32, device='cuda')
    from libsafecrypto_amd import EngineError
    with pytest.raises(EngineError, match='no private key'):
        f.sign(msgs, tapes, fixed_len=64)
    with pytest.raises(EngineError, match='no public key'):
        f.verify(msgs, torch.zeros((2, 784), dtype=torch.uint8, device='cuda'), torch.zeros(2, dtype=torch.int32, device='cuda'), fixed_len=64)
    f.set_privkey(np.asarray(G['falcon0_sk'], dtype=np.int64).view(np.float64))
    bad_tapes = torch.zeros((2, 2050), dtype=torch.int32, device='cuda')      # stride not a multiple of 4 words
    with pytest.raises(EngineError, match='16-byte aligned'):
        f.sign(msgs, bad_tapes, fixed_len=64)
    sigs, siglens, status = f.sign(msgs, tapes, fixed_len=64)                 # the handle survived the failures
    assert status.cpu().tolist() in ([1, 1], [4, 4], [1, 4], [4, 1])
    f.close()
    bad = np.array(G['falcon0_params']).copy(); bad[4] = 1                     # the ternary ring is not served
    with pytest.raises(EngineError, match='ternary'):
        eng.falcon(bad)
    K = load_golden('kyber_ref.txt')
    ky = eng.kyber(K['kyber1_params'])
    ky.ct_bytes = 1184
    with pytest.raises(EngineError, match='public key'):
        ky.encaps(torch.zeros((2, 32), dtype=torch.uint8, device='cuda'))
    ky.close()


def test_vtable_counts_errors_instead_of_dropping_them(eng):
    from libsafecrypto_amd import load_library
    from tests.test_gpu_vtable import VT, Params, P
    L = load_library()
    L.sc_hip_utils_arith_ntt32.restype = C.POINTER(VT)
    L.sc_hip_vtable_errors.restype = C.c_ulong
    vt = L.sc_hip_utils_arith_ntt32().contents
    e0 = L.sc_hip_vtable_errors()
    p = Params(); p.n = 256; p.u.ntt32.q = 7683                                # not a prime with 2n | q-1: no plan can exist
    v = np.zeros(256, dtype=np.int32)
    vt.fwd_ntt_32_32(P(v), C.byref(p), P(v), P(v))
    assert L.sc_hip_vtable_errors() == e0 + 1 and 'plan_create' in err(L)
    assert vt.invert_32(P(v), C.byref(p), 256) == 1                            # fallible slots report SC_FUNC_FAILURE
