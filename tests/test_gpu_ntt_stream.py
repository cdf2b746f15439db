"""GPU parity for the two I/O forms of the batched NTT kernels (ntt_kernel<..., STREAM>, csrc/sc_hip.hip): the streaming form (16 bytes per lane,
non-temporal, transposed through LDS; taken by working set in production, forced here through sc_hip_bench_set_ntt_stream) must give exactly what the
plain form gives - and that is pinned against the oracle / the reference's fixtures in test_gpu_ntt.py - for every operation, ring size, modulus class and
for batches that end in a ragged group."""
import numpy as np
import pytest

from tests import oracle_lib as O

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def eng():
    from libsafecrypto_amd import Engine
    e = Engine(0)
    yield e
    e.L.sc_hip_bench_set_ntt_stream(1)
    e.close()


CASES = [(7681, 256), (8380417, 256), (12289, 512), (8399873, 512), (12289, 1024), (134348801, 1024)]


@pytest.mark.parametrize('q,n', CASES)
@pytest.mark.parametrize('batch', [1, 3, 4, 5, 64, 1027])
def test_stream_form_equals_plain_form(eng, q, n, batch):
    import torch
    rng = np.random.default_rng(q % 1000 + n + batch)
    a = torch.from_numpy(rng.integers(0, q, size=(batch, n)).astype(np.int32)).cuda()
    b = torch.from_numpy(rng.integers(0, q, size=(batch, n)).astype(np.int32)).cuda()
    small = torch.from_numpy(rng.integers(-12, 13, size=(batch, n)).astype(np.int32)).cuda()      # the reference's small_rand_dist_32 inputs (unit_ntt.c:501-524)
    plan = eng.plan(q, n)
    got = {}
    for mode in (0, 2):
        eng.L.sc_hip_bench_set_ntt_stream(mode)
        f = plan.fwd(a)
        got[mode] = (f, plan.inv(f), plan.polymul_ntt(a, b), plan.polymul(a, small), plan.fwd(small))
        torch.cuda.synchronize()
    eng.L.sc_hip_bench_set_ntt_stream(1)
    for x, y in zip(got[0], got[2]):
        assert torch.equal(x, y)
    assert torch.equal(got[2][1], a)                                   # inv(fwd(a)) == a, canonical
    if batch <= 5:                                                     # and the oracle on the small cases (the plain form is pinned at every size elsewhere)
        w, r = O.tables(q, n)
        want = O.polymul_ntt(a.cpu().numpy(), b.cpu().numpy(), q, w, r)
        assert np.array_equal(np.mod(got[2][2].cpu().numpy(), q), np.mod(want, q))
    plan.close()


def test_stream_form_in_place(eng):
    """out may alias in0 (groups are disjoint in memory): both forms"""
    import torch
    q, n, batch = 7681, 256, 515
    rng = np.random.default_rng(5)
    a = torch.from_numpy(rng.integers(0, q, size=(batch, n)).astype(np.int32)).cuda()
    plan = eng.plan(q, n)
    want = plan.fwd(a)
    for mode in (0, 2):
        eng.L.sc_hip_bench_set_ntt_stream(mode)
        x = a.clone()
        plan.fwd(x, x)
        assert torch.equal(x, want)
    eng.L.sc_hip_bench_set_ntt_stream(1)
    plan.close()
