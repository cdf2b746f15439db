"""CPU: the `-m gpu` parity tests run against the product library BUILT FOR THE CPU - every source of libsafecrypto_amd/csrc/Makefile, host code and kernels,
compiled as host C++ over tests/emu/hip_host_emu.h (a host-side HIP shim: allocations are host memory, a launch runs to completion with a host thread per
lane - tests/emu/wave_emu.h) with AddressSanitizer + UBSan, linked into a library with the product's C ABI (tests/emu_harness.py build_library), loaded
through the product's own Python binding with torch's "cuda" pointed at the CPU (tests/conftest.py, SC_EMU_LIB).  The tests are the GPU suite's, unchanged:
what they call is sc_hip_* - key loading, workspace sizing, the round loops, the stream / event logic degenerated to one in-order stream - down to the
kernels' own text.  The GPU pool cannot run a sanitizer; this can, and it exercises the host side the single-scheme emulations (tests/test_emu_*.py) restate.

Here: a subset that finishes in about a minute.  tools/emu_gpu_suite.sh runs the whole `-m gpu` suite this way (about an hour; the tests sized for the real
device - full BASELINE batches, the bench line, a C host linked against the device library, graph capture - are deselected there and listed)."""
import os
import re
import subprocess

from tests.emu_harness import ROOT, build_library, sanitizer_env

QUICK = [
    'tests/test_gpu_errors.py',
    'tests/test_gpu_cdt.py',
    'tests/test_gpu_ops.py',
    'tests/test_gpu_drbg.py',
    'tests/test_gpu_vtable.py',
    'tests/test_gpu_zx_mlwe_xof.py',
    'tests/test_gpu_zx_cdt_hp.py',
    'tests/test_gpu_zx_bliss_hp.py::test_reference_signatures',
    'tests/test_gpu_ntt.py::test_golden_fwd_inv_mul',
    'tests/test_gpu_ntt.py::test_tables_match_reference',
    'tests/test_gpu_ntt.py::test_extreme_inputs_every_mode',
    'tests/test_gpu_ntt.py::test_inplace_and_empty',
    'tests/test_gpu_ntt.py::test_elementwise_ops',
    'tests/test_gpu_ntt.py::test_host_buffer_entry_points',
    'tests/test_gpu_ntt.py::test_plan_rejects_bad_parameters',
    'tests/test_gpu_ntt.py::test_large_modulus_variants_vs_reference',
    'tests/test_gpu_kyber.py::test_enc_dec_vs_reference',
    'tests/test_gpu_dilithium.py::test_verify_matches_reference_verdicts',
]


def test_gpu_suite_subset_on_the_cpu_build_asan_ubsan():
    so = build_library('address,undefined')
    env = sanitizer_env('address,undefined')
    env['SC_EMU_LIB'] = so
    r = subprocess.run(['python3', '-m', 'pytest'] + QUICK + ['-m', 'gpu', '-q', '-n', '4', '-p', 'no:cacheprovider', '--timeout', '900', '--timeout-method=thread',
                        '--deselect', 'tests/test_gpu_ops.py::test_batches_vs_oracle'],          # (70 s of emulation; tools/emu_gpu_suite.sh runs it)
                       cwd=ROOT, env=env, capture_output=True, text=True, timeout=2400)
    tail = r.stdout[-3000:]
    m = re.search(r'(\d+) passed', tail)
    assert r.returncode == 0 and m and int(m.group(1)) >= 103 and 'failed' not in tail.splitlines()[-1], (tail, r.stderr[-3000:])
