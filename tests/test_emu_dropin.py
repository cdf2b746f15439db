"""CPU: sections of the drop-in proof (oracle/_ref/dropin_driver: examples/safecrypto_glue/dropin_driver.c + sc_hip_glue.c, linked with the REFERENCE library) run against
the product library built for the CPU (tests/emu_harness.py build_library; the dynamic linker is given that build under the name libsc_hip.so).  The reference's own
safecrypto_sign / safecrypto_verify - and its bliss_b_sign over the engine's sampler table - beside the glue's engine entries: identical bytes.

Here: BLISS-B-I with the 128-bit sampler, the configuration of the reference's own harness (/root/reference/test/functional/func_alg_bliss_b.c:118), inner and outer boundary
(a context created with SC_FLAG_0_SAMPLE_128BIT holds a table of noise in the MPFR-less build of the reference: the driver installs one of known content, install_hp_table).
The whole driver takes the device (tests/test_gpu_dropin.py): a Dilithium batch is hours of emulation."""
import json
import os
import subprocess

import pytest

from tests.emu_harness import ROOT, build_library

DRV = os.path.join(ROOT, 'oracle', '_ref', 'dropin_driver')


@pytest.mark.skipif(not os.access(DRV, os.X_OK), reason='oracle/_ref/dropin_driver not built (needs the reference tree: make -C oracle ref)')
def test_bliss_with_the_harness_128_bit_sampler_through_the_glue(tmp_path):
    so = build_library(None)
    os.symlink(so, tmp_path / 'libsc_hip.so')
    env = dict(os.environ)
    env['LD_LIBRARY_PATH'] = os.pathsep.join([str(tmp_path), os.path.join(ROOT, 'oracle', '_ref'), env.get('LD_LIBRARY_PATH', '')])
    env['SC_DROPIN_BATCH'] = '16'
    p = subprocess.run([DRV, '128-bit sampler'], capture_output=True, text=True, timeout=1500, env=env)
    lines = [json.loads(x) for x in p.stdout.splitlines() if x.startswith('{')]
    assert lines and p.returncode == 0, p.stdout[-2000:] + p.stderr[-2000:]
    assert not [x for x in lines if x.get('ok') is False], lines
    summary = lines[-1]
    assert summary.get('summary') and summary['failed'] == 0 and summary['vtable_errors'] == 0 and summary['checks'] == 10, summary
    names = ' | '.join(x['check'] for x in lines if 'check' in x)
    for must in ("signatures over the engine's tables == over the CPU tables", 'sc_hip_glue sign == safecrypto_sign from the same CSPRNG state', 'a batch of 16 through sc_hip_glue_sign_batch'):
        assert must in names, must


def test_plain_c_host_on_the_cpu_build(tmp_path):
    """examples/c_host.c (C99, no HIP headers, no Python in the process): the reference-shaped vtable and the batched host entries, the probe vectors SURVEY.md lists for the
    reference - linked against the CPU build of the library under the product's name (tests/test_gpu_c_host.py is the same program on the device)"""
    import shutil
    gcc = shutil.which('gcc')
    assert gcc, 'gcc not found'
    so = build_library(None)
    os.symlink(so, tmp_path / 'libsc_hip.so')
    exe = str(tmp_path / 'c_host')
    subprocess.check_call([gcc, '-std=c99', '-O2', '-I' + os.path.join(ROOT, 'include'), os.path.join(ROOT, 'examples', 'c_host.c'),
                           '-L' + str(tmp_path), '-lsc_hip', '-Wl,-rpath,' + str(tmp_path), '-o', exe])
    r = subprocess.run([exe], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and 'all reference probe vectors reproduced' in r.stdout, r.stdout + r.stderr
