"""GPU: a plain C host program (examples/c_host.c: gcc, no HIP headers, no Python in the process) links libsc_hip.so, drives the
reference-shaped vtable and the batched host entries, and checks the probe vectors SURVEY.md lists for the reference."""
import os
import shutil
import subprocess

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_c_host_reproduces_reference_probe_vectors(tmp_path):
    gcc = shutil.which('gcc')
    assert gcc, 'gcc not found'
    lib = os.path.join(ROOT, 'libsafecrypto_amd', 'lib')
    exe = str(tmp_path / 'c_host')
    subprocess.check_call([gcc, '-std=c99', '-O2', '-I' + os.path.join(ROOT, 'include'), os.path.join(ROOT, 'examples', 'c_host.c'),
                           '-L' + lib, '-lsc_hip', '-Wl,-rpath,' + lib, '-Wl,-rpath,/opt/rocm/lib', '-o', exe])
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert 'all reference probe vectors reproduced' in r.stdout
