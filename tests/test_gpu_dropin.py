"""GPU: the drop-in proof.  oracle/_ref/dropin_driver (built in the dev container by oracle/build_ref.sh from
examples/safecrypto_glue/{dropin_driver,sc_hip_glue}.c against the reference's headers, linked with the reference library AND
libsc_hip.so) runs the reference's OWN scheme code - dilithium_sign, bliss_b_sign, ring_tesla_sign, kyber / ENS encapsulation,
key generation - first over its CPU tables and then with sc->sc_ntt / sc_poly / sc_vec / sc_gauss pointed at the engine's
layout-identical tables (include/sc_hip_vtable.h), and requires identical bytes; then the scheme-table-shaped entries of the glue
(whole operation on the device) against safecrypto_sign / safecrypto_verify."""
import json
import os
import subprocess

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DRV = os.path.join(ROOT, 'oracle', '_ref', 'dropin_driver')


def test_reference_scheme_code_runs_over_the_engine_tables():
    if not os.path.exists(DRV):
        pytest.skip('oracle/_ref/dropin_driver was not built (it needs the reference tree: run __graft_entry__.build() in the dev container)')
    env = dict(os.environ)
    env['LD_LIBRARY_PATH'] = os.pathsep.join([os.path.join(ROOT, 'oracle', '_ref'), os.path.join(ROOT, 'libsafecrypto_amd', 'lib'), '/opt/rocm/lib',
                                              env.get('LD_LIBRARY_PATH', '')])
    p = subprocess.run([DRV], capture_output=True, text=True, timeout=900, env=env)
    lines = [json.loads(x) for x in p.stdout.splitlines() if x.startswith('{')]
    assert lines, p.stdout + p.stderr
    bad = [x for x in lines if x.get('ok') is False]
    assert not bad, bad
    summary = lines[-1]
    assert summary.get('summary') and summary['failed'] == 0 and summary['vtable_errors'] == 0, summary
    assert p.returncode == 0
    names = ' | '.join(x['check'] for x in lines if 'check' in x)
    for must in ('Dilithium set 3 (Dilithium-III): signatures over the engine', 'BLISS-B-I: signatures over the engine', 'ring-TESLA set 0: signatures',
                 'Kyber KEM set 1: ciphertexts', 'Dilithium set 3: key pair generated', 'outer boundary, Dilithium set 3: a batch'):
        assert must in names, must
    assert summary['checks'] >= 35
