"""CPU: tests/golden/*.txt are what the reference itself writes, byte for byte, every time.

oracle/_ref/ref_driver is the reference's own code (built by oracle/build_ref.sh from the sources where they lie) behind a driver that seeds every input:
the arithmetic-level cases from fixed seeds, the scheme contexts through safecrypto_entropy_callback + SC_FLAG_1_CSPRNG_USE_CALLBACK_RANDOM
(/root/reference/include/safecrypto.h:108,281) with a fixed stream per generator and context - so `ref_driver all <dir>` (= `make -C oracle golden`)
regenerates all twenty files identically, keys included, and a reader can byte-compare every one of them."""
import filecmp
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DRIVER = os.path.join(ROOT, 'oracle', '_ref', 'ref_driver')
GOLDEN = os.path.join(ROOT, 'tests', 'golden')


@pytest.mark.skipif(not os.access(DRIVER, os.X_OK), reason='oracle/_ref/ref_driver not built (needs the reference tree: make -C oracle ref)')
def test_every_fixture_regenerates_byte_for_byte(tmp_path):
    subprocess.run([DRIVER, 'all', str(tmp_path)], check=True, stdout=subprocess.DEVNULL, timeout=300)
    made = sorted(os.listdir(tmp_path))
    committed = sorted(f for f in os.listdir(GOLDEN) if f.endswith('_ref.txt'))
    assert made == committed
    differing = [f for f in made if not filecmp.cmp(os.path.join(tmp_path, f), os.path.join(GOLDEN, f), shallow=False)]
    assert differing == []
