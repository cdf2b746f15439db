"""CPU: the oracle against fixtures the reference has JUST written from other seeds.

tests/golden/ is one draw: one key pair per scheme and set, one handful of messages, tapes and table cases.  oracle/_ref/ref_driver takes a salt (SC_REF_SALT) that
perturbs every seeded stream it has - the generator of messages and inputs, the entropy the contexts draw their keys from, the constant per-case CSPRNG seeds - and writes a
set of the same shape with everything in it different; the whole oracle suite (tests/test_oracle*.py, unchanged) then runs against that directory (SC_GOLDEN_DIR).
An oracle that had been fitted to the committed vectors rather than restating the reference's code would not survive this.

Salt 1 by default (about 25 s; SC_TEST_SALTS="3,20261004,..." for others - tried in round 6: 1, 2, 3, 12345, 20261004 all 156 tests green).  What a salt can run
into, seen while trying them, and what it is NOT: salt 777 - the driver's own search for a BLISS key pair of a particular shape gives up (no fixture, nothing to compare);
salt 3 - the ENS set-1 key drawn makes the REFERENCE reject its own three signatures, and the oracle returns the reference's verdict for each (the committed-fixture tests
additionally assert "valid", a property of the committed keys)."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DRIVER = os.path.join(ROOT, 'oracle', '_ref', 'ref_driver')
SALTS = [int(x) for x in os.environ.get('SC_TEST_SALTS', '1').split(',') if x.strip()]


@pytest.mark.skipif(not os.access(DRIVER, os.X_OK), reason='oracle/_ref/ref_driver not built (needs the reference tree: make -C oracle ref)')
@pytest.mark.skipif(bool(os.environ.get('SC_GOLDEN_DIR')), reason='already running against another fixture directory')
@pytest.mark.parametrize('salt', SALTS)
def test_oracle_suite_on_an_independently_seeded_fixture_set(tmp_path, salt):
    env = dict(os.environ, SC_REF_SALT=str(salt))
    subprocess.run([DRIVER, 'all', str(tmp_path)], check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=300, env=env)
    committed = os.path.join(ROOT, 'tests', 'golden')
    made = sorted(os.listdir(tmp_path))
    assert made == sorted(f for f in os.listdir(committed) if f.endswith('_ref.txt'))
    differing = [f for f in made if open(os.path.join(tmp_path, f), 'rb').read() != open(os.path.join(committed, f), 'rb').read()]
    assert len(differing) == len(made), sorted(set(made) - set(differing))            # every file is another draw
    suites = sorted(f for f in os.listdir(os.path.join(ROOT, 'tests')) if f.startswith('test_oracle') and f.endswith('.py') and f != os.path.basename(__file__))
    r = subprocess.run([sys.executable, '-m', 'pytest', '-q', '-p', 'no:cacheprovider'] + [os.path.join('tests', f) for f in suites], cwd=ROOT,
                       env=dict(os.environ, SC_GOLDEN_DIR=str(tmp_path)), capture_output=True, text=True, timeout=900)
    tail = r.stdout[-1500:]
    assert r.returncode == 0 and ' passed' in tail and 'failed' not in tail.splitlines()[-1], tail
