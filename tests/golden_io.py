"""Loader for the text fixtures under tests/golden/ (format: see oracle/ref_driver.c)."""
import os
import numpy as np

# SC_GOLDEN_DIR: another directory of fixtures of the same shape (tests/test_oracle_fresh_fixtures.py: a set the reference has just written from other seeds)
GOLDEN_DIR = os.environ.get('SC_GOLDEN_DIR') or os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')


def load_golden(name):
    out = {}
    cur = None
    vals = []
    with open(name if os.path.isabs(name) else os.path.join(GOLDEN_DIR, name)) as f:
        for line in f:
            if line.startswith('@'):
                if cur is not None:
                    out[cur] = vals
                _, cur, _cnt = line.split()
                vals = []
            else:
                vals.extend(line.split())
    if cur is not None:
        out[cur] = vals
    res = {}
    for k, v in out.items():
        ints = [int(x) for x in v]
        if any(x > 2**63 - 1 for x in ints):
            res[k] = np.array(ints, dtype=np.uint64)
        elif k.endswith('_cdf') or k.endswith('_uniform'):
            res[k] = np.array(ints, dtype=np.uint64)
        else:
            res[k] = np.array(ints, dtype=np.int64)
    return res
