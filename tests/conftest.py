import os
import sys

import numpy as np
import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)
GOLDEN = os.path.join(REPO, 'tests', 'golden')


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


def load_golden(name):
    return np.load(os.path.join(GOLDEN, name + '.npz'))


@pytest.fixture(scope='session')
def golden():
    return load_golden


def flips(label, got, ref, observed=0):
    """A float-stage product (cc, protus) against the oracle's: prints how many pixels differ and by how much, and holds the count
    to `observed` -- what this comparison gives on MI355X today (0 unless the test says otherwise), not a tolerance: a change that
    makes a single further pixel flip fails.  A flip is a last-bit difference of a host float moving one truncation: 1 LSB at most."""
    got, ref = np.asarray(got), np.asarray(ref)
    assert got.shape == ref.shape, (label, got.shape, ref.shape)
    d = np.abs(got.astype(np.int64) - ref.astype(np.int64))
    n, worst = int(np.count_nonzero(d)), int(d.max()) if d.size else 0
    print('PARITY %-60s %9d px  %d differ  max %d LSB' % (label, d.size, n, worst))
    assert n <= observed and worst <= (1 if observed else 0), (label, n, worst)
