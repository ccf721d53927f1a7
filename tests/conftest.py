import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, 'tests', 'golden')


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu)')


def load_golden(name):
    """read one committed golden fixture (tests/golden/<name>.npz) into a dict of ndarrays"""
    with np.load(os.path.join(GOLDEN, name + '.npz')) as f:
        return {k: f[k] for k in f.files}


@pytest.fixture
def golden():
    return load_golden
