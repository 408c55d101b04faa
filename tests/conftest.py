import os
import sys

import numpy as np
import pytest

# the MV-CNN twin's convolutions go through MIOpen: no exhaustive kernel search in tests
os.environ.setdefault("MIOPEN_FIND_MODE", "2")

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

GOLDEN = os.path.join(REPO, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


def load_cases(fname):
    """npz with 'case/field' keys -> {case: {field: array}}"""
    z = np.load(os.path.join(GOLDEN, fname))
    out = {}
    for k in z.files:
        case, field = k.split("/", 1)
        out.setdefault(case, {})[field] = z[k]
    return out


@pytest.fixture(scope="session")
def oracle_mod():
    from oracle import oracle
    oracle.build()
    return oracle
