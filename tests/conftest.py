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


def assert_depth_flips_are_near_ties(depth, depth_ref, S_new_ref, tie, what=""):
    """Arg-max depth maps are discontinuous: a depth may differ from the reference's by more
    than 1e-4 ONLY where the reference's two best probabilities are within `tie` of each
    other (a near-tie that a last-bit difference decides).  Returns the number of such
    pixels.  depth, depth_ref: [n]; S_new_ref: [n, M] final distributions of the reference."""
    flips = np.where(np.abs(np.asarray(depth).ravel() - np.asarray(depth_ref).ravel()) > 1e-4)[0]
    for r in flips:
        top = np.sort(S_new_ref[r])[::-1]
        assert top[0] - top[1] <= tie, "%s ray %d: depth differs away from a near-tie (%g vs %g)" % (
            what, r, top[0], top[1])
    return len(flips)
