import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run by the driver with -m gpu)")


@pytest.fixture(scope="session")
def golden():
    def _load(name):
        return np.load(os.path.join(GOLDEN, name), allow_pickle=False)
    return _load


def rel_err(a, ref, floor_frac=1e-3):
    """SURVEY 8d metric: max |a-ref| / max(|ref|, floor_frac*max|ref|)."""
    a = np.asarray(a, dtype=np.float64)
    ref = np.asarray(ref, dtype=np.float64)
    mx = np.max(np.abs(ref)) if ref.size else 0.0
    if mx == 0.0:  # all-zero reference: any non-zero is an infinite relative error
        return 0.0 if not np.any(a) else float("inf")
    den = np.maximum(np.abs(ref), floor_frac * mx)
    return float(np.max(np.abs(a - ref) / den))
