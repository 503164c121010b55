import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden():
    cache = {}

    def load(name):
        if name not in cache:
            cache[name] = dict(np.load(os.path.join(GOLDEN, name + ".npz")))
        return cache[name]

    return load


def rel_err(a, b):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    scale = max(np.max(np.abs(b)) if b.size else 0.0, 1e-300)
    return float(np.max(np.abs(a - b)) / scale) if b.size else 0.0


def elem_err(a, b, floor_frac=1e-5):
    """Element-wise relative error max_i |a_i - b_i| / max(|b_i|, floor) with the absolute floor
    floor_frac * max|b|: weak entries (small-p sweep values, weak periods' powers) are held to
    the same relative bar as the strongest one, down to entries 1e-5 of it; below the floor the
    bar degrades to the max-norm one (BLAS nrm2 / the float32 reference carry absolute noise)."""
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    if not b.size:
        return 0.0
    floor = max(floor_frac * float(np.max(np.abs(b))), 1e-300)
    return float(np.max(np.abs(a - b) / np.maximum(np.abs(b), floor)))
