import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


@pytest.fixture(scope="session")
def golden():
    """Loader for the committed fixtures captured from the reference (tools/make_golden.py)."""
    cache = {}

    def load(name):
        if name not in cache:
            with np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False) as z:
                cache[name] = {k: z[k] for k in z.files}
        return cache[name]

    return load


@pytest.fixture(scope="session")
def scenes():
    from oracle import nerf_oracle as O

    cache = {}

    def get(name):
        if name not in cache:
            cache[name] = O.make_scene(name)
        return cache[name]

    return get
