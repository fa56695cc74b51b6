import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden():
    path = os.path.join(ROOT, "tests", "golden", "golden_v1.npz")
    return dict(np.load(path))


@pytest.fixture(scope="session")
def oracle():
    """The CPU oracle (test infrastructure).  Never imported by the product."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import oracle as _oracle
    _oracle.build()
    return _oracle
