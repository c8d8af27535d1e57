import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs an MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    from _oracle import Oracle
    return Oracle()


@pytest.fixture(scope="session")
def ref():
    """The reference's own code, where oracle/_ref/libptref.so was built (else skip)."""
    from _oracle import Ref
    if not Ref.available():
        pytest.skip("oracle/_ref/libptref.so not built here")
    return Ref()


def bits(a):
    return np.ascontiguousarray(a).view(np.uint32)
