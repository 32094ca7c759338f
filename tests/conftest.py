import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    from reflibs import Oracle
    return Oracle()


@pytest.fixture(scope="session")
def reference_c():
    from reflibs import Reference, have_reference
    if not have_reference():
        pytest.skip("oracle/_ref/libhavoc_ref.so not built (needs /root/reference; `make -C oracle ref`)")
    return Reference(0)


@pytest.fixture(scope="session")
def reference_jit():
    from reflibs import Reference, have_reference
    if not have_reference():
        pytest.skip("oracle/_ref/libhavoc_ref.so not built (needs /root/reference; `make -C oracle ref`)")
    return Reference(1)
