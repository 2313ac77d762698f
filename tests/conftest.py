import os
import sys

import pytest

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, os.path.abspath(ROOT))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def sda():
    """The product package.  Importing it dlopens sparse_dot_amd/libmi_sparse.so (build it first)."""
    import sparse_dot_amd
    return sparse_dot_amd


@pytest.fixture(scope="session")
def oracle():
    from oracle import cpu_oracle
    cpu_oracle.build()
    return cpu_oracle


@pytest.fixture(scope="session")
def gpu(sda):
    if sda.mi_get_device_count() < 1:
        pytest.fail("this test is marked gpu but no HIP device is visible: " + sda.mi_get_version_string())
    return sda
