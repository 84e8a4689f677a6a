import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


@pytest.fixture(scope="session")
def orc():
    from oracle import oracle
    oracle.load()
    return oracle


@pytest.fixture(scope="session")
def k():
    import kissabc_jl_amd
    return kissabc_jl_amd


@pytest.fixture(scope="session")
def gpu_ctx(k):
    """Fails loudly (no skip, no fallback) when the HIP library or device is missing."""
    return k.default_context(0)
