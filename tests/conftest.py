import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


@pytest.fixture(scope="session", autouse=True)
def _built_library():
    """Make sure the HIP library and the C oracle exist (hipcc cross-compiles without a GPU)."""
    import laplace_amd.build as b
    lib = os.path.join(ROOT, "laplace-gnn-recommendation_amd", "liblaplace_hip.so")
    if not os.path.exists(lib):
        b.build_hip()
    if not os.path.exists(os.path.join(ROOT, "oracle", "liboracle_ref.so")):
        b.build_oracle()
    yield
