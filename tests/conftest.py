import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import __graft_entry__ as graft  # noqa: E402


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "slow: CPU test of a few minutes (skipped with GSDF_SKIP_SLOW=1)")


@pytest.fixture(scope="session")
def pkg():
    return graft.package()


@pytest.fixture(scope="session")
def O():
    """CPU oracle wrapper (the checker)."""
    return graft.oracle_module()


@pytest.fixture(scope="session")
def gpu_lib(pkg):
    """libgsdf.so loaded; skips nothing: on the GPU box a missing .so must fail loudly."""
    return pkg.binding.load()


def pose7_from(O, R, t):
    return np.concatenate([np.asarray(t, np.float32), O.R_to_quat(R)]).astype(np.float32)
