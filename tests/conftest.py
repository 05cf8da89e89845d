import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def hip_ctx():
    """One nyxhip context on cuda:0.  The HIP library must load -- no fallback."""
    from nyxus_amd import _lib
    ctx = _lib.Context(0)
    yield ctx
    ctx.close()
