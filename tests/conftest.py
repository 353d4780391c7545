import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session", autouse=True)
def _built():
    """Both shared libraries must exist (built by __graft_entry__.build()); build them if missing."""
    from oracle import api as oracle_api
    from zkir_amd import build as zbuild
    oracle_api.build()
    if not os.path.exists(zbuild.OUT):
        zbuild.build()
    yield
