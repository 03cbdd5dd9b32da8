import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def ctx():
    """One liblbvh context on cuda:0 for the whole GPU session.  Fails (does not skip) if the
    library or the device is missing: GPU tests must never pass on a fallback."""
    from unitysimpleraytracing_amd.host import Context
    c = Context(0)
    yield c
    c.close()
