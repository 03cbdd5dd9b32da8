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


@pytest.fixture(autouse=True)
def _poisoned_hit_buffers(monkeypatch):
    """Every full-frame trace issued through RaytracingMeshDrawer.update() in a test starts from a hit buffer filled
    with NaN patterns: a tile the library fails to trace can then never pass on values left there by an earlier
    frame (the repeated-frame paths — cost-ordered dispatch, XCD regions, cooperative tiles — reuse the buffer)."""
    try:
        from unitysimpleraytracing_amd import host
    except Exception:       # no library (CPU-only run): nothing to patch
        yield
        return
    plain = host.RaytracingMeshDrawer.update

    def update(self, camera, rect=None, mode=host.L.TRACE_FAST, stats=False):
        if self._hits is not None:
            self._hits.fill_u32(0x7FC00000, mirror=False)
        out = plain(self, camera, rect=rect, mode=mode, stats=stats)
        return out

    monkeypatch.setattr(host.RaytracingMeshDrawer, "update", update)
    yield
