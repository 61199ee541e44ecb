import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


@pytest.fixture(scope="session")
def rt():
    """The product package (loads librt1w.so; builds it if the .so is missing)."""
    lib = os.path.join(ROOT, "raytracing-1w_amd", "librt1w.so")
    if not os.path.exists(lib):
        import __graft_entry__
        __graft_entry__.build()
    import orc
    return orc.rt()


@pytest.fixture(scope="session")
def gpu_ctx_factory(rt):
    """Context factory that FAILS (not skips) when the HIP path is unavailable."""
    made = []

    def make(scene, device=0):
        assert rt.device_count() >= 1, "no HIP device visible: GPU tests must run on the GPU box"
        c = rt.Context(scene, device)
        made.append(c)
        return c

    yield make
    for c in made:
        c.close()
