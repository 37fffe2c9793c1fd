import os, sys
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session", autouse=True)
def _native_built():
    """Build the native pieces once if a fresh checkout has none (hipcc cross-compiles gfx950 without a GPU)."""
    need = [os.path.join(ROOT, "spartan_amd", "lib", "libspartan_hip.so"), os.path.join(ROOT, "spartan_amd", "lib", "libspartan_host.so"),
            os.path.join(ROOT, "oracle", "liboracle.so"), os.path.join(ROOT, "tests", "csrc", "libhostcheck.so")]
    if not all(os.path.exists(p) for p in need):
        import __graft_entry__
        __graft_entry__.build()


@pytest.fixture(scope="session")
def orc():
    from tests import helpers
    return helpers.load_oracle()


@pytest.fixture(scope="session")
def hc():
    from tests import helpers
    return helpers.load_hostcheck()
