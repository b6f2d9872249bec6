import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


# wass_stereo hands its frame to a per-GPU resident worker when it can (wass_amd/host/stereo_server.hpp).  The tests of the executable
# itself run it in-process; tests/test_server.py switches the server on, with a socket directory and an idle time-out of its own.
os.environ.setdefault("WASS_NO_SERVER", "1")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    """The CPU oracle (test infrastructure only)."""
    from oracle import oracle as O
    O.build()
    return O


@pytest.fixture(scope="session")
def gpu_ctx():
    """A libwassgpu context on device 0; fails loudly if the HIP library or the GPU is missing."""
    import wass_amd
    ctx = wass_amd.Context(0)
    yield ctx
    ctx.close()
