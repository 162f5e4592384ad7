import os
import sys

import pytest

# The coarse tail of a single-rank V-cycle as a HIP graph is off by default (it stopped paying once the smallest levels ran
# as one kernel: DESIGN.md section 0) but stays a feature: the suite runs with it on — recording, replay and invalidation
# are what many tests are about — and switches it off where a test says so.  Read when a solver is created; inherited by the
# worker processes of the multi-rank tests.
os.environ.setdefault("HYPRE_AMD_CYCLE_GRAPH_ROWS", "100000")

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def lib():
    from hypre_amd import binding
    return binding.load_library(build_if_missing=True)


@pytest.fixture(scope="session")
def gpu_lib(lib):
    if not lib.hypre_amd_DeviceAvailable():
        pytest.fail("no HIP device: the gpu-marked tests must run on the GPU box")
    return lib


@pytest.fixture(scope="session")
def oracle():
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import pyoracle
    pyoracle.load()
    return pyoracle


def free_port():
    """A TCP port nobody is listening on right now (the kernel picks it), for a torch.distributed rendezvous on
    127.0.0.1: fixed or hashed port numbers can collide between tests and leave a rank waiting for a store."""
    import socket
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sock:
        sock.bind(("127.0.0.1", 0))
        return sock.getsockname()[1]
