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
def oracle_port():
    """Our C restatement (oracle/orc_*.c), built on demand with gcc."""
    from oracle import loader
    return loader.port()


@pytest.fixture(scope="session")
def oracle_ref():
    """The real reference build (oracle/_ref); skip when it was not prebuilt."""
    from oracle import loader
    if not loader.have_ref():
        pytest.skip("oracle/_ref/libwmixref.so not present (built only where /root/reference exists)")
    return loader.ref()


@pytest.fixture(scope="session")
def wmx():
    """libwmix_amd.so via ctypes.  Fails (does not skip) when it is not built."""
    from wmix_amd import _lib
    return _lib.lib()


@pytest.fixture(scope="session")
def cuda():
    import torch
    assert torch.cuda.is_available(), "gpu-marked test needs a GPU"
    return torch.device("cuda:0")
