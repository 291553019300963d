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


_HOST_POWF = None


def host_powf_is_the_products():
    """The float AEC is bit-exact against a reference that links THIS libm: the device evaluates glibc 2.35's powf (the
    fused-multiply-add build its ifunc picks on x86-64 with FMA), and the oracle calls the host's powf.  On a host whose powf is another
    function -- no FMA, another glibc, musl -- the two differ by one float ulp in ~0.1 % of arguments and the float path falls back to
    the class rounds 1-4 were in (<= 1 LSB on a 2e-5 fraction of samples).  Probed once per session (round-5 ADVICE): 2 M arguments of
    the AEC's own domain through wmx_debug_pow (the kernel's source, compiled for the host) and through the host's powf."""
    global _HOST_POWF
    if _HOST_POWF is None:
        import ctypes as C
        import warnings

        import numpy as np
        from oracle import loader
        from wmix_amd import _lib
        rng = np.random.default_rng(5)
        x = np.concatenate([rng.random(1_000_000), 1 - rng.random(1_000_000) * 1e-2]).astype(np.float32)
        e = (1 + rng.random(2_000_000) * 29).astype(np.float32)
        got, flt = np.zeros_like(x), np.zeros_like(x)
        assert _lib.lib().wmx_debug_pow(x.ctypes.data, e.ctypes.data, got.ctypes.data, x.size) == 0
        p = lambda a: a.ctypes.data_as(C.c_void_p)  # noqa: E731
        loader.port().orc_libm_powf(p(x), p(e), p(flt), C.c_size_t(x.size))
        _HOST_POWF = bool(np.array_equal(got.view(np.uint32), flt.view(np.uint32)))
        if not _HOST_POWF:
            warnings.warn("this host's powf is not glibc's FMA build (%d of %d probe arguments differ): the float-path tests fall back to "
                          "<= 1 LSB on a 2e-5 fraction of samples" % (int((got.view(np.uint32) != flt.view(np.uint32)).sum()), x.size))
    return _HOST_POWF
