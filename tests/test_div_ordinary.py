"""div_ordinary (wmix_amd/csrc/libm_dev.h): the compiler's fp32 division sequence without its rescaling and special-case
instructions, used by the NS kernels where the operand ranges are known.  It must give the IEEE quotient, bit for bit, for
every ORDINARY pair: a finite (or 0), b normal, both within 2^+-96, exponents less than 96 apart, quotient normal.  Swept on
the host (the same source compiled for the CPU, numpy's float32 division as the reference) and, under `-m gpu`, on the device
against the compiler's own `a / b`."""
import numpy as np
import pytest


def _pairs(seed, n, emax):
    """random signs, mantissas and exponents in [-emax, emax], plus the value classes the NS feeds in"""
    rng = np.random.default_rng(seed)
    a = np.ldexp(1 + rng.random(n), rng.integers(-emax, emax + 1, n)) * rng.choice([-1.0, 1.0], n)
    b = np.ldexp(1 + rng.random(n), rng.integers(-emax, emax + 1, n)) * rng.choice([-1.0, 1.0], n)
    a, b = a.astype(np.float32), b.astype(np.float32)
    k = n // 10
    a[:k] = 0.0                                                      # a stream's first frame: magnPrev == 0
    a[k:2 * k] = (1 + rng.random(k) * 8.4e6).astype(np.float32)      # magnitudes |X| + 1
    b[k:2 * k] = (1e-4 + np.exp(rng.random(k) * 17)).astype(np.float32)  # noise + 0.0001
    a[2 * k:3 * k] = (rng.random(k) * 40).astype(np.float32)         # quantile steps over counters
    b[2 * k:3 * k] = rng.integers(1, 202, k).astype(np.float32)
    b[3 * k:4 * k] = a[3 * k:4 * k]                                  # quotient exactly 1
    b[4 * k:5 * k] = np.nextafter(a[4 * k:5 * k], np.float32(np.inf))  # ... and one ulp off
    b[a == b] = np.where(b[a == b] == 0, np.float32(1), b[a == b])
    b[b == 0] = np.float32(1)
    return a, b


@pytest.mark.parametrize("seed", [1, 2, 3, 4])
def test_host_sweep_against_ieee(wmx, seed):
    a, b = _pairs(seed, 5_000_000, 40)
    q = np.zeros_like(a)
    assert wmx.wmx_debug_div_host(a.ctypes.data, b.ctypes.data, q.ctypes.data, a.size) == 0
    with np.errstate(all="ignore"):
        want = a / b
    assert np.array_equal(q.view(np.uint32), want.view(np.uint32))


@pytest.mark.gpu
@pytest.mark.parametrize("emax", [40, 46])
def test_device_sweep_against_the_compilers_division(wmx, cuda, emax):
    """10^8 pairs: the device's own reciprocal estimate in div_ordinary, the compiler's full sequence beside it, IEEE from the host"""
    import torch
    bad = 0
    for seed in range(5):
        a, b = _pairs(100 * emax + seed, 10_000_000, emax)
        da, db = torch.from_numpy(a).to(cuda), torch.from_numpy(b).to(cuda)
        q0, q1 = torch.empty_like(da), torch.empty_like(da)
        assert wmx.wmx_debug_div(da.data_ptr(), db.data_ptr(), q0.data_ptr(), q1.data_ptr(), a.size, None) == 0
        torch.cuda.synchronize()
        bad += int((q0.view(torch.int32) != q1.view(torch.int32)).sum())
        with np.errstate(all="ignore"):
            want = a / b
        assert np.array_equal(q1.cpu().numpy().view(np.uint32), want.view(np.uint32))  # the compiler's division is IEEE
    assert bad == 0
