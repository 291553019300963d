"""The NS kernels' table-driven log / exp (wmix_amd/csrc/libm_dev.h), evaluated on the host from the same source
through wmx_debug_ns_libm, swept against the reference's own expression float(log((double)x)) / float(exp((double)x))
with glibc (oracle/orc_libm.c).  No GPU needed: the functions are plain IEEE double arithmetic with explicit fma."""
import ctypes as C

import numpy as np
import pytest


def _run(fn, x):
    x = np.ascontiguousarray(x, np.float32)
    y = np.zeros_like(x)
    fn(x.ctypes.data_as(C.c_void_p), y.ctypes.data_as(C.c_void_p), C.c_size_t(x.size))
    return y


def _product(wmx, kind, x):
    x = np.ascontiguousarray(x, np.float32)
    y = np.zeros_like(x)
    assert wmx.wmx_debug_ns_libm(kind, x.ctypes.data, y.ctypes.data, x.size) == 0
    return y


def test_log_sweep(wmx, oracle_port):
    """every argument class the NS produces: magn = |X| + 1 in [1, 1e8], 1 + 2 snr; dense next to 1 where log -> 0"""
    rng = np.random.default_rng(11)
    x = np.concatenate([
        1 + rng.random(3_000_000) * 1e-4, 1 + rng.random(3_000_000), np.exp(rng.random(6_000_000) * 18.5),
        np.float32(1) + np.arange(0, 4096, dtype=np.float32) * np.float32(2 ** -23),  # the first floats above 1
        2.0 ** np.arange(0, 40), np.array([1.0, 3.0e7, 3.3e38, np.inf, 0.5, 0.0, -1.0, np.nan]),  # last ones: fallback path
    ]).astype(np.float32)
    with np.errstate(all="ignore"):
        got, want = _product(wmx, 0, x), _run(oracle_port.orc_libm_log, x)
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32))


def test_exp_sweep(wmx, oracle_port):
    """exp(-logLrt), exp(lquantile), exp(flatness): [-120, 90] incl. float-denormal results, overflow, specials"""
    rng = np.random.default_rng(12)
    x = np.concatenate([
        rng.random(6_000_000) * -120, rng.random(4_000_000) * 40 - 10, rng.random(1_000_000) * 2e-3 - 1e-3,
        np.arange(-110, 95, 0.25), np.array([0.0, -0.0, 88.72, 88.73, 100, 710, -710, -800, np.inf, -np.inf, np.nan]),
    ]).astype(np.float32)
    with np.errstate(all="ignore"):
        got, want = _product(wmx, 1, x), _run(oracle_port.orc_libm_exp, x)
    nan = np.isnan(want)
    assert np.array_equal(np.isnan(got), nan)
    assert np.array_equal(got.view(np.uint32)[~nan], want.view(np.uint32)[~nan])


def test_tanh_sweep(wmx, oracle_port):
    """the three indicator arguments width * (feature - threshold) of ns_core.c:700-731: a few units either side of 0,
    dense next to 0 (tanh x ~ x) and in the saturating tail; plus the high-band gain argument and specials"""
    rng = np.random.default_rng(14)
    x = np.concatenate([
        rng.random(4_000_000) * 16 - 8, rng.random(2_000_000) * 60 - 30, (rng.random(2_000_000) * 2 - 1) * 1e-2,
        (rng.random(1_000_000) * 2 - 1) * 1e-6, np.arange(-25, 25, 1 / 64), 10.0 ** np.arange(-40, 3, 0.5),
        np.array([0.0, -0.0, 0.0054, 0.0055, 19.9, 20.0, 20.1, 88, -88, 1e30, -1e30, np.inf, -np.inf, np.nan]),
    ]).astype(np.float32)
    with np.errstate(all="ignore"):
        got, want = _product(wmx, 2, x), _run(oracle_port.orc_libm_tanh, x)
    nan = np.isnan(want)
    assert np.array_equal(np.isnan(got), nan)
    assert np.array_equal(got.view(np.uint32)[~nan], want.view(np.uint32)[~nan])


def _pow_arguments():
    """6 M arguments of the AEC's domain (base in (0, 1], exponent in [1, 30]), 3 M general positive ones, and the cases powf decides
    by rule: zeros, infinities, NaNs, negative bases with integer and non-integer exponents, subnormals, overflow and underflow."""
    rng = np.random.default_rng(13)
    n = 6_000_000
    sx = np.array([1.0, 0.5, 1e-30, 1e-45, 0.0, 2.0, np.inf, np.nan, -0.5, -0.5, -0.5, -0.5, -2.0, -2.0, -0.0, -0.0, 0.0, 1e-40, 3e-39, 1e30, 1e-30,
                   1e30, 1.5, -np.inf, -np.inf, np.inf, 0.999, 2.0, 2.0, 2.0, 2.0, -1.0, -1.0, 1.0, np.nan, 0.5, 7.0])
    se = np.array([5.0, 7.5, 3.0, 2.0, 5.0, 10.0, 2.0, 2.0, 2.0, 3.0, 2.5, -3.0, 127.0, 128.0, 3.0, -3.0, -2.0, 1.5, 0.25, 5.0, 5.0,
                   -5.0, np.inf, 3.0, 2.0, -1.0, -np.inf, 127.99, 128.0, -149.0, -150.0, np.inf, 1e10, np.nan, 0.0, 16777217.0, 0.0])
    x = np.concatenate([rng.random(n // 2), 1 - rng.random(n // 2) * 1e-2, np.exp(rng.random(n // 2) * 40 - 20), sx]).astype(np.float32)
    e = np.concatenate([1 + rng.random(n) * 29, rng.random(n // 2) * 16 - 8, se]).astype(np.float32)
    return x, e


def test_pow_sweep(wmx, oracle_port):
    """The AEC's hNl ^ (overDriveSm * curve): base in (0, 1], exponent in [1, 30].  Since round 5 the product evaluates glibc's own
    powf algorithm (exp2(y log2 x) in double, 16- and 32-entry tables, fused multiply-adds; wmix_amd/csrc/libm_dev.h): it must equal
    the host's powf -- which is NOT correctly rounded -- bit for bit, over the AEC's domain, over general positive arguments, and in
    the cases powf decides by rule (zeros, infinities, NaNs, negative bases with integer and non-integer exponents, subnormals,
    overflow and underflow).  The rounded double pow, which rounds 1-4 matched instead, differs from it in ~0.1 % of arguments."""
    x, e = _pow_arguments()
    got = np.zeros_like(x)
    assert wmx.wmx_debug_pow(x.ctypes.data, e.ctypes.data, got.ctypes.data, x.size) == 0
    p = lambda a: a.ctypes.data_as(C.c_void_p)
    dbl, flt = np.zeros_like(x), np.zeros_like(x)
    with np.errstate(all="ignore"):
        oracle_port.orc_libm_pow_d(p(x), p(e), p(dbl), C.c_size_t(x.size))
        oracle_port.orc_libm_powf(p(x), p(e), p(flt), C.c_size_t(x.size))
    ok = ~np.isnan(flt)
    assert np.array_equal(np.isnan(got), ~ok)
    assert np.array_equal(got.view(np.uint32)[ok], flt.view(np.uint32)[ok])
    # and the sweep is not vacuous: the correctly rounded power is a different function on these arguments
    d = got.view(np.int32)[ok].astype(np.int64) - dbl.view(np.int32)[ok].astype(np.int64)
    assert np.abs(d).max() == 1 and 1e-4 < (d != 0).mean() < 2e-3


@pytest.mark.gpu
def test_pow_sweep_on_the_device(wmx, cuda, oracle_port):
    """The same 9 M arguments through the kernel's own aec_powf on the GPU (wmx_debug_pow_device): the vector ALU's fused multiply-adds,
    conversions, subnormals and the double pow behind the rule cases give the host's powf bit for bit."""
    import torch
    x, e = _pow_arguments()
    dx, de = torch.from_numpy(x).to(cuda), torch.from_numpy(e).to(cuda)
    dy = torch.empty_like(dx)
    rc = wmx.wmx_debug_pow_device(dx.data_ptr(), de.data_ptr(), dy.data_ptr(), x.size, None)
    assert rc == 0, (rc, wmx.wmx_last_error())
    got = dy.cpu().numpy()
    flt = np.zeros_like(x)
    p = lambda a: a.ctypes.data_as(C.c_void_p)
    with np.errstate(all="ignore"):
        oracle_port.orc_libm_powf(p(x), p(e), p(flt), C.c_size_t(x.size))
    ok = ~np.isnan(flt)
    assert np.array_equal(np.isnan(got), ~ok)
    assert np.array_equal(got.view(np.uint32)[ok], flt.view(np.uint32)[ok])


def test_rejects_bad_arguments(wmx):
    assert wmx.wmx_debug_ns_libm(5, None, None, 0) == -10001
