"""CPU: oracle/orc_fft.c (Ooura rdft restated over complex indices) against known answers
produced by the real WebRtc_rdft / aec_rdft_*_128 (tests/golden/fft_golden.npz), bit for bit,
and against oracle/_ref on fresh random vectors when it is present."""
import ctypes as C
import os

import numpy as np
import pytest

from conftest import GOLDEN

G = np.load(os.path.join(GOLDEN, "fft_golden.npz"))
fp = np.ctypeslib.ndpointer(np.float32, flags="C")


def _bind(port):
    port.orc_rdft.argtypes = [C.c_int, C.c_int, fp]
    port.orc_rdft.restype = None
    port.orc_aec_rdft.argtypes = [C.c_int, fp]
    port.orc_aec_rdft.restype = None


def bits(a):
    return np.ascontiguousarray(a).view(np.uint32)


@pytest.mark.parametrize("n", [128, 256])
@pytest.mark.parametrize("isgn,tag", [(1, "fwd"), (-1, "inv")])
def test_ooura_known_answers(oracle_port, n, isgn, tag):
    _bind(oracle_port)
    y = G["in_%d" % n].copy()
    for r in y:
        oracle_port.orc_rdft(n, isgn, r)
    assert np.array_equal(bits(y), bits(G["ooura_%s_%d" % (tag, n)]))


@pytest.mark.parametrize("isgn,tag", [(1, "fwd"), (-1, "inv")])
def test_aec_rdft_known_answers(oracle_port, isgn, tag):
    _bind(oracle_port)
    y = G["in_128"].copy()
    for r in y:
        oracle_port.orc_aec_rdft(isgn, r)
    assert np.array_equal(bits(y), bits(G["aec_%s_128" % tag]))


def test_round_trip_scaling(oracle_port):
    """inverse(forward(x)) * 2/n == x up to float rounding (the caller's scaling, ns_core.c:941-943)."""
    _bind(oracle_port)
    rng = np.random.default_rng(3)
    for n in (128, 256):
        x = rng.standard_normal(n).astype(np.float32) * 1000
        y = x.copy()
        oracle_port.orc_rdft(n, 1, y)
        oracle_port.orc_rdft(n, -1, y)
        assert np.allclose(y * (2.0 / n), x, rtol=0, atol=1e-3)


def test_against_real_reference_random(oracle_port, oracle_ref):
    _bind(oracle_port)
    ip_t = np.ctypeslib.ndpointer(np.int32, flags="C")
    oracle_ref.WebRtc_rdft.argtypes = [C.c_int, C.c_int, fp, ip_t, fp]
    oracle_ref.WebRtc_rdft.restype = None
    oracle_ref.aec_rdft_init.restype = None
    oracle_ref.aec_rdft_init()
    oracle_ref.aec_rdft_forward_128.argtypes = [fp]
    oracle_ref.aec_rdft_inverse_128.argtypes = [fp]
    rng = np.random.default_rng(7)
    for n in (128, 256):
        ip, w = np.zeros(128, np.int32), np.zeros(128, np.float32)
        for isgn in (1, -1):
            for _ in range(100):
                x = (rng.standard_normal(n) * rng.choice([1, 100, 30000])).astype(np.float32)
                a, b = x.copy(), x.copy()
                oracle_ref.WebRtc_rdft(n, isgn, a, ip, w)
                oracle_port.orc_rdft(n, isgn, b)
                assert np.array_equal(bits(a), bits(b))
    for fn, isgn in ((oracle_ref.aec_rdft_forward_128, 1), (oracle_ref.aec_rdft_inverse_128, -1)):
        for _ in range(100):
            x = (rng.standard_normal(128) * rng.choice([1, 100, 30000])).astype(np.float32)
            a, b = x.copy(), x.copy()
            fn(a)
            oracle_port.orc_aec_rdft(isgn, b)
            assert np.array_equal(bits(a), bits(b))
