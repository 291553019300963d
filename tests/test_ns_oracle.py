"""CPU: oracle/orc_ns.c (float NS as wmix drives it) against golden outputs of the real
reference (tests/golden/ns_golden.npz, made by tests/golden/make_ns_golden.py) -- bit-exact --
and against oracle/_ref on longer runs when it is present."""
import os
import sys

import numpy as np
import pytest

from conftest import GOLDEN
from oracle import loader as L

sys.path.insert(0, GOLDEN)
from make_ns_golden import NS_CASES, ns_case_input  # noqa: E402

G = np.load(os.path.join(GOLDEN, "ns_golden.npz"))


@pytest.mark.parametrize("chn,freq,nf", NS_CASES)
def test_oracle_matches_reference_golden_synthetic(oracle_port, chn, freq, nf):
    x = ns_case_input(chn, freq, nf)
    got = L.run_ns(oracle_port, chn, freq, x, freq // 100, prefix="orc")
    want = G["synth_%dx%d" % (chn, freq)]
    assert np.array_equal(got, want)
    if freq == 32000:  # second half of every 10 ms packet is zero
        assert not got.reshape(nf, 320, chn)[:, 160:, :].any()


@pytest.mark.parametrize("name,chn,freq", [("speech_1x8000", 1, 8000), ("speech_2x16000", 2, 16000)])
def test_oracle_matches_reference_golden_speech(oracle_port, name, chn, freq):
    got = L.run_ns(oracle_port, chn, freq, G[name + "_in"], freq // 100, prefix="orc")
    assert np.array_equal(got, G[name + "_out"])


def test_multi_packet_calls_equal_single_packet_calls(oracle_port):
    x = ns_case_input(1, 16000, 40)
    a = L.run_ns(oracle_port, 1, 16000, x, 160, prefix="orc")
    b = L.run_ns(oracle_port, 1, 16000, x, 320, prefix="orc")  # 20 ms calls = 2 packets each (daemon default)
    assert np.array_equal(a, b)


def test_unsupported_rates_are_rejected(oracle_port):
    import ctypes as C
    oracle_port.orc_ns_init.restype = C.c_void_p
    assert oracle_port.orc_ns_init(1, 44100) is None
    assert oracle_port.orc_ns_init(1, 48000) is None  # > 32000 (src/webrtc.c:563)


@pytest.mark.parametrize("chn,freq", [(1, 16000), (1, 8000), (2, 32000)])
def test_oracle_equals_real_reference_long_run(oracle_port, oracle_ref, chn, freq):
    nf = 1500  # crosses the 50 / 200 / 500 / 1000 block boundaries
    x = ns_case_input(chn, freq, nf, seed=99)
    a = L.run_ns(oracle_ref, chn, freq, x, freq // 100)
    b = L.run_ns(oracle_port, chn, freq, x, freq // 100, prefix="orc")
    assert np.array_equal(a, b)
