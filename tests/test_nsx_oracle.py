"""CPU: oracle/orc_nsx.c (the fixed-point noise suppressor the reference selects with MAKE_WEBRTC_NSX, src/webrtc.c:512-521)
against goldens of the real reference (tests/golden/nsx_golden.npz) -- bit-exact, integer path -- against the real
WebRtcSpl real FFT, and against oracle/_ref on other inputs when it is present."""
import ctypes as C
import os
import sys

import numpy as np
import pytest

from conftest import GOLDEN
from oracle import loader as L

sys.path.insert(0, GOLDEN)
from make_nsx_golden import HEAD, NSX_CASES, TAIL, case_key, nsx_case_input, packet_crcs  # noqa: E402

G = np.load(os.path.join(GOLDEN, "nsx_golden.npz"))
NSG = np.load(os.path.join(GOLDEN, "ns_golden.npz"))


def check_against_golden(y, chn, freq, amp):
    per, k = freq // 100 * chn, case_key(chn, freq, amp)
    assert np.array_equal(y[: HEAD * per], G[k + "_head"])
    assert np.array_equal(y[-TAIL * per:], G[k + "_tail"])
    assert np.array_equal(packet_crcs(y, per), G[k + "_crc"])


@pytest.mark.parametrize("chn,freq,nf,amp", NSX_CASES)
def test_oracle_matches_reference_golden(oracle_port, chn, freq, nf, amp):
    y = L.run_nsx(oracle_port, chn, freq, nsx_case_input(chn, freq, nf, amp), freq // 100, prefix="orc")
    check_against_golden(y, chn, freq, amp)
    if freq == 32000:  # SURVEY quirk 3 holds for the NSX wrapper too: second half of every 10 ms packet is zero
        assert not y.reshape(nf, 320, chn)[:, 160:, :].any()


@pytest.mark.parametrize("name,chn,freq", [("speech_1x8000", 1, 8000), ("speech_2x16000", 2, 16000)])
def test_oracle_matches_reference_golden_speech(oracle_port, name, chn, freq):
    got = L.run_nsx(oracle_port, chn, freq, NSG[name + "_in"], freq // 100, prefix="orc")
    assert np.array_equal(got, G[name + "_out"])


@pytest.mark.parametrize("order", [7, 8])
def test_spl_real_fft_known_answers(oracle_port, order):
    n = 1 << order
    i16p = np.ctypeslib.ndpointer(np.int16, flags="C")
    oracle_port.orc_spl_real_fft.argtypes = [C.c_int, i16p, i16p]
    oracle_port.orc_spl_real_ifft.argtypes = [C.c_int, i16p, i16p]
    oracle_port.orc_spl_real_ifft.restype = C.c_int
    for x, want in zip(G["fft_in_%d" % order], G["fft_fwd_%d" % order]):
        got = np.zeros(n + 2, np.int16)
        oracle_port.orc_spl_real_fft(order, np.ascontiguousarray(x), got)
        assert np.array_equal(got, want)
    for x, want, sc in zip(G["ifft_in_%d" % order], G["ifft_out_%d" % order], G["ifft_scale_%d" % order]):
        got = np.zeros(n, np.int16)
        assert oracle_port.orc_spl_real_ifft(order, np.ascontiguousarray(x), got) == sc
        assert np.array_equal(got, want)
    assert G["ifft_scale_%d" % order].min() == 0 and G["ifft_scale_%d" % order].max() >= 4  # the stage shifts were exercised


def test_multi_packet_calls_equal_single_packet_calls(oracle_port):
    x = nsx_case_input(1, 16000, 60, 3000)
    a = L.run_nsx(oracle_port, 1, 16000, x, 160, prefix="orc")
    b = L.run_nsx(oracle_port, 1, 16000, x, 320, prefix="orc")
    assert np.array_equal(a, b)


@pytest.mark.parametrize("chn,freq,amp", [(1, 16000, 800), (2, 8000, 12000), (1, 32000, 20000), (1, 8000, 3)])
def test_oracle_equals_real_reference_other_inputs(oracle_port, oracle_ref, chn, freq, amp):
    nf = 1200
    x = nsx_case_input(chn, freq, nf, amp, seed=31337)
    a = L.run_nsx(oracle_ref, chn, freq, x, freq // 100)
    b = L.run_nsx(oracle_port, chn, freq, x, freq // 100, prefix="orc")
    assert np.array_equal(a, b)
