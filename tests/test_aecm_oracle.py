"""CPU: oracle/orc_aecm.c (the fixed-point echo canceller the reference selects with its AECM switch, src/webrtc.c:168-191)
against goldens of the real reference (tests/golden/aecm_golden.npz) -- bit-exact, integer path -- and against oracle/_ref
on other inputs when it is present (10 ms packets only: with 20 ms packets at 8 kHz the reference reads uninitialised
memory, see make_aecm_golden.py; that mode is pinned by the goldens, generated on a zero-filled heap)."""
import os
import sys

import numpy as np
import pytest

from conftest import GOLDEN
from oracle import loader as L

sys.path.insert(0, GOLDEN)
from make_aecm_golden import AECM_CASES, HEAD, TAIL, aecm_case_input, case_key, packet_crcs, speech_case  # noqa: E402

G = np.load(os.path.join(GOLDEN, "aecm_golden.npz"))
NSG = np.load(os.path.join(GOLDEN, "ns_golden.npz"))


def check_against_golden(y, key, per):
    assert np.array_equal(y[: HEAD * per], G[key + "_head"])
    assert np.array_equal(y[-TAIL * per:], G[key + "_tail"])
    assert np.array_equal(packet_crcs(y, per), G[key + "_crc"])


@pytest.mark.parametrize("chn,freq,iv,n,delay,split", AECM_CASES)
def test_oracle_matches_reference_golden(oracle_port, chn, freq, iv, n, delay, split):
    far, near, pkt = aecm_case_input(chn, freq, iv, n)
    y = L.run_aecm(oracle_port, chn, freq, iv, far, near, pkt, delay, split, prefix="orc")
    check_against_golden(y, case_key(chn, freq, iv, delay, split), pkt * chn)
    d = (y.astype(np.int32) - near).reshape(n, -1)[:, ::chn]  # left channel (the output is duplicated to the others)
    assert np.abs(d[:3]).max() == 0      # start-up: pass-through until the far-end buffer has filled (echo_control_mobile.c:341-408)
    assert np.abs(d[-100:]).mean() > 50  # ... and a working canceller afterwards


def test_oracle_matches_reference_golden_speech(oracle_port):
    far, near = speech_case(NSG)
    assert np.array_equal(L.run_aecm(oracle_port, 1, 8000, 10, far, near, 80, prefix="orc"), G["speech_1x8000_out"])


def test_rejected_arguments(oracle_port):
    """32 kHz -> NULL like aec_init (src/webrtc.c:220); a delay outside [0, 500] ms makes WebRtcAecm_Process return -1
    after processing, and the wrapper stops there without writing the packet (src/webrtc.c:382-387)."""
    import ctypes as C
    oracle_port.orc_aecm_init.restype = C.c_void_p
    assert oracle_port.orc_aecm_init(1, 32000, 10) is None and oracle_port.orc_aecm_init(1, 44100, 10) is None
    far, near, pkt = aecm_case_input(1, 16000, 10, 20)
    y = L.run_aecm(oracle_port, 1, 16000, 10, far, near, pkt, 600, 0, prefix="orc", expect_rc=-1)
    assert not y.any()


@pytest.mark.parametrize("chn,freq,delay,split", [(1, 16000, 0, 1), (2, 8000, 0, 0), (1, 8000, 300, 0), (1, 16000, 20, 0)])
def test_oracle_equals_real_reference_other_inputs(oracle_port, oracle_ref, chn, freq, delay, split):
    far, near, pkt = aecm_case_input(chn, freq, 10, 1200, seed=424242)
    a = L.run_aecm(oracle_ref, chn, freq, 10, far, near, pkt, delay, split)
    b = L.run_aecm(oracle_port, chn, freq, 10, far, near, pkt, delay, split, prefix="orc")
    assert np.array_equal(a, b)
