"""CPU: oracle/orc_aec.c (+ orc_chain.c) against golden outputs of the real reference
(tests/golden/aec_golden.npz) and against oracle/_ref on longer runs when present.  The oracle keeps
the reference's operation order, so the comparison is bit-exact."""
import ctypes as C
import os
import sys

import numpy as np
import pytest

from conftest import GOLDEN
from oracle import loader as L

sys.path.insert(0, GOLDEN)
from make_aec_golden import AEC_CASES, CHAIN_CASES, aec_input, aec_pkg  # noqa: E402

G = np.load(os.path.join(GOLDEN, "aec_golden.npz"))


@pytest.mark.parametrize("chn,freq,ims,delay,n", AEC_CASES)
def test_aec_golden(oracle_port, chn, freq, ims, delay, n):
    far, near = aec_input(chn, freq, ims, n)
    got = L.run_aec(oracle_port, chn, freq, ims, far, near, aec_pkg(freq, ims), delay, prefix="orc")
    want = G["aec_%dx%d_%dms_d%d" % (chn, freq, ims, delay)]
    assert np.array_equal(got, want)
    # the canceller does cancel: residual after convergence is well below the near-end level
    if chn != 1:
        return
    tail = slice(want.size // 2, None)
    assert np.abs(want[tail].astype(np.float64)).mean() < 0.7 * np.abs(near[tail].astype(np.float64)).mean()


@pytest.mark.parametrize("chn,freq,stages,n", CHAIN_CASES)
def test_chain_golden(oracle_port, chn, freq, stages, n):
    far, near = aec_input(chn, freq, 10, n, seed=4100)
    got = L.run_chain(oracle_port, chn, freq, 5, stages, far, near, freq // 100, prefix="orc")
    assert np.array_equal(got, G["chain_%dx%d_s%d" % (chn, freq, stages)])


def test_speech_goldens(oracle_port):
    far, near = G["speech_far"], G["speech_near"]
    assert np.array_equal(L.run_aec(oracle_port, 1, 8000, 10, far, near, 80, 0, prefix="orc"), G["speech_aec"])
    assert np.array_equal(L.run_chain(oracle_port, 1, 8000, 5, 15, far, near, 80, prefix="orc"), G["speech_chain"])


def test_startup_phase_is_pass_through_and_rates_are_checked(oracle_port):
    far, near = aec_input(1, 16000, 10, 20)
    out = L.run_aec(oracle_port, 1, 16000, 10, far, near, 160, 0, prefix="orc")
    assert np.array_equal(out[:160 * 6], near[:160 * 6])  # AEC disabled until the buffer size is decided (echo_cancellation.c:651-727)
    oracle_port.orc_aec_init.restype = C.c_void_p
    assert oracle_port.orc_aec_init(1, 32000, 10) is None  # src/webrtc.c:220
    assert oracle_port.orc_aec_init(1, 11025, 10) is None


def test_bad_delay_aborts_like_the_wrapper(oracle_port):
    far, near = aec_input(1, 8000, 10, 5)
    out = np.full_like(near, 7)
    fn = oracle_port.orc_run_aec
    fn.restype = C.c_int
    i16p = np.ctypeslib.ndpointer(np.int16, flags="C_CONTIGUOUS")
    fn.argtypes = [C.c_int, C.c_int, C.c_int, i16p, i16p, i16p, C.c_int, C.c_int, C.c_int]
    assert fn(1, 8000, 10, far, near, out, 80, 5, 600) == -1  # msInSndCardBuf > 500 (echo_cancellation.c:369-373)
    assert (out == 7).all()  # nothing written: the wrapper returns before the copy-out (src/webrtc.c:463-472)


@pytest.mark.parametrize("chn,freq,ims,delay", [(1, 16000, 10, 0), (1, 8000, 20, 40), (2, 8000, 10, 300)])
def test_against_real_reference_long(oracle_port, oracle_ref, chn, freq, ims, delay):
    n = 3200 * (freq // 100) // aec_pkg(freq, ims)  # 32 s: crosses the 500*mult noise-init blocks
    far, near = aec_input(chn, freq, ims, n, seed=5000)
    a = L.run_aec(oracle_ref, chn, freq, ims, far, near, aec_pkg(freq, ims), delay)
    b = L.run_aec(oracle_port, chn, freq, ims, far, near, aec_pkg(freq, ims), delay, prefix="orc")
    assert np.array_equal(a, b)


def test_chain_against_real_reference_long(oracle_port, oracle_ref):
    far, near = aec_input(1, 16000, 10, 2500, seed=5100)
    a = L.run_chain(oracle_ref, 1, 16000, 5, 15, far, near, 160)
    b = L.run_chain(oracle_port, 1, 16000, 5, 15, far, near, 160, prefix="orc")
    assert np.array_equal(a, b)


@pytest.mark.parametrize("chn,freq", [(1, 16000), (1, 8000), (2, 16000), (2, 8000)])
def test_chain_at_the_daemons_interval_against_real_reference(oracle_port, oracle_ref, chn, freq):
    """The daemon hands WMIX_INTERVAL_MS = 20 to aec_init / agc_init / vad_init (src/wmixConf.h:112, src/wmix.c:636,684,703)
    and calls the chain with 20 ms per heartbeat: VAD packets of 20 ms, AEC packets of 20 ms at 8 kHz."""
    n_calls, per_call = 700, freq // 50  # frames per 20 ms heartbeat
    far, near = aec_input(chn, freq, 20 if freq == 8000 else 10, n_calls * (2 if freq == 16000 else 1), seed=5200 + chn + freq // 8000)
    a = L.run_chain(oracle_ref, chn, freq, 5, 15, far, near, per_call, interval_ms=20)
    b = L.run_chain(oracle_port, chn, freq, 5, 15, far, near, per_call, prefix="orc", interval_ms=20)
    assert a.size == n_calls * per_call * chn and np.array_equal(a, b)
    # and the cadence matters: 10 ms handles fed the same audio give another result (the VAD packet differs)
    c = L.run_chain(oracle_port, chn, freq, 5, 15, far, near, per_call, prefix="orc", interval_ms=10)
    assert not np.array_equal(b, c)


@pytest.mark.parametrize("freq,ims", [(16000, 10), (8000, 20)])
def test_reported_delay_that_changes_from_call_to_call_against_real_reference(oracle_port, oracle_ref, freq, ims):
    """aec_process2's delayms is a per-call argument (src/webrtc.c:410-483): EstBufDelay filters it, knownDelay follows with its
    hysteresis and the far-end read pointer moves (W: echo_cancellation.c:821-872, aec_core.c:1753-1760).  A wandering delay, steps
    across the hysteresis, and the clamp at 0: the port against the rebuilt reference, call by call."""
    pkg = aec_pkg(freq, ims)
    n = 1800
    far, near = aec_input(1, freq, ims, n, seed=5300 + freq // 8000)
    t = np.arange(n)
    delays = (20 + 15 * np.sin(t / 37.0) + (t % 7)).astype(np.int32)
    delays[600:900] = 260   # a step far beyond the 224-sample threshold ...
    delays[900:1200] = 0    # ... and back below 96
    delays[1500:] = 480
    a = L.run_aec_delays(oracle_ref, 1, freq, ims, far, near, pkg, delays)
    b = L.run_aec_delays(oracle_port, 1, freq, ims, far, near, pkg, delays, prefix="orc")
    assert np.array_equal(a, b)
    c = L.run_aec(oracle_port, 1, freq, ims, far, near, pkg, 20, prefix="orc")
    assert not np.array_equal(b, c)
