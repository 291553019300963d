"""Extreme inputs through every module, GPU vs oracle: full-scale square waves (every intermediate of the fixed-point code at its
largest -- the reference's int32 products wrap there, digital_agc.c:633, and the kernels must wrap the same way), the Nyquist
tone, constant -32768 / +32767, full-range noise, isolated full-scale impulses, loud bursts alternating with digital silence.
None of this is in the synthetic recipe of the main tests.  (The oracle itself is checked on these signals against the real
reference in tests/test_oracle_extremes.py and under UBSan by tools_dev/sanitize_cpu.sh.)"""
import numpy as np
import pytest

from oracle import loader as L

pytestmark = pytest.mark.gpu


def extreme_signals(freq, n):
    pkt = freq // 100
    rng = np.random.default_rng(5)
    t = np.arange(n * pkt)
    sig = {
        "square40": np.where((t // 40) % 2 == 0, 32767, -32768),
        "nyquist": np.where(t % 2 == 0, 32767, -32768),
        "loud": rng.integers(-32768, 32768, n * pkt),
        "min": np.full(n * pkt, -32768),
        "max": np.full(n * pkt, 32767),
        "impulses": (t % 997 == 0) * 32767 - (t % 1013 == 0) * 32768,
        "burst": np.where((t // (pkt * 50)) % 2 == 0, rng.integers(-32768, 32768, n * pkt), 0),
    }
    names = sorted(sig)
    return pkt, names, np.stack([sig[k] for k in names]).astype(np.int16)


def _same(got, want, names, what):
    for i, k in enumerate(names):
        d = np.abs(got[i].astype(np.int32) - want[i].astype(np.int32))
        assert d.max() == 0, "%s, signal %s: max |d| = %d LSB, first at sample %d" % (what, k, d.max(), int(np.argmax(d > 0)))


@pytest.mark.parametrize("freq", [8000, 16000, 32000])
def test_agc_vad_on_extreme_signals(cuda, oracle_port, freq):
    from test_vadagc_gpu import gpu_agc, gpu_vad
    pkt, names, x = extreme_signals(freq, 700)
    got = gpu_agc(cuda, 1, freq, 5, x.copy())
    _same(got, np.stack([L.run_agc(oracle_port, 1, freq, 5, s, pkt, prefix="orc") for s in x]), names, "agc %d" % freq)
    got = gpu_vad(cuda, 1, freq, 10, 1, x.copy())
    _same(got, np.stack([L.run_vad(oracle_port, 1, freq, 10, s, pkt, prefix="orc") for s in x]), names, "vad %d" % freq)


@pytest.mark.parametrize("freq", [8000, 16000, 32000])
def test_ns_on_extreme_signals(cuda, oracle_port, freq):
    from test_ns_gpu import run_gpu
    pkt, names, x = extreme_signals(freq, 500)
    _same(run_gpu(cuda, 1, freq, x.copy()), np.stack([L.run_ns(oracle_port, 1, freq, s, pkt, prefix="orc") for s in x]), names, "ns %d" % freq)


@pytest.mark.parametrize("freq", [8000, 16000])
def test_nsx_on_extreme_signals(cuda, oracle_port, freq):
    from test_nsx_gpu import run_gpu
    pkt, names, x = extreme_signals(freq, 500)
    _same(run_gpu(cuda, 1, freq, x.copy()), np.stack([L.run_nsx(oracle_port, 1, freq, s, pkt, prefix="orc") for s in x]), names, "nsx %d" % freq)


@pytest.mark.parametrize("freq", [8000, 16000])
@pytest.mark.parametrize("far_kind", ["loud", "square40", "min"])
def test_aecm_on_extreme_signals(cuda, oracle_port, freq, far_kind):
    from test_aecm_gpu import run_gpu
    pkt, names, x = extreme_signals(freq, 500)
    far = x[names.index(far_kind)]
    got, rc = run_gpu(cuda, 1, freq, 10, far, x.copy())
    assert rc == 0
    _same(got, np.stack([L.run_aecm(oracle_port, 1, freq, 10, far, s, pkt, prefix="orc") for s in x]), names, "aecm %d far=%s" % (freq, far_kind))


@pytest.mark.parametrize("freq", [8000, 16000])
@pytest.mark.parametrize("far_kind", ["loud", "square40", "min"])
def test_aec_on_extreme_signals(cuda, oracle_port, freq, far_kind):
    from test_aec_gpu import check_float_path, gpu_aec
    pkt, names, x = extreme_signals(freq, 500)
    far = x[names.index(far_kind)]
    got = gpu_aec(cuda, 1, freq, 10, 0, far, x.copy())
    want = np.stack([L.run_aec(oracle_port, 1, freq, 10, far, s, pkt, prefix="orc") for s in x])
    check_float_path(got, want, max_fraction=1e-4)
