"""The oracle restatement against the REAL reference (oracle/_ref, built where /root/reference exists) on the extreme signals
of tests/test_extremes_gpu.py: full-scale square waves, the Nyquist tone, constant extremes, full-range noise, impulses,
bursts.  The fixed-point modules wrap their int32 arithmetic on these (the reference through undefined behaviour that x86
resolves by wrapping, the restatement explicitly): they must agree bit for bit, the float path within its +-1 LSB."""
import numpy as np
import pytest

from oracle import loader as L


def _signals(freq, n):
    pkt = freq // 100
    rng = np.random.default_rng(5)
    t = np.arange(n * pkt)
    sig = {
        "square40": np.where((t // 40) % 2 == 0, 32767, -32768),
        "nyquist": np.where(t % 2 == 0, 32767, -32768),
        "loud": rng.integers(-32768, 32768, n * pkt),
        "min": np.full(n * pkt, -32768),
        "impulses": (t % 997 == 0) * 32767 - (t % 1013 == 0) * 32768,
        "burst": np.where((t // (pkt * 50)) % 2 == 0, rng.integers(-32768, 32768, n * pkt), 0),
    }
    return pkt, {k: v.astype(np.int16) for k, v in sig.items()}


@pytest.mark.parametrize("freq", [8000, 16000, 32000])
def test_port_equals_reference_on_extreme_signals(oracle_port, oracle_ref, freq):
    pkt, sig = _signals(freq, 300)
    far = sig["loud"]
    for name, x in sig.items():
        for fn, args in ((L.run_agc, (1, freq, 5, x, pkt)), (L.run_vad, (1, freq, 10, x, pkt)), (L.run_ns, (1, freq, x, pkt))):
            a, b = fn(oracle_port, *args, prefix="orc"), fn(oracle_ref, *args, prefix="ref")
            assert np.array_equal(a, b), "%s %s %d" % (fn.__name__, name, freq)
        if freq < 32000:
            for fn, args in ((L.run_nsx, (1, freq, x, pkt)), (L.run_aecm, (1, freq, 10, far, x, pkt)), (L.run_aec, (1, freq, 10, far, x, pkt)),
                             (L.run_chain, (1, freq, 5, 15, far, x, pkt))):
                a, b = fn(oracle_port, *args, prefix="orc"), fn(oracle_ref, *args, prefix="ref")
                assert np.abs(a.astype(np.int32) - b.astype(np.int32)).max() <= (0 if fn in (L.run_nsx, L.run_aecm) else 1), \
                    "%s %s %d" % (fn.__name__, name, freq)


@pytest.mark.parametrize("kind", range(4))
def test_mfft_port_equals_reference_on_extreme_values(oracle_port, oracle_ref, kind):
    n = 256
    rng = np.random.default_rng(9)
    base_re, base_im = rng.standard_normal(n).astype(np.float32), rng.standard_normal(n).astype(np.float32)
    spike = np.zeros(n, np.float32)
    spike[17] = 32767
    for re, im in ((base_re * np.float32(1e18), base_im * np.float32(1e18)), (base_re * np.float32(1e-41), base_im * np.float32(1e-41)),
                   (spike, np.zeros(n, np.float32)), (np.zeros(n, np.float32), np.zeros(n, np.float32))):
        a, b = L.mfft(oracle_port, kind, re, im, n, prefix="orc"), L.mfft(oracle_ref, kind, re, im, n, prefix="ref")
        for k in a:
            assert np.array_equal(a[k].view(np.uint32), b[k].view(np.uint32)), (kind, k)


@pytest.mark.parametrize("freq", [8000, 16000])
def test_port_equals_reference_on_perfect_echoes_and_tones(oracle_port, oracle_ref, freq):
    pkt, n = freq // 100, 500
    rng = np.random.default_rng(21)
    far = rng.integers(-20000, 20001, n * pkt).astype(np.int16)
    t = np.arange(n * pkt)
    for near in (far, -far, np.roll(far, 3), far // 2, np.zeros_like(far)):
        a, b = L.run_aec(oracle_port, 1, freq, 10, far, near, pkt, prefix="orc"), L.run_aec(oracle_ref, 1, freq, 10, far, near, pkt, prefix="ref")
        assert np.array_equal(a, b)
        a, b = L.run_aecm(oracle_port, 1, freq, 10, far, near, pkt, prefix="orc"), L.run_aecm(oracle_ref, 1, freq, 10, far, near, pkt, prefix="ref")
        assert np.array_equal(a, b)
    for x in (np.round(12000 * np.sin(2 * np.pi * t * (16 * freq / 256) / freq)), np.round(12000 * np.sin(2 * np.pi * t * (16.5 * freq / 256) / freq)),
              (t % pkt == 0) * 30000.0):
        x = x.astype(np.int16)
        assert np.array_equal(L.run_ns(oracle_port, 1, freq, x, pkt, prefix="orc"), L.run_ns(oracle_ref, 1, freq, x, pkt, prefix="ref"))
        assert np.array_equal(L.run_nsx(oracle_port, 1, freq, x, pkt, prefix="orc"), L.run_nsx(oracle_ref, 1, freq, x, pkt, prefix="ref"))
