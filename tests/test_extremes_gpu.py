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


def _stereo(x):
    """two-channel versions of the extreme signals: every ordered pair (left = signal i, right = signal j) for a few i, j --
    opposite full-scale signs in the two channels is what the wrappers' (L + R) / 2 averaging must round like the reference"""
    pairs = [(0, 1), (1, 0), (2, 3), (3, 4), (4, 3), (5, 6), (6, 2), (3, 3)]
    out = np.empty((len(pairs), x.shape[1] * 2), np.int16)
    for r, (i, j) in enumerate(pairs):
        out[r, 0::2] = x[i]
        out[r, 1::2] = x[j]
    return ["%d|%d" % p for p in pairs], out


@pytest.mark.parametrize("freq", [8000, 16000, 32000])
def test_two_channel_agc_vad_ns_on_extreme_signals(cuda, oracle_port, freq):
    from test_ns_gpu import run_gpu as gpu_ns
    from test_vadagc_gpu import gpu_agc, gpu_vad
    pkt, _, x = extreme_signals(freq, 400)
    names, s2 = _stereo(x)
    # the oracle runners take FRAMES (samples per channel) per call
    _same(gpu_agc(cuda, 2, freq, 5, s2.copy()), np.stack([L.run_agc(oracle_port, 2, freq, 5, s, pkt, prefix="orc") for s in s2]), names, "agc 2ch %d" % freq)
    _same(gpu_vad(cuda, 2, freq, 10, 1, s2.copy()), np.stack([L.run_vad(oracle_port, 2, freq, 10, s, pkt, prefix="orc") for s in s2]), names, "vad 2ch %d" % freq)
    _same(gpu_ns(cuda, 2, freq, s2.copy()), np.stack([L.run_ns(oracle_port, 2, freq, s, pkt, prefix="orc") for s in s2]), names, "ns 2ch %d" % freq)


@pytest.mark.parametrize("value", [0, 1, 30, 90])
def test_agc_gain_settings_on_extreme_signals(cuda, oracle_port, value):
    """agc_init's compression gain at its ends (src/webrtc.c:726-760 passes it straight to WebRtcAgc_set_config)"""
    from test_vadagc_gpu import gpu_agc
    pkt, names, x = extreme_signals(16000, 400)
    _same(gpu_agc(cuda, 1, 16000, value, x.copy()), np.stack([L.run_agc(oracle_port, 1, 16000, value, s, pkt, prefix="orc") for s in x]), names, "agc gain %d" % value)


@pytest.mark.parametrize("freq,ims,k", [(8000, 20, 1), (16000, 30, 1), (16000, 10, 3), (32000, 20, 2)])
def test_vad_intervals_on_extreme_signals(cuda, oracle_port, freq, ims, k):
    from test_vadagc_gpu import gpu_vad, vad_pkg
    pkt = vad_pkg(freq, ims)  # the wrapper's own packet: 20 ms when the interval allows it and the rate is at most 16 kHz, else 10 ms
    _, names, x = extreme_signals(freq, 480)
    x = x[:, : (x.shape[1] // (pkt * k)) * pkt * k]
    _same(gpu_vad(cuda, 1, freq, ims, k, x.copy()), np.stack([L.run_vad(oracle_port, 1, freq, ims, s, k * pkt, prefix="orc") for s in x]), names,
          "vad %d Hz %d ms x%d" % (freq, ims, k))


@pytest.mark.parametrize("freq,ims,delay", [(8000, 20, 0), (16000, 10, 120), (16000, 20, 500), (8000, 10, 40)])
def test_aec_aecm_intervals_and_delays_on_extreme_signals(cuda, oracle_port, freq, ims, delay):
    from test_aec_gpu import check_float_path, gpu_aec
    from test_aecm_gpu import run_gpu as gpu_aecm
    pkt = freq // 1000 * ims
    _, names, x = extreme_signals(freq, 300 * ims // 10)
    x = x[:, : (x.shape[1] // pkt) * pkt]
    far = x[names.index("loud")]
    got = gpu_aec(cuda, 1, freq, ims, delay, far, x.copy())
    want = np.stack([L.run_aec(oracle_port, 1, freq, ims, far, s, pkt, delay_ms=delay, prefix="orc") for s in x])
    check_float_path(got, want, max_fraction=1e-4)
    got, rc = gpu_aecm(cuda, 1, freq, ims, far, x.copy(), delay=delay)
    assert rc == 0
    _same(got, np.stack([L.run_aecm(oracle_port, 1, freq, ims, far, s, pkt, delay_ms=delay, prefix="orc") for s in x]), names,
          "aecm %d Hz %d ms delay %d" % (freq, ims, delay))


@pytest.mark.parametrize("kind", ["min", "max", "alternating", "full_range"])
def test_mixer_on_extreme_sources(cuda, oracle_port, kind):
    """wmix_load_data / wmix_pcm_zoom with sources at the ends of int16: the N-way sum saturates at every sample, the
    averaging of channel pairs and the volume reduction see their largest operands (src/wmix.c:49-222, 113-127)."""
    from test_mix_gpu import gpu_load
    from test_mix_oracle import _bind, orc_load, orc_zoom
    import torch
    from wmix_amd import mix
    _bind(oracle_port)
    rng = np.random.default_rng(31)

    def make(n):
        if kind == "min":
            return np.full(n, -32768, np.int16)
        if kind == "max":
            return np.full(n, 32767, np.int16)
        if kind == "alternating":
            return np.where(np.arange(n) % 2 == 0, 32767, -32768).astype(np.int16)
        return rng.integers(-32768, 32768, n).astype(np.int16)

    for (ring_chn, ring_freq, freq, chn, rmode, rarg, nsrc, sbytes, start) in (
            (1, 8000, 32000, 2, 1, 1, 8, 1280, 0), (1, 8000, 8000, 1, 2, 2, 8, 320, 64), (2, 16000, 16000, 2, 1, 3, 6, 1280, 0),
            (1, 8000, 44100, 2, 3, 1, 5, 1764, 32), (1, 8000, 11025, 1, 1, 1, 7, 440, 0)):
        src = make(nsrc * sbytes // 2 + 8)
        want, meta = orc_load(oracle_port, ring_chn, ring_freq, freq, chn, rmode, rarg, nsrc, sbytes, start, src)
        rings, h, t = gpu_load(cuda, ring_chn, ring_freq, freq, chn, rmode, rarg, nsrc, sbytes, start, src)
        assert np.array_equal(rings[0], want), (kind, ring_chn, ring_freq, freq, chn, rmode, rarg)
        assert (t, h) == tuple(meta[-1])
    for (ic, ifr, oc, ofr, n) in ((1, 8000, 2, 16000, 640), (2, 32000, 1, 8000, 2560), (2, 44100, 2, 8000, 3528), (1, 16000, 1, 48000, 640),
                                  (2, 8000, 2, 8000, 640)):
        x = make(n // 2)
        got = mix.pcm_zoom(ic, ifr, torch.from_numpy(x[None, :]).to(cuda), oc, ofr).cpu().numpy()[0]
        assert np.array_equal(got, orc_zoom(oracle_port, ic, ifr, x, oc, ofr)), (kind, ic, ifr, oc, ofr)


@pytest.mark.parametrize("kind", range(4))
@pytest.mark.parametrize("what", ["huge", "tiny", "spike", "zeros"])
def test_mfft_on_extreme_values(cuda, oracle_port, kind, what):
    """math/fft.c on values at the ends of float: 1e18 (the amplitude's re*re + im*im stays finite in double, as in the
    reference), float denormals, one full-scale spike in zeros, all zeros (atan2(0, 0))."""
    import torch
    from test_mfft_gpu import check_outputs
    from wmix_amd import mfft
    n = 256
    rng = np.random.default_rng(9)
    re, im = rng.standard_normal(n).astype(np.float32), rng.standard_normal(n).astype(np.float32)
    if what == "huge":
        re, im = re * np.float32(1e18), im * np.float32(1e18)
    elif what == "tiny":
        re, im = re * np.float32(1e-41), im * np.float32(1e-41)
    elif what == "spike":
        re[:], im[:] = 0, 0
        re[17] = 32767
    else:
        re[:], im[:] = 0, 0
    got = mfft.transform(kind, torch.from_numpy(re[None]).to(cuda), torch.from_numpy(im[None]).to(cuda))
    want = L.mfft(oracle_port, kind, re, im, n, prefix="orc")
    check_outputs(got, lambda k: want[k], (kind, what))


@pytest.mark.parametrize("freq", [8000, 16000])
def test_echo_cancellers_on_perfect_echoes(cuda, oracle_port, freq):
    """near == far, bit for bit (coherence exactly 1: the NLP's `1 - cohxd` lands on 0 or just below it, and powf of a negative
    gain is a NaN the reference then converts to 0), near == -far, a pure delay, an exact half, and far-end only (near = 0)."""
    from test_aec_gpu import check_float_path, gpu_aec
    from test_aecm_gpu import run_gpu as gpu_aecm
    pkt, n = freq // 100, 900
    rng = np.random.default_rng(21)
    far = rng.integers(-20000, 20001, n * pkt).astype(np.int16)
    near = np.stack([far, -far, np.roll(far, 3), far // 2, np.zeros_like(far), np.roll(far, 40) // 4 + rng.integers(-3, 4, far.size).astype(np.int16)])
    got = gpu_aec(cuda, 1, freq, 10, 0, far, near.copy())
    want = np.stack([L.run_aec(oracle_port, 1, freq, 10, far, s, pkt, prefix="orc") for s in near])
    check_float_path(got, want, max_fraction=1e-4)
    got, rc = gpu_aecm(cuda, 1, freq, 10, far, near.copy())
    assert rc == 0
    want = np.stack([L.run_aecm(oracle_port, 1, freq, 10, far, s, pkt, prefix="orc") for s in near])
    assert np.array_equal(got, want)


@pytest.mark.parametrize("freq", [8000, 16000, 32000])
def test_noise_suppressors_on_tones_and_trains(cuda, oracle_port, freq):
    """a sine exactly on an FFT bin, one between two bins, a sweep, a click train at the block rate, white noise switched on and off:
    spectral flatness, the LRT features and the quantile noise estimate at their ends"""
    from test_ns_gpu import run_gpu as gpu_ns
    pkt, n = freq // 100, 600
    t = np.arange(n * pkt)
    fs = min(freq, 16000)
    rng = np.random.default_rng(4)
    x = np.stack([
        np.round(12000 * np.sin(2 * np.pi * t * (16 * fs / 256) / freq)),
        np.round(12000 * np.sin(2 * np.pi * t * (16.5 * fs / 256) / freq)),
        np.round(9000 * np.sin(2 * np.pi * (100 + t * 3000.0 / t.size) * t / freq)),
        (t % (pkt if freq <= 16000 else pkt // 2) == 0) * 30000.0,
        np.where((t // (pkt * 30)) % 2 == 0, rng.integers(-8000, 8001, t.size), 0),
    ]).astype(np.int16)
    _same(gpu_ns(cuda, 1, freq, x.copy()), np.stack([L.run_ns(oracle_port, 1, freq, s, pkt, prefix="orc") for s in x]), list("abcde"), "ns %d" % freq)
    if freq < 32000:
        from test_nsx_gpu import run_gpu as gpu_nsx
        _same(gpu_nsx(cuda, 1, freq, x.copy()), np.stack([L.run_nsx(oracle_port, 1, freq, s, pkt, prefix="orc") for s in x]), list("abcde"), "nsx %d" % freq)


@pytest.mark.parametrize("freq", [16000, 8000])
def test_every_stage_subset_of_the_chain_on_extreme_signals(cuda, oracle_port, freq):
    """wmx_chain_process with each of the 15 stage masks (the daemon switches NS / AEC / AGC / VAD on and off independently,
    src/wmix.c:617-703) on the extreme signals, one C call per tick, against the oracle chain with the same mask."""
    import torch
    from test_aec_gpu import check_float_path
    from wmix_amd.chain import ChainBatch
    pkt, names, x = extreme_signals(freq, 260)
    far = x[names.index("loud")]
    n = far.size // pkt
    dfar = torch.from_numpy(far.reshape(n, pkt).copy()).to(cuda)
    for stages in range(1, 16):
        cb = ChainBatch(x.shape[0], 1, freq, 10, 5, stages)
        d = torch.from_numpy(x.reshape(x.shape[0], n, pkt).copy()).to(cuda)
        for f in range(n):
            rc, _, _ = cb.process(dfar[f:f + 1] if stages & 2 else None, d[:, f:f + 1])
            assert rc == 0
        got = d.cpu().numpy().reshape(x.shape[0], -1)
        cb.close()
        want = np.stack([L.run_chain(oracle_port, 1, freq, 5, stages, far, s, pkt, prefix="orc") for s in x])
        if stages & 3:  # a float stage in the path
            check_float_path(got, want, max_fraction=1e-4)
        else:
            assert np.array_equal(got, want), stages
