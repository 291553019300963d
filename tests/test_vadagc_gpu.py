"""GPU parity: wmix_amd/csrc/vad.hip and agc.hip through the C ABI vs the goldens of the real
reference and vs the oracle on many independent streams.  Integer paths: bit-exact."""
import ctypes as C
import os
import sys

import numpy as np
import pytest

from conftest import GOLDEN
from oracle import loader as L

sys.path.insert(0, GOLDEN)
from make_vadagc_golden import AGC_CASES, VAD_CASES, agc_input, agc_pkg, vad_input, vad_pkg  # noqa: E402

pytestmark = pytest.mark.gpu
G = np.load(os.path.join(GOLDEN, "vadagc_golden.npz"))


def gpu_vad(cuda, chn, freq, ims, k, x_streams, calls_per_launch=50, packet_major=False):
    import torch
    from wmix_amd.vad import VadBatch
    S = x_streams.shape[0]
    vb = VadBatch(S, chn, freq, ims)
    per_call = k * vb.pkt
    n_calls = x_streams.shape[1] // per_call
    if packet_major:
        d = torch.from_numpy(np.ascontiguousarray(x_streams.reshape(S, n_calls, per_call).transpose(1, 0, 2))).to(cuda)
        for c in range(0, n_calls, calls_per_launch):
            vb.process_packet_major(d[c:c + calls_per_launch], k)
        out = d.cpu().numpy().transpose(1, 0, 2).reshape(S, -1)
    else:
        d = torch.from_numpy(np.ascontiguousarray(x_streams.reshape(S, n_calls, per_call))).to(cuda)
        for c in range(0, n_calls, calls_per_launch):
            vb.process(d[:, c:c + calls_per_launch], k)
        out = d.cpu().numpy().reshape(S, -1)
    vb.close()
    return out


def gpu_agc(cuda, chn, freq, value, x_streams, packets_per_launch=64, packet_major=False, in_place=True):
    import torch
    from wmix_amd.agc import AgcBatch
    S = x_streams.shape[0]
    ab = AgcBatch(S, chn, freq, value)
    n = x_streams.shape[1] // ab.pkt
    if packet_major:
        d = torch.from_numpy(np.ascontiguousarray(x_streams.reshape(S, n, ab.pkt).transpose(1, 0, 2))).to(cuda)
        o = d if in_place else torch.zeros_like(d)
        for c in range(0, n, packets_per_launch):
            ab.process_packet_major(d[c:c + packets_per_launch], None if in_place else o[c:c + packets_per_launch])
        out = o.cpu().numpy().transpose(1, 0, 2).reshape(S, -1)
    else:
        d = torch.from_numpy(np.ascontiguousarray(x_streams.reshape(S, n, ab.pkt))).to(cuda)
        o = d if in_place else torch.zeros_like(d)
        for c in range(0, n, packets_per_launch):
            ab.process(d[:, c:c + packets_per_launch], None if in_place else o[:, c:c + packets_per_launch])
        out = o.cpu().numpy().reshape(S, -1)
    ab.close()
    return out


@pytest.mark.parametrize("chn,freq,ims,k", VAD_CASES)
def test_vad_golden(cuda, chn, freq, ims, k):
    x = vad_input(chn, freq, ims, k)
    got = gpu_vad(cuda, chn, freq, ims, k, x[None, :])
    assert np.array_equal(got[0], G["vad_%dx%d_%dms_k%d" % (chn, freq, ims, k)])


@pytest.mark.parametrize("chn,freq,value", AGC_CASES)
def test_agc_golden(cuda, chn, freq, value):
    x = agc_input(chn, freq)
    got = gpu_agc(cuda, chn, freq, value, x[None, :])
    assert np.array_equal(got[0], G["agc_%dx%d_v%d" % (chn, freq, value)])


def test_speech_goldens(cuda):
    sp = G["speech_in"]
    assert np.array_equal(gpu_vad(cuda, 1, 8000, 20, 1, sp[None, :])[0], G["speech_vad_20ms"])
    assert np.array_equal(gpu_agc(cuda, 1, 8000, 5, sp[None, :], in_place=False)[0], G["speech_agc_v5"])


@pytest.mark.parametrize("chn,freq,ims", [(1, 16000, 10), (1, 8000, 20), (2, 32000, 10)])
def test_vad_many_streams_vs_oracle(cuda, oracle_port, chn, freq, ims):
    S, n_calls = 130, 700  # 130: a partially filled third wave; 700 calls cross the 100-frame ageing of FindMinimum
    x = np.stack([vad_input(chn, freq, ims, 1, n_calls=n_calls, seed=200 + 7 * s) for s in range(S)])
    x[3] = 0  # a silent stream
    x[4] = (np.arange(x.shape[1]) % 251 * 131).astype(np.int16)  # a loud periodic one
    want = np.stack([L.run_vad(oracle_port, chn, freq, ims, x[s], vad_pkg(freq, ims), prefix="orc") for s in range(S)])
    got = gpu_vad(cuda, chn, freq, ims, 1, x, calls_per_launch=64, packet_major=(chn == 1))
    assert np.array_equal(got, want)


@pytest.mark.parametrize("chn,freq,value", [(1, 16000, 5), (1, 8000, 12), (2, 32000, 40)])
def test_agc_many_streams_vs_oracle(cuda, oracle_port, chn, freq, value):
    S, n = 130, 900
    x = np.stack([agc_input(chn, freq, n_calls=n, seed=300 + 11 * s) for s in range(S)])
    x[3] = 0
    x[4] = np.where(np.arange(x.shape[1]) % 2 == 0, 32767, -32768).astype(np.int16)  # full-scale square wave
    want = np.stack([L.run_agc(oracle_port, chn, freq, value, x[s], agc_pkg(freq), prefix="orc") for s in range(S)])
    got = gpu_agc(cuda, chn, freq, value, x, packets_per_launch=100, packet_major=(chn == 1))
    assert np.array_equal(got, want)


@pytest.mark.parametrize("chn,freq,packet_major", [(1, 16000, True), (1, 8000, False), (2, 32000, False), (2, 16000, True), (3, 16000, False)])
def test_agc_gain_per_stream(cuda, oracle_port, chn, freq, packet_major):
    """The compression gain is per handle in the reference (agc_init's value, agc_addition(fp, value): src/webrtc.c:694-753,
    824-839; src/wmix.c:684, 1068-1070): streams of ONE batch run with three different gains from the start, two groups get an
    agc_addition in mid-life (one of them twice, one back to the batch's own value), some are re-created with a gain of their
    own -- and every stream equals its own per-handle oracle run, bit for bit."""
    import torch
    from wmix_amd.agc import AgcBatch
    from wmix_amd._lib import WmxError
    S, n, step = 150, 360, 40  # 150: a partially filled third wave
    pkt = agc_pkg(freq)
    x = np.stack([agc_input(chn, freq, n_calls=n, seed=1300 + 7 * s) for s in range(S)])
    x[3] = 0
    sid = np.arange(S)
    v0 = np.where(sid % 3 == 0, 5, np.where(sid % 3 == 1, 20, 40))  # three gains from the first packet on
    ab = AgcBatch(S, chn, freq, 5)
    ab.reset_streams_gain(sid[sid % 3 == 1], 20)  # before the first packet: agc_init(.., 20, ..)
    ab.reset_streams_gain(sid[sid % 3 == 2], 40)
    ga, gb, gc = sid[sid % 5 == 1], sid[sid % 7 == 2], sid[sid % 11 == 4]
    events = {120: [("add", ga, 30)], 200: [("add", gb, 0), ("add", ga, 5)], 280: [("reinit", gc, 12)]}
    per = pkt * chn
    if packet_major:
        d = torch.from_numpy(np.ascontiguousarray(x.reshape(S, n, per).transpose(1, 0, 2))).to(cuda)
    else:
        d = torch.from_numpy(np.ascontiguousarray(x.reshape(S, n, per))).to(cuda)
    for c in range(0, n, step):
        for kind, who, v in events.get(c, []):
            (ab.set_gain_streams if kind == "add" else ab.reset_streams_gain)(who, v)
        if c == 200:
            with pytest.raises(WmxError):
                ab.set_gain_streams(ga, 200)  # WebRtcAgc_set_config refuses: agc_addition prints and nothing changes
        (ab.process_packet_major(d[c:c + step]) if packet_major else ab.process(d[:, c:c + step]))
    got = (d.cpu().numpy().transpose(1, 0, 2) if packet_major else d.cpu().numpy()).reshape(S, -1)
    assert ab.stream_gain(int(ga[0])) == 5 and ab.stream_gain(int(gc[0])) == 12
    for s in range(S):
        adds, cuts = {}, [0]
        for c, evs in sorted(events.items()):
            for kind, who, v in evs:
                if s in who:
                    if kind == "add":
                        adds[c] = v
                    else:
                        cuts.append(c)
        # a re-created handle starts over at its cut with its own gain
        lo = cuts[-1]
        val = int(v0[s]) if lo == 0 else 12
        adds = {c - lo: v for c, v in adds.items() if c >= lo}
        want = L.run_agc_handle(oracle_port, chn, freq, val, x[s, lo * per:], pkt, adds, prefix="orc")
        assert np.array_equal(got[s, lo * per:], want), (s, val, adds)
        if lo:  # what the old handle produced until it was released
            old = {c: v for c, v in ((c, v) for c, evs in events.items() for kind, who, v in evs if kind == "add" and s in who) if c < lo}
            want0 = L.run_agc_handle(oracle_port, chn, freq, int(v0[s]), x[s, : lo * per], pkt, old, prefix="orc")
            assert np.array_equal(got[s, : lo * per], want0), (s, "before the re-creation")
    # export / import carries the gain: a stream moved into a fresh batch continues bit for bit
    mover = int(gc[0])  # runs with 12 dB: a value the other batch has no table for yet
    blob = ab.export_stream(mover)
    other = AgcBatch(4, chn, freq, 9)
    other.import_stream(2, blob)
    assert other.stream_gain(2) == 12 and other.stream_gain(0) == 9
    tail = agc_input(chn, freq, n_calls=50, seed=77).reshape(1, 50, per)
    a = torch.from_numpy(np.repeat(tail, S, 0)).to(cuda)
    b = torch.from_numpy(np.repeat(tail, 4, 0)).to(cuda)
    ab.process(a)
    other.process(b)
    assert torch.equal(a[mover], b[2]) and not torch.equal(a[mover], b[1])
    # one value for the whole batch again: back on the single-table kernels, same results as a batch that never had another
    ab.set_gain(9)
    assert all(ab.stream_gain(s) == 9 for s in (0, 1, 2, int(gc[0])))
    other.close()
    ab.close()


def test_agc_addition_changes_the_table_like_the_reference(cuda, oracle_port):
    from wmix_amd.agc import AgcBatch
    from wmix_amd._lib import WmxError
    ab = AgcBatch(4, 1, 16000, 5)
    t5 = ab.gain_table()
    want = (C.c_int32 * 32)()
    oracle_port.orc_agc_gain_table(want, C.c_int16(5), C.c_int16(0), 0, C.c_int16(4 + (5 * 5 + 5) // 11))
    assert list(t5) == list(want)
    ab.set_gain(30)
    oracle_port.orc_agc_gain_table(want, C.c_int16(30), C.c_int16(0), 0, C.c_int16(4 + (5 * 30 + 5) // 11))
    assert list(ab.gain_table()) == list(want)
    with pytest.raises(WmxError):
        ab.set_gain(200)  # WebRtcAgc_set_config fails: diffGain >= 128
    assert list(ab.gain_table()) == list(want)  # the old table stays, like in the reference
    ab.close()
    with pytest.raises(WmxError):
        AgcBatch(1, 1, 16000, 250)  # agc_init returns NULL
    with pytest.raises(WmxError):
        AgcBatch(1, 1, 44100, 5)


def test_reference_host_signatures(wmx, oracle_port):
    """vad_* and agc_* over HOST buffers (src/webrtc.h:32-36,55-60), daemon call pattern: 20 ms calls."""
    assert wmx.vad_init(1, 48000, 20, None) is None and wmx.agc_init(1, 44100, 20, 5, None) is None
    assert wmx.agc_init(1, 16000, 20, 250, None) is None
    chn, freq = 1, 8000
    x = vad_input(chn, freq, 20, 1, n_calls=150, seed=9)
    want = L.run_vad(oracle_port, chn, freq, 20, x, 160, prefix="orc")
    h = wmx.vad_init(chn, freq, 20, None)
    buf = x.copy()
    for off in range(0, buf.size, 160):
        wmx.vad_process(h, C.c_void_p(buf.ctypes.data + 2 * off), 160)
    wmx.vad_release(h)
    assert np.array_equal(buf, want)
    chn, freq = 2, 16000
    x = agc_input(chn, freq, n_calls=200, seed=10)
    h = wmx.agc_init(chn, freq, 20, 5, None)
    buf = x.copy()
    step = 320 * chn  # 20 ms = two 10 ms packets per call
    want = L.run_agc(oracle_port, chn, freq, 5, x, 320, prefix="orc")
    for off in range(0, buf.size, step):
        p = C.c_void_p(buf.ctypes.data + 2 * off)
        assert wmx.agc_process(h, p, p, 320) == 0
    wmx.agc_addition(h, 9)
    wmx.agc_release(h)
    assert np.array_equal(buf, want)


def test_full_size_batch_properties(cuda):
    """65536 streams (configs[2] size): streams with equal input produce equal output wherever they sit."""
    import torch
    from wmix_amd.agc import AgcBatch
    from wmix_amd.vad import VadBatch
    S, n, pkt = 65536, 24, 160
    base = np.stack([agc_input(1, 16000, n_calls=n, seed=500 + s) for s in range(8)])
    idx = np.arange(S) % 8
    d = torch.from_numpy(base[idx].reshape(S, n, pkt)).to(cuda)
    ab, vb = AgcBatch(S, 1, 16000, 5), VadBatch(S, 1, 16000, 10)
    ab.process(d)
    vb.process(d)
    out = d.cpu().numpy()
    ab.close()
    vb.close()
    small = gpu_vad(cuda, 1, 16000, 10, 1, gpu_agc(cuda, 1, 16000, 5, base))
    assert np.array_equal(out[:8].reshape(8, -1), small)
    for k in range(8):
        assert (out[idx == k] == out[k]).all()


@pytest.mark.parametrize("freq", [8000, 16000])
@pytest.mark.parametrize("S,k", [(1, 1), (65, 2), (130, 3)])
def test_pipeline_kernels_equal_the_one_lane_kernels(cuda, monkeypatch, freq, S, k):
    """The four-wave VAD / AGC pipelines (mono 10 ms packets, aligned rows) against the one-lane-per-stream kernels they
    replace, on stream counts that leave lanes and workgroups partly empty and on calls of 1, 2 and 3 packets (the
    wrapper's analyse-packet-0-again quirk); then the AGC out of place."""
    n_calls = 150
    xv = np.stack([vad_input(1, freq, 10, k, n_calls=n_calls, seed=900 + 3 * s) for s in range(S)])
    xa = np.stack([agc_input(1, freq, n_calls=n_calls, seed=950 + 5 * s) for s in range(S)])
    res = {}
    for one_lane in ("1", "0"):
        monkeypatch.setenv("WMIX_AMD_VAD_ONE_LANE", one_lane)
        monkeypatch.setenv("WMIX_AMD_AGC_ONE_LANE", one_lane)
        res[one_lane] = (gpu_vad(cuda, 1, freq, 10, k, xv, calls_per_launch=40, packet_major=(S > 1)),
                         gpu_agc(cuda, 1, freq, 5, xa, packets_per_launch=64, packet_major=(S > 1)),
                         gpu_agc(cuda, 1, freq, 5, xa, packets_per_launch=7, in_place=False))
    for a, b in zip(res["1"], res["0"]):
        assert np.array_equal(a, b)


@pytest.mark.parametrize("chn,freq,ims", [(2, 16000, 10), (2, 8000, 10), (2, 32000, 10), (1, 32000, 10), (2, 16000, 20), (2, 8000, 20)])
def test_vad_pipeline_for_every_shape_vad_init_accepts(cuda, oracle_port, monkeypatch, chn, freq, ims):
    """vad_init takes any channel count and 8 / 16 / 32 kHz (src/webrtc.c:40-82); vad_process averages the channels in place,
    analyses and attenuates the mean and writes it back to every channel (:104-150); 32 kHz goes through two chained decimators
    (W: vad_core.c:623-644).  The four-wave pipeline now takes interleaved two-channel packets and 32 kHz: against the one-lane
    kernel on partly filled waves, and against the oracle's per-handle run."""
    S, n_calls = 130, 260
    x = np.stack([vad_input(chn, freq, ims, 1, n_calls=n_calls, seed=1500 + 7 * s) for s in range(S)])
    x[3] = 0
    x[4] = (np.arange(x.shape[1]) % 251 * 131).astype(np.int16)
    res = {}
    for one_lane in ("1", "0"):
        monkeypatch.setenv("WMIX_AMD_VAD_ONE_LANE", one_lane)
        res[one_lane] = gpu_vad(cuda, chn, freq, ims, 1, x, calls_per_launch=64, packet_major=True)
    assert np.array_equal(res["1"], res["0"])
    for s in (0, 3, 4, 63, 64, 129):
        assert np.array_equal(res["0"][s], L.run_vad(oracle_port, chn, freq, ims, x[s], vad_pkg(freq, ims), prefix="orc")), s
    # the gate moved: some packets came back attenuated, some not
    assert not np.array_equal(res["0"][0], x[0])


@pytest.mark.parametrize("freq", [8000, 32000])
def test_two_channel_agc_pipeline_equals_the_one_lane_kernel(cuda, monkeypatch, freq):
    """Interleaved stereo through the AGC pipeline (pair averaged on the way in, result written to both channels)."""
    S = 70
    xa = np.stack([agc_input(2, freq, n_calls=120, seed=990 + 5 * s) for s in range(S)])
    res = {}
    for one_lane in ("1", "0"):
        monkeypatch.setenv("WMIX_AMD_AGC_ONE_LANE", one_lane)
        res[one_lane] = (gpu_agc(cuda, 2, freq, 9, xa, packets_per_launch=50), gpu_agc(cuda, 2, freq, 9, xa, packets_per_launch=3, in_place=False))
    for a, b in zip(res["1"], res["0"]):
        assert np.array_equal(a, b)
