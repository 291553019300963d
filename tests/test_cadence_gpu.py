"""GPU parity at the DAEMON'S OWN cadence.  wmix passes WMIX_INTERVAL_MS = 20 to aec_init / agc_init / vad_init
(src/wmixConf.h:112, src/wmix.c:636, 684, 703) and calls the record chain with 20 ms of audio per heartbeat
(src/wmix.c:613-709): the VAD packet is then 20 ms (src/webrtc.c:57-66), the AEC packet 20 ms at 8 kHz
(src/webrtc.c:239-248), NS and AGC keep 10 ms.  Everything else in the suite runs 10 ms handles (round-3 VERDICT).
Also: chains of two interleaved channels (the wrapper's in-place downmix / re-expansion of the VAD, the AEC's left channel
copied to both, the AGC's channel mean), which need the heartbeat's packets contiguous per stream (round-3 ADVICE)."""
import numpy as np
import pytest
import torch

from oracle import loader as L
from test_aec_gpu import check_float_path
from wmix_amd import synth
from wmix_amd._lib import WmxError
from wmix_amd.chain import ChainBatch

pytestmark = pytest.mark.gpu


def _inputs(seed, S, n10, chn, freq):
    """far int16 [n10 * pkt10 * chn], near [S, n10 * pkt10 * chn]; the right channel carries something else (the AEC
    and the NS treat the channels differently)"""
    p = freq // 100
    far = synth.far_end(seed, n10, p)
    near = synth.near_end(seed + 1, S, n10, p, far=far)
    if chn == 2:
        far = np.stack([far, far // 2], -1).reshape(-1)
        near = np.stack([near, near // 3], -1).reshape(S, -1)
    return np.ascontiguousarray(far, np.int16), np.ascontiguousarray(near, np.int16)


def _run_ticks(cuda, cb, far, near, n10_per_tick, packet_major=False):
    S, pkt = near.shape[0], cb.pkt
    T = near.shape[1] // (pkt * n10_per_tick)
    dfar = torch.from_numpy(far.reshape(T, n10_per_tick, pkt).copy()).to(cuda)
    d = torch.from_numpy(near.reshape(S, T, n10_per_tick, pkt).copy()).to(cuda)
    if packet_major:
        d = d.permute(1, 2, 0, 3).contiguous()  # [T, n10, S, pkt]
        for t in range(T):
            rc, _, _ = cb.process_packet_major(dfar[t], d[t])
            assert rc == 0
        return d.permute(2, 0, 1, 3).contiguous().cpu().numpy().reshape(S, -1)
    for t in range(T):
        rc, _, _ = cb.process(dfar[t], d[:, t])
        assert rc == 0
    return d.cpu().numpy().reshape(S, -1)


@pytest.mark.parametrize("freq,packet_major", [(16000, False), (16000, True), (8000, False)])
def test_chain_interval20_mono_vs_oracle(cuda, oracle_port, freq, packet_major):
    """20 ms per heartbeat through wmx_chain_process with interval_ms = 20 handles, every stream's heartbeat in one piece.  In a
    packet-major batch the VAD's 20 ms packet (and at 8 kHz the AEC's) would straddle two rows: refused before anything runs."""
    S, T = 40, 260
    far, near = _inputs(900 + freq // 8000, S, 2 * T, 1, freq)
    cb = ChainBatch(S, 1, freq, interval_ms=20)
    if packet_major:
        # the VAD's 20 ms packet would straddle two rows of a packet-major batch
        with pytest.raises(WmxError):
            _run_ticks(cuda, cb, far, near, 2, packet_major=True)
        cb.close()
        return
    got = _run_ticks(cuda, cb, far, near, 2)
    cb.close()
    for s in range(0, S, 3):
        want = L.run_chain(oracle_port, 1, freq, 5, 15, far, near[s], freq // 50, prefix="orc", interval_ms=20)
        check_float_path(got[s], want)


@pytest.mark.parametrize("freq,interval_ms,n10", [(16000, 10, 1), (16000, 10, 2), (16000, 20, 2), (8000, 20, 2)])
def test_two_channel_chain_vs_oracle(cuda, oracle_port, freq, interval_ms, n10):
    S, T = 24, 220
    far, near = _inputs(930 + freq // 8000 + interval_ms + n10, S, n10 * T, 2, freq)
    cb = ChainBatch(S, 2, freq, interval_ms=interval_ms)
    got = _run_ticks(cuda, cb, far, near, n10)
    cb.close()
    for s in range(0, S, 2):
        want = L.run_chain(oracle_port, 2, freq, 5, 15, far, near[s], freq // 100 * n10, prefix="orc", interval_ms=interval_ms)
        check_float_path(got[s], want)


def test_two_channel_packet_major_tick_is_refused_not_corrupted(cuda):
    """vad_process averages the channels of the WHOLE call in place and expands them again (src/webrtc.c:104-116, 145-150):
    with two packets per heartbeat lying in different rows that would rewrite the neighbouring streams' rows (round-3 ADVICE)."""
    S, freq = 16, 16000
    far, near = _inputs(960, S, 2, 2, freq)
    cb = ChainBatch(S, 2, freq, interval_ms=10)
    d = torch.from_numpy(near.reshape(S, 2, cb.pkt).transpose(1, 0, 2).copy()).to(cuda)
    before = d.clone()
    dfar = torch.from_numpy(far.reshape(2, cb.pkt).copy()).to(cuda)
    with pytest.raises(WmxError):
        cb.process_packet_major(dfar, d)
    cb.close()
    # one packet per heartbeat is fine in any layout
    cb = ChainBatch(S, 2, freq, interval_ms=10)
    rc, _, _ = cb.process_packet_major(dfar[:1], d[:1])
    assert rc == 0 and torch.equal(d[1], before[1])
    cb.close()
    # and the VAD alone refuses strides that do not cover what a two-channel call touches
    from wmix_amd._lib import lib
    import ctypes as C
    h = C.c_void_p()
    assert lib().wmx_vad_create(C.byref(h), S, 2, freq, 10) == 0
    pkt = 2 * 160
    rc = lib().wmx_vad_process(h, d.data_ptr(), 2, 1, pkt, S * pkt, torch.cuda.current_stream().cuda_stream)
    assert rc != 0
    lib().wmx_vad_destroy(h)


def test_config2_full_size_chain_interval20(cuda, oracle_port):
    """configs[2] at full size at the daemon's cadence: 65 536 streams, interval_ms = 20 handles, 20 ms per heartbeat through
    ONE wmx_chain_process call per tick.  Streams with equal input give equal output wherever they sit, and sampled streams
    agree with the oracle chain built from 20 ms handles."""
    S, T, U, freq = 65536, 16, 64, 16000
    far, uniq = _inputs(990, U, 2 * T, 1, freq)
    uniq[9] = 0
    cb = ChainBatch(S, 1, freq, interval_ms=20)
    pkt = cb.pkt
    d = torch.from_numpy(uniq.reshape(U, T, 2, pkt)).to(cuda)[torch.arange(S, device=cuda) % U].contiguous()  # [S, T, 2, pkt]
    dfar = torch.from_numpy(far.reshape(T, 2, pkt).copy()).to(cuda)
    for t in range(T):
        rc, _, _ = cb.process(dfar[t], d[:, t])
        assert rc == 0
    cb.close()
    first = d[:U]
    assert torch.equal(d.view(S // U, U, T, 2, pkt), first.unsqueeze(0).expand(S // U, U, T, 2, pkt))
    got = first.cpu().numpy().reshape(U, -1)
    for s in (0, 9, 31, 63):
        want = L.run_chain(oracle_port, 1, freq, 5, 15, far, uniq[s].reshape(-1), freq // 50, prefix="orc", interval_ms=20)
        check_float_path(got[s], want)


def test_legacy_adapter_heartbeat_latency_is_measured_and_bounded(cuda):
    """The unchanged daemon goes through include/wmix_compat.h: a batch of one, H2D -> launch -> D2H per call, four calls per
    20 ms heartbeat (src/wmix.c:613-709).  That path exists for link compatibility, not speed -- the CPU reference does a whole
    heartbeat in tens of microseconds -- but it must keep real time with a wide margin: a heartbeat is 20 000 us of audio."""
    import os
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools_dev"))
    from legacy_latency import heartbeat_latency
    r = heartbeat_latency(8000, 1, n_beats=300, warm=60)
    assert r["parity_max_lsb_vs_oracle"] == 0
    assert r["adapters_us_per_heartbeat"]["median"] < 2000.0, r   # 10 % of the heartbeat's 20 ms
    assert r["adapters_us_per_heartbeat"]["p99"] < 10000.0, r


@pytest.mark.parametrize("freq,interval_ms,n10", [(16000, 10, 1), (16000, 20, 2), (8000, 20, 2), (8000, 10, 1)])
def test_fixed_point_chain_is_bit_exact(cuda, oracle_port, freq, interval_ms, n10):
    """The heartbeat of the reference's OTHER builds -- MAKE_WEBRTC_NSX (src/webrtc.c:512-521) and the AECM switch
    (src/webrtc.c:168-191): WebRtcNsx_* and WebRtcAecm_* behind the same ns_* / aec_* calls -- as ONE wmx_chain_process call per
    tick (WMX_CHAIN_NSX | WMX_CHAIN_AECM).  Every stage is integer: the chain is bit-exact.  The oracle side composes the stages
    over the whole signal (a stage's output for a call depends on its input up to that call only)."""
    from wmix_amd.chain import AEC, AECM, AGC, NS, NSX, VAD
    S, T = 20, 240
    far, near = _inputs(1200 + freq // 8000 + interval_ms, S, n10 * T, 1, freq)
    cb = ChainBatch(S, 1, freq, interval_ms=interval_ms, stages=NS | AEC | AGC | VAD | NSX | AECM)
    got = _run_ticks(cuda, cb, far, near, n10)
    cb.close()
    per_call = freq // 100 * n10
    for s in range(0, S, 3):
        x = L.run_nsx(oracle_port, 1, freq, near[s], per_call, prefix="orc")
        x = L.run_aecm(oracle_port, 1, freq, interval_ms, far, x, per_call, 0, prefix="orc")
        x = L.run_agc(oracle_port, 1, freq, 5, x, per_call, prefix="orc")
        x = L.run_vad(oracle_port, 1, freq, interval_ms, x, per_call, prefix="orc")
        assert np.array_equal(got[s], x), "stream %d: %d samples differ" % (s, int((got[s] != x).sum()))


def test_fixed_point_stage_bits_need_their_stage(cuda):
    from wmix_amd.chain import AGC, NSX, VAD
    with pytest.raises(WmxError):
        ChainBatch(4, 1, 16000, stages=AGC | VAD | NSX)  # WMX_CHAIN_NSX without WMX_CHAIN_NS
