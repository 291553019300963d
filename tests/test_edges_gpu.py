"""Edge cases of the batched API on the GPU: ragged stream counts (not multiples of the 4 / 8 waves per workgroup or the
64 lanes of the lane-per-stream kernels), empty calls, padded (strided) layouts, launches longer than the per-launch
plan limit -- all against the oracle, same parity criteria as the main tests."""
import numpy as np
import pytest
import torch

from oracle import loader as L
from test_aec_gpu import check_float_path, gpu_chain
from wmix_amd import synth
from wmix_amd.aec import AecBatch
from wmix_amd.agc import AgcBatch
from wmix_amd.ns import NsBatch
from wmix_amd.vad import VadBatch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("S", [1, 3, 5, 7, 9, 63, 65])
def test_ragged_stream_counts_full_chain(cuda, oracle_port, S):
    n, freq = 60, 16000
    far = synth.far_end(900 + S, n, 160)
    near = synth.near_end(950 + S, S, n, 160, far=far)
    got = gpu_chain(cuda, 1, freq, 15, far, near, pkts_per_launch=7)
    for s in sorted({0, S // 2, S - 1}):
        want = L.run_chain(oracle_port, 1, freq, 5, 15, far, near[s], 160, prefix="orc")
        check_float_path(got[s], want)


def test_empty_calls_change_nothing(cuda):
    S = 5
    x = torch.randint(-3000, 3000, (S, 0, 160), dtype=torch.int16, device=cuda)
    far = torch.zeros((0, 160), dtype=torch.int16, device=cuda)
    ns, aec, agc, vad = NsBatch(S, 1, 16000), AecBatch(S, 1, 16000, 10), AgcBatch(S, 1, 16000, 5), VadBatch(S, 1, 16000, 10)
    before = ns.export_state(0)
    ns.process(x)
    rc, _ = aec.process2(far, x)
    assert rc == 0
    agc.process(x)
    vad.process(x)
    after = ns.export_state(0)
    # integer fields live in the float block as bit patterns (blockInd = -1 reads as NaN): compare the bits
    assert np.array_equal(before[0].view(np.uint32), after[0].view(np.uint32)) and np.array_equal(before[1], after[1])
    for b in (ns, aec, agc, vad):
        b.close()


def test_padded_strides_and_long_launches(cuda, oracle_port):
    """rows padded to 200 int16 per packet and 37 packets per launch (the AEC plans at most 16 per kernel launch and
    must chunk); only the 160 live samples of a packet may be touched"""
    S, n, pkt, pad = 6, 74, 160, 200
    far = synth.far_end(77, n, pkt)
    near = synth.near_end(78, S, n, pkt, far=far)
    buf = torch.full((S, n, pad), 12345, dtype=torch.int16, device=cuda)
    buf[:, :, :pkt] = torch.from_numpy(near.reshape(S, n, pkt)).to(cuda)
    dfar = torch.from_numpy(far.reshape(n, pkt).copy()).to(cuda)
    ns, aec, agc, vad = NsBatch(S, 1, 16000), AecBatch(S, 1, 16000, 10), AgcBatch(S, 1, 16000, 5), VadBatch(S, 1, 16000, 10)
    for f in range(0, n, 37):
        view = buf[:, f:f + 37, :pkt]  # strides (n*pad, pad, 1)
        ns.process(view)
        rc, _ = aec.process2(dfar[f:f + 37], view)
        assert rc == 0
        agc.process(view)
        vad.process(view)
    out = buf.cpu().numpy()
    assert (out[:, :, pkt:] == 12345).all()
    for s in (0, 5):
        want = L.run_chain(oracle_port, 1, 16000, 5, 15, far, near[s], pkt, prefix="orc")
        check_float_path(out[s, :, :pkt].reshape(-1), want)
    for b in (ns, aec, agc, vad):
        b.close()


def test_24khz_is_rejected_everywhere(cuda, wmx):
    """24000 passes the reference wrappers' `freq % 8000` test, but WebRtcNs_Init refuses it and WebRtcVad / WebRtcAgc
    fail on every packet (round-1 ADVICE: we used to build a 240-sample packet and process 160 of it)."""
    import ctypes as C
    from wmix_amd._lib import WmxError
    for mk in (lambda: NsBatch(2, 1, 24000), lambda: AgcBatch(2, 1, 24000, 5), lambda: VadBatch(2, 1, 24000, 10),
               lambda: AecBatch(2, 1, 24000, 10)):
        with pytest.raises(WmxError):
            mk()
    assert not wmx.ns_init(1, 24000, None) and not wmx.agc_init(1, 24000, 10, 5, None)
    assert not wmx.vad_init(1, 24000, 10, None) and not wmx.aec_init(1, 24000, 10, None)


def test_handles_know_their_device(cuda, wmx):
    ns = NsBatch(1, 1, 8000)
    assert wmx.wmx_handle_device(ns._h) == torch.cuda.current_device()
    assert wmx.wmx_handle_device(None) == -10001
    ns.close()


def _edge_signals(freq, n):
    """Per-stream (far, near) pairs the synthetic recipe never produces: activity that falls into digital silence (the AEC's
    smoothed spectra and the NS's noise estimate decay geometrically -- into float denormals -- for the rest of the run),
    full-scale square waves (every sample clips), a DC offset, and silence from the first sample on."""
    pkt = freq // 100
    rng = np.random.default_rng(77)
    t = np.arange(n * pkt)
    noise = lambda a: rng.integers(-a, a + 1, n * pkt).astype(np.int16)
    far = noise(8000)
    far[60 * pkt:] = 0                                   # far-end falls silent after 0.6 s
    near = np.zeros((5, n * pkt), np.int16)
    near[0] = (far.astype(np.int32) // 2 + noise(200)).astype(np.int16)
    near[0, 60 * pkt:] = 0                               # ... and so does the microphone: exact zeros to the end
    near[1] = np.where((t // 40) % 2 == 0, 32767, -32768).astype(np.int16)   # full-scale square wave
    near[2] = (12000 + noise(50)).astype(np.int16)       # DC offset
    near[3] = 0                                          # never anything
    near[4] = noise(30000)                               # loud noise, no echo in it
    return far, near


@pytest.mark.parametrize("freq", [16000, 8000])
def test_silence_clipping_dc_vs_oracle(cuda, oracle_port, freq):
    """3 000 packets (30 s): long enough for 0.9^n / 0.93^n recursions to run through the denormal range into zero.  The float path
    must follow the reference there too (x86 keeps denormals; so does gfx950 in the mode the library's kernels are built for)."""
    from test_aec_gpu import check_float_path, gpu_chain
    n, pkt = 3000, freq // 100
    far, near = _edge_signals(freq, n)
    got = gpu_chain(cuda, 1, freq, 15, far, near)
    want = np.stack([L.run_chain(oracle_port, 1, freq, 5, 15, far, near[s], pkt, prefix="orc") for s in range(near.shape[0])])
    check_float_path(got, want, max_fraction=1e-4)


def test_round5_entry_points_refuse_bad_arguments(cuda):
    """wmx_tick_*, wmx_pipe_*, the per-stream AGC gain and the per-cohort far-end of the chain: bad arguments come back as WMX_E*
    with nothing launched, like the reference's *_init returning NULL."""
    import ctypes as C
    import torch
    from wmix_amd._lib import WmxError, lib
    from wmix_amd.agc import AgcBatch
    from wmix_amd.chain import ChainBatch
    from wmix_amd.pipeline import RtpChain, StreamingPipe
    from wmix_amd.tick import TickBatch
    W = lib()
    h = C.c_void_p()
    assert W.wmx_tick_create(C.byref(h), 0, 1, 1, 8000, 20, 400, 5, 15) < 0              # no groups
    assert W.wmx_tick_create(C.byref(h), 2, 1, 1, 8000, 20, 410, 5, 15) < 0              # a delay that is not whole packages
    assert W.wmx_tick_create(C.byref(h), 2, 1, 1, 44100, 20, 400, 5, 15) < 0             # ns_init / aec_init return NULL at 44.1 kHz
    assert W.wmx_tick_create(C.byref(h), 2, 1, 1, 32000, 20, 400, 5, 15) < 0             # aec_init: freq > 16000
    assert W.wmx_pipe_create(C.byref(h), 4, 0, 0, 5, 15) < 0 and W.wmx_pipe_create(C.byref(h), 4, 3, 1, 5, 15) < 0  # slots, law
    tb = TickBatch(2, 2)
    rec = torch.zeros((4, 160), dtype=torch.int16, device=cuda)
    assert W.wmx_tick_record(tb._h, rec.data_ptr(), 100, None, 0, 0, None, None) < 0    # rows shorter than a package
    assert W.wmx_tick_play(tb._h, rec.data_ptr(), 100, None) < 0
    tb.close()
    ab = AgcBatch(8, 1, 16000, 5)
    with pytest.raises(WmxError):
        ab.set_gain_streams([0, 8], 9)      # stream 8 of 8
    with pytest.raises(WmxError):
        ab.reset_streams_gain([1], 250)     # agc_init returns NULL for this gain: nothing is reset
    assert ab.stream_gain(1) == 5 and ab.stream_gain(0) == 5
    ab.set_gain_streams([], 9)
    ab.close()
    with pytest.raises(WmxError):
        ChainBatch(4, 1, 16000, 10, 5, n_cohorts=2, stream_cohort=[0, 1, 2, 0])  # cohort 2 of 2
    pc = RtpChain(3, cuda, slots=2)
    pipe = StreamingPipe(pc)
    pipe.wait(-1)                           # nothing in flight: returns at once
    assert W.wmx_pipe_wait(pc._h, 2) < 0    # slot 2 of 2
    d = torch.zeros((3, 172), dtype=torch.uint8, device=cuda)
    far = torch.zeros((2, 80), dtype=torch.int16, device=cuda)
    assert W.wmx_pipe_step_resident(pc._h, d.data_ptr(), 100, far.data_ptr(), d.data_ptr(), 172, None) < 0  # rows shorter than a datagram
    pc.close()
    # later in round 5: the platform constants, rwTest, the PCM pipeline, the device-side pow sweep
    from wmix_amd.mix import MixBatch
    from wmix_amd.pipeline import PcmChain
    mb = MixBatch(2, 1, 8000)
    for bad in (16000, 3201, 1 << 20):      # not inside the ring / not a whole frame
        with pytest.raises(WmxError):
            mb.set_play_correct(bad)
    mb.set_play_correct(0)
    mb.close()
    assert W.wmx_mix_set_play_correct(None, 0) < 0 and W.wmx_tick_set_play_correct(None, 0) < 0 and W.wmx_tick_rw_test(None, 1) < 0
    with pytest.raises(KeyError):
        TickBatch.for_platform("qnx", 2)
    pcm = PcmChain(3, cuda, 1, 16000, 10, 5, 15, slots=1)
    x = torch.zeros((3, 200), dtype=torch.int16, device=cuda)
    farp = torch.zeros((1, 160), dtype=torch.int16, device=cuda)
    assert W.wmx_pipe_step_resident(pcm._h, x.data_ptr(), 400, farp.data_ptr(), x.data_ptr(), 320, None) < 0   # in place: one stride
    assert W.wmx_pipe_step_resident(pcm._h, x.data_ptr(), 300, farp.data_ptr(), x.data_ptr(), 300, None) < 0   # rows shorter than a package
    assert W.wmx_pipe_datagram_bytes(pcm._h) == 320
    pcm.close()
    assert W.wmx_debug_pow_device(None, None, None, 4, None) < 0 and W.wmx_debug_pow_device(x.data_ptr(), x.data_ptr(), x.data_ptr(), 0, None) == 0
