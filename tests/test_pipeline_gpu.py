"""SURVEY 8f-1 end to end on the GPU: RTP/PCMA datagrams -> decode -> NS -> AEC -> AGC -> VAD -> encode -> RTP datagrams,
bit for bit against the oracle composition orc_rtp_ingest -> orc_run_chain -> orc_rtp_egress (the float stages in their
+-1 LSB class, which G.711 quantises away or keeps: compared code for code with a tolerance of one code step where the
PCM differed by 1 LSB).  Resident and streaming (pinned host buffers, overlapped copies) forms give identical datagrams."""
import ctypes as C

import numpy as np
import pytest
import torch

from oracle import loader as L
from wmix_amd import synth

pytestmark = pytest.mark.gpu


def make_datagrams(port, S, n, seed):
    """n datagrams per stream: A-law of a synthetic near-end (echo of the shared far-end + noise + gated tone)."""
    far = synth.far_end(seed, 2 * n, 80)
    near = synth.near_end(seed + 1, S, 2 * n, 80, far=far)
    enc = port.orc_PCM2G711a
    enc.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int]
    pk = np.zeros((S, n, 172), np.uint8)
    pk[:, :, 0], pk[:, :, 1] = 0x80, 0x88
    for s in range(S):
        x = np.ascontiguousarray(near[s])
        codes = np.zeros(x.size, np.uint8)
        enc(x.ctypes.data, codes.ctypes.data, x.size * 2, 0)
        pk[s, :, 12:] = codes.reshape(n, 160)
        pk[s, :, 3] = np.arange(n) & 0xFF
    return far, pk


def oracle_pipeline(port, far, pk_stream):
    return L.run_rtp_chain(port, far, pk_stream)


def test_rtp_chain_vs_oracle_resident_and_streaming(cuda, oracle_port):
    from wmix_amd.pipeline import RtpChain, StreamingPipe
    S, n = 37, 150  # 3 s: past the AEC start-up and the NS start-up blocks
    far, pk = make_datagrams(oracle_port, S, n, seed=8800)
    dfar = torch.from_numpy(far.reshape(n, 2, 80).copy()).to(cuda)
    # resident form
    ch = RtpChain(S, cuda)
    d_in = torch.from_numpy(pk.transpose(1, 0, 2).copy()).to(cuda)  # [n, S, 172]
    d_out = torch.zeros_like(d_in)
    for k in range(n):
        ch.step(d_in[k], dfar[k], d_out[k])
    got = d_out.cpu().numpy().transpose(1, 0, 2)
    ch.close()
    # streaming form: datagrams start and end in (pinned) host memory
    ch2 = RtpChain(S, cuda)
    pipe = StreamingPipe(ch2)
    got2 = np.zeros_like(got)
    pending = []
    for k in range(n):
        slot = k % pipe.SLOTS
        if len(pending) == pipe.SLOTS:  # the slot is about to be reused: collect its result first
            kk, ss = pending.pop(0)
            pipe.wait(ss)
            got2[:, kk] = pipe.h_out[ss]
        pipe.h_in[slot][:] = pk[:, k]
        if k % 2:  # the far-end from host memory too, every other step (wmx_pipe_far), else from the device
            pipe.h_far[slot][:] = far.reshape(n, 2, 80)[k]
            got_slot = pipe.submit(None)
        else:
            got_slot = pipe.submit(dfar[k])
        assert got_slot == slot
        pending.append((k, got_slot))
    pipe.drain()
    for kk, ss in pending:
        got2[:, kk] = pipe.h_out[ss]
    ch2.close()
    assert np.array_equal(got, got2)
    for s in (0, 1, 17, 36):
        want = oracle_pipeline(oracle_port, far, pk[s])
        assert np.array_equal(got[s][:, :12], want[:, :12])  # headers: v/m/pt, running sequence number, timestamp
        # payload: the float stages may differ from the CPU by 1 LSB on a few samples (AEC powf); a 1-LSB PCM
        # difference moves an A-law code by at most one step.  Observed: identical.
        diff = got[s][:, 12:].astype(np.int16) - want[:, 12:].astype(np.int16)
        assert not diff.any()
    assert got[0][5, 3] == 5 and got[0][5, 1] == 0x88


@pytest.mark.parametrize("chn,freq,interval_ms,stages,slots", [(1, 16000, 10, 15, 3), (1, 8000, 20, 15, 2), (2, 32000, 10, 1 | 4 | 8, 1)])
def test_pcm_pipe_vs_oracle_resident_and_streaming(cuda, oracle_port, chn, freq, interval_ms, stages, slots):
    """wmx_pipe_create_pcm: the heartbeat's own boundary -- a package in HOST memory worked on in place (buffSrc, src/wmix.c:609-709) --
    for a batch: pinned rows in, pinned rows out, copies overlapped by the library.  Resident and streaming forms against per-handle
    oracle runs (the daemon's cadence: what aec_init / agc_init / vad_init are given is interval_ms), and the far-end from the slot's
    own host samples as well as from the device."""
    from wmix_amd.pipeline import PcmChain, StreamingPipe
    from wmix_amd._lib import WmxError
    S, n = 23, 130
    pkt10, ppc = freq // 100 * chn, interval_ms // 10
    far1 = synth.far_end(8900 + freq, n * ppc, freq // 100)           # mono signal ...
    near1 = synth.near_end(8901 + freq, S, n * ppc, freq // 100, far=far1).reshape(S, -1)
    far = np.repeat(far1, chn)                                        # ... interleaved to chn channels (the AEC takes channel 0)
    near = np.repeat(near1, chn, axis=1)
    if chn == 2:
        near[:, 1::2] = near[:, 1::2] // 2                            # a right channel of its own
    want = np.stack([L.run_chain(oracle_port, chn, freq, 5, stages, far, near[s], freq // 100 * ppc, prefix="orc", interval_ms=interval_ms)
                     for s in range(S)])
    rows = near.reshape(S, n, pkt10 * ppc).transpose(1, 0, 2)         # [n, S, package]
    dfar = torch.from_numpy(far.reshape(n, ppc, pkt10).copy()).to(cuda)
    ch = PcmChain(S, cuda, chn, freq, interval_ms, 5, stages, slots=slots)
    d = torch.from_numpy(np.ascontiguousarray(rows)).to(cuda)
    for k in range(n):
        ch.step(d[k], dfar[k])
    got = d.cpu().numpy().transpose(1, 0, 2).reshape(S, -1)
    ch.close()
    assert np.array_equal(got, want)
    ch2 = PcmChain(S, cuda, chn, freq, interval_ms, 5, stages, slots=slots)
    pipe = StreamingPipe(ch2)
    got2 = np.zeros((n, S, pkt10 * ppc), np.int16)
    pending = []
    for k in range(n):
        slot = k % pipe.SLOTS
        if len(pending) == pipe.SLOTS:
            kk, ss = pending.pop(0)
            pipe.wait(ss)
            got2[kk] = pipe.h_out[ss]
        pipe.h_in[slot][:] = rows[k]
        if k % 2:
            pipe.h_far[slot][:] = far.reshape(n, ppc, pkt10)[k]
            assert pipe.submit(None) == slot
        else:
            assert pipe.submit(dfar[k]) == slot
        pending.append((k, slot))
    for kk, ss in pending:
        pipe.wait(ss)
        got2[kk] = pipe.h_out[ss]
    ch2.close()
    assert np.array_equal(got2.transpose(1, 0, 2).reshape(S, -1), want)
    for bad in ((3, 16000, 10), (1, 16050, 10), (1, 16000, 15), (1, 48000, 10)):  # the last: a rate the AEC / NS refuse (freq > 32000)
        with pytest.raises(WmxError):
            PcmChain(4, cuda, bad[0], bad[1], bad[2], 5, 15)
