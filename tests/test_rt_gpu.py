"""wmx_rt_*: the paced heartbeat over S streams in host memory (src/wmix.c:536-538, 613-709, 820) -- sub-batches of one wmx_pipe each on one
upload and one download stream.  Every stream of every sub-batch, bit for bit against its own handle of the oracle; the streamed form
(pinned rows, tick by tick), the pipelined form (submit, submit, wait) and the resident form agree; the RTP form gives the datagrams
of the single-pipe pipeline."""
import numpy as np
import pytest
import torch

from oracle import loader as L
from wmix_amd import synth

pytestmark = pytest.mark.gpu


def _pcm_case(oracle_port, S, n, chn, freq, interval_ms, stages, seed):
    pkt10, ppc = freq // 100 * chn, interval_ms // 10
    far1 = synth.far_end(seed, n * ppc, freq // 100)
    near1 = synth.near_end(seed + 1, S, n * ppc, freq // 100, far=far1).reshape(S, -1)
    far, near = np.repeat(far1, chn), np.repeat(near1, chn, axis=1)
    want = np.stack([L.run_chain(oracle_port, chn, freq, 5, stages, far, near[s], freq // 100 * ppc, prefix="orc", interval_ms=interval_ms)
                     for s in range(S)])
    rows = np.ascontiguousarray(near.reshape(S, n, pkt10 * ppc).transpose(1, 0, 2))  # [n, S, package]
    return far.reshape(n, ppc, pkt10), rows, want


@pytest.mark.parametrize("chn,freq,interval_ms,stages,sub,slots", [(1, 16000, 20, 15, 16, 2), (1, 8000, 20, 15, 7, 1), (2, 16000, 10, 1 | 4 | 8, 40, 3)])
def test_rt_pcm_every_stream_vs_its_own_oracle_handle(cuda, oracle_port, chn, freq, interval_ms, stages, sub, slots):
    from wmix_amd.realtime import RtBatch
    S, n = 37, 90
    far, rows, want = _pcm_case(oracle_port, S, n, chn, freq, interval_ms, stages, 9100 + freq + chn)
    dfar = torch.from_numpy(far.copy()).to(cuda)
    rt = RtBatch(S, cuda, sub_batch=sub, slots=slots, kind="pcm", chn=chn, freq=freq, interval_ms=interval_ms, stages=stages)
    assert rt.B == -(-S // sub) and sum(rt.batch_n) == S
    got = np.zeros_like(rows)
    for k in range(n):
        slot = k % slots
        rt.fill(slot, rows[k])
        if k % 3 == 0:  # the far-end from the device ...
            assert rt.tick(dfar[k]) == slot
        else:           # ... or from the tick's own host samples (uploaded once, by the first sub-batch)
            rt.h_far[slot][:] = far[k]
            assert rt.tick(None) == slot
        got[k] = rt.gather(slot)
    assert rt.failed_steps() == 0
    rt.close()
    assert np.array_equal(got.transpose(1, 0, 2).reshape(S, -1), want)
    # pipelined: ticks queued `slots` deep before anybody waits (the bench's streaming form); resident: the same launches on rows in HBM
    rt2 = RtBatch(S, cuda, sub_batch=sub, slots=slots, kind="pcm", chn=chn, freq=freq, interval_ms=interval_ms, stages=stages, compute_streams=2)
    got2 = np.zeros_like(rows)
    k = 0
    while k < n:
        m = min(slots, n - k)
        for j in range(m):
            rt2.fill((k + j) % slots, rows[k + j])
            assert rt2.submit(dfar[k + j]) == (k + j) % slots
        rt2.wait()
        for j in range(m):
            got2[k + j] = rt2.gather((k + j) % slots)
        k += m
    rt2.close()
    assert np.array_equal(got2, got)
    rt3 = RtBatch(S, cuda, sub_batch=sub, slots=1, kind="pcm", chn=chn, freq=freq, interval_ms=interval_ms, stages=stages, compute_streams=3)
    d = torch.from_numpy(rows).to(cuda)
    for k in range(n):
        rt3.step_resident(d[k], dfar[k])
    rt3.close()
    assert np.array_equal(d.cpu().numpy(), got)


def test_rt_rtp_equals_the_single_pipe(cuda, oracle_port):
    from test_pipeline_gpu import make_datagrams
    from wmix_amd.pipeline import RtpChain
    from wmix_amd.realtime import RtBatch
    S, n = 29, 60
    far, pk = make_datagrams(oracle_port, S, n, seed=9300)
    dfar = torch.from_numpy(far.reshape(n, 2, 80).copy()).to(cuda)
    ch = RtpChain(S, cuda)
    d_in = torch.from_numpy(pk.transpose(1, 0, 2).copy()).to(cuda)
    d_out = torch.zeros_like(d_in)
    for k in range(n):
        ch.step(d_in[k], dfar[k], d_out[k])
    want = d_out.cpu().numpy()
    ch.close()
    rt = RtBatch(S, cuda, sub_batch=10, slots=2, kind="rtp")
    got = np.zeros_like(want)
    for k in range(n):
        slot = k % 2
        rt.fill(slot, pk[:, k])
        rt.h_far[slot][:] = far.reshape(n, 2, 80)[k]
        assert rt.tick(None) == slot
        got[k] = rt.gather(slot)
    rt.close()
    assert np.array_equal(got, want)
    rt = RtBatch(S, cuda, sub_batch=10, slots=1, kind="rtp")
    d_out.zero_()
    for k in range(n):
        rt.step_resident(d_in[k], dfar[k], d_out[k])
    rt.close()
    assert np.array_equal(d_out.cpu().numpy(), want)


def test_rt_bad_arguments(cuda):
    from wmix_amd._lib import WmxError
    from wmix_amd.realtime import RtBatch
    for kw in (dict(n_streams=0), dict(n_streams=8, sub_batch=0), dict(n_streams=8, slots=0), dict(n_streams=8, slots=17), dict(n_streams=8, freq=16050),
               dict(n_streams=8, chn=3), dict(n_streams=10 ** 9, sub_batch=1)):
        args = dict(dict(n_streams=8, dev=cuda), **kw)
        with pytest.raises(WmxError):
            RtBatch(args.pop("n_streams"), args.pop("dev"), **args)


def test_paced_loop_small(cuda, oracle_port):
    """the paced loop itself at a small S: every tick inside the reference's budget (tick - 2 ms), outputs equal to the oracle's"""
    from wmix_amd.realtime import GpuClock, RtBatch, latency_summary, paced_loop
    S, n, tick_ms = 512, 60, 20
    far, rows, want = _pcm_case(oracle_port, 8, n, 1, 16000, 20, 15, 9400)
    rows = np.ascontiguousarray(np.tile(rows, (1, S // 8, 1)))  # 8 distinct streams, tiled
    rt = RtBatch(S, cuda, sub_batch=200, slots=2, kind="pcm", chn=1, freq=16000, interval_ms=20)
    got = np.zeros((n, 8, rows.shape[2]), np.int16)
    sample = [0, 1, 2, 203, 204, 405, 510, 511]

    def tick(k):
        slot = k % 2
        rt.fill(slot, rows[k])
        rt.h_far[slot][:] = far[k]
        rt.tick(None)
        got[k] = rt.gather(slot, sample)
    lat, lag, clk = paced_loop(tick, tick_ms, n, GpuClock())
    s = latency_summary(lat, lag, tick_ms, clk)
    rt.close()
    for col, st in enumerate(sample):
        assert np.array_equal(got[:, col].reshape(-1), want[st % 8])
    assert s["ticks"] == n and s["misses"] <= 3, s  # (a late wake-up or two of a shared test box are not the library's: DESIGN.md 5a)


@pytest.mark.parametrize("freq,interval_ms,sub,slots", [(16000, 20, 5, 2), (8000, 20, 16, 1), (16000, 10, 3, 3)])
def test_rt_calls_every_stream_its_own_far_end(cuda, oracle_port, freq, interval_ms, sub, slots):
    """wmx_rt_create_pcm_calls: every stream hears a far-end of its own, like every handle of the reference (aec_process2(fp, far, near, ..),
    src/webrtc.c:410-483) -- far rows beside the near rows, from host memory (uploaded per sub-batch) and from the device, streamed and
    resident: every stream against an oracle handle fed ITS far-end."""
    from wmix_amd.realtime import RtBatch
    S, n = 13, 80
    pkt10, ppc = freq // 100, interval_ms // 10
    fars = np.stack([synth.far_end(9600 + 7 * s, n * ppc, pkt10) for s in range(S)])                          # [S, n * package]
    near = np.stack([synth.near_end(9700 + s, 1, n * ppc, pkt10, far=fars[s])[0] for s in range(S)])
    want = np.stack([L.run_chain(oracle_port, 1, freq, 5, 15, fars[s], near[s], pkt10 * ppc, prefix="orc", interval_ms=interval_ms) for s in range(S)])
    rows = np.ascontiguousarray(near.reshape(S, n, pkt10 * ppc).transpose(1, 0, 2))                         # [n, S, package]
    frows = np.ascontiguousarray(fars.reshape(S, n, pkt10 * ppc).transpose(1, 0, 2))
    dfar = torch.from_numpy(frows).to(cuda)
    rt = RtBatch(S, cuda, sub_batch=sub, slots=slots, kind="pcm", chn=1, freq=freq, interval_ms=interval_ms, far_rows=True)
    got = np.zeros_like(rows)
    for k in range(n):
        slot = k % slots
        rt.fill(slot, rows[k])
        if k % 2:
            rt.fill_far(slot, frows[k])
            assert rt.tick(None) == slot
        else:
            assert rt.tick(dfar[k]) == slot
        got[k] = rt.gather(slot)
    rt.close()
    assert np.array_equal(got.transpose(1, 0, 2).reshape(S, -1), want)
    rt2 = RtBatch(S, cuda, sub_batch=sub, slots=1, kind="pcm", chn=1, freq=freq, interval_ms=interval_ms, far_rows=True)
    d = torch.from_numpy(rows).to(cuda)
    for k in range(n):
        rt2.step_resident(d[k], dfar[k])
    rt2.close()
    assert np.array_equal(d.cpu().numpy(), got)
