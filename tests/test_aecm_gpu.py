"""GPU parity: wmix_amd/csrc/aecm.hip (fixed-point echo canceller, the reference's AECM build of aec_process2 & co) through
the C ABI vs the goldens of the real reference and vs the oracle.  Integer path: bit-exact everywhere."""
import ctypes as C
import os
import sys

import numpy as np
import pytest

from conftest import GOLDEN
from oracle import loader as L

sys.path.insert(0, GOLDEN)
from make_aecm_golden import AECM_CASES, aecm_case_input, case_key, speech_case  # noqa: E402
from test_aecm_oracle import NSG, G, check_against_golden  # noqa: E402

pytestmark = pytest.mark.gpu


def run_gpu(cuda, chn, freq, iv, far, near_streams, delay=0, split=0, packets_per_launch=50, packet_major=False):
    """far int16 [n*pkt*chn] (shared), near_streams int16 [S, n*pkt*chn] -> outputs [S, n*pkt*chn], return code."""
    import torch
    from wmix_amd.aecm import AecmBatch
    S = near_streams.shape[0]
    ab = AecmBatch(S, chn, freq, iv)
    per = ab.pkt
    n = far.size // per
    dfar = torch.from_numpy(np.ascontiguousarray(far.reshape(n, per))).to(cuda)
    if packet_major:
        d = torch.from_numpy(np.ascontiguousarray(near_streams.reshape(S, n, per).transpose(1, 0, 2))).to(cuda)
    else:
        d = torch.from_numpy(np.ascontiguousarray(near_streams.reshape(S, n, per))).to(cuda)
    rc = 0
    for f in range(0, n, packets_per_launch):
        fa = dfar[f:f + packets_per_launch]
        if split:
            rc = ab.set_frame_far(fa)
            if rc == 0:
                rc, _ = ab.process(d[:, f:f + packets_per_launch], delay_ms=delay)
        elif packet_major:
            rc, _ = ab.process2_packet_major(fa, d[f:f + packets_per_launch], delay_ms=delay)
        else:
            rc, _ = ab.process2(fa, d[:, f:f + packets_per_launch], delay_ms=delay)
        if rc != 0:
            break
    out = d.cpu().numpy()
    ab.close()
    return (out.transpose(1, 0, 2) if packet_major else out).reshape(S, -1), rc


@pytest.mark.parametrize("chn,freq,iv,n,delay,split", AECM_CASES)
def test_golden_bit_exact(cuda, chn, freq, iv, n, delay, split):
    far, near, pkt = aecm_case_input(chn, freq, iv, n)
    # > the 32-packet plan chunk; the split form (aec_setFrameFar, then aec_process) is driven packet by packet like the
    # generator drove the reference: buffering 70 far packets ahead would be a different, equally legal, call pattern
    got, rc = run_gpu(cuda, chn, freq, iv, far, near[None, :], delay, split, packets_per_launch=1 if split else 70)
    assert rc == 0
    check_against_golden(got[0], case_key(chn, freq, iv, delay, split), pkt * chn)


def test_golden_speech_bit_exact(cuda):
    far, near = speech_case(NSG)
    got, rc = run_gpu(cuda, 1, 8000, 10, far, near[None, :], packets_per_launch=7)
    assert rc == 0 and np.array_equal(got[0], G["speech_1x8000_out"])


@pytest.mark.parametrize("chn,freq,iv", [(1, 16000, 10), (1, 8000, 10), (2, 16000, 10), (1, 8000, 20)])
def test_many_streams_long_run_vs_oracle(cuda, oracle_port, chn, freq, iv):
    """66 near-end streams (ragged: not a multiple of 4) against one far-end, 1 100 packets: different echo delays and
    gains, near-end talkers, one silent stream, one clipping; packet-major and stream-major, 1 and 100 packets per launch."""
    from wmix_amd import synth
    S, n = 66, 1100
    far, _, pkt = aecm_case_input(chn, freq, iv, n, seed=777)
    f0 = far[::chn].astype(np.int32)
    near = np.zeros((S, n * pkt * chn), np.int16)
    rng = np.random.default_rng(9)
    for s in range(S):
        delay, gain = int(rng.integers(8, 900)), float(rng.uniform(0.1, 0.9))
        echo = np.zeros_like(f0)
        echo[delay:] = (f0[:-delay] * gain).astype(np.int32)
        x = echo + synth.lcg_noise([4000 + s], n * pkt, 150)[0].astype(np.int32)
        if s % 3 == 0:
            x += np.trunc(synth.gated_tone(n, pkt, amp=4000.0 * (1 + s % 5), period=37 + s)).astype(np.int32)
        if s == 7:
            x[:] = 0
        if s == 8:
            x *= 20
        x = np.clip(x, -32768, 32767).astype(np.int16)
        near[s] = np.repeat(x, chn) if chn == 2 else x
    want = np.stack([L.run_aecm(oracle_port, chn, freq, iv, far, near[s], pkt, prefix="orc") for s in range(S)])
    got, rc = run_gpu(cuda, chn, freq, iv, far, near, packets_per_launch=100, packet_major=(chn == 1))
    assert rc == 0 and np.array_equal(got, want)
    k = 150 * pkt * chn
    got1, rc = run_gpu(cuda, chn, freq, iv, far[:k], near[:9, :k], packets_per_launch=1)
    assert rc == 0 and np.array_equal(got1, want[:9, :k])


def test_bad_delay_processes_but_does_not_write(cuda, oracle_port):
    """delay outside [0, 500] ms: WebRtcAecm_Process clamps, processes and returns -1; the wrapper then leaves the packet's
    output unwritten and stops (src/webrtc.c:382-387).  The state must have advanced exactly like the reference's."""
    far, near, pkt = aecm_case_input(1, 16000, 10, 40)
    got, rc = run_gpu(cuda, 1, 16000, 10, far, near[None, :], delay=600, packets_per_launch=40)
    assert rc == -1 and np.array_equal(got[0], near)  # in place: nothing written, nothing after the first packet ran


def test_reference_host_signatures_with_aecm_switch(wmx, oracle_port, monkeypatch):
    """aec_init / aec_setFrameFar / aec_process / aec_process2 / aec_release over HOST buffers pick the fixed-point path
    when WMIX_AMD_AECM=1 -- the run-time form of the reference's source switch -- and the float path otherwise."""
    for chn, freq, iv in ((1, 8000, 20), (2, 16000, 10)):
        far, near, pkt = aecm_case_input(chn, freq, iv, 120, seed=5)
        monkeypatch.setenv("WMIX_AMD_AECM", "1")
        want = L.run_aecm(oracle_port, chn, freq, iv, far, near, pkt, prefix="orc")
        h = wmx.aec_init(chn, freq, iv, None)
        assert h
        buf = near.copy()
        step = pkt * chn * 2  # two packets per call
        for off in range(0, buf.size, step):
            p, f = C.c_void_p(buf.ctypes.data + 2 * off), C.c_void_p(far.ctypes.data + 2 * off)
            if (off // step) % 2:
                assert wmx.aec_setFrameFar(h, f, 2 * pkt) == 0 and wmx.aec_process(h, p, p, 2 * pkt, 0) == 0
            else:
                assert wmx.aec_process2(h, f, p, p, 2 * pkt, 0) == 0
        wmx.aec_release(h)
        # the split form buffers both far packets before it processes the two near packets: compare with the oracle driven
        # the same way
        a = L.port().orc_aecm_init
        a.restype = C.c_void_p
        o = a(chn, freq, iv)
        run = L.port().orc_aecm_run
        run.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int]
        ref = near.copy()
        for off in range(0, ref.size, step):
            p, f = C.c_void_p(ref.ctypes.data + 2 * off), C.c_void_p(far.ctypes.data + 2 * off)
            if (off // step) % 2:
                assert run(o, 1, f, None, None, 2 * pkt, 0) == 0 and run(o, 2, None, p, p, 2 * pkt, 0) == 0
            else:
                assert run(o, 3, f, p, p, 2 * pkt, 0) == 0
        L.port().orc_aecm_release.argtypes = [C.c_void_p]
        L.port().orc_aecm_release(o)
        assert np.array_equal(buf, ref)
        assert not np.array_equal(ref, want) or chn  # (the all-process2 run differs once the split calls reorder far / near)
    monkeypatch.setenv("WMIX_AMD_AECM", "1")
    assert wmx.aec_init(1, 32000, 10, None) is None


def test_full_size_batch_properties(cuda, oracle_port):
    """65 536 streams: identical streams give identical outputs wherever they sit in the batch; sampled streams equal
    the oracle."""
    import torch
    from wmix_amd.aecm import AecmBatch
    S, n = 65536, 24
    far, _, pkt = aecm_case_input(1, 16000, 10, n, seed=31)
    base = np.stack([aecm_case_input(1, 16000, 10, n, seed=31 + 0)[1], (aecm_case_input(1, 16000, 10, n, seed=31)[1] // 2),
                     np.zeros(n * pkt, np.int16), -aecm_case_input(1, 16000, 10, n, seed=31)[1]])
    idx = np.arange(S) % 4
    d = torch.from_numpy(base).to(cuda)[torch.from_numpy(idx).to(cuda)].reshape(S, n, pkt).transpose(0, 1).contiguous()
    dfar = torch.from_numpy(far.reshape(n, pkt).copy()).to(cuda)
    ab = AecmBatch(S, 1, 16000, 10)
    for f in range(n):
        rc, _ = ab.process2_packet_major(dfar[f:f + 1], d[f:f + 1])
        assert rc == 0
    out = d.cpu().numpy()
    ab.close()
    for k in range(4):
        want = L.run_aecm(oracle_port, 1, 16000, 10, far, base[k], pkt, prefix="orc").reshape(n, pkt)
        same = out[:, idx == k, :]
        assert (same == same[:, :1, :]).all() and np.array_equal(same[:, 0, :], want)
