"""GPU parity: wmix_amd/csrc/nsx.hip (fixed-point noise suppressor, the reference's MAKE_WEBRTC_NSX build of ns_process)
through the C ABI vs the goldens of the real reference and vs the oracle.  Integer path: bit-exact everywhere."""
import ctypes as C
import os
import sys

import numpy as np
import pytest

from conftest import GOLDEN
from oracle import loader as L

sys.path.insert(0, GOLDEN)
from make_nsx_golden import NSX_CASES, nsx_case_input  # noqa: E402
from test_nsx_oracle import NSG, G, check_against_golden  # noqa: E402

pytestmark = pytest.mark.gpu


def run_gpu(cuda, chn, freq, x_streams, packets_per_launch=64, packet_major=False):
    """x_streams: int16 [S, n_frames*pkt*chn] -> same shape, through NsxBatch."""
    import torch
    from wmix_amd.nsx import NsxBatch
    S = x_streams.shape[0]
    per = freq // 100 * chn
    nf = x_streams.shape[1] // per
    nb = NsxBatch(S, chn, freq)
    if packet_major:
        d = torch.from_numpy(np.ascontiguousarray(x_streams.reshape(S, nf, per).transpose(1, 0, 2))).to(cuda)
        for f in range(0, nf, packets_per_launch):
            nb.process_packet_major(d[f:f + packets_per_launch])
        out = d.cpu().numpy().transpose(1, 0, 2).reshape(S, -1)
    else:
        d = torch.from_numpy(np.ascontiguousarray(x_streams.reshape(S, nf, per))).to(cuda)
        for f in range(0, nf, packets_per_launch):
            nb.process(d[:, f:f + packets_per_launch])
        out = d.cpu().numpy().reshape(S, -1)
    nb.close()
    return out


@pytest.mark.parametrize("chn,freq,nf,amp", NSX_CASES)
def test_golden_bit_exact(cuda, chn, freq, nf, amp):
    # 700 packets per launch: one launch spans a 512-block threshold update (histogram scan after L2 atomics)
    got = run_gpu(cuda, chn, freq, nsx_case_input(chn, freq, nf, amp)[None, :], packets_per_launch=700)
    check_against_golden(got[0], chn, freq, amp)


@pytest.mark.parametrize("name,chn,freq", [("speech_1x8000", 1, 8000), ("speech_2x16000", 2, 16000)])
def test_golden_speech_bit_exact(cuda, name, chn, freq):
    got = run_gpu(cuda, chn, freq, NSG[name + "_in"][None, :], packets_per_launch=7)
    assert np.array_equal(got[0], G[name + "_out"])


@pytest.mark.parametrize("chn,freq", [(1, 16000), (1, 8000), (2, 16000), (2, 32000)])
def test_many_streams_long_run_vs_oracle(cuda, oracle_port, chn, freq):
    """66 independent streams (ragged: not a multiple of the 4 streams per workgroup), 1 100 packets (blocks 50 / 200 and
    two 512-block threshold updates), loudness from near-silence to clipping, one stream with long silences, packet-major
    and stream-major layouts, one packet and many packets per launch."""
    S, nf = 66, 1100
    amps = [3, 40, 800, 3000, 12000, 30000]
    x = np.stack([nsx_case_input(chn, freq, nf, amps[s % len(amps)], seed=2000 + 37 * s) for s in range(S)])
    x[5].reshape(nf, -1)[300:420] = 0
    want = np.stack([L.run_nsx(oracle_port, chn, freq, x[s], freq // 100, prefix="orc") for s in range(S)])
    got = run_gpu(cuda, chn, freq, x, packets_per_launch=100, packet_major=(chn == 1))
    assert np.array_equal(got, want)
    got1 = run_gpu(cuda, chn, freq, x[:9, : 130 * (freq // 100) * chn], packets_per_launch=1)
    assert np.array_equal(got1, want[:9, : 130 * (freq // 100) * chn])


def test_full_size_batch_properties(cuda, oracle_port):
    """65 536 streams (the chain's batch size): identical streams give identical outputs wherever they sit in the batch,
    and sampled streams equal the oracle."""
    import torch
    from wmix_amd.nsx import NsxBatch
    S, nf, pkt = 65536, 24, 160
    base = np.stack([nsx_case_input(1, 16000, nf, 3000, seed=77 + s) for s in range(4)])
    idx = np.arange(S) % 4
    d = torch.from_numpy(base).to(cuda)[torch.from_numpy(idx).to(cuda)].reshape(S, nf, pkt).transpose(0, 1).contiguous()
    nb = NsxBatch(S, 1, 16000)
    for f in range(nf):
        nb.process_packet_major(d[f:f + 1])
    out = d.cpu().numpy()  # [nf, S, pkt]
    nb.close()
    for k in range(4):
        want = L.run_nsx(oracle_port, 1, 16000, base[k], pkt, prefix="orc").reshape(nf, pkt)
        same = out[:, idx == k, :]
        assert (same == same[:, :1, :]).all()
        assert np.array_equal(same[:, 0, :], want)


def test_reference_host_signatures_with_nsx_switch(wmx, oracle_port, monkeypatch):
    """ns_init / ns_process / ns_release over HOST buffers pick the fixed-point path when WMIX_AMD_NSX=1 -- the run-time
    form of the reference's build-time MAKE_WEBRTC_NSX (src/webrtc.c:512-521) -- and the float path otherwise."""
    for chn, freq in ((1, 8000), (2, 16000), (1, 32000)):
        x = nsx_case_input(chn, freq, 120, 3000, seed=5)
        for switch, runner in (("1", L.run_nsx), ("0", L.run_ns)):
            monkeypatch.setenv("WMIX_AMD_NSX", switch)
            want = runner(oracle_port, chn, freq, x, freq // 100, prefix="orc")
            h = wmx.ns_init(chn, freq, None)
            assert h
            buf = x.copy()
            step = 2 * (freq // 100) * chn  # WMIX_FRAME_NUM: 20 ms per call
            for off in range(0, buf.size, step):
                p = C.c_void_p(buf.ctypes.data + 2 * off)
                wmx.ns_process(h, p, p, 2 * (freq // 100))
            wmx.ns_release(h)
            assert np.array_equal(buf, want)
    monkeypatch.setenv("WMIX_AMD_NSX", "1")
    assert wmx.ns_init(1, 44100, None) is None and wmx.ns_init(1, 24000, None) is None
