"""GPU parity: wmix_amd/csrc/aec.hip (+ the whole NS->AEC->AGC->VAD chain through the batched
C ABI) vs the reference goldens and vs the oracle on many streams sharing one far-end.
Tolerance stated by BASELINE.json for the float AEC / chain path: max |d| <= 1 LSB and RMS <= 1e-3
of full scale; the kernels keep the reference's operation order and (round 5) the reference's powf, so
the tests require every sample bit-identical -- see check_float_path."""
import ctypes as C
import os
import sys

import numpy as np
import pytest

from conftest import GOLDEN
from oracle import loader as L
from wmix_amd import synth

sys.path.insert(0, GOLDEN)
from make_aec_golden import AEC_CASES, CHAIN_CASES, aec_input, aec_pkg  # noqa: E402

pytestmark = pytest.mark.gpu
G = np.load(os.path.join(GOLDEN, "aec_golden.npz"))


def check_float_path(got, want, max_fraction=None):
    """Bit for bit (round 5).  Through round 4 this allowed 1 LSB in a 2e-5 fraction of samples: the kernels kept the reference's
    operation order but rounded a double-precision pow() where the reference calls the host's powf, which is not correctly rounded
    (measured then: 4 - 9 isolated samples of 2e7).  The product now evaluates glibc's powf algorithm itself
    (wmix_amd/csrc/libm_dev.h, tests/test_libm_tables.py::test_pow_sweep), and the float path is held to the same bar as the
    integer ones.  `max_fraction` is what callers used to pass; it no longer loosens anything -- except on a host whose own powf is
    another function than the one the device restates (conftest.host_powf_is_the_products: the parity claim is tied to the reference's
    libm, not to the test host's), where the bar is the old one: <= 1 LSB on a 2e-5 fraction of samples."""
    from conftest import host_powf_is_the_products
    d = got.astype(np.int32) - want.astype(np.int32)
    n_diff = int((d != 0).sum())
    if host_powf_is_the_products():
        assert n_diff == 0, "%d of %d samples differ, max |d| = %d LSB" % (n_diff, d.size, np.abs(d).max())
    else:
        assert np.abs(d).max() <= 1 and n_diff <= max(4, 2e-5 * d.size), "%d of %d samples differ, max |d| = %d LSB" % (n_diff, d.size, np.abs(d).max())


def gpu_aec(cuda, chn, freq, ims, delay, far, near_streams, pkts_per_launch=23, packet_major=False):
    import torch
    from wmix_amd.aec import AecBatch
    S = near_streams.shape[0]
    ab = AecBatch(S, chn, freq, ims)
    n = far.size // ab.pkt
    dfar = torch.from_numpy(np.ascontiguousarray(far.reshape(n, ab.pkt))).to(cuda)
    if packet_major:
        d = torch.from_numpy(np.ascontiguousarray(near_streams.reshape(S, n, ab.pkt).transpose(1, 0, 2))).to(cuda)
        for f in range(0, n, pkts_per_launch):
            rc, _ = ab.process2_packet_major(dfar[f:f + pkts_per_launch], d[f:f + pkts_per_launch], delay_ms=delay)
            assert rc == 0
        out = d.cpu().numpy().transpose(1, 0, 2).reshape(S, -1)
    else:
        d = torch.from_numpy(np.ascontiguousarray(near_streams.reshape(S, n, ab.pkt))).to(cuda)
        for f in range(0, n, pkts_per_launch):
            rc, _ = ab.process2(dfar[f:f + pkts_per_launch], d[:, f:f + pkts_per_launch], delay_ms=delay)
            assert rc == 0
        out = d.cpu().numpy().reshape(S, -1)
    ab.close()
    return out


def gpu_chain(cuda, chn, freq, stages, far, near_streams, agc_value=5, pkts_per_launch=16):
    """The daemon's record chain (src/wmix.c:613-709) on the batch API: NS -> AEC -> AGC -> VAD, in place."""
    import torch
    from wmix_amd.aec import AecBatch
    from wmix_amd.agc import AgcBatch
    from wmix_amd.ns import NsBatch
    from wmix_amd.vad import VadBatch
    S = near_streams.shape[0]
    pkt = freq // 100 * chn
    n = far.size // pkt
    ns = NsBatch(S, chn, freq) if stages & 1 else None
    aec = AecBatch(S, chn, freq, 10) if stages & 2 else None
    agc = AgcBatch(S, chn, freq, agc_value) if stages & 4 else None
    vad = VadBatch(S, chn, freq, 10) if stages & 8 else None
    dfar = torch.from_numpy(np.ascontiguousarray(far.reshape(n, pkt))).to(cuda)
    d = torch.from_numpy(np.ascontiguousarray(near_streams.reshape(S, n, pkt))).to(cuda)
    for f in range(0, n, pkts_per_launch):
        blk = d[:, f:f + pkts_per_launch]
        # every stage is causal per packet and only touches its own state, so running a stage over several
        # packets before the next stage equals the daemon's packet-by-packet interleaving
        if ns:
            ns.process(blk)
        if aec:
            rc, _ = aec.process2(dfar[f:f + pkts_per_launch], blk)
            assert rc == 0
        if agc:
            agc.process(blk)
        if vad:
            vad.process(blk)
    out = d.cpu().numpy().reshape(S, -1)
    for b in (ns, aec, agc, vad):
        if b:
            b.close()
    return out


@pytest.mark.parametrize("chn,freq,ims,delay,n", AEC_CASES)
def test_aec_golden(cuda, chn, freq, ims, delay, n):
    far, near = aec_input(chn, freq, ims, n)
    got = gpu_aec(cuda, chn, freq, ims, delay, far, near[None, :])
    check_float_path(got[0], G["aec_%dx%d_%dms_d%d" % (chn, freq, ims, delay)])


@pytest.mark.parametrize("chn,freq,stages,n", CHAIN_CASES)
def test_chain_golden(cuda, chn, freq, stages, n):
    far, near = aec_input(chn, freq, 10, n, seed=4100)
    got = gpu_chain(cuda, chn, freq, stages, far, near[None, :])
    check_float_path(got[0], G["chain_%dx%d_s%d" % (chn, freq, stages)])


def test_speech_goldens(cuda):
    far, near = G["speech_far"], G["speech_near"]
    check_float_path(gpu_aec(cuda, 1, 8000, 10, 0, far, near[None, :])[0], G["speech_aec"])
    check_float_path(gpu_chain(cuda, 1, 8000, 15, far, near[None, :])[0], G["speech_chain"])


@pytest.mark.parametrize("freq", [16000, 8000])
def test_many_streams_shared_far_vs_oracle(cuda, oracle_port, freq):
    """96 near-end streams against one far-end, 1300 packets (crosses the 500*mult noise-init blocks at 8 kHz
    and many NLP state changes); one silent stream, one stream that is pure echo."""
    from wmix_amd import synth
    S, n = 96, 1300
    pkg = freq // 100
    far = synth.far_end(6001, n, pkg)
    near = synth.near_end(6100, S, n, pkg, far=far)
    near[7] = 0
    near[8, 40:] = far[:-40] // 2
    want = np.stack([L.run_aec(oracle_port, 1, freq, 10, far, near[s], pkg, 0, prefix="orc") for s in range(S)])
    got = gpu_aec(cuda, 1, freq, 10, 0, far, near, pkts_per_launch=50, packet_major=True)
    check_float_path(got, want)


def test_chain_many_streams_vs_oracle(cuda, oracle_port):
    from wmix_amd import synth
    S, n, freq = 48, 800, 16000
    far = synth.far_end(7001, n, 160)
    near = synth.near_end(7100, S, n, 160, far=far)
    want = np.stack([L.run_chain(oracle_port, 1, freq, 5, 15, far, near[s], 160, prefix="orc") for s in range(S)])
    check_float_path(gpu_chain(cuda, 1, freq, 15, far, near), want)


def test_chain_parity_gate_3000_frames(cuda, oracle_port):
    """SURVEY section 8d parity gate: the full NS -> AEC -> AGC -> VAD chain over a 3 000-frame (30 s) run, 64 streams
    of a larger batch (256) picked at random, <= 1 LSB and <= 1e-3 RMS against the generic-C oracle.  Crosses every
    start-up phase (NS 50 / 200 / 500 / 1000 / ... blocks, AEC 1000 noise-init blocks, AGC and VAD hang-overs)."""
    from wmix_amd import synth
    S, n, freq = 256, 3000, 16000
    far = synth.far_end(8001, n, 160)
    near = synth.near_end(8100, S, n, 160, far=far)
    got = gpu_chain(cuda, 1, freq, 15, far, near, pkts_per_launch=50)
    pick = np.random.default_rng(8).choice(S, 64, replace=False)
    want = np.stack([L.run_chain(oracle_port, 1, freq, 5, 15, far, near[s], 160, prefix="orc") for s in pick])
    check_float_path(got[pick], want)


def test_reference_host_signatures(wmx, oracle_port):
    """aec_init / aec_process2 / aec_setFrameFar + aec_process / aec_release over HOST buffers (src/webrtc.h:40-45)."""
    assert wmx.aec_init(1, 32000, 10, None) is None
    far, near = aec_input(2, 16000, 10, 150, seed=77)
    want = L.run_aec(oracle_port, 2, 16000, 10, far, near, 320, 0, prefix="orc")  # 20 ms calls like the daemon
    h = wmx.aec_init(2, 16000, 20, None)
    buf = near.copy()
    for off in range(0, buf.size, 640):
        p = C.c_void_p(buf.ctypes.data + 2 * off)
        assert wmx.aec_process2(h, C.c_void_p(far.ctypes.data + 2 * off), p, p, 320, 0) == 0
    wmx.aec_release(h)
    assert np.array_equal(buf, want)
    # split form: aec_setFrameFar then aec_process gives the same stream as aec_process2
    far, near = aec_input(1, 8000, 10, 120, seed=78)
    want = L.run_aec(oracle_port, 1, 8000, 10, far, near, 80, 0, prefix="orc")
    h = wmx.aec_init(1, 8000, 10, None)
    out = np.zeros_like(near)
    for off in range(0, near.size, 80):
        assert wmx.aec_setFrameFar(h, C.c_void_p(far.ctypes.data + 2 * off), 80) == 0
        assert wmx.aec_process(h, C.c_void_p(near.ctypes.data + 2 * off), C.c_void_p(out.ctypes.data + 2 * off), 80, 0) == 0
    # out-of-range delay: the reference returns -1 (src/webrtc.c:382-387)
    assert wmx.aec_process(h, C.c_void_p(near.ctypes.data), C.c_void_p(out.ctypes.data), 80, 900) == -1
    wmx.aec_release(h)
    assert np.array_equal(out, want)


def test_full_size_batch_properties(cuda):
    """configs[2] size: 65536 streams, shared far-end.  Streams with equal near input give equal output wherever
    they sit; an all-zero near stream stays silent apart from comfort noise bounded by the noise floor."""
    import torch
    from wmix_amd import synth
    from wmix_amd.aec import AecBatch
    S, n, pkg = 65536, 30, 160
    far = synth.far_end(8001, n, pkg)
    base = synth.near_end(8100, 8, n, pkg, far=far)
    idx = np.arange(S) % 8
    d = torch.from_numpy(base[idx].reshape(S, n, pkg)).to(cuda)
    dfar = torch.from_numpy(far.reshape(n, pkg)).to(cuda)
    ab = AecBatch(S, 1, 16000)
    rc, _ = ab.process2(dfar, d)
    assert rc == 0
    out = d.cpu().numpy()
    ab.close()
    small = gpu_aec(cuda, 1, 16000, 10, 0, far, base)
    assert np.array_equal(out[:8].reshape(8, -1), small)
    for k in range(8):
        assert (out[idx == k] == out[k]).all()


def test_far_end_groups_vs_per_handle_oracle(cuda, oracle_port):
    """The reference handle owns its far-end (aec_process2(fp, far, near, ...), src/webrtc.c:410-483): one batch with 4
    far-ends x 24 streams (interleaved, so neighbouring waves of a workgroup use different far-ends) must give every stream
    what a per-handle oracle run with its own far-end gives.  VERDICT r01 item 8."""
    import torch
    from wmix_amd.aec import AecBatch
    n_far, per, n, pkt = 4, 24, 420, 160
    S = n_far * per
    fars = np.stack([synth.far_end(7000 + 13 * g, n, pkt, amp=2000 * (g + 1)) for g in range(n_far)])
    stream_far = np.arange(S) % n_far
    near = np.stack([synth.near_end(7100 + s, 1, n, pkt, far=fars[stream_far[s]], delay=20 + 7 * (s % 9))[0] for s in range(S)])
    ab = AecBatch(S, 1, 16000, 10, stream_far=stream_far)
    d = torch.from_numpy(near.reshape(S, n, pkt).copy()).to(cuda)
    dfar = torch.from_numpy(fars.reshape(n_far, n, pkt).copy()).to(cuda)
    for f in range(0, n, 37):
        rc, _ = ab.process2(dfar[:, f:f + 37], d[:, f:f + 37])
        assert rc == 0
    got = d.cpu().numpy().reshape(S, -1)
    ab.close()
    for s in list(range(0, S, 5)) + [S - 1]:
        want = L.run_aec(oracle_port, 1, 16000, 10, fars[stream_far[s]], near[s], pkt, prefix="orc")
        check_float_path(got[s], want)
    # and a map is refused when it points outside the groups
    from wmix_amd._lib import lib
    import ctypes as C
    hnd = C.c_void_p()
    bad = np.array([0, 1, 4], np.int32)
    assert lib().wmx_aec_create_groups(C.byref(hnd), 3, 1, 16000, 10, 4, bad.ctypes.data) == -10001
